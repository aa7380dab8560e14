# -*- coding: utf-8 -*-
''' The N > 1 path on CPU: world_size = 2 over gloo. Each rank takes its shard of a queue,
    computes per-configuration metric rows (here with the oracle, the GPU being absent) and the
    rows are all-gathered back into queue order. '''
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from conftest import ROOT
from pysonic_amd.parallel import shard_bounds, shard_queue, all_gather_rows


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 64, 4096, 4097):
        for world in (1, 2, 3, 8):
            blocks = [shard_bounds(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    assert shard_queue(list(range(10)), 1, 3) == [4, 5, 6]
    with pytest.raises(ValueError):
        shard_bounds(10, 3, 3)
    rows = np.arange(12.).reshape(4, 3)
    np.testing.assert_array_equal(all_gather_rows(rows, 4), rows)


WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, {root!r})
    import torch.distributed as dist
    from pysonic_amd.parallel import shard_bounds, all_gather_rows
    from oracle import oracle as O
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    d = np.load(os.path.join({root!r}, 'pysonic_amd', 'lookups', 'tables_RS_32nm_500kHz.npz'))
    tables = np.array([d['tab_' + str(k)] for k in d['keys']])
    queue = [(a, dc) for a in (50e3, 200e3, 400e3) for dc in (0.5, 1.0)][:5]   # 5 configs: uneven
    a, b = shard_bounds(len(queue), rank, world)
    rows = []
    for amp, dc in queue[a:b]:
        ev, tstop = O.pulsed_events(5e-3, 1e-3, 1000., dc)
        out = O.sim_sonic('RS', d['A'], d['Q'], tables, amp, ev, tstop)
        rows.append([amp, dc, out['Qm'].max(), out['Qm'][-1], float(out['t'].size)])
    full = all_gather_rows(np.array(rows).reshape(-1, 5), len(queue), dist)
    if rank == 0:
        np.save(sys.argv[1], full)
    dist.barrier()
    dist.destroy_process_group()
''')


def test_two_rank_gather_gloo(tmp_path):
    script = os.path.join(tmp_path, 'worker.py')
    with open(script, 'w') as fh:
        fh.write(WORKER.format(root=ROOT))
    out = os.path.join(tmp_path, 'gathered.npy')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port), script, out]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    full = np.load(out)
    assert full.shape == (5, 5)
    queue = [(a, dc) for a in (50e3, 200e3, 400e3) for dc in (0.5, 1.0)][:5]
    np.testing.assert_array_equal(full[:, :2], np.array(queue))      # queue order preserved
    assert np.all(full[:, 4] == full[0, 4]) and np.all(np.isfinite(full))
    assert full[4, 2] > full[0, 2]                                   # stronger drive -> higher peak


def test_weighted_bounds():
    from pysonic_amd.parallel import weighted_bounds, run_sharded
    for n in (0, 1, 5, 100):
        for world in (1, 2, 3, 8):
            b = weighted_bounds(np.ones(n), world)
            assert len(b) == world and b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
    # one costly item gets a rank of its own; the cheap tail is spread over the others
    costs = np.array([100.] + [1.] * 99)
    b = weighted_bounds(costs, 4)
    assert b[0] == (0, 1) and all(y > x for x, y in b[1:])
    sums = [costs[x:y].sum() for x, y in b]
    assert max(sums[1:]) <= 40
    # single process: run_sharded is the local launch
    rows = run_sharded(lambda a, b_: np.arange(a, b_, dtype=float)[:, None] * [1., 2.], 7)
    np.testing.assert_array_equal(rows, np.arange(7.)[:, None] * [1., 2.])


def test_dealt_shards():
    from pysonic_amd.parallel import dealt_shards, run_sharded
    rng = np.random.default_rng(3)
    for n in (0, 1, 5, 100):
        for world in (1, 2, 3, 8):
            sh = dealt_shards(rng.random(n), world)
            assert len(sh) == world and all(np.all(np.diff(x) > 0) for x in sh)          # ascending within a rank
            np.testing.assert_array_equal(np.sort(np.concatenate(sh)), np.arange(n))     # a partition of the queue
            assert max(len(x) for x in sh) - min(len(x) for x in sh) <= 1
    # balanced whatever the estimate misses: the true cost = the estimate times a factor that depends on where the
    # item sits in the sweep (what a firing threshold does to an amplitude sweep)
    n = 4096
    est = np.linspace(1., 2., n)
    true = est * np.where(np.arange(n) > n // 3, 8., 1.)
    for world in (2, 4, 8):
        sums = np.array([true[x].sum() for x in dealt_shards(est, world)])
        assert sums.max() / sums.mean() < 1.01
    # single process: the local launch, on the index array
    rows = run_sharded(lambda idx: np.asarray(idx, dtype=float)[:, None] * [1., 2.], 7, dealt=True)
    np.testing.assert_array_equal(rows, np.arange(7.)[:, None] * [1., 2.])


WORKER2 = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, {root!r})
    import torch.distributed as dist
    from pysonic_amd.parallel import run_sharded, weighted_bounds
    from pysonic_amd import Batch
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    n = 11
    costs = np.array([8., 1, 1, 1, 1, 1, 1, 1, 1, 1, 1])
    seen = []
    def launch(a, b):                      # stub launcher: what a metrics-only kernel launch returns
        seen.append((a, b))
        i = np.arange(a, b, dtype=float)
        return np.stack([i, i * i, np.full(b - a, float(rank))], axis=1)
    rows = run_sharded(launch, n, costs=costs, dist=dist)
    assert seen == [weighted_bounds(costs, world)[rank]]
    # dealt by estimated cost: launch(indices), rows back in item order
    from pysonic_amd.parallel import dealt_shards
    got = []
    def launch_items(idx):
        got.append(np.asarray(idx))
        i = np.asarray(idx, dtype=float)
        return np.stack([i, i * i, np.full(len(idx), float(rank))], axis=1)
    rows_d = run_sharded(launch_items, n, costs=costs, dist=dist, dealt=True)
    assert len(got) == 1 and np.array_equal(got[0], dealt_shards(costs, world)[rank])
    assert np.array_equal(rows_d[:, 0], np.arange(n)) and np.array_equal(rows_d[:, 1], np.arange(n)**2.)
    assert set(rows_d[:, 2]) == set(float(r) for r in range(world)) and rows_d[0, 2] == 0.     # the costliest item: rank 0

    class Stub:                            # owner with a batched implementation, like NeuronalBilayerSonophore
        calls = 0
        def work(self, x, y=0):
            return (x + y, rank)
        def _batched_work(self, calls):
            Stub.calls += len(calls)
            return [(a[0] + k.get('y', 0), rank) for a, k in calls]
        def _queueCosts(self, calls):
            return [1.0 + a[0] for a, _ in calls]
    stub = Stub()
    queue = [[i] for i in range(6)] + [([10], dict(y=5))]
    out = Batch(stub.work, queue).run(mpi=True)
    if rank == 0:
        np.save(sys.argv[1], rows)
        np.save(sys.argv[1] + '.batch.npy', np.array(out, dtype=float))
        np.save(sys.argv[1] + '.calls.npy', np.array([Stub.calls]))
    dist.barrier()
    dist.destroy_process_group()
''')


def test_run_sharded_and_batch_two_ranks_gloo(tmp_path):
    ''' run_sharded (cost-weighted split, one all-gather of the rows) and Batch.run(mpi=True) under a
        process group (queue split over the ranks, results back in queue order on every rank), with
        stub launchers: the N > 1 path of the product without a GPU '''
    script = os.path.join(tmp_path, 'worker2.py')
    with open(script, 'w') as fh:
        fh.write(WORKER2.format(root=ROOT))
    out = os.path.join(tmp_path, 'rows.npy')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port), script, out]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    rows = np.load(out)
    np.testing.assert_array_equal(rows[:, 0], np.arange(11.))
    np.testing.assert_array_equal(rows[:, 1], np.arange(11.)**2)
    # the costly first item went to rank 0 with few others; the rest to rank 1
    assert rows[0, 2] == 0 and rows[-1, 2] == 1 and np.count_nonzero(rows[:, 2] == 0) < 5
    batch = np.load(out + '.batch.npy')
    np.testing.assert_array_equal(batch[:, 0], [0, 1, 2, 3, 4, 5, 15])      # queue order, kwargs honoured
    assert set(batch[:, 1]) == {0., 1.}                                       # both ranks worked
    assert int(np.load(out + '.calls.npy')[0]) < 7                            # rank 0 ran only its share


WORKER_GPU = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, {root!r})
    import torch.distributed as dist
    from pysonic_amd import (NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, Batch, getPointNeuron)
    from pysonic_amd.parallel import run_sharded
    from pysonic_amd import _native as N
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron('RS'))
    queue = [[AcousticDrive(500e3, float(a)), PulsedProtocol(20e-3, 5e-3, 100., float(dc))]
             for a in (30e3, 100e3, 300e3, 600e3) for dc in (0.3, 1.0)][:7]          # 7 configurations: uneven
    out = Batch(nbls.simulate, queue).run(mpi=True, gather=True)      # sharded over the ranks, gathered on every rank
    mine = Batch(nbls.simulate, queue).run(mpi=True)           # default for simulate: a rank keeps what it computed
    held = [i for i, o in enumerate(mine) if o is not None]
    assert len(mine) == len(queue) and 0 < len(held) < len(queue)
    for i in held:
        assert np.array_equal(mine[i][0]['Qm'].values, out[i][0]['Qm'].values)
    # a metrics-only sweep split by cost with one all-gather of the metric rows
    cfgs = [(q[0], q[1]) for q in queue]
    costs = NeuronalBilayerSonophore._queueCosts([([d, pp], {{}}) for d, pp in cfgs])
    sizes = []
    def launch(a, b):
        sizes.append(b - a)
        return nbls.runSonicBatch(500e3, 1., cfgs[a:b], traces=False)[1]
    rows = run_sharded(launch, len(cfgs), costs=costs)
    np.savez(sys.argv[1] + f'.rank{{rank}}.npz', qm=np.stack([d['Qm'].values for d, _ in out]), rows=rows,
             share=np.array(sizes), device=np.array([nbls._device()]), held=np.array(held))
    dist.barrier()
    dist.destroy_process_group()
''')


@pytest.mark.gpu
def test_product_path_two_ranks_on_one_gpu(tmp_path):
    ''' The N > 1 path of the product with the real kernels: two ranks (gloo process group, both on the one
        GPU of the test box -- device = LOCAL_RANK modulo the device count) run Batch(nbls.simulate, queue)
        and a cost-split metrics-only sweep; every rank ends with the whole result, equal to what one
        process computes. '''
    from pysonic_amd import _native as N
    N.require_gpu()
    from pysonic_amd import (NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, Batch, getPointNeuron)
    script = os.path.join(tmp_path, 'worker_gpu.py')
    with open(script, 'w') as fh:
        fh.write(WORKER_GPU.format(root=ROOT))
    out = os.path.join(tmp_path, 'res')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(port), script, out]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron('RS'))
    queue = [[AcousticDrive(500e3, float(a)), PulsedProtocol(20e-3, 5e-3, 100., float(dc))]
             for a in (30e3, 100e3, 300e3, 600e3) for dc in (0.3, 1.0)][:7]
    single = Batch(nbls.simulate, queue).run(mpi=True)
    qm = np.stack([d['Qm'].values for d, _ in single])
    rows1 = nbls.runSonicBatch(500e3, 1., [(q[0], q[1]) for q in queue], traces=False)[1]
    r = [np.load(out + f'.rank{k}.npz') for k in range(2)]
    for k in range(2):
        # the same configurations on the same kernel, packed into other wavefronts: identical results
        np.testing.assert_array_equal(r[k]['qm'], qm)
        np.testing.assert_array_equal(r[k]['rows'][:, :11], rows1[:, :11])
    assert r[0]['share'][0] + r[1]['share'][0] == 7 and min(r[0]['share'][0], r[1]['share'][0]) >= 1
    assert sorted(list(r[0]['held']) + list(r[1]['held'])) == list(range(7))       # ungathered: a partition of the queue


WORKER_NCCL = textwrap.dedent('''
    import os, sys
    import numpy as np
    sys.path.insert(0, {root!r})
    from pysonic_amd.parallel import init_process_group, run_sharded, collective_device
    dist = init_process_group('nccl')          # before any GPU call of the process: RCCL, cuda:<LOCAL_RANK>
    assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1
    from pysonic_amd import (NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, Batch, getPointNeuron)
    from pysonic_amd import _native as N
    nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron('RS'))
    cfgs = [(AcousticDrive(500e3, float(a)), PulsedProtocol(20e-3, 5e-3, 100., float(dc)))
            for a in (30e3, 100e3, 300e3, 600e3) for dc in (0.3, 1.0)]
    rows_local = nbls.runSonicBatch(500e3, 1., cfgs, traces=False)[1]
    launch = lambda a, b: nbls.runSonicBatch(500e3, 1., cfgs[a:b], traces=False)[1]
    # the product's sharded sweep over the RCCL group: the gather buffers must sit on the GPU
    assert str(collective_device(dist)).startswith('cuda')
    rows = run_sharded(launch, len(cfgs), force_collective=True)
    assert np.array_equal(rows[:, :11], rows_local[:, :11], equal_nan=True)      # (NaN: first-spike time of a silent cell)
    # ... and dealt by estimated cost (what Batch.run and bench.py --scaling strong do)
    costs = NeuronalBilayerSonophore._queueCosts([([d, pp], {{}}) for d, pp in cfgs])
    rows_d = run_sharded(lambda idx: nbls.runSonicBatch(500e3, 1., [cfgs[i] for i in idx], traces=False)[1],
                         len(cfgs), costs=costs, force_collective=True, dealt=True)
    assert np.array_equal(rows_d[:, :11], rows_local[:, :11], equal_nan=True)
    # the lookup cells of config 3 through the same entry point
    f, A, Q = np.full(6, 500e3), np.array([0., 1e3, 50e3, 100e3, 300e3, 600e3]), np.full(6, -71.9e-5)
    mech = lambda a, b: nbls.runMechBatch(f[a:b], A[a:b], Q[a:b], [1.])[0][:, 0, :]
    eff = run_sharded(mech, 6, force_collective=True)
    assert np.array_equal(eff, mech(0, 6)) and np.all(np.isfinite(eff))
    out = Batch(nbls.simulate, [[d, pp] for d, pp in cfgs[:3]]).run(mpi=True)
    assert all(o is not None for o in out)
    thr = Batch(nbls.titrate, [[AcousticDrive(500e3), PulsedProtocol(20e-3, 5e-3)]]).run(mpi=True)
    assert 1e3 < thr[0] < 600e3, thr
    np.save(sys.argv[1], rows)
    dist.barrier()
    dist.destroy_process_group()
''')


@pytest.mark.gpu
def test_product_path_over_rccl(tmp_path):
    ''' The sharded entry points of the product (parallel.run_sharded with the sonic and the lookup kernels,
        Batch.run(mpi=True)) over an RCCL (`nccl`) process group, one rank on the one GPU of the test box: the
        group is created by parallel.init_process_group, the all-gather of the metric rows runs on device
        buffers (an nccl group cannot move host tensors), and the gathered rows equal the ungathered ones. '''
    from pysonic_amd import _native as N
    N.require_gpu()
    script = os.path.join(tmp_path, 'worker_nccl.py')
    with open(script, 'w') as fh:
        fh.write(WORKER_NCCL.format(root=ROOT))
    out = os.path.join(tmp_path, 'rows.npy')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               PYSONIC_AMD_TITRATIONS=os.path.join(tmp_path, 'titrations.log'))
    res = subprocess.run([sys.executable, script, out], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    assert np.load(out).shape == (8, 16)
