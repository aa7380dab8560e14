''' Measurements of the BASELINE.json configurations other than the headline one (bench.py = config 2),
    on one GPU. Prints one JSON line per configuration; the committed copies live in profiles/.

      config 3  lookup generation: RS, a in {16, 32, 64} nm x 7 frequencies x 51 A x 158 Q (run_lookups.py grid)
      config 4  mixed sweep: {RS, FS, LTS, TC, RE, STN} x 10 000 (f, A, PRF, DC), sonic, spike metrics only
      config 5  full NICE integration: 256 RS configurations (16 A x 16 DC), f = 500 kHz, PRF = 1 kHz
      6         the same 256 configurations with method='hybrid'

    usage: python tools/bench_configs.py [3] [4] [5] [--tstim-full 1e-3]
    N GPUs (configs 3 and 4 shard over the ranks, metric rows / effective variables all-gathered over RCCL):
           python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                  --master-port P tools/bench_configs.py 3 4
'''
import sys, os, time, json, argparse
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import (NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron)
from pysonic_amd import _native as N


def config3():
    pn = getPointNeuron('RS')
    radii = [16e-9, 32e-9, 64e-9]
    freqs = np.array([20., 100., 500., 1e3, 2e3, 3e3, 4e3]) * 1e3
    amps = np.insert(np.logspace(np.log10(100.), np.log10(600e3), 50), 0, 0.)
    charges = np.arange(pn.Qbounds[0], pn.Qbounds[1] + 1e-5, 1e-5) if hasattr(pn, 'Qbounds') else \
        np.arange(-107e-5, 50e-5 + 1e-5, 1e-5)
    # one launch per radius (the sonophore parameters differ), issued from one host thread each:
    # mech_batch_run runs on a private stream, so the three kernels share the GPU instead of
    # running one after the other with a third of the SIMDs idle each
    from concurrent.futures import ThreadPoolExecutor
    wall0 = time.perf_counter()

    from pysonic_amd.parallel import run_sharded, _group
    world = _group(None)[2]

    def one(a):
        return NeuronalBilayerSonophore(a, pn).computeLookup(freqs, amps, charges)
    if world > 1:
        # one process per GPU: the cells of each radius are split over the ranks by estimated cost (one acoustic
        # period each: 1 / f) and the effective variables all-gathered; radii in sequence (one collective each)
        return config3_sharded(pn, radii, freqs, amps, charges, wall0)
    with ThreadPoolExecutor(len(radii)) as pool:
        lkps = list(pool.map(one, radii))
    ncell = sum(l.ncycles.size for l in lkps)
    kms = max(l.kernel_ms for l in lkps)
    ncyc = np.zeros(16, dtype=np.int64)
    for l in lkps:
        ncyc += np.bincount(l.ncycles.ravel(), minlength=16)[:16]
        assert np.all(np.isfinite(l['V']))
    wall = time.perf_counter() - wall0
    return {'config': 3, 'workload': f'BLS mechanical lookup generation, RS: {len(radii)} radii x '
            f'{freqs.size} f x {amps.size} A x {charges.size} Q, fs=1', 'cells': int(ncell),
            'kernel_ms_longest': kms, 'wall_s': wall, 'launches': 'one per radius, concurrent',
            'cells_per_s_wall': ncell / wall,
            'cycles_histogram': {str(i): int(c) for i, c in enumerate(ncyc) if c}}


def config3_sharded(pn, radii, freqs, amps, charges, wall0):
    from pysonic_amd.parallel import run_sharded, _group
    _, rank, world = _group(None)
    ncell, kms = 0, 0.
    ncyc = np.zeros(16, dtype=np.int64)
    for a in radii:
        nbls = NeuronalBilayerSonophore(a, pn)
        F, A, Q = [g.ravel() for g in np.meshgrid(freqs, amps, charges, indexing='ij')]
        ms_box = []

        def launch(idx):
            eff, ncy, status, ms = nbls.runMechBatch(F[idx], A[idx], Q[idx], [1.])
            ms_box.append(ms)
            return np.column_stack([eff[:, 0, :], ncy, status])
        rows = run_sharded(launch, F.size, costs=1. / F + 1e-6 * A / 600e3, dealt=True)
        assert np.all(np.isfinite(rows[:, 0])) and np.all(rows[:, -1].astype(int) & 2 == 0)
        ncell += F.size
        kms += ms_box[0]
        ncyc += np.bincount(rows[:, -2].astype(int), minlength=16)[:16]
    wall = time.perf_counter() - wall0
    return {'config': 3, 'workload': f'BLS mechanical lookup generation, RS: {len(radii)} radii x '
            f'{freqs.size} f x {amps.size} A x {charges.size} Q, fs=1', 'cells': int(ncell), 'ranks': world,
            'kernel_ms_sum_rank0': kms, 'wall_s': wall, 'launches': 'one per radius and rank, radii in sequence',
            'cells_per_s_wall': ncell / wall,
            'cycles_histogram': {str(i): int(c) for i, c in enumerate(ncyc) if c}}


def config4(n_per_neuron=10000):
    # the sweep of BASELINE config 4: 5 frequencies x 20 amplitudes x 10 PRFs x 10 duty cycles per neuron;
    # tables of the four frequencies without a packaged lookup are generated on the device first
    # (timed apart: they are cached afterwards). Under a process group (torchrun, one process per GPU)
    # every (neuron, frequency) block is split over the ranks by parallel.run_sharded and its spike
    # metric rows are all-gathered. kernel_ms = the longest of the neuron's concurrent launches.
    from pysonic_amd.parallel import run_sharded
    freqs = [100e3, 500e3, 1e6, 2e6, 4e6]
    amps = np.logspace(np.log10(10e3), np.log10(600e3), 20)
    PRFs = np.logspace(1, 3, 10)
    DCs = np.linspace(0.05, 1.0, 10)
    reps = max(1, n_per_neuron // (len(freqs) * amps.size * PRFs.size * DCs.size))
    out = {'config': 4, 'workload': f'mixed sweep, sonic, metrics only: per neuron {reps} x '
           f'({len(freqs)} f x {amps.size} A x {PRFs.size} PRF x {DCs.size} DC), tstim=100 ms, '
           f'toffset=50 ms, a=32 nm', 'per_neuron': {}}
    import torch.distributed as dist
    sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    names = ['RS', 'FS', 'LTS', 'TC', 'RE', 'STN']
    models = {}
    t0 = time.perf_counter()
    for name in names:
        models[name] = NeuronalBilayerSonophore(32e-9, getPointNeuron(name))
        for f in freqs:
            models[name]._sonicModel(f, 1.)         # lookup (generated on the device if needed) + upload
    t_tables = time.perf_counter() - t0

    # the queue of every neuron, built before any clock starts: 60 000 Drive / Protocol objects with their bounds
    # checks are ~1 s of Python (the reference's object model, as a caller of either implementation pays it)
    t0 = time.perf_counter()
    queues = {}
    for name in names:
        cfgs = [(f, AcousticDrive(f, float(a)), PulsedProtocol(100e-3, 50e-3, float(prf), float(dc)))
                for f in freqs for a in amps for prf in PRFs for dc in DCs] * reps
        queues[name] = (cfgs, NeuronalBilayerSonophore._queueCosts([([d, pp], {}) for _, d, pp in cfgs]))
    t_queue = time.perf_counter() - t0

    def one(name):
        # the 10 000 configurations of the neuron as ONE sweep: its five frequency groups are launched
        # together (nbls.runSonicBatches: one stream each), so they cost the longest group, not the sum
        nbls = models[name]
        cfgs, costs = queues[name]
        ms_box, span_box = [], []

        def launch(items):
            part = [cfgs[i] for i in items]
            fs_here = sorted({f for f, _, _ in part})
            idx = {f: [i for i, c in enumerate(part) if c[0] == f] for f in fs_here}
            t_in = time.perf_counter()
            res = nbls.runSonicBatches([(f, 1., [(part[i][1], part[i][2]) for i in idx[f]], None) for f in fs_here],
                                       traces=False)
            span_box.append((t_in, time.perf_counter()))
            ms_box.append(max(r[3] for r in res))
            rows = np.empty((len(part), N.SONIC_NMETRICS + 1))
            for f, (_, met, st, _) in zip(fs_here, res):
                rows[idx[f], :-1] = met
                rows[idx[f], -1] = st
            return rows
        launch(range(64))                               # warm-up (module load, allocations)
        ms_box.clear(); span_box.clear()
        t0 = time.perf_counter()
        rows = run_sharded(launch, len(cfgs), costs=costs, dealt=True)
        wall = time.perf_counter() - t0
        kms = ms_box[0]                                 # the longest of the concurrent launches (HIP events)
        steps = rows[:, N.M_NSTEPS]
        return name, {
            'configs': len(cfgs), 'kernel_ms': kms, 'wall_s': wall, 'span': span_box[0],
            'bad_status': int(np.count_nonzero(rows[:, -1])), 'mean_steps': float(steps.mean()),
            'max_steps': float(steps.max()), 'spiking_fraction': float(np.mean(rows[:, N.M_NSPIKES] > 0))}

    t0 = time.perf_counter()
    if sharded:
        # the ranks meet in one collective per neuron: neurons one after the other
        results = [one(name) for name in names]
        out['launches'] = 'per neuron: five frequency groups concurrent; neurons in sequence (process group)'
    else:
        # one process: the six neurons from six host threads, thirty launches sharing the GPU
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(len(names)) as pool:
            results = list(pool.map(one, names))
        out['launches'] = 'six neurons x five frequency groups, all concurrent (one stream each)'
    wall_all = time.perf_counter() - t0
    # the interval the device side was at work: from the first thread entering its prepare + launch to the last
    # one leaving its fetch (host clock; HIP-event durations of launches issued from different threads at
    # different times do not add up to an interval)
    spans = [r.pop('span') for _, r in results]
    busy = max(b for _, b in spans) - min(a for a, _ in spans) if not sharded else sum(b - a for a, b in spans)
    out['per_neuron'] = dict(results)
    tot_cfg = sum(r['configs'] for _, r in results)
    out['configs'] = tot_cfg
    out['kernel_ms_longest'] = max(r['kernel_ms'] for _, r in results)     # a lower bound of the busy time
    out['launch_to_fetch_span_s'] = busy
    out['wall_s_total'] = wall_all
    out['lookup_generation_and_upload_s'] = t_tables
    out['queue_objects_s'] = t_queue
    out['configs_per_s'] = tot_cfg / busy                                   # prepare + kernels + fetch, measured
    out['configs_per_s_wall'] = tot_cfg / wall_all
    if sharded:
        out['ranks'] = dist.get_world_size()
    return out


def config5(tstim):
    nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron('RS'))
    amps = np.logspace(np.log10(10e3), np.log10(600e3), 16)
    DCs = np.linspace(0.1, 1.0, 16)
    cfgs = [(AcousticDrive(500e3, float(a)), PulsedProtocol(tstim, tstim / 4, 1e3, float(dc)), 1.)
            for a in amps for dc in DCs]
    t0 = time.perf_counter()
    frames, status, ms = nbls.runFullBatch(cfgs)
    wall = time.perf_counter() - t0
    rows = int(sum(len(f) for f in frames))
    return {'config': 5, 'workload': f'full NICE, RS, 256 configurations (16 A x 16 DC), f=500 kHz, '
            f'PRF=1 kHz, tstim={tstim * 1e3:g} ms + {tstim * 0.25e3:g} ms offset, traces resampled at 10 ns',
            'configs': len(cfgs), 'kernel_ms': ms, 'wall_s': wall, 'rows': rows,
            'simulated_ms_per_config': tstim * 1.25e3,
            'kernel_s_per_simulated_ms': ms * 1e-3 / (tstim * 1.25e3),
            'bad_status': int(np.count_nonzero(status))}


def config5_hybrid(tstim):
    nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron('RS'))
    amps = np.logspace(np.log10(10e3), np.log10(600e3), 16)
    DCs = np.linspace(0.1, 1.0, 16)
    cfgs = [(AcousticDrive(500e3, float(a)), PulsedProtocol(tstim, tstim / 4, 1e3, float(dc)), 1.)
            for a in amps for dc in DCs]
    t0 = time.perf_counter()
    frames, status, ncycles, ms = nbls.runHybridBatch(cfgs)
    wall = time.perf_counter() - t0
    return {'config': '5-hybrid', 'workload': f'hybrid NICE (method=hybrid), RS, 256 configurations '
            f'(16 A x 16 DC), f=500 kHz, PRF=1 kHz, tstim={tstim * 1e3:g} ms + {tstim * 0.25e3:g} ms '
            'offset, traces resampled at 10 ns', 'configs': len(cfgs), 'kernel_ms': ms, 'wall_s': wall,
            'rows': int(sum(len(f) for f in frames)), 'simulated_ms_per_config': tstim * 1.25e3,
            'kernel_s_per_simulated_ms': ms * 1e-3 / (tstim * 1.25e3),
            'dense_periods_mean': float(ncycles.mean()), 'dense_periods_max': int(ncycles.max()),
            'bad_status': int(np.count_nonzero(status))}


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('which', nargs='*', type=int, default=[3, 4, 5, 6])
    ap.add_argument('--tstim-full', type=float, default=1e-3)
    ap.add_argument('--n-per-neuron', type=int, default=10000)
    args = ap.parse_args()
    # under torchrun (one process per GPU): join the RCCL group before anything touches the GPU
    from pysonic_amd.parallel import init_process_group
    _dist = init_process_group()
    _rank = _dist.get_rank() if _dist is not None else 0
    N.require_gpu()
    # long launches (config 5 and its hybrid variant run for minutes): a heartbeat on stderr once a
    # minute, so that a watchdog on silent runs does not take the process for hung
    import threading

    def heartbeat():
        t0 = time.perf_counter()
        while True:
            time.sleep(60)
            print(f'[bench_configs] running, {time.perf_counter() - t0:.0f} s', file=sys.stderr, flush=True)
    threading.Thread(target=heartbeat, daemon=True).start()
    for w in args.which:
        res = {3: config3, 4: lambda: config4(args.n_per_neuron), 5: lambda: config5(args.tstim_full),
               6: lambda: config5_hybrid(args.tstim_full)}[w]()
        if _rank == 0:
            print(json.dumps(res), flush=True)
    if _dist is not None:
        _dist.barrier()
        _dist.destroy_process_group()
