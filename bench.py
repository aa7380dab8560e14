#!/usr/bin/env python3
# -*- coding: utf-8 -*-
''' bench.py -- headline benchmark of the hot path (BASELINE.json metric):
        stimulus-configs/sec (sonic, CorticalRS, 100 ms)

    One "step" = one pass of the batched SONIC integration over one activation-map batch:
    BASELINE config 2 = CorticalRS, a = 32 nm, f = 500 kHz, 64 x 64 (A x DC) grid,
    A = logspace(10 kPa, 600 kPa), DC = linspace(0.05, 1), PRF = 100 Hz, tstim = 100 ms,
    toffset = 0 (plt/actmap.py:29-34) = 4096 configurations per GPU, full traces written to HBM.

    N = 1 (default): the 4096-configuration map itself (`"scaling": "weak"`).
    N > 1 defaults to --scaling strong: ONE fixed sweep of 65 536 configurations (256 amplitudes x 256
        duty cycles, the same protocol) is dealt over the N ranks by estimated cost (pysonic_amd.parallel.dealt_shards),
        each rank integrates its block, the metric rows are all-gathered over RCCL inside every timed step.
        Its N = 1 point is the `saturated` figure of the N = 1 line; every N > 1 line carries it too
        (`strong_n1`: rank 0 integrates the whole sweep alone after the timed region), so the line states
        its own speed-up. One 4096-cell map cannot scale -- it is one latency-bound launch that lasts as
        long as its slowest configuration -- which is why the map is not what N > 1 splits.
    --scaling weak with N GPUs: every rank integrates its own 4096-configuration map (amplitude grid
        interleaved across ranks: the global sweep is 64 N x 64), no data-path collective; the metric rows
        are all-gathered inside every timed step. Trivially N x: kept for comparison only.

    Inputs (segment schedules, projected lookups) are resident in HBM before the timed region;
    the timed region is K x (kernel launch [+ metric all-gather]) between barrier+synchronize.

    JSON extras:
      roofline     HBM roofline of the integration kernel: algorithmic bytes per launch (output
                   rows x (n_states + 4) x 8 B + inputs) / mean kernel duration (HIP events on the
                   kernel's own stream, measured in this run) vs 8 TB/s peak. `traffic` is NOT measured
                   in this run: it is the constant of the rocprofv3 PMC passes that profiles/CURRENT.json
                   names, and null when that file was made from other kernel sources than this build's.
      saturated    (N = 1, weak) the same kernel on 16 maps at once (65 536 configurations): every SIMD
                   busy, GB/s and fraction of the HBM roofline.
      valu         (N = 1, weak) FP64 VALU instruction rate of the headline launch against the issue
                   peak of gfx950, from the SQ counters of the profile named in `source`.
      end_to_end   (N = 1, weak) configurations per second through the public API,
                   Batch(nbls.simulate, queue).run(mpi=True): prepare + launch + fetch + one
                   TimeSeries per configuration, for the same 4096-configuration map.
      cpu_baseline the oracle (scipy LSODA + C right-hand side, oracle/) timed on this host's cores
                   on a bounded stratified sample of the same 4096 configurations (rank 0, N=1).
'''
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec
# FP64 VALU issue peak: 256 CUs x 4 SIMDs x one wave64 instruction per 4 cycles x 2.4 GHz (= 78.6 TFLOP/s
# of FMA, the vector FP64 figure of the same guide)
VALU_PEAK_WAVE_INSTS = 256 * 4 * 2.4e9 / 4
N_AMPS, N_DCS = 64, 64
TSTIM, TOFFSET, PRF, FREQ, RADIUS = 100e-3, 0., 100., 500e3, 32e-9


def activation_map(rank, world, n_amps=N_AMPS, n_dcs=N_DCS):
    ''' (A, DC) list of this rank: amplitude grid of n_amps * world points interleaved over ranks '''
    amps = np.logspace(np.log10(10e3), np.log10(600e3), n_amps * world)[rank::world]
    DCs = np.linspace(0.05, 1.0, n_dcs)
    return [(float(a), float(dc)) for a in amps for dc in DCs]


def _oracle_worker(args):
    ''' cpu_baseline leg: one configuration through the oracle (checker used as CPU baseline) '''
    from oracle import oracle as O
    name, amp, dc = args
    global _ORC_TABLES
    if '_ORC_TABLES' not in globals():
        d = np.load(os.path.join(ROOT, 'pysonic_amd', 'lookups', f'tables_{name}_32nm_500kHz.npz'))
        keys = [str(k) for k in d['keys']]
        _ORC_TABLES = (d['A'], d['Q'], np.array([d[f'tab_{k}'] for k in keys]))
    A, Q, tables = _ORC_TABLES
    ev, tstop = O.pulsed_events(TSTIM, TOFFSET, PRF, dc)
    out = O.sim_sonic(name, A, Q, tables, amp, ev, tstop)
    return float(out['Qm'][-1])


def cpu_baseline(cfgs, budget_s=30.0):
    ''' Oracle ("port") throughput on the host cores over a stratified sample of the workload:
        every second configuration of the map (2048), stopped early at `budget_s`. The worker pool is
        started and warmed (library loaded, tables read) before the clock. '''
    import multiprocessing as mp
    from oracle import oracle as O
    O.build()
    cores = max(1, min(os.cpu_count() or 1, 32))
    sample = [cfgs[i] for i in range(0, len(cfgs), 2)]
    # interleave so that an early stop still covers all amplitudes
    order = np.argsort([(i * 37) % len(sample) for i in range(len(sample))], kind='stable')
    sample = [sample[i] for i in order]
    done = 0
    with mp.get_context('fork').Pool(cores) as pool:
        pool.map(_oracle_worker, [('RS', 20e3, 0.5)] * cores)            # warm-up, not timed
        t0 = time.perf_counter()
        chunk = cores * 4
        for i in range(0, len(sample), chunk):
            pool.map(_oracle_worker, [('RS', a, dc) for a, dc in sample[i:i + chunk]])
            done += len(sample[i:i + chunk])
            if time.perf_counter() - t0 > budget_s:
                break
        el = time.perf_counter() - t0
    return {'value': done / el, 'unit': 'configs/s', 'cores': cores, 'kind': 'port',
            'sample': f'{done} of the 4096 activation-map configurations (stratified over the A x DC '
                      f'grid), oracle = scipy odeint (LSODA, default tolerances) + C right-hand side, '
                      f'{cores} worker processes started and warmed before the clock, {el:.1f} s wall'}


def step_limits(metrics, N):
    ''' accepted steps sized by the node predictor / by the error controller, rejections by the error estimate /
        for ending too far past a node, cells of the charge grid crossed (the floor of the step count) '''
    def one(m):
        nsteps, nrej, ncap, nover, ncross = (float(m[N.M_NSTEPS]), float(m[N.M_NREJ]), float(m[N.M_NCAPPED]),
                                             float(m[N.M_NREJ_NODE]), float(m[N.M_NCROSS]))
        return {'attempts': nsteps, 'accepted_node_capped': ncap, 'accepted_error_controlled': nsteps - nrej - ncap,
                'rejected_by_error': nrej - nover, 'rejected_past_node': nover, 'cells_crossed': ncross}
    return {'costliest_configuration': one(metrics[int(np.argmax(metrics[:, N.M_NSTEPS]))]),
            'whole_map': one(metrics.sum(axis=0))}


def current_profiles():
    ''' profiles/CURRENT.json names the counter summaries of the benchmarked kernel and the digest of the
        native sources they were measured on (tools/profile_summary.py writes it). A summary of another
        build does not describe this kernel: (None, reason) then. '''
    path = os.path.join(ROOT, 'profiles', 'CURRENT.json')
    try:
        with open(path) as fh:
            cur = json.load(fh)
    except (OSError, ValueError):
        return None, 'profiles/CURRENT.json missing'
    from pysonic_amd.build import source_hash
    if cur.get('source_hash') != source_hash():
        return None, (f'profiles/CURRENT.json was made from other kernel sources ({str(cur.get("source_hash"))[:12]}) '
                      f'than this build ({source_hash()[:12]}): counters not quoted')
    return cur, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--scaling', choices=('weak', 'strong'), default=None,
                    help='default: weak (the 4096-cell map) with one GPU, strong (one 65 536-cell sweep split '
                         'over the ranks) with more')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip saturated / end_to_end (profiling runs)')
    ap.add_argument('--force-collective', action='store_true',
                    help='run the RCCL gather of the metric rows even with one rank (test hook)')
    args = ap.parse_args()
    if args.scaling is None:
        args.scaling = 'weak' if args.gpus == 1 else 'strong'
    # stdout carries ONE line, the JSON of rank 0: whatever libraries write to file descriptor 1 on the way
    # (RCCL prints its version banner there when the first communicator is created) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch multi-GPU runs with python -m torch.distributed.run '
                             '--nproc-per-node N bench.py --gpus N ...')
        raise SystemExit(f'--gpus {args.gpus} does not match WORLD_SIZE {world}')

    # CPU baseline first: its worker pool is forked before this process touches the GPU
    baseline = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        baseline = cpu_baseline(activation_map(0, 1))

    # torch only where a process group is needed (RCCL): the single-GPU path is the library and numpy
    dist = None
    use_dist = world > 1 or args.force_collective
    if use_dist:
        import torch
        from pysonic_amd.parallel import init_process_group
        os.environ.setdefault('MASTER_PORT', '29531')
        dist = init_process_group('nccl')         # cuda:<LOCAL_RANK>, before the first GPU call

    import __graft_entry__ as entry
    if not os.path.isfile(os.path.join(ROOT, 'pysonic_amd', '_lib', 'libpysonic_amd.so')):
        entry.build()
    from pysonic_amd import _native as N
    from pysonic_amd import (NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, Batch,
                             getPointNeuron)
    from pysonic_amd.parallel import dealt_shards
    N.require_gpu()

    pneuron = getPointNeuron('RS')
    nbls = NeuronalBilayerSonophore(RADIUS, pneuron)
    nbls.device = local_rank
    lkp = nbls.getLookup2D(FREQ, 1.)
    tables = np.array([lkp[k] for k in ['V'] + pneuron.rates])
    model = N.SonicModel('RS', pneuron.device_params(), tables, lkp.refs['A'], lkp.refs['Q'],
                         device=local_rank)

    def make_batch(cfgs, traces=True):
        configs = [(AcousticDrive(FREQ, a), PulsedProtocol(TSTIM, TOFFSET, PRF, dc)) for a, dc in cfgs]
        return model.prepare(*nbls._packConfigs(configs), nbls.initialConditionsSonic(),
                             N.default_opts(write_traces=int(traces)))

    if args.scaling == 'weak':
        cfgs = activation_map(rank, world)
        n_global = world * len(cfgs)
    else:
        sweep = activation_map(0, 1, 256, 256)                 # one fixed 65 536-configuration sweep
        costs = NeuronalBilayerSonophore._queueCosts(
            [([AcousticDrive(FREQ, a), PulsedProtocol(TSTIM, TOFFSET, PRF, dc)], {}) for a, dc in sweep])
        # dealt over the ranks in order of falling estimated cost (not cut into blocks of equal estimated cost: the
        # estimate does not know where the neuron starts to fire, and a block of low amplitudes is cheap -- measured
        # on this sweep: 1.49 x the mean for the costliest of 8 blocks, 1.004 x for the costliest of 8 dealt shards)
        cfgs = [sweep[i] for i in dealt_shards(costs, world)[rank]]
        n_global = len(sweep)
    batch = make_batch(cfgs)
    opts = batch.opts
    n_cfg = batch.n_cfg
    ncol = model.ncol

    # metric rows as a torch tensor over the library's HBM buffer (no copy), for the RCCL gather
    gather = None
    if use_dist:
        _, mptr, _ = batch.device_ptrs()

        class _Dev:
            __cuda_array_interface__ = {'shape': (n_cfg, N.SONIC_NMETRICS), 'typestr': '<f8',
                                        'data': (mptr, False), 'version': 2}
        metrics_t = torch.as_tensor(_Dev(), device=torch.device('cuda', local_rank))
        if args.scaling == 'weak':
            gather_out = torch.empty((world * n_cfg, N.SONIC_NMETRICS), dtype=torch.float64,
                                     device=metrics_t.device)
            gather = lambda: dist.all_gather_into_tensor(gather_out, metrics_t)      # noqa: E731
        else:
            sizes = torch.tensor([n_cfg], device=metrics_t.device)
            all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
            dist.all_gather(all_sizes, sizes)
            nmax = int(max(int(s.item()) for s in all_sizes))
            padded = torch.zeros((nmax, N.SONIC_NMETRICS), dtype=torch.float64, device=metrics_t.device)
            gather_out = torch.empty((world * nmax, N.SONIC_NMETRICS), dtype=torch.float64,
                                     device=metrics_t.device)

            def gather():
                padded[:n_cfg].copy_(metrics_t)
                dist.all_gather_into_tensor(gather_out, padded)

    def step():
        batch.launch()
        ms = batch.sync()                      # kernel done (its own stream) before the gather
        if gather is not None:
            gather()
        return ms

    def fence():
        # every step ends in batch.sync(): the kernel's stream is idle here; with a process group, the ranks
        # meet and torch's stream (the gather) drains too
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    kernel_ms = [step() for _ in range(args.steps)]
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=torch.device('cuda', local_rank))
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    tr, metrics, status = batch.fetch(traces=False)
    if np.any(status != 0):
        raise SystemExit(f'rank {rank}: {np.count_nonzero(status)} configurations failed')
    if use_dist and args.scaling == 'weak':
        g = gather_out.cpu().numpy()
        # the gathered block of this rank equals its own metric rows (col 11 = diagnostics, NaN-free)
        assert np.array_equal(g[rank * n_cfg:(rank + 1) * n_cfg], metrics, equal_nan=True)

    def bytes_per_launch(b):
        out_bytes = float(b.total_rows) * ncol * 8
        in_bytes = float(b.n_cfg) * (64 + 16 * 2 * 100 * 0.525)   # descriptor + mean event bytes
        return out_bytes + in_bytes

    if rank == 0:
        kms = float(np.mean(kernel_ms))
        alg_bytes = bytes_per_launch(batch)
        achieved = alg_bytes / (kms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        cur, why = current_profiles()
        if cur is None:
            traffic_src = why
        elif args.scaling == 'weak' and world == 1:
            with open(os.path.join(ROOT, 'profiles', cur['hbm_traffic'])) as fh:
                traffic = json.load(fh).get('hbm_bytes_per_launch')
            traffic_src = (f'profiles/{cur["hbm_traffic"]}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes '
                           'of this command on this build (not measured in this run)')
        res = {
            'metric': 'stimulus-configs/sec (sonic, RS, 100 ms)',
            'value': n_global * args.steps / elapsed,
            'unit': 'configs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': ('activation map 64x64 (A x DC) per GPU: CorticalRS sonic, '
                                    'a=32nm f=500kHz PRF=100Hz tstim=100ms toffset=0, traces '
                                    'written (BASELINE config 2)') if args.scaling == 'weak' else
                                   ('one 256x256 (A x DC) sweep of the same protocol, 65 536 configurations '
                                    'dealt over the GPUs by estimated cost, traces written'),
                       'configs_per_gpu': n_cfg, 'rows_per_gpu': int(batch.total_rows),
                       'integrator': 'Rosenbrock ROS4 (Shampine) adaptive, order 4(3)', 'rtol': batch.rtol, 'atol': batch.atol,
                       'parallelism': f'shard{world}' if world > 1 else 'single'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'traffic_source': traffic_src,
                         'kernel': 'sonic_integrate_quad_kernel<false> (level tables in L2)',
                         'kernel_ms': kms, 'algorithmic_bytes_per_launch': alg_bytes,
                         'mean_steps_per_config': float(metrics[:, 0].mean()),
                         'max_steps_per_config': float(metrics[:, 0].max()),
                         # the launch lasts as long as its slowest configuration: a sequential
                         # chain of max_steps Rosenbrock steps
                         'critical_path_us_per_step': kms * 1e3 / float(metrics[:, 0].max()),
                         # what set the steps (metric columns NCAPPED, NREJ_NODE, NCROSS): of the costliest
                         # configuration and summed over the map
                         'step_limits': step_limits(metrics, N)},
        }
        if world == 1 and args.scaling == 'weak' and not args.no_extras:
            # FP64 VALU rate of the headline launch, from the SQ counters of the committed profile
            vfile = os.path.join(ROOT, 'profiles', cur['sq_counters']) if cur else None
            if vfile:
                with open(vfile) as fh:
                    sq = json.load(fh)
                rate = sq['valu_wave_insts_per_launch'] / (kms * 1e-3)
                res['valu'] = {'wave_insts_per_launch': sq['valu_wave_insts_per_launch'],
                               'rate_wave_insts_per_s': rate, 'peak_wave_insts_per_s': VALU_PEAK_WAVE_INSTS,
                               'frac': rate / VALU_PEAK_WAVE_INSTS,
                               'source': f'profiles/{os.path.basename(vfile)} (SQ_INSTS_VALU of this command under '
                                         'rocprofv3 --pmc; the rate uses the kernel time of THIS run)'}
            # the same kernel with every SIMD busy: 16 maps in one launch
            batch.close()
            big = make_batch(activation_map(0, 1, 256, 256))
            big.launch(); big.sync()
            ms_big = []
            for _ in range(3):
                big.launch(); ms_big.append(big.sync())
            _, _, st_big = big.fetch(traces=False)
            bb = bytes_per_launch(big)
            gbs = bb / (np.mean(ms_big) * 1e-3) / 1e9
            res['saturated'] = {'configs': big.n_cfg, 'kernel_ms': float(np.mean(ms_big)),
                                'value': big.n_cfg / (np.mean(ms_big) * 1e-3), 'unit': 'configs/s',
                                'achieved': gbs, 'frac': gbs / HBM_PEAK_GBS, 'bad_status': int(np.count_nonzero(st_big)),
                                'workload': 'the 256 x 256 (A x DC) sweep of --scaling strong in one launch on one GPU '
                                            '(65 536 configurations = 16 maps, traces written)'}
            big.close()
            # the public API, end to end, on the headline map
            queue = [[AcousticDrive(FREQ, a), PulsedProtocol(TSTIM, TOFFSET, PRF, dc), 1., 'sonic', None]
                     for a, dc in cfgs]
            import logging
            from pysonic_amd.utils import logger
            logger.setLevel(logging.WARNING)
            Batch(nbls.simulate, queue[:64]).run(mpi=True, loglevel=logging.WARNING)      # warm-up
            # two consecutive sweeps: the first maps the page-locked block the traces land in (0.66 GB), the
            # second finds it in the pool -- the state of a session that sweeps repeatedly. Both are reported.
            els = []
            for _ in range(2):
                t0 = time.perf_counter()
                outputs = Batch(nbls.simulate, queue).run(mpi=True, loglevel=logging.WARNING)
                els.append(time.perf_counter() - t0)
                assert len(outputs) == len(queue) and outputs[-1][0].shape[0] == 2005
                del outputs
            el = els[1]
            # the stages of the same sweep, timed one by one on the calls Batch.run makes
            import time as _t
            cfg_objs = [(q[0], q[1]) for q in queue]
            t0 = _t.perf_counter(); calls = [Batch.resolve(q) for q in queue]
            resolved = nbls._resolveSimulateCalls(calls); t1 = _t.perf_counter()
            packed = nbls._packConfigs(cfg_objs); t2 = _t.perf_counter()
            eb = model.prepare(*packed, nbls.initialConditionsSonic()); t3 = _t.perf_counter()
            eb.launch(to_host=True); kms_e = eb.sync(); t4 = _t.perf_counter()
            eb.fetch(traces=False); blk = eb.host_traces; eb.close(); t5 = _t.perf_counter()
            del blk, resolved, calls
            res['end_to_end'] = {'value': len(queue) / el, 'unit': 'configs/s', 'wall_s': el,
                                 'first_call_wall_s': els[0], 'first_call_value': len(queue) / els[0],
                                 'stages_ms': {'resolve_and_check_calls': (t1 - t0) * 1e3, 'pack_events': (t2 - t1) * 1e3,
                                               'prepare_schedule_upload': (t3 - t2) * 1e3,
                                               'kernel_then_copy_to_host': (t4 - t3) * 1e3, 'kernel': kms_e,
                                               'metrics_and_release': (t5 - t4) * 1e3},
                                 'path': 'Batch(nbls.simulate, queue).run(mpi=True): host schedule + upload, '
                                         'kernel, the traces copied to a page-locked block behind the kernel, a result '
                                         'sequence whose (TimeSeries, meta) pairs are views of that block built on '
                                         'access; second of two consecutive sweeps (the first, which also maps the '
                                         'page-locked block, is first_call_*); stages_ms: the same calls timed one by one'}
        if world > 1 and args.scaling == 'strong':
            # the N = 1 point of this curve, measured here: rank 0 integrates the whole sweep alone (the other
            # ranks wait at the closing barrier)
            batch.close()
            whole = make_batch(sweep)
            whole.launch(); whole.sync()
            ms1 = []
            for _ in range(3):
                whole.launch(); ms1.append(whole.sync())
            _, _, st1 = whole.fetch(traces=False)
            v1 = whole.n_cfg / (np.mean(ms1) * 1e-3)
            res['strong_n1'] = {'value': v1, 'unit': 'configs/s', 'kernel_ms': float(np.mean(ms1)),
                                'bad_status': int(np.count_nonzero(st1)),
                                'speedup': res['value'] / v1, 'efficiency': res['value'] / v1 / world,
                                'what': 'the same 65 536-configuration sweep in one launch on rank 0 alone, same run '
                                        '(kernel time; no gather needed with one rank)'}
            whole.close()
        if baseline is not None:
            res['cpu_baseline'] = baseline
        # the reference itself on the build container's cores: a committed fixture, not measured in this run
        try:
            with open(os.path.join(ROOT, 'tests', 'golden', 'reference_timing.json')) as fh:
                rt = json.load(fh)
            res['reference_timing'] = {
                'source': 'tests/golden/reference_timing.json (tests/golden/make_reference_timing.py)',
                'what': 'PySONIC Batch(nbls.simulate, queue).run(mpi=True) on a 64-cell slice of this map',
                'value': rt['config2_slice']['configs_per_s'], 'unit': 'configs/s', 'cores': rt['cores'],
                'cpu': rt['cpu']}
        except (OSError, KeyError, ValueError):
            pass
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + '\n').encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
