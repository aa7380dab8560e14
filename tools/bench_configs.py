''' Measurements of the BASELINE.json configurations other than the headline one (bench.py = config 2),
    on one GPU. Prints one JSON line per configuration; the committed copies live in profiles/.

      config 3  lookup generation: RS, a in {16, 32, 64} nm x 7 frequencies x 51 A x 158 Q (run_lookups.py grid)
      config 4  mixed sweep: {RS, FS, LTS, TC, RE, STN} x 10 000 (f, A, PRF, DC), sonic, spike metrics only
      config 5  full NICE integration: 256 RS configurations (16 A x 16 DC), f = 500 kHz, PRF = 1 kHz
      6         the same 256 configurations with method='hybrid'

    usage: python tools/bench_configs.py [3] [4] [5] [--tstim-full 1e-3]
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

    def one(a):
        return NeuronalBilayerSonophore(a, pn).computeLookup(freqs, amps, charges)
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


def config4(n_per_neuron=10000):
    freqs = [500e3]
    amps = np.logspace(np.log10(10e3), np.log10(600e3), 20)
    PRFs = np.logspace(1, 3, 10)
    DCs = np.linspace(0.05, 1.0, 10)
    reps = max(1, n_per_neuron // (len(freqs) * amps.size * PRFs.size * DCs.size))
    out = {'config': 4, 'workload': f'mixed sweep, sonic, metrics only: per neuron {reps} x '
           f'({len(freqs)} f x {amps.size} A x {PRFs.size} PRF x {DCs.size} DC), tstim=100 ms, '
           f'toffset=50 ms, a=32 nm', 'per_neuron': {}}
    tot_cfg, tot_ms = 0, 0.
    for name in ['RS', 'FS', 'LTS', 'TC', 'RE', 'STN']:
        pn = getPointNeuron(name)
        nbls = NeuronalBilayerSonophore(32e-9, pn)
        cfgs = [(AcousticDrive(f, float(a)), PulsedProtocol(100e-3, 50e-3, float(prf), float(dc)))
                for f in freqs for a in amps for prf in PRFs for dc in DCs] * reps
        t0 = time.perf_counter()
        batch = nbls._sonicBatch(cfgs, write_traces=False) if hasattr(nbls, '_sonicBatch') else None
        if batch is None:
            lkp = nbls.getLookup2D(freqs[0], 1.)
            tables = np.array([lkp[k] for k in ['V'] + pn.rates])
            model = N.SonicModel(name, pn.device_params(), tables, lkp.refs['A'], lkp.refs['Q'])
            batch = model.prepare(*nbls._packConfigs(cfgs), nbls.initialConditionsSonic(),
                                  N.default_opts(write_traces=0))
        prep = time.perf_counter() - t0
        ms = []
        for _ in range(2):
            batch.launch(); ms.append(batch.sync())
        _, met, st = batch.fetch(traces=False)
        out['per_neuron'][name] = {
            'configs': len(cfgs), 'kernel_ms': min(ms), 'prepare_s': prep,
            'configs_per_s': len(cfgs) / (min(ms) * 1e-3), 'bad_status': int(np.count_nonzero(st)),
            'mean_steps': float(met[:, N.M_NSTEPS].mean()), 'max_steps': float(met[:, N.M_NSTEPS].max()),
            'spiking_fraction': float(np.mean(met[:, N.M_NSPIKES] > 0))}
        tot_cfg += len(cfgs); tot_ms += min(ms)
    out['configs'] = tot_cfg
    out['kernel_ms_total'] = tot_ms
    out['configs_per_s'] = tot_cfg / (tot_ms * 1e-3)
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
        print(json.dumps(res), flush=True)
