''' Detailed model of LTS / RE / TC / STN: the row-cooperative kernel (one configuration per 16 lanes, 8(5,3) pair,
    csrc/full_row.hpp) against the lane kernel (one per lane, 5(4) pair) on a 256-configuration batch (16 amplitudes
    10 - 600 kPa x 16 duty cycles, four pulses): kernel ms, steps, microseconds per step of the slowest configuration,
    and the distance between the two kernels' rows relative to each variable's range.

    --hybrid: the same comparison for method='hybrid' (csrc/hybrid_row.hpp against the lane kernel), with the number
    of dense periods of each.

    usage (GPU box): python tools/row_probe.py [--neurons TC,LTS] [--tstim 2e-4] [--n 256] [--hybrid]
'''
import os
import sys
import json
import argparse
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron  # noqa: E402
from pysonic_amd import _native as N  # noqa: E402

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--neurons', default='TC,LTS,RE,STN')
    ap.add_argument('--tstim', type=float, default=2e-4)
    ap.add_argument('--n', type=int, default=256)
    ap.add_argument('--amax', type=float, default=600e3)
    ap.add_argument('--hybrid', action='store_true')
    args = ap.parse_args()
    N.require_gpu()
    na = int(round(np.sqrt(args.n)))
    for name in args.neurons.split(','):
        nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron(name))
        amps = np.logspace(np.log10(10e3), np.log10(args.amax), na)
        DCs = np.linspace(0.1, 1.0, args.n // na)
        cfgs = [(AcousticDrive(500e3, float(a)), PulsedProtocol(args.tstim, args.tstim / 4, 4. / args.tstim, float(dc)), 1.)
                for a in amps for dc in DCs]
        if args.hybrid:
            pn = nbls.pneuron
            A_, tstop_, _, ev_t, ev_x, ev_off = nbls._packConfigs([(d, pp) for d, pp, _ in cfgs])
            n = len(cfgs)
            out = {}
            for kernel, label in ((2, 'row'), (1, 'lane')):
                out[label] = N.hybrid_batch_run(name, pn.device_params(), nbls.device_params(), [500e3] * n, A_, [1.] * n,
                                                tstop_, ev_t, ev_x, ev_off, nbls.initialConditionsSonic(),
                                                N.full_default_opts(kernel=kernel))
            (tr, ro, sa, nsa, nca, msa), (ref, _, sb, nsb, ncb, msb) = out['row'], out['lane']
            worst = np.zeros(tr.shape[1])
            for i in range(n):
                if sa[i] or sb[i]:
                    continue
                a, b = tr[ro[i]:ro[i + 1]], ref[ro[i]:ro[i + 1]]
                rng = np.maximum(np.ptp(b, axis=0), 1e-300)
                worst = np.maximum(worst, np.sqrt(np.mean((a - b)**2, axis=0)) / rng)
            print(json.dumps({'neuron': name, 'method': 'hybrid', 'configs': n, 'tstim_us': args.tstim * 1e6,
                              'row_kernel_ms': msa, 'lane_kernel_ms': msb, 'speedup': msb / msa,
                              'steps_row': int(nsa.sum()), 'steps_lane': int(nsb.sum()),
                              'dense_periods_row': int(nca.sum()), 'dense_periods_lane': int(ncb.sum()),
                              'configs_with_other_period_count': int(np.count_nonzero(nca != ncb)),
                              'status_row': {int(k): int(v) for k, v in zip(*np.unique(sa, return_counts=True))},
                              'status_lane': {int(k): int(v) for k, v in zip(*np.unique(sb, return_counts=True))},
                              'worst_rms_over_range': [float(f'{v:.2e}') for v in worst[2:]]}), flush=True)
            continue
        res = {}
        for kernel, label in ((0, 'row'), (1, 'lane')):
            frames, status, ms = nbls.runFullBatch(cfgs, opts={'kernel': kernel})
            res[label] = (frames, status, ms)
        # which configurations the explicit row kernel gives up as stiff (they went to the lane kernel above)
        A_, tstop_, _, ev_t, ev_x, ev_off = nbls._packConfigs([(d, pp) for d, pp, _ in cfgs])
        _, _, st0, ns0, ms0 = N.full_batch_run(name, nbls.pneuron.device_params(), nbls.device_params(), [500e3] * len(cfgs),
                                               A_, [1.] * len(cfgs), tstop_, ev_t, ev_x, ev_off,
                                               nbls.initialConditionsSonic(), N.full_default_opts(kernel=2, stiff=0))
        stiff_amps = sorted({round(float(d.A) * 1e-3, 1) for (d, _, _), s_ in zip(cfgs, st0) if s_ & 64})
        cols = list(res['row'][0][0].columns)
        worst = {}
        for fa, fb, sa, sb in zip(res['row'][0], res['lane'][0], res['row'][1], res['lane'][1]):
            if sa or sb:
                continue
            for c in cols[2:]:
                x, y = fa[c].values, fb[c].values
                d = float(np.sqrt(np.mean((x - y)**2)) / (np.ptp(y) or 1.))
                worst[c] = max(worst.get(c, 0.), d)
        print(json.dumps({'neuron': name, 'configs': len(cfgs), 'tstim_us': args.tstim * 1e6,
                          'row_kernel_ms': res['row'][2], 'lane_kernel_ms': res['lane'][2],
                          'speedup': res['lane'][2] / res['row'][2],
                          'row_kernel_alone_ms': ms0, 'given_up_as_stiff': int(np.count_nonzero(st0 & 64)),
                          'stiff_amplitudes_kPa': stiff_amps,
                          'max_steps_row_alone': int(ns0[(st0 & 64) == 0].max()) if np.any((st0 & 64) == 0) else None,
                          'status_row': {int(k): int(v) for k, v in zip(*np.unique(res['row'][1], return_counts=True))},
                          'status_lane': {int(k): int(v) for k, v in zip(*np.unique(res['lane'][1], return_counts=True))},
                          'worst_rms_over_range': {k: float(f'{v:.2e}') for k, v in worst.items()}}), flush=True)
