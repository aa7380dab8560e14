''' Round-2 probes on the GPU box (one gpurun call, results under gpurun_out/r02/):
      stn     the 7 OtsukaSTN configurations of tests/golden/golden_sonic_STN_range.npz (amplitudes
              314 - 600 kPa): status, metrics, error against the reference's converged traces
      axes    mech_batch_run on the cells of tests/golden/golden_mech_axes.npz (16 / 64 nm,
              20 kHz - 4 MHz): relative errors and cycle counts against the reference
      tables  (A, Q) lookups generated on the device for a second frequency
              (RS 32 nm 100 kHz, LTS 32 nm 2 MHz) -> gpurun_out/r02/tables_*.npz
    usage: python tools/r02_probe.py stn axes tables
'''
import os
import sys
import json
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pysonic_amd import (NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron)  # noqa: E402
from pysonic_amd import _native as N  # noqa: E402

OUT = os.path.join(ROOT, 'gpurun_out', 'r02')
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def rms(a, b):
    return float(np.sqrt(np.mean((np.asarray(a) - np.asarray(b))**2)))


def stn():
    g = np.load(os.path.join(GOLDEN, 'golden_sonic_STN_range.npz'))
    nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron('STN'))
    cfgs = [(AcousticDrive(500e3, float(c[0])), PulsedProtocol(*[float(x) for x in c[1:]])) for c in g['configs']]
    rows, met, st, ms = nbls.runSonicBatch(500e3, 1., cfgs)
    for i, c in enumerate(g['configs']):
        r = rows[i]
        bad = ~np.isfinite(r[:, 2])
        line = {'cfg': [float(x) for x in c], 'status': int(st[i]), 'nsteps': float(met[i, N.M_NSTEPS]),
                'nrej': float(met[i, N.M_NREJ]), 'Qmin': float(met[i, N.M_QMIN]), 'Qmax': float(met[i, N.M_QMAX]),
                'first_nan_row': int(np.argmax(bad)) if bad.any() else -1,
                'ref_default_raised': bool(g[f'c{i}_default_raised']), 'ref_tight_raised': bool(g[f'c{i}_tight_raised'])}
        if f'c{i}_tight_Qm' in g and not bad.any():
            line['rms_vs_tight'] = rms(r[:, 2], g[f'c{i}_tight_Qm'])
            if f'c{i}_default_Qm' in g:
                line['ref_spread'] = rms(g[f'c{i}_default_Qm'], g[f'c{i}_tight_Qm'])
        if bad.any():
            j = int(np.argmax(bad))
            line['t_first_nan'] = float(r[j, 0])
            line['Q_before'] = [float(x) for x in r[max(0, j - 3):j, 2]]
        print('stn', json.dumps(line), flush=True)


def axes():
    g = np.load(os.path.join(GOLDEN, 'golden_mech_axes.npz'))
    cells = g['cells']
    pn = getPointNeuron('RS')
    res = []
    for a in sorted(set(cells[:, 0])):
        idx = np.where(cells[:, 0] == a)[0]
        nbls = NeuronalBilayerSonophore(float(a), pn)
        t0 = time.perf_counter()
        eff, ncyc, status, ms = nbls.runMechBatch(cells[idx, 1], cells[idx, 2], cells[idx, 3], [1.0])
        wall = time.perf_counter() - t0
        for k, i in enumerate(idx):
            tight, default = g[f'c{i}_tight_eff'], g[f'c{i}_default_eff']
            ok = np.isfinite(tight) & (tight != 0)
            e_t = float(np.max(np.abs(eff[k, 0][ok] / tight[ok] - 1))) if ok.any() else 0.
            sp = float(np.max(np.abs(default[ok] / tight[ok] - 1))) if ok.any() else 0.
            line = {'a_nm': a * 1e9, 'f_kHz': cells[i, 1] * 1e-3, 'A_kPa': cells[i, 2] * 1e-3,
                    'Q_nC': cells[i, 3] * 1e5, 'err_vs_tight': e_t, 'ref_spread': sp, 'ncyc': int(ncyc[k]),
                    'ref_ncyc_tight': (int(g[f'c{i}_tight_nrows']) - 2) // 999,
                    'ref_ncyc_default': (int(g[f'c{i}_default_nrows']) - 2) // 999, 'status': int(status[k])}
            res.append(line)
            print('axes', json.dumps(line), flush=True)
        print('axes', f'a = {a * 1e9:.0f} nm: {len(idx)} cells, kernel {ms:.1f} ms, wall {wall:.2f} s', flush=True)


def tables():
    os.makedirs(OUT, exist_ok=True)
    for name, f in (('RS', 100e3), ('LTS', 2e6)):
        pn = getPointNeuron(name)
        nbls = NeuronalBilayerSonophore(32e-9, pn)
        ref = nbls.getLookup()
        amps, charges = ref.refs['A'], ref.refs['Q']
        t0 = time.perf_counter()
        lkp3 = nbls.computeLookup([f], amps, charges)
        lkp = lkp3.project('f', f)
        wall = time.perf_counter() - t0
        keys = ['V'] + list(pn.rates)
        np.savez_compressed(os.path.join(OUT, f'tables_{name}_32nm_{f * 1e-3:.0f}kHz.npz'), A=amps, Q=charges,
                            a=32e-9, f=f, fs=1., keys=np.array(keys), ncycles=lkp3.ncycles[0],
                            **{f'tab_{k}': lkp[k] for k in keys})
        print('tables', name, f, 'wall %.2f s' % wall, 'finite', bool(np.all([np.isfinite(lkp[k]).all() for k in keys])), flush=True)


if __name__ == '__main__':
    N.require_gpu()
    os.makedirs(OUT, exist_ok=True)
    for w in sys.argv[1:] or ['stn', 'axes', 'tables']:
        {'stn': stn, 'axes': axes, 'tables': tables}[w]()
