''' Development (GPU box): errors of method='hybrid' against the reference's goldens (tests/golden/golden_hybrid_*.npz),
    per variable, in units of the variable's range and of the reference's own default-vs-tightened spread, for a list
    of tolerances.  usage: python tools/hybrid_parity_probe.py [rtol ...] '''
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, Batch, getPointNeuron
from pysonic_amd import _native as N
N.require_gpu()
rms = lambda a, b: float(np.sqrt(np.mean((np.asarray(a) - np.asarray(b))**2)))
for name in ['RS', 'FS']:
    g = np.load(os.path.join(ROOT, 'tests', 'golden', f'golden_hybrid_{name}.npz'))
    for rtol in [float(x) for x in sys.argv[1:]] or [None]:
        nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron(name))
        if rtol is not None:
            nbls.full_opts['rtol'] = rtol
        queue = [[AcousticDrive(500e3, float(A)), PulsedProtocol(float(ts), float(to), float(prf), float(dc)), 1., 'hybrid', None]
                 for A, ts, to, prf, dc in g['configs']]
        t0 = time.perf_counter()
        out = Batch(nbls.simulate, queue).run(mpi=True)
        wall = time.perf_counter() - t0
        for ic, (data, meta) in enumerate(out):
            ref, tight, dec = g[f'c{ic}_default'], g[f'c{ic}_tight'], int(g['decimation'])
            cols = [str(c) for c in g[f'c{ic}_columns']]
            line = []
            for i, k in enumerate(cols[2:], start=2):
                ptp, spread = np.ptp(tight[:, i]), rms(ref[:, i], tight[:, i])
                e_t = rms(data[k].values[::dec], tight[:, i])
                line.append(f'{k} {e_t / ptp:.1e} ({e_t / max(spread, 1e-300):.2f}x)')
            print(f'{name} rtol {rtol} cfg {ic} ({wall:.2f} s): ' + '  '.join(line), flush=True)
