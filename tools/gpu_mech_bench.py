''' Development: time the full RS 2-D lookup (51 x 158 cells) on the GPU and compare to the shipped table '''
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, getPointNeuron
name = sys.argv[1] if len(sys.argv) > 1 else 'RS'
d = np.load(f'pysonic_amd/lookups/tables_{name}_32nm_500kHz.npz')
nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron(name))
t0 = time.perf_counter()
lkp = nbls.computeLookup([500e3], d['A'], d['Q'])
el = time.perf_counter() - t0
print(f'{name}: {d["A"].size * d["Q"].size} cells in {el:.2f} s wall, kernel {lkp.kernel_ms:.1f} ms; reference sum tcomp = {d["tcomp"].sum():.0f} core-s')
for k in [str(x) for x in d['keys']]:
    ref = d[f'tab_{k}']; mine = lkp[k][0]
    ok = np.abs(ref) > 1e-12
    r = np.abs(mine[ok] / ref[ok] - 1)
    print(f'  {k:8s} median rel diff {np.median(r):.1e}  p99 {np.quantile(r, 0.99):.1e}  max {r.max():.1e}')
print('cycles histogram', np.bincount(lkp.ncycles.ravel()))
