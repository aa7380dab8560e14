''' Development: CPU harness build of the mech core vs golden_mech.npz '''
import ctypes, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from pysonic_amd import NeuronalBilayerSonophore, getPointNeuron
lib = ctypes.CDLL('/root/repo/tests/native/libharness.so')
dp = ctypes.POINTER(ctypes.c_double)
g = np.load('/root/repo/tests/golden/golden_mech.npz')
nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron('RS'))
P = np.ascontiguousarray(nbls.device_params())
rtol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-10
fs = np.array([1.0])
for i, (A, Q) in enumerate(g['pairs']):
    zs = np.zeros(999); ngs = np.zeros(999); eff = np.zeros(9); st = ctypes.c_int()
    t0 = time.perf_counter()
    nc = lib.harness_mech(0, P.ctypes.data_as(dp), ctypes.c_double(float(g['f'])), ctypes.c_double(A), ctypes.c_double(np.pi), ctypes.c_double(Q),
                          fs.ctypes.data_as(dp), 1, ctypes.c_double(rtol), 100000000, zs.ctypes.data_as(dp), ngs.ctypes.data_as(dp), eff.ctypes.data_as(dp), ctypes.byref(st))
    el = time.perf_counter() - t0
    ref = g[f'p{i}_tight_eff']; refd = g[f'p{i}_default_eff']
    ok = np.isfinite(ref)
    rel = np.max(np.abs(eff[ok] - ref[ok]) / np.maximum(np.abs(ref[ok]), 1e-300)) if ok.any() else 0
    spread = np.max(np.abs(refd[ok] - ref[ok]) / np.maximum(np.abs(ref[ok]), 1e-300)) if ok.any() else 0
    zerr = np.max(np.abs(zs - g[f'p{i}_tight_Z'][1:])) / max(np.ptp(g[f'p{i}_tight_Z']), 1e-30)
    ncr = (int(g[f'p{i}_tight_nrows']) - 2) // 999
    print(f'{i:2d} A={A:8.0f} Q={Q:+.5f} ncyc={nc} (ref tight {ncr}, default {(int(g[f"p{i}_default_nrows"])-2)//999}) st={st.value} {el*1e3:6.1f} ms  eff relerr vs tight {rel:.2e} (ref default spread {spread:.2e})  Z err/ptp {zerr:.2e}')
