''' the stiff path of full_core.hpp on the CPU harness against golden_full_stiff.npz / golden_<neuron>.npz
    usage: python proto_stiff.py NAME stiff|std [mode] [rtol]      (mode: 0 explicit, 1 automatic, 2 RODAS4) '''
import ctypes, sys, os, time
import numpy as np
sys.path.insert(0, '/root/repo')
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron
from oracle import oracle as O
lib = ctypes.CDLL(os.environ.get('HARNESS', '/root/repo/tests/native/libharness.so'))
dp = ctypes.POINTER(ctypes.c_double); ip = ctypes.POINTER(ctypes.c_int)
name, key = sys.argv[1], sys.argv[2]
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rtol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-8
if key == 'stiff':
    g = np.load('/root/repo/tests/golden/golden_full_stiff.npz')
    f, A, tstim, toffset, PRF, DC = [float(x) for x in g[f'{name}_cfg']]
    ref, tight = g[f'{name}_default'], g[f'{name}_tight']; cols = [str(c) for c in g[f'{name}_columns']]
else:
    g = np.load(f'/root/repo/tests/golden/golden_{name}.npz')
    f, A, tstim, toffset = 500e3, 120e3, 4e-6, 1e-6
    ref, tight = g['full_default'], g['full_tight']; cols = [str(c) for c in g['full_columns']]
pn = getPointNeuron(name); nbls = NeuronalBilayerSonophore(32e-9, pn)
ev, tstop = O.pulsed_events(tstim, toffset)
dt = 1 / (1000 * f)
t0s, t1s, xs, ns = [], [], [], []
tnow, xcur = 0., 0.
for te, xe in ev + [(tstop, None)]:
    t0s.append(tnow); t1s.append(te); xs.append(xcur); ns.append(O.get_nsamples(tnow, te, dt))
    if xe is not None: xcur = xe
    tnow = te
t0s, t1s, xs = [np.array(v) for v in (t0s, t1s, xs)]; ns = np.array(ns, dtype=np.int32)
M = O.get_nsamples(0., tstop, 1e-8)
ncol = len(pn.statesNames()) + 6
tr = np.zeros((M, ncol)); st = ctypes.c_int(); nst = ctypes.c_int()
P = np.ascontiguousarray(pn.device_params()); B = np.ascontiguousarray(nbls.device_params()); y0 = np.ascontiguousarray(nbls.initialConditionsSonic())
t0 = time.perf_counter()
lib.harness_full(pn.native_id, P.ctypes.data_as(dp), B.ctypes.data_as(dp), ctypes.c_double(f), ctypes.c_double(A), ctypes.c_double(1.), ctypes.c_double(tstop),
                 t0s.ctypes.data_as(dp), t1s.ctypes.data_as(dp), xs.ctypes.data_as(dp), ns.ctypes.data_as(ip), len(ns), ctypes.c_longlong(M),
                 y0.ctypes.data_as(dp), ctypes.c_double(rtol), -(mode + 1), tr.ctypes.data_as(dp), ctypes.byref(st), ctypes.byref(nst))
print(name, key, 'mode', mode, 'status', st.value, 'nsteps', nst.value, 'rows', M, f'{time.perf_counter() - t0:.1f} s')
ok = True
for i, k in enumerate(cols):
    if i < 2: continue
    spread = np.sqrt(np.mean((ref[:, i] - tight[:, i])**2)); ptp = np.ptp(tight[:, i])
    e = np.sqrt(np.nanmean((tr[:, i] - tight[:, i])**2))
    bar = max(3 * spread, 1e-6 * ptp, 1e-13 * np.abs(tight[:, i]).max())
    flag = '' if e <= bar else '   <-- FAIL'
    ok &= e <= bar
    print(f'  {k:4s} e {e:.3e}  e/ptp {e / max(ptp, 1e-300):.2e}  spread/ptp {spread / max(ptp, 1e-300):.2e}{flag}')
print('OK' if ok and st.value == 0 else 'NOT OK')
