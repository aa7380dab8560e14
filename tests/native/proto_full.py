import ctypes, sys
import numpy as np
sys.path.insert(0, '/root/repo')
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron
from oracle import oracle as O
np.set_printoptions(linewidth=250, precision=5)
import os
lib = ctypes.CDLL(os.environ.get('HARNESS', '/root/repo/tests/native/libharness.so'))
dp = ctypes.POINTER(ctypes.c_double); ip = ctypes.POINTER(ctypes.c_int)
name = sys.argv[1]; A = float(sys.argv[2]); tstim = float(sys.argv[3]); rtol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-8
pn = getPointNeuron(name); nbls = NeuronalBilayerSonophore(32e-9, pn)
ev, tstop = O.pulsed_events(tstim, tstim)
dt = 1 / (1000 * 500e3)
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
lib.harness_full(pn.native_id, P.ctypes.data_as(dp), B.ctypes.data_as(dp), ctypes.c_double(500e3), ctypes.c_double(A), ctypes.c_double(1.), ctypes.c_double(tstop),
                 t0s.ctypes.data_as(dp), t1s.ctypes.data_as(dp), xs.ctypes.data_as(dp), ns.ctypes.data_as(ip), len(ns), ctypes.c_longlong(M),
                 y0.ctypes.data_as(dp), ctypes.c_double(rtol), 50000000, tr.ctypes.data_as(dp), ctypes.byref(st), ctypes.byref(nst))
print(name, 'status', st.value, 'nsteps', nst.value, 'rows', M)
bad = np.where(np.isnan(tr[:, 2]))[0]
print('first nan row', bad[:1])
i = bad[0] if bad.size else M
print(tr[max(0, i - 4):i + 1])
