''' Development (CPU): the row-cooperative detailed-model core (full_row.hpp, emulated) against the lane core
    (full_core.hpp) on one configuration.   usage: python tests/native/proto_row.py <neuron> <A> <tstim> [rtol_row] '''
import ctypes, sys, os, time
import numpy as np
sys.path.insert(0, '/root/repo')
from pysonic_amd import NeuronalBilayerSonophore, getPointNeuron
from oracle import oracle as O
np.set_printoptions(linewidth=250, precision=5)
lib = ctypes.CDLL(os.environ.get('HARNESS', '/tmp/libharness.so'))
dp = ctypes.POINTER(ctypes.c_double); ip = ctypes.POINTER(ctypes.c_int)
name = sys.argv[1]; A = float(sys.argv[2]); tstim = float(sys.argv[3]); rtol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-7
pn = getPointNeuron(name); nbls = NeuronalBilayerSonophore(32e-9, pn)
PRF = float(os.environ.get('PRF', '100')); DC = float(os.environ.get('DC', '1'))
ev, tstop = O.pulsed_events(tstim, tstim / 4, PRF, DC)
dt = 1 / (1000 * 500e3)
t0s, t1s, xs, ns = [], [], [], []
tnow, xcur = 0., 0.
for te, xe in ev + [(tstop, None)]:
    t0s.append(tnow); t1s.append(te); xs.append(xcur); ns.append(O.get_nsamples(tnow, te, dt))
    if xe is not None: xcur = xe
    tnow = te
t0s, t1s, xs = [np.array(v) for v in (t0s, t1s, xs)]; ns = np.array(ns, dtype=np.int32)
M = O.get_nsamples(0., tstop, 1e-8)
cols = ['t', 'stim', 'Z', 'ng', 'Qm'] + pn.statesNames() + ['Vm']
P = np.ascontiguousarray(pn.device_params()); B = np.ascontiguousarray(nbls.device_params()); y0 = np.ascontiguousarray(nbls.initialConditionsSonic())
out = {}
ms = int(os.environ.get('MAXSTEPS', '50000000'))
for fn, rt in (('harness_full', 1e-8), ('harness_full_row', rtol)):
    tr = np.zeros((M, len(cols))); st = ctypes.c_int(); nst = ctypes.c_int()
    t0 = time.time()
    getattr(lib, fn)(pn.native_id, P.ctypes.data_as(dp), B.ctypes.data_as(dp), ctypes.c_double(500e3), ctypes.c_double(A), ctypes.c_double(1.), ctypes.c_double(tstop),
                     t0s.ctypes.data_as(dp), t1s.ctypes.data_as(dp), xs.ctypes.data_as(dp), ns.ctypes.data_as(ip), len(ns), ctypes.c_longlong(M),
                     y0.ctypes.data_as(dp), ctypes.c_double(rt), ms, tr.ctypes.data_as(dp), ctypes.byref(st), ctypes.byref(nst))
    print(fn, name, 'status', st.value, 'nsteps', nst.value, 'rows', M, f'{time.time() - t0:.1f} s', 'nan rows', int(np.isnan(tr).any(axis=1).sum()))
    out[fn] = tr
a, b = out['harness_full'], out['harness_full_row']
assert np.array_equal(a[:, :2], b[:, :2]), 't / stim columns differ'
for j, c in enumerate(cols[2:], start=2):
    rng_ = np.ptp(a[:, j]) or 1.
    print(f'{c:>4}: rms diff / range {np.sqrt(np.mean((a[:, j] - b[:, j])**2)) / rng_:.2e}   max {np.abs(a[:, j] - b[:, j]).max() / rng_:.2e}')
