''' Development (CPU): the row-cooperative hybrid core (hybrid_row.hpp, emulated) against the lane hybrid core
    (hybrid_core.hpp) on one configuration.   usage: python tests/native/proto_hybrid_row.py <neuron> <A> <tstim> [PRF DC] '''
import ctypes, sys, os, time
import numpy as np
sys.path.insert(0, '/root/repo')
from pysonic_amd import NeuronalBilayerSonophore, getPointNeuron
from oracle import oracle as O
lib = ctypes.CDLL(os.environ.get('HARNESS', '/tmp/libharness.so'))
dp = ctypes.POINTER(ctypes.c_double)
name = sys.argv[1]; A = float(sys.argv[2]); tstim = float(sys.argv[3])
PRF = float(sys.argv[4]) if len(sys.argv) > 4 else 100.; DC = float(sys.argv[5]) if len(sys.argv) > 5 else 1.
pn = getPointNeuron(name); nbls = NeuronalBilayerSonophore(32e-9, pn)
P = np.ascontiguousarray(pn.device_params()); B = np.ascontiguousarray(nbls.device_params())
y0 = np.ascontiguousarray(nbls.initialConditionsSonic())
ev, tstop = O.pulsed_events(tstim, tstim / 4, PRF, DC)
ev_t = np.array([e[0] for e in ev]); ev_x = np.array([e[1] for e in ev])
M = O.get_nsamples(0., tstop, 1e-8)
cols = ['t', 'stim', 'Z', 'ng', 'Qm'] + pn.statesNames() + ['Vm']
out = {}
for fn in ('harness_hybrid', 'harness_hybrid_row'):
    tr = np.zeros((M, len(cols))); st = ctypes.c_int(); nst = ctypes.c_int(); ncy = ctypes.c_int()
    scratch = np.zeros(lib.harness_hybrid_scratch_doubles())
    t0 = time.time()
    getattr(lib, fn)(pn.native_id, P.ctypes.data_as(dp), B.ctypes.data_as(dp), ctypes.c_double(500e3), ctypes.c_double(A),
                     ctypes.c_double(1.), ctypes.c_double(tstop), ev_t.ctypes.data_as(dp), ev_x.ctypes.data_as(dp), len(ev),
                     ctypes.c_longlong(M), y0.ctypes.data_as(dp), ctypes.c_double(1e-8), (-2000000000 if fn.endswith('row') and os.environ.get('STIFF') else 2000000000),
                     tr.ctypes.data_as(dp), scratch.ctypes.data_as(dp), ctypes.byref(st), ctypes.byref(nst), ctypes.byref(ncy))
    print(fn, name, 'status', st.value, 'steps', nst.value, 'dense periods', ncy.value, f'{time.time() - t0:.1f} s', 'nan rows', int(np.isnan(tr).any(axis=1).sum()))
    out[fn] = tr
a, b = out['harness_hybrid'], out['harness_hybrid_row']
print('t / stim identical:', np.array_equal(a[:, :2], b[:, :2]))
for j, c in enumerate(cols[2:], start=2):
    rng_ = max(np.ptp(a[:, j]), 1e-3 * np.abs(a[:, j]).max(), 1e-300)
    print(f'{c:>4}: rms diff / range {np.sqrt(np.mean((a[:, j] - b[:, j])**2)) / rng_:.2e}')
