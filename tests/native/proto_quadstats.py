''' what limits the steps of the quad kernel on a configuration of the headline map (CPU harness built with
    -DSONIC_QUAD_STATS): usage python proto_quadstats.py A_kPa DC [rtol] '''
import ctypes, sys, os, time
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests/native')
from oracle import oracle as O
import proto_check as PC
lib = PC.lib
dp = ctypes.POINTER(ctypes.c_double); ip = ctypes.POINTER(ctypes.c_int)
A = float(sys.argv[1]) * 1e3; DC = float(sys.argv[2]); rtol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-6
atol = rtol * 1e-2
d = np.load('/root/repo/pysonic_amd/lookups/tables_RS_32nm_500kHz.npz')
keys = [str(k) for k in d['keys']]
tables = np.array([d[f'tab_{k}'] for k in keys]); Aref, Qref = d['A'], d['Q']
from pysonic_amd.neurons import getPointNeuron
pn = getPointNeuron('RS'); P = np.ascontiguousarray(pn.device_params(), dtype=float)
y0 = np.concatenate(([pn.Qm0], pn.getSteadyStates(pn.Vm0)))
recs = PC.quad_recs(PC.build_recs(Aref, Qref, tables, [0., A]))
events, tstop = O.pulsed_events(100e-3, 0., 100., DC)
t0s, t1s, xs, ns, lv = PC.schedule(events, tstop, 5e-5, {0.: 0, 1.: 1})
LOG = np.zeros((200000, 6)); lib.harness_quad_log(LOG.ctypes.data_as(dp), 200000)
N = 1 + int(ns.sum()); rows = np.zeros((N, 8)); nst = ctypes.c_int(); nrj = ctypes.c_int()
st = lib.harness_run_quad(P.ctypes.data_as(dp), recs.ctypes.data_as(dp), Qref.size - 1, ctypes.c_double(Qref[0]), ctypes.c_double(Qref[-1]),
    ctypes.c_double(1 / 1e-5), t0s.ctypes.data_as(dp), t1s.ctypes.data_as(dp), xs.ctypes.data_as(dp), ns.ctypes.data_as(ip), lv.ctypes.data_as(ip),
    len(ns), y0.ctypes.data_as(dp), ctypes.c_double(rtol), ctypes.c_double(atol), ctypes.c_double(1e-6), ctypes.c_double(1e-30), 10000000,
    rows.ctypes.data_as(dp), ctypes.byref(nst), ctypes.byref(nrj))
lib.harness_quad_nlog.restype = ctypes.c_long; nlog = lib.harness_quad_nlog(); LOG = LOG[:nlog]; lib.harness_quad_log(None, 0)
np.save('/tmp/quadlog.npy', LOG)
dom = (ctypes.c_long * 10)(); lib.harness_quad_dom(dom); print('  dominant error component (Q m h n p): free steps', list(dom)[:5], ' capped', list(dom)[5:])
out = (ctypes.c_long * 26)(); sums = (ctypes.c_double * 2)()
lib.harness_quad_stats(out, sums, 1)
v = list(out)
names = ['steps', 'capped', 'capped_acc', 'errlim_acc', 'last_acc', 'rej_err', 'rej_over', 'cross']
print(f'A {A/1e3:.0f} kPa DC {DC}: status {st} nsteps {nst.value} nrej {nrj.value} segments {len(ns)}')
print('  ' + '  '.join(f'{n} {x}' for n, x in zip(names, v[:8])))
print('  err of accepted node-capped steps [<1e-4 <1e-3 <1e-2 <.1 <.3 <.6 <1 >=1]:', v[8:16])
print('  err of accepted free steps                                              :', v[16:24])
print(f'  mean h capped {sums[0]/max(v[2],1):.3e}  free {sums[1]/max(v[3],1):.3e}')
if '--ref' in sys.argv:
    ref = O.sim_sonic('RS', Aref, Qref, tables, A, sorted(events, key=lambda e: e[0]), tstop, odeint_kwargs=dict(rtol=1e-12, atol=1e-15, mxstep=1000000))
    e = rows[:, 2] - ref['Qm']
    print(f'  Qm RMS vs converged oracle {np.sqrt(np.mean(e**2)):.3e} C/m2, max {np.abs(e).max():.3e}')
