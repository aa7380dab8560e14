''' Development script: run the CPU harness build of the integrator core on the golden RS configs
    and report accuracy (vs the reference's tight-tolerance run) and step counts. '''
import ctypes, sys, os, time
import numpy as np
sys.path.insert(0, '/root/repo')
from oracle import oracle as O

HERE = '/root/repo'
lib = ctypes.CDLL(os.environ.get('HARNESS', os.path.join(HERE, 'tests/native/libharness.so')))
dp = ctypes.POINTER(ctypes.c_double); ip = ctypes.POINTER(ctypes.c_int)

def build_recs(Aref, Qref, tables, amps):
    ''' tables (ntab, nA, nQ) -> recs (nlev, ncell, 2+2*ntab) '''
    ntab = tables.shape[0]; nQ = Qref.size
    recs = np.empty((len(amps), nQ - 1, 2 + 2 * ntab))
    for l, A in enumerate(amps):
        t1d = O.project_A(Aref, tables, A)   # (ntab, nQ)
        recs[l, :, 0] = Qref[:-1]; recs[l, :, 1] = Qref[1:]
        slope = (t1d[:, 1:] - t1d[:, :-1]) / (Qref[1:] - Qref[:-1])
        recs[l, :, 2::2] = t1d[:, :-1].T
        recs[l, :, 3::2] = slope.T
    return np.ascontiguousarray(recs)

def quad_recs(recs):
    ''' lane-per-config records (.., 2 + 2*9) -> quad layout (.., 20) '''
    q = np.empty(recs.shape[:-1] + (20,))
    q[..., 0:4] = recs[..., 0:4]            # Q_j, Q_j+1, V value, V slope
    q[..., 4:] = recs[..., 4:]              # (alpha v,s, beta v,s) per gate: same order
    return np.ascontiguousarray(q)

QUAD = bool(int(os.environ.get('QUAD', '0')))

def schedule(events, tstop, dt, levels_of_x):
    events = sorted(events, key=lambda e: e[0]) + [(tstop, None)]
    t0s, t1s, xs, ns, lv = [], [], [], [], []
    tnow, xcur = 0., 0.
    for te, xe in events:
        n = O.get_nsamples(tnow, te, dt)
        t0s.append(tnow); t1s.append(te); xs.append(xcur); ns.append(n); lv.append(levels_of_x[xcur])
        if xe is not None: xcur = xe
        tnow = te
    return (np.array(t0s), np.array(t1s), np.array(xs), np.array(ns, dtype=np.int32), np.array(lv, dtype=np.int32))

def run(name='RS', rtol=1e-8, atol=1e-10, h0=1e-6, which=None):
    d = np.load(f'{HERE}/pysonic_amd/lookups/tables_{name}_32nm_500kHz.npz')
    g = np.load(f'{HERE}/tests/golden/golden_sonic_{name}.npz')
    keys = [str(k) for k in d['keys']]
    tables = np.array([d[f'tab_{k}'] for k in keys])
    Aref, Qref = d['A'], d['Q']
    from pysonic_amd.neurons import getPointNeuron
    P = np.ascontiguousarray(getPointNeuron(name).device_params(), dtype=float)
    pn = getPointNeuron(name)
    y0 = np.concatenate(([pn.Qm0], pn.getSteadyStates(pn.Vm0)))
    ncol = y0.size + 3
    res = []
    for i, (A, tstim, toffset, PRF, DC) in enumerate(g['configs']):
        if which is not None and i not in which: continue
        recs = build_recs(Aref, Qref, tables, [0., O.is_within(A, (Aref.min(), Aref.max()))])
        events, tstop = O.pulsed_events(tstim, toffset, PRF, DC)
        t0s, t1s, xs, ns, lv = schedule(events, tstop, 5e-5, {0.: 0, 1.: 1})
        N = 1 + int(ns.sum())
        rows = np.zeros((N, ncol)); nst = ctypes.c_int(); nrj = ctypes.c_int()
        tic = time.perf_counter()
        if QUAD:
            recs = quad_recs(recs)
            call = lambda *a: lib.harness_run_quad(a[1], a[2], *a[4:])
        else:
            call = lib.harness_run
        st = call(pn.native_id, P.ctypes.data_as(dp), recs.ctypes.data_as(dp), 2, Qref.size - 1,
            ctypes.c_double(Qref[0]), ctypes.c_double(Qref[-1]), ctypes.c_double(1 / 1e-5),
            t0s.ctypes.data_as(dp), t1s.ctypes.data_as(dp), xs.ctypes.data_as(dp),
            ns.ctypes.data_as(ip), lv.ctypes.data_as(ip), len(ns), y0.ctypes.data_as(dp),
            ctypes.c_double(rtol), ctypes.c_double(atol), ctypes.c_double(h0), ctypes.c_double(1e-14), 10000000,
            rows.ctypes.data_as(dp), ctypes.byref(nst), ctypes.byref(nrj))
        el = time.perf_counter() - tic
        ref = g[f'c{i}_default']; tight = g[f'c{i}_tight']
        assert ref.shape[0] == N, (ref.shape, N)
        rms_t = np.sqrt(np.mean((rows[:, 2] - tight[:, 0])**2))
        rms_d = np.sqrt(np.mean((rows[:, 2] - ref[:, 2])**2))
        mx_t = np.abs(rows[:, 2] - tight[:, 0]).max()
        tdiff = np.abs(rows[:, 0] - ref[:, 0]).max(); sdiff = np.abs(rows[:, 1] - ref[:, 1]).max()
        vdiff = np.nanmax(np.abs(rows[:, ncol - 1] - ref[:, ncol - 1]))
        print(f'cfg {i}: st={st} steps={nst.value} rej={nrj.value} {el*1e3:.1f} ms | Qm rms vs tight {rms_t:.2e} (max {mx_t:.2e}) vs default {rms_d:.2e} | t,stim exact: {tdiff==0},{sdiff==0} | Vm maxdiff {vdiff:.2e}')
        res.append((nst.value, rms_t))
    return res

if __name__ == '__main__':
    rtol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-8
    atol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-10
    run(name=(sys.argv[3] if len(sys.argv) > 3 else 'RS'), rtol=rtol, atol=atol)
