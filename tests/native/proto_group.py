''' Development: the group-cooperative integrator (sonic_group.hpp, 16-array emulation) against the
    lane-per-configuration core and the goldens, on the CPU harness.
    usage: python tests/native/proto_group.py [rtol] [atol] [neurons...] '''
import ctypes, sys, os, time
import numpy as np
sys.path.insert(0, '/root/repo')
sys.path.insert(0, '/root/repo/tests/native')
from oracle import oracle as O
import proto_check as PC
lib = PC.lib; dp = PC.dp; ip = PC.ip

def run(name, rtol, atol, h0=1e-6):
    HERE = PC.HERE
    d = np.load(f'{HERE}/pysonic_amd/lookups/tables_{name}_32nm_500kHz.npz')
    g = np.load(f'{HERE}/tests/golden/golden_sonic_{name}.npz')
    keys = [str(k) for k in d['keys']]
    tables = np.array([d[f'tab_{k}'] for k in keys])
    Aref, Qref = d['A'], d['Q']
    from pysonic_amd.neurons import getPointNeuron
    pn = getPointNeuron(name)
    P = np.ascontiguousarray(pn.device_params(), dtype=float)
    y0 = np.concatenate(([pn.Qm0], [pn.steadyStates()[k](pn.Vm0) for k in pn.statesNames()]))
    ncol = y0.size + 3
    for i, (A, tstim, toffset, PRF, DC) in enumerate(g['configs']):
        recs = PC.build_recs(Aref, Qref, tables, [0., O.is_within(A, (Aref.min(), Aref.max()))])
        events, tstop = O.pulsed_events(tstim, toffset, PRF, DC)
        t0s, t1s, xs, ns, lv = PC.schedule(events, tstop, pn.chooseTimeStep(), {0.: 0, 1.: 1})
        N = 1 + int(ns.sum())
        out = {}
        for kind, fn in (('lane', lib.harness_run), ('group', lib.harness_run_group)):
            rows = np.zeros((N, ncol)); nst = ctypes.c_int(); nrj = ctypes.c_int()
            tic = time.perf_counter()
            st = fn(pn.native_id, P.ctypes.data_as(dp), recs.ctypes.data_as(dp), 2, Qref.size - 1,
                    ctypes.c_double(Qref[0]), ctypes.c_double(Qref[-1]), ctypes.c_double(1 / 1e-5),
                    t0s.ctypes.data_as(dp), t1s.ctypes.data_as(dp), xs.ctypes.data_as(dp),
                    ns.ctypes.data_as(ip), lv.ctypes.data_as(ip), len(ns), y0.ctypes.data_as(dp),
                    ctypes.c_double(rtol), ctypes.c_double(atol), ctypes.c_double(h0), ctypes.c_double(1e-30), 10000000,
                    rows.ctypes.data_as(dp), ctypes.byref(nst), ctypes.byref(nrj))
            out[kind] = (rows, st, nst.value, nrj.value, time.perf_counter() - tic)
        rl, rg = out['lane'][0], out['group'][0]
        tight = g[f'c{i}_tight']
        if tight.shape[0] != rl.shape[0]:       # golden stored resampled: lane against group only
            tight = rl[:, 2:3]
        e_l = np.sqrt(np.nanmean((rl[:, 2] - tight[:, 0])**2)); e_g = np.sqrt(np.nanmean((rg[:, 2] - tight[:, 0])**2))
        dmax = np.nanmax(np.abs(rl - rg), axis=0)
        scale = np.nanmax(np.abs(rl), axis=0) + 1e-300
        print(f'{name} cfg {i} A={A:.0f}: status {out["lane"][1]}/{out["group"][1]} steps {out["lane"][2]}/{out["group"][2]} '
              f'rej {out["lane"][3]}/{out["group"][3]} | Qm rms vs tight: lane {e_l:.2e} group {e_g:.2e} | '
              f'max |lane - group| / max|col|: {np.max(dmax / scale):.2e} (col {int(np.argmax(dmax / scale))}) '
              f'nan rows {int(np.isnan(rl[:, 2]).sum())}/{int(np.isnan(rg[:, 2]).sum())}')

if __name__ == '__main__':
    rtol = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-6
    atol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-8
    for name in (sys.argv[3:] or ['LTS', 'RE', 'TC', 'STN']):
        run(name, rtol, atol)
