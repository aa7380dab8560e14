''' Development script (GPU box): 4096-configuration map -- step counts and time per step of the critical path. '''
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _common as O
from pysonic_amd import _native as N
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = np.load(f'{HERE}/pysonic_amd/lookups/tables_RS_32nm_500kHz.npz')
tables = np.array([d[f'tab_{k}'] for k in [str(k) for k in d['keys']]])
P = np.array([560.0, 50.0, 60.0, -90.0, 0.75, 0.205, -70.3])
y0 = np.concatenate(([O.neuron_Qm0('RS')], O.steady_states('RS')))
model = N.SonicModel('RS', P, tables, d['A'], d['Q'])
def pack(cfgs):
    A, tstop, dt, ev_t, ev_x, ev_off = [], [], [], [], [], [0]
    for (a, tstim, toffset, PRF, DC) in cfgs:
        ev, ts = O.pulsed_events(tstim, toffset, PRF, DC)
        A.append(a); tstop.append(ts); dt.append(5e-5)
        ev_t += [e[0] for e in ev]; ev_x += [e[1] for e in ev]; ev_off.append(len(ev_t))
    return np.array(A), np.array(tstop), np.array(dt), np.array(ev_t), np.array(ev_x), np.array(ev_off)
amps = np.logspace(np.log10(10e3), np.log10(600e3), 64)
DCs = np.linspace(0.05, 1.0, 64)
cfgs = [(a, 100e-3, 0., 100., dc) for a in amps for dc in DCs]
for rtol in [float(a) for a in sys.argv[1:]] or [1e-6]:
    b = model.prepare(*pack(cfgs), y0, N.default_opts(rtol=rtol, atol=rtol * 1e-2))
    ms = []
    for _ in range(3):
        b.launch(); ms.append(b.sync())
    _, met, st = b.fetch(traces=False)
    ns = met[:, N.M_NSTEPS]; nr = met[:, N.M_NREJ] if hasattr(N, 'M_NREJ') else np.zeros_like(ns)
    mhz = met[:, 11]
    print(f'clock {mhz.min():.0f}-{mhz.max():.0f} (of the longest: {mhz[np.argmax(ns)]:.0f}) rtol {rtol:g}: kernel {min(ms):.2f} ms | steps max {ns.max():.0f} mean {ns.mean():.0f} | '
          f'rejected max {nr.max():.0f} | {min(ms) * 1e3 / ns.max():.3f} us per step of the longest | bad {np.count_nonzero(st)}')
