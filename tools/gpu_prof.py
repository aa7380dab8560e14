''' Development script: one activation-map launch (4096 cfgs) + one 65536 launch, for rocprofv3. '''
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
b = model.prepare(*pack(cfgs * n), y0)
for _ in range(2):
    b.launch(); print('kernel ms', b.sync())
