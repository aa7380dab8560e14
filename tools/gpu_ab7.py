''' Development script (GPU box): sub-maps of the activation map (heaviest amplitude rows), timing only. '''
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
out = []
for nrows in [int(a) for a in sys.argv[1:]]:
    cfgs = [(a, 100e-3, 0., 100., dc) for a in amps[64 - nrows:] for dc in DCs] if nrows <= 64 else \
           [(a, 100e-3, 0., 100., dc) for a in amps for dc in DCs] * (nrows // 64)
    b = model.prepare(*pack(cfgs), y0, N.default_opts(write_traces=1))
    ms = []
    for _ in range(3):
        b.launch(); ms.append(b.sync())
    out.append(f'{len(cfgs)}: {min(ms):.2f}')
print(f'lds={os.environ.get("PYSONIC_AMD_LDS", "auto")} qpw={os.environ.get("PYSONIC_AMD_QPW", "auto")} | ' + ' | '.join(out), flush=True)
