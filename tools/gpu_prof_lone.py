''' Development (GPU box): 256 heavy configurations, one per wavefront (PYSONIC_AMD_QPW=1 must be set by
    the caller) -- for rocprofv3 --pmc: per-iteration instruction counts are then exact (iterations = steps). '''
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
cfgs = [(a, 100e-3, 0., 100., dc) for a in amps[32:] for dc in DCs[56:]] if os.environ.get("LONE_N", "256") == "256" else [(a, 100e-3, 0., 100., dc) for a in amps for dc in DCs[48:]]
b = model.prepare(*pack(cfgs), y0)
for _ in range(2):
    b.launch(); ms = b.sync()
tr, met, st = b.fetch(traces=False)
print(f'QPW={os.environ.get("PYSONIC_AMD_QPW")} kernel {ms:.2f} ms, sum steps {met[:,0].sum():.0f} max {met[:,0].max():.0f} rows {met[:,2].sum():.0f} rej {met[:,1].sum():.0f}')
