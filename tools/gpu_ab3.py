''' Development (GPU box): 256 heavy configurations x 4 copies, one per wavefront, with 1 or 4 wavefronts
    per workgroup: do identical wavefronts on one CU run faster than different ones? '''
import sys, os, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) < 2:
    for wpb in (1, 4):
        for mode in ('dup_adjacent', 'dup_strided', 'distinct'):
            env = dict(os.environ, PYSONIC_AMD_QPW='1', PYSONIC_AMD_WPB=str(wpb))
            subprocess.run([sys.executable, __file__, mode], env=env)
    sys.exit(0)
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
base = [(a, 100e-3, 0., 100., dc) for a in amps[32:] for dc in DCs[56:]]
mode = sys.argv[1]
if mode == 'dup_adjacent':       # stable cost sort keeps the 4 copies adjacent -> same workgroup when WPB=4
    cfgs = [c for c in base for _ in range(4)]
elif mode == 'dup_strided':
    cfgs = base * 4
else:
    cfgs = [(a, 100e-3, 0., 100., dc) for a in amps for dc in DCs[48:]]
b = model.prepare(*pack(cfgs), y0, N.default_opts(write_traces=int(os.environ.get("WT", "1"))))
ms = []
for _ in range(3):
    b.launch(); ms.append(b.sync())
tr, met, st = b.fetch(traces=False)
print(f'WPB={os.environ["PYSONIC_AMD_WPB"]} {mode}: {len(cfgs)} cfgs kernel {min(ms):.2f} ms; {min(ms)*1e3/met[:,0].max():.3f} us/step of the slowest')
