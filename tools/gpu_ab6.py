''' Development script (GPU box): which wavefront of the 4096-configuration map finishes last, and
    what its 8 configurations cost when they run on their own. '''
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
def run(cfgs, label):
    b = model.prepare(*pack(cfgs), y0, N.default_opts(write_traces=0))
    ms = []
    for _ in range(3):
        b.launch(); ms.append(b.sync())
    _, met, st = b.fetch(traces=False)
    ns = met[:, N.M_NSTEPS]
    print(f'{label}: {min(ms):.2f} ms, steps max {ns.max():.0f} -> {min(ms) * 1e3 / ns.max():.3f} us/step', flush=True)
    return met
amps = np.logspace(np.log10(10e3), np.log10(600e3), 64)
DCs = np.linspace(0.05, 1.0, 64)
cfgs = [(a, 100e-3, 0., 100., dc) for a in amps for dc in DCs]
met = run(cfgs, 'map')
steps = met[:, N.M_NSTEPS].reshape(64, 64); rej = met[:, N.M_NREJ].reshape(64, 64)
ia, idc = np.unravel_index(np.argmax(steps), steps.shape)
print('longest configuration: A index', ia, 'DC index', idc, 'steps', steps[ia, idc], 'rejected', rej[ia, idc])
print('row of that amplitude, steps per DC group of 8:', [int(steps[ia, g * 8:(g + 1) * 8].max()) for g in range(8)])
print('steps of its group:', steps[ia, (idc // 8) * 8:(idc // 8 + 1) * 8].astype(int).tolist())
print('max steps per amplitude (every 8th):', steps.max(axis=1)[::8].astype(int).tolist())
g0 = (idc // 8) * 8
grp = [(amps[ia], 100e-3, 0., 100., DCs[g0 + k]) for k in range(8)]
run(grp * 32, 'its group of 8 alone, x32 copies')
run([cfgs[ia * 64 + idc]] * 256, 'the longest configuration alone, x256')
run([(amps[ia], 100e-3, 0., 100., dc) for dc in DCs] * 4, 'its amplitude row (64 DC) x4')
for lo in (56, 48, 32, 16, 0):
    run([(a, 100e-3, 0., 100., dc) for a in amps[lo:] for dc in DCs], f'amplitude rows {lo}..63 ({(64 - lo) * 64} configurations)')
run([(a, 100e-3, 0., 100., dc) for a in amps[:56] for dc in DCs], 'amplitude rows 0..55')
run([(a, 100e-3, 0., 100., dc) for a in amps[:32] for dc in DCs] + [(amps[62], 100e-3, 0., 100., dc) for dc in DCs], 'amplitude rows 0..31 + row 62')
