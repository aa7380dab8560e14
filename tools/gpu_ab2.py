''' Development script (GPU box): 1024 heaviest activation-map configurations at several PYSONIC_AMD_QPW. '''
import sys, os, time, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) < 2:
    for q in (1, 2, 4, 16):
        env = dict(os.environ, PYSONIC_AMD_QPW=str(q))
        subprocess.run([sys.executable, __file__, str(q)], env=env)
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
for name, cfgs in (('top-quarter DC (1024)', [(a, 100e-3, 0., 100., dc) for a in amps for dc in DCs[48:]]),
                   ('top amps x top DC (256)', [(a, 100e-3, 0., 100., dc) for a in amps[32:] for dc in DCs[56:]])):
    b = model.prepare(*pack(cfgs), y0)
    ms = []
    for _ in range(3):
        b.launch(); ms.append(b.sync())
    tr, met, st = b.fetch()
    if os.environ.get('PYSONIC_AMD_DIAG') == '1':
        i = int(np.argmax(met[:, 0]))
        print(f'   shader clock MHz: slowest config {met[i, 11]:.0f}, min {met[:, 11].min():.0f} max {met[:, 11].max():.0f}')
        met[:, 11] = 0
    hw = met[:, 11].astype(np.uint64)
    xcc = (hw >> np.uint64(32)) & np.uint64(0xf); hwid = hw & np.uint64(0xffffffff)
    simd = (hwid >> np.uint64(4)) & np.uint64(3); cu = (hwid >> np.uint64(8)) & np.uint64(0xf); sh = (hwid >> np.uint64(12)) & np.uint64(1); se = (hwid >> np.uint64(13)) & np.uint64(7)
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    nsimd = len(set(zip(key.tolist(), simd.tolist()))); ncu = len(set(key.tolist()))
    from collections import Counter
    per_simd = Counter(zip(key.tolist(), simd.tolist()))
    order = np.argsort(-met[:, 0])[:8]
    print(f'   placement: {ncu} CUs, {nsimd} SIMDs used, max waves/SIMD {max(per_simd.values())}, hist {sorted(Counter(per_simd.values()).items())}; heaviest 8: ' + ' '.join(f'x{xcc[i]}s{se[i]}c{cu[i]}m{simd[i]}' for i in order))
    print(f'QPW={sys.argv[1]} {name}: kernel {min(ms):.2f} ms; max steps {met[:,0].max():.0f} -> {min(ms)*1e3/met[:,0].max():.3f} us/step of the slowest')
