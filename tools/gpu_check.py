''' Development script (GPU box): golden RS configs through libpysonic_amd + activation-map timing. '''
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _common as O
from pysonic_amd import _native as N

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = 'RS'
d = np.load(f'{HERE}/pysonic_amd/lookups/tables_{name}_32nm_500kHz.npz')
g = np.load(f'{HERE}/tests/golden/golden_sonic_{name}.npz')
keys = [str(k) for k in d['keys']]
tables = np.array([d[f'tab_{k}'] for k in keys])
P = np.array([560.0, 50.0, 60.0, -90.0, 0.75, 0.205, -70.3])
y0 = np.concatenate(([O.neuron_Qm0(name)], O.steady_states(name)))
model = N.SonicModel(name, P, tables, d['A'], d['Q'])

def pack(cfgs):
    A, tstop, dt, ev_t, ev_x, ev_off = [], [], [], [], [], [0]
    for (a, tstim, toffset, PRF, DC) in cfgs:
        ev, ts = O.pulsed_events(tstim, toffset, PRF, DC)
        A.append(a); tstop.append(ts); dt.append(5e-5)
        ev_t += [e[0] for e in ev]; ev_x += [e[1] for e in ev]; ev_off.append(len(ev_t))
    return np.array(A), np.array(tstop), np.array(dt), np.array(ev_t), np.array(ev_x), np.array(ev_off)

cfgs = [tuple(c) for c in g['configs']]
b = model.prepare(*pack(cfgs), y0)
tr, met, st = b.run()
for i in range(len(cfgs)):
    r = tr[b.row_off[i]:b.row_off[i + 1]]
    ref = g[f'c{i}_default']; tight = g[f'c{i}_tight']
    print(f'cfg {i}: st={st[i]} steps={met[i,0]:.0f} rej={met[i,1]:.0f} rows={r.shape[0]}/{ref.shape[0]} '
          f'rms tight {np.sqrt(np.mean((r[:,2]-tight[:,0])**2)):.2e} default {np.sqrt(np.mean((r[:,2]-ref[:,2])**2)):.2e} '
          f't exact {np.array_equal(r[:,0], ref[:,0])} stim exact {np.array_equal(r[:,1], ref[:,1])}')

# activation map 64 x 64
amps = np.logspace(np.log10(10e3), np.log10(600e3), 64)
DCs = np.linspace(0.05, 1.0, 64)
cfgs = [(a, 100e-3, 0., 100., dc) for a in amps for dc in DCs]
for rep in range(3):
    t0 = time.perf_counter()
    b = model.prepare(*pack(cfgs), y0)
    t1 = time.perf_counter()
    b.launch(); ms = b.sync()
    t2 = time.perf_counter()
    tr, met, st = b.fetch()
    t3 = time.perf_counter()
    print(f'actmap 4096: prepare {t1-t0:.3f}s kernel {ms:.2f} ms (wall {t2-t1:.3f}s) fetch {t3-t2:.3f}s -> {4096/(ms*1e-3):.0f} cfg/s; '
          f'steps mean {met[:,0].mean():.0f} max {met[:,0].max():.0f} rej mean {met[:,1].mean():.0f}; bad status {np.count_nonzero(st)}')
# large batch: 16 x the map
cfgs16 = cfgs * 16
b = model.prepare(*pack(cfgs16), y0, N.default_opts(write_traces=0))
b.launch(); ms = b.sync()
print(f'65536 cfgs metrics-only: kernel {ms:.2f} ms -> {65536/(ms*1e-3):.0f} cfg/s')
b = model.prepare(*pack(cfgs16), y0)
b.launch(); ms = b.sync()
print(f'65536 cfgs traces: kernel {ms:.2f} ms -> {65536/(ms*1e-3):.0f} cfg/s  ({b.total_rows*8*8/1e9:.2f} GB)')
