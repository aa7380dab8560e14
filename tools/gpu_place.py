''' Development script (GPU box): where the wavefronts of a small batch run (HW_ID / XCC_ID of metric 11). '''
import sys, os, subprocess, collections
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import _common as O
    from pysonic_amd import _native as N
    HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = np.load(f'{HERE}/pysonic_amd/lookups/tables_RS_32nm_500kHz.npz')
    tables = np.array([d[f'tab_{k}'] for k in [str(k) for k in d['keys']]])
    P = np.array([560.0, 50.0, 60.0, -90.0, 0.75, 0.205, -70.3])
    y0 = np.concatenate(([O.neuron_Qm0('RS')], O.steady_states('RS')))
    model = N.SonicModel('RS', P, tables, d['A'], d['Q'])
    A, tstop, dt, ev_t, ev_x, ev_off = [], [], [], [], [], [0]
    for dc in list(np.linspace(0.9, 1.0, 8)) * 32:
        ev, ts = O.pulsed_events(100e-3, 0., 100., dc)
        A.append(600e3); tstop.append(ts); dt.append(5e-5)
        ev_t += [e[0] for e in ev]; ev_x += [e[1] for e in ev]; ev_off.append(len(ev_t))
    b = model.prepare(np.array(A), np.array(tstop), np.array(dt), np.array(ev_t), np.array(ev_x),
                      np.array(ev_off), y0, N.default_opts(write_traces=0))
    b.launch(); ms = b.sync()
    _, met, st = b.fetch(traces=False)
    ids = met[:, 11].astype(np.int64)
    hw = ids & 0xffffffff; xcc = (ids >> 32) & 0xf
    # gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
    wave = hw & 0xf; simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    percu = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    persimd = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist(), simd.tolist()))
    nw = len(set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist(), simd.tolist(), wave.tolist())))
    print(f'qpw={os.environ.get("PYSONIC_AMD_QPW")}: {ms:.2f} ms; distinct wave slots {nw}, CUs used {len(percu)}, '
          f'configs per CU {sorted(collections.Counter(percu.values()).items())}, SIMDs used {len(persimd)}, XCCs {sorted(set(xcc.tolist()))}', flush=True)
else:
    for qpw in (1, 4, 8, 16):
        env = dict(os.environ, PYSONIC_AMD_QPW=str(qpw), PYSONIC_AMD_DIAG='0')
        subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=env, check=True)
