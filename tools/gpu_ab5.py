''' Development script (GPU box): cost of wavefront divergence between the quads of one wavefront.
    256 heavy configurations (8 duty cycles x 32 copies); PYSONIC_AMD_QPW = quads per wavefront. '''
import sys, os, subprocess
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
    same = sys.argv[2] == 'same'
    A, tstop, dt, ev_t, ev_x, ev_off = [], [], [], [], [], [0]
    DCs = [0.95] * 8 if same else np.linspace(0.9, 1.0, 8)
    for dc in list(DCs) * int(os.environ.get('COPIES', '32')):
        ev, ts = O.pulsed_events(100e-3, 0., 100., dc)
        A.append(600e3); tstop.append(ts); dt.append(5e-5)
        ev_t += [e[0] for e in ev]; ev_x += [e[1] for e in ev]; ev_off.append(len(ev_t))
    for wt in (0,):
        b = model.prepare(np.array(A), np.array(tstop), np.array(dt), np.array(ev_t), np.array(ev_x),
                          np.array(ev_off), y0, N.default_opts(write_traces=wt))
        ms = []
        for _ in range(3):
            b.launch(); ms.append(b.sync())
        _, met, st = b.fetch(traces=False)
        ns = met[:, N.M_NSTEPS]
        mhz = met[:, 11] if os.environ.get('PYSONIC_AMD_DIAG') == '1' else np.zeros(1)
        print(f'  {sys.argv[2]:5s} qpw={os.environ.get("PYSONIC_AMD_QPW")} lds={os.environ.get("PYSONIC_AMD_LDS", "auto")} '
              f'traces={wt}: {min(ms):.2f} ms, steps max {ns.max():.0f} -> {min(ms) * 1e3 / ns.max():.3f} us/step, shader clock {mhz.min():.0f}-{mhz.max():.0f} MHz', flush=True)
else:
    for qpw, copies in ((16, 32), (16, 512), (8, 32), (8, 512), (4, 16), (4, 512), (1, 2), (1, 32), (1, 96)):
        print(f'{copies * 8 // qpw} wavefronts:', flush=True)
        env = dict(os.environ, PYSONIC_AMD_QPW=str(qpw), PYSONIC_AMD_DIAG='1', COPIES=str(copies))
        subprocess.run([sys.executable, os.path.abspath(__file__), 'child', 'mixed'], env=env, check=True)
