''' Saturated regime of the RS sonic kernels: 65 536 configurations (1024 A x 64 DC, 100 ms, traces
    written) per launch under the library's development switches (quad kernel with 8 / 16 configurations per
    wavefront, lane-per-configuration kernel). usage (GPU box): python tools/sat_probe.py [n_amps] '''
import os, sys, json, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == 'child':
    from pysonic_amd import _native as N
    from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron
    n_amps = int(sys.argv[2]); traces = int(sys.argv[3])
    pn = getPointNeuron('RS'); nbls = NeuronalBilayerSonophore(32e-9, pn)
    lkp = nbls.getLookup2D(500e3, 1.)
    tables = np.array([lkp[k] for k in ['V'] + pn.rates])
    model = N.SonicModel('RS', pn.device_params(), tables, lkp.refs['A'], lkp.refs['Q'])
    amps = np.logspace(np.log10(10e3), np.log10(600e3), n_amps); DCs = np.linspace(0.05, 1.0, 64)
    cfgs = [(AcousticDrive(500e3, float(a)), PulsedProtocol(100e-3, 0., 100., float(dc))) for a in amps for dc in DCs]
    b = model.prepare(*nbls._packConfigs(cfgs), nbls.initialConditionsSonic(), N.default_opts(write_traces=traces))
    b.launch(); b.sync()
    ms = []
    for _ in range(3):
        b.launch(); ms.append(b.sync())
    _, met, st = b.fetch(traces=False)
    print(json.dumps({'env': {k: v for k, v in os.environ.items() if k.startswith('PYSONIC_AMD_')}, 'configs': len(cfgs), 'traces': traces,
                      'kernel_ms': float(np.mean(ms)), 'configs_per_s': len(cfgs) / (np.mean(ms) * 1e-3),
                      'mean_steps': float(met[:, 0].mean()), 'bad': int(np.count_nonzero(st))}), flush=True)
else:
    n_amps = sys.argv[1] if len(sys.argv) > 1 else '1024'
    for env in ({}, {'PYSONIC_AMD_QPW': '16'}, {'PYSONIC_AMD_QPW': '8'}, {'PYSONIC_AMD_QUAD': '0'}, {'PYSONIC_AMD_QUAD': '0', 'PYSONIC_AMD_LPW': '64'}):
        for traces in ('1', '0'):
            subprocess.run([sys.executable, os.path.abspath(__file__), 'child', n_amps, traces], env={**os.environ, **env})
