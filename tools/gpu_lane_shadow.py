''' Development (GPU box): lane-per-configuration kernels with and without shadow lanes (PYSONIC_AMD_DIAG=3). '''
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import (NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron)
from pysonic_amd import _native as N
amps = np.logspace(np.log10(10e3), np.log10(600e3), 20)
PRFs = np.logspace(1, 3, 10)
DCs = np.linspace(0.05, 1.0, 10)
for name in sys.argv[1:] or ['LTS', 'TC', 'STN']:
    pn = getPointNeuron(name)
    nbls = NeuronalBilayerSonophore(32e-9, pn)
    cfgs = [(AcousticDrive(500e3, float(a)), PulsedProtocol(100e-3, 50e-3, float(prf), float(dc)))
            for a in amps for prf in PRFs for dc in DCs]
    lkp = nbls.getLookup2D(500e3, 1.)
    tables = np.array([lkp[k] for k in ['V'] + pn.rates])
    model = N.SonicModel(name, pn.device_params(), tables, lkp.refs['A'], lkp.refs['Q'])
    for n in (2000, 100, 1):
        packed = nbls._packConfigs(cfgs[-n:] if n > 1 else [cfgs[-1]])
        out = []
        for diag in ('0', '3'):
            os.environ['PYSONIC_AMD_DIAG'] = diag
            batch = model.prepare(*packed, nbls.initialConditionsSonic(), N.default_opts(write_traces=0))
            ms = []
            for _ in range(2):
                batch.launch(); ms.append(batch.sync())
            out.append(f'{"shadows" if diag == "0" else "no shadows"}: {min(ms):.1f} ms')
        print(name, n, 'configurations |', ' | '.join(out), flush=True)
