''' Development script (GPU box): lane-per-configuration kernels -- cost of a step of the slowest
    configuration as a function of the configurations per wavefront (PYSONIC_AMD_LPW), mixed sweep. '''
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import (NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron)
from pysonic_amd import _native as N
amps = np.logspace(np.log10(10e3), np.log10(600e3), 20)
PRFs = np.logspace(1, 3, 10)
DCs = np.linspace(0.05, 1.0, 10)
reps = int(os.environ.get('REPS', '1'))
for name in sys.argv[1:] or ['LTS', 'RE', 'TC', 'STN']:
    pn = getPointNeuron(name)
    nbls = NeuronalBilayerSonophore(32e-9, pn)
    cfgs = [(AcousticDrive(500e3, float(a)), PulsedProtocol(100e-3, 50e-3, float(prf), float(dc)))
            for a in amps for prf in PRFs for dc in DCs] * reps
    lkp = nbls.getLookup2D(500e3, 1.)
    tables = np.array([lkp[k] for k in ['V'] + pn.rates])
    model = N.SonicModel(name, pn.device_params(), tables, lkp.refs['A'], lkp.refs['Q'])
    packed = nbls._packConfigs(cfgs)
    out = []
    for lpw in ['1', '2', '4', '8', '16', '32', '64', 'auto']:
        if lpw == 'auto':
            os.environ.pop('PYSONIC_AMD_LPW', None)
        else:
            os.environ['PYSONIC_AMD_LPW'] = lpw
        batch = model.prepare(*packed, nbls.initialConditionsSonic(), N.default_opts(write_traces=0))
        ms = []
        for _ in range(2):
            batch.launch(); ms.append(batch.sync())
        _, met, st = batch.fetch(traces=False)
        ns = met[:, N.M_NSTEPS]
        out.append(f'{lpw}: {min(ms):.1f} ms ({min(ms) * 1e3 / ns.max():.2f})')
    print(f'{name} {len(cfgs)} cfgs, steps max {ns.max():.0f} mean {ns.mean():.0f} | ' + ' | '.join(out), flush=True)
