''' Development script (GPU box): BASELINE config 4, RS share (10 000 configurations, PRF sweep): packing vs uniform q. '''
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import (NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron)
from pysonic_amd import _native as N
name = sys.argv[1] if len(sys.argv) > 1 else 'RS'
amps = np.logspace(np.log10(10e3), np.log10(600e3), 20)
PRFs = np.logspace(1, 3, 10)
DCs = np.linspace(0.05, 1.0, 10)
pn = getPointNeuron(name)
nbls = NeuronalBilayerSonophore(32e-9, pn)
cfgs = [(AcousticDrive(500e3, float(a)), PulsedProtocol(100e-3, 50e-3, float(prf), float(dc)))
        for a in amps for prf in PRFs for dc in DCs] * 5
lkp = nbls.getLookup2D(500e3, 1.)
tables = np.array([lkp[k] for k in ['V'] + pn.rates])
model = N.SonicModel(name, pn.device_params(), tables, lkp.refs['A'], lkp.refs['Q'])
packed = nbls._packConfigs(cfgs)
for q in ['auto', '4', '8']:
    if q == 'auto':
        os.environ.pop('PYSONIC_AMD_QPW', None)
    else:
        os.environ['PYSONIC_AMD_QPW'] = q
    batch = model.prepare(*packed, nbls.initialConditionsSonic(), N.default_opts(write_traces=0))
    ms = []
    for _ in range(2):
        batch.launch(); ms.append(batch.sync())
    _, met, st = batch.fetch(traces=False)
    ns = met[:, N.M_NSTEPS]
    print(f'{name} q={q}: {min(ms):.1f} ms, steps max {ns.max():.0f} mean {ns.mean():.0f}', flush=True)
# how good is the a-priori cost estimate? rank correlation with the measured steps
A = np.array([c[0].A for c in cfgs]); DC = np.array([c[1].DC for c in cfgs]); PRF = np.array([c[1].PRF for c in cfgs])
est = 0.1 * DC * (0.1 + A / (A + 40e3)) + 0.02 * 0.15 + 4.4e-4 * (2 * np.round(0.1 * PRF) + 1)
from scipy.stats import spearmanr
print('spearman(est, steps) =', spearmanr(est, ns).correlation)
top = np.argsort(-ns)[:10]
print('top steps:', [(int(ns[i]), round(A[i] / 1e3), round(PRF[i]), round(DC[i], 2), int((est > est[i]).sum())) for i in top])
