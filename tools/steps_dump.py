''' Development (GPU box): steps per configuration over an (A, DC) sweep, for the cost model of pack_wavefronts '''
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron
from pysonic_amd import _native as N
out = {}
for name in sys.argv[1:] or ['RS']:
    nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron(name))
    amps = np.logspace(np.log10(10e3), np.log10(600e3), 48); DCs = np.linspace(0.05, 1.0, 24); PRFs = [10., 100., 1000.]
    cfgs = [(AcousticDrive(500e3, float(a)), PulsedProtocol(100e-3, 50e-3, prf, float(dc))) for prf in PRFs for a in amps for dc in DCs]
    _, met, st, ms = nbls.runSonicBatch(500e3, 1., cfgs, traces=False)
    out[name + '_steps'] = met[:, 0].reshape(3, 48, 24); out[name + '_status'] = st
    print(name, ms, met[:, 0].mean(), met[:, 0].max())
out['amps'] = amps; out['DCs'] = DCs; out['PRFs'] = np.array(PRFs)
os.makedirs('gpurun_out', exist_ok=True)
np.savez('gpurun_out/steps_dump.npz', **out)
