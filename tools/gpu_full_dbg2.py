''' Development (GPU box): full-model runs of every neuron, status and first NaN row. '''
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron
np.set_printoptions(linewidth=250, precision=6)
for name in ['RS', 'FS', 'LTS', 'RE', 'TC', 'STN']:
    nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron(name))
    frames, status, ms = nbls.runFullBatch([(AcousticDrive(500e3, 80e3), PulsedProtocol(4e-6, 1e-6), 1.)])
    v = frames[0].values
    bad = np.where(np.isnan(v[:, 2]))[0]
    print(name, 'status', status, 'first nan row', bad[:1], 'of', v.shape[0])
    if bad.size:
        i = bad[0]
        print(v[max(0, i - 2):i + 1])
