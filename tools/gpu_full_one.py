''' Development (GPU box): one full-model run of one neuron; saves the rows (for traced / alternative builds). '''
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron
name, tag = sys.argv[1], sys.argv[2]
nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron(name))
frames, status, ms = nbls.runFullBatch([(AcousticDrive(500e3, 120e3), PulsedProtocol(4e-6, 1e-6), 1.)])
print(name, tag, status, ms)
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', f'full_{name}_{tag}.npy'), frames[0].values)
print(list(frames[0].columns))
