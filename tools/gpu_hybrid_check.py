''' Development (GPU box): method='hybrid' against method='full' for the neurons added late in round 1. '''
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron
for name in ['HHseg','MRGnode','IB']:
    nbls=NeuronalBilayerSonophore(32e-9,getPointNeuron(name))
    d,pp=AcousticDrive(500e3,100e3),PulsedProtocol(1.2e-3,0.4e-3)
    h,_=nbls.simulate(d,pp,1.,'hybrid')
    f,_=nbls.simulate(d,pp,1.,'full')
    print(name, h.shape, f.shape, 'Qm max diff %.2e (ptp %.2e)'%(np.abs(h['Qm'].values-f['Qm'].values).max(), np.ptp(f['Qm'].values)), 'nan', np.isnan(h.values).sum())
