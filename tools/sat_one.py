''' one configuration of tools/sat_probe.py for profiling: python tools/sat_one.py [WPS] '''
import sys, os
os.environ['PYSONIC_AMD_WPS'] = sys.argv[1] if len(sys.argv) > 1 else '2'
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron
from pysonic_amd import _native as N
nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron('RS'))
model, _ = nbls._sonicModel(500e3, 1.)
amps = np.logspace(np.log10(10e3), np.log10(600e3), 256); DCs = np.linspace(0.05, 1.0, 256)
cfgs = [(AcousticDrive(500e3, float(a)), PulsedProtocol(100e-3, 0., 100., float(dc))) for a in amps for dc in DCs]
b = model.prepare(*nbls._packConfigs(cfgs), nbls.initialConditionsSonic())
for _ in range(3):
    b.launch(); print(b.sync())
