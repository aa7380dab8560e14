''' Development (GPU box): phases of sonic_batch_prepare for the 4096-cell map (PYSONIC_AMD_DIAG=2), three times '''
import sys, os, time
import numpy as np
os.environ['PYSONIC_AMD_DIAG'] = '2'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron
from pysonic_amd import _native as N
N.require_gpu()
pn = getPointNeuron('RS'); nbls = NeuronalBilayerSonophore(32e-9, pn)
amps = np.logspace(np.log10(10e3), np.log10(600e3), 64); DCs = np.linspace(0.05, 1.0, 64)
cfgs = [(AcousticDrive(500e3, float(a)), PulsedProtocol(100e-3, 0., 100., float(dc))) for a in amps for dc in DCs]
model, lkp = nbls._sonicModel(500e3, 1.)
packed = nbls._packConfigs(cfgs); y0 = nbls.initialConditionsSonic()
for i in range(3):
    t0 = time.perf_counter(); b = model.prepare(*packed, y0); t1 = time.perf_counter()
    b.launch(); b.sync(); t2 = time.perf_counter(); b.close(); t3 = time.perf_counter()
    print(f'--- prepare {1e3 * (t1 - t0):.1f} ms, run {1e3 * (t2 - t1):.1f} ms, destroy {1e3 * (t3 - t2):.1f} ms', file=sys.stderr, flush=True)
