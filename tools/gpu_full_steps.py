''' Development (GPU box): step counts and time per step of the full kernel. '''
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron
from pysonic_amd import _native
nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron('RS'))
orig = _native.full_batch_run
last = {}
def spy(*a, **k):
    r = orig(*a, **k); last['nsteps'] = r[3]; return r
_native.full_batch_run = spy
import pysonic_amd.core.nbls as M
for ncfg, tstim in ((1, 0.1e-3), (64, 0.1e-3), (256, 0.1e-3)):
    amps = np.logspace(np.log10(10e3), np.log10(600e3), max(1, int(np.sqrt(ncfg))))
    DCs = np.linspace(0.1, 1.0, max(1, ncfg // amps.size))
    cfgs = [(AcousticDrive(500e3, float(a)), PulsedProtocol(tstim, tstim / 4, 1e4, float(dc)), 1.) for a in amps for dc in DCs]
    t0 = time.perf_counter()
    frames, status, ms = nbls.runFullBatch(cfgs)
    ns = last['nsteps']
    print(f'{len(cfgs)} cfgs x {tstim*1.25e3:.3f} ms: kernel {ms:.0f} ms, steps min/mean/max {ns.min()} {ns.mean():.0f} {ns.max()}, '
          f'{ms*1e3/ns.max():.2f} us per step of the slowest, steps per us simulated {ns.max()/(tstim*1.25e6):.0f}')
