''' Development (GPU box): where the host time of a metrics-only sweep and of a Batch(simulate) queue goes.
    usage: python tools/host_profile.py [neuron] [n] '''
import sys, os, time, cProfile, pstats, io
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron, Batch
from pysonic_amd import _native as N
N.require_gpu()
name = sys.argv[1] if len(sys.argv) > 1 else 'RS'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron(name))
amps = np.logspace(np.log10(10e3), np.log10(600e3), 20)
PRFs = np.logspace(1, 3, 10); DCs = np.linspace(0.05, 1.0, 10)
cfgs = [(AcousticDrive(500e3, float(a)), PulsedProtocol(100e-3, 50e-3, float(prf), float(dc)))
        for a in amps for prf in PRFs for dc in DCs]
cfgs = (cfgs * (n // len(cfgs) + 1))[:n]
nbls.runSonicBatches([(500e3, 1., cfgs[:64], None)], traces=False)      # warm-up
for label, fn in (('metrics-only sweep', lambda: nbls.runSonicBatches([(500e3, 1., cfgs, None)], traces=False)),
                  ('Batch(simulate) x 1024', lambda: Batch(nbls.simulate, [[d, pp, 1., 'sonic', None] for d, pp in cfgs[:1024]]).run(mpi=True))):
    t0 = time.perf_counter(); fn(); t1 = time.perf_counter()
    pr = cProfile.Profile(); pr.enable(); out = fn(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28)
    print(f'==== {label}: {t1 - t0:.3f} s unprofiled'); print(s.getvalue()[:6000])
