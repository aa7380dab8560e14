''' Development (GPU box): where the host time of a metrics-only sweep with many events goes (BASELINE config 4: one
    neuron, five frequency groups of 2000 configurations, PRF up to 1 kHz). Times the stages of
    NeuronalBilayerSonophore.runSonicBatches one by one, single-threaded, and prints the phases of
    sonic_batch_prepare (PYSONIC_AMD_DIAG=2).   usage: python tools/config4_host_probe.py [neuron] '''
import os
import sys
import time
import numpy as np
os.environ.setdefault('PYSONIC_AMD_DIAG', '2')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron  # noqa: E402
from pysonic_amd import _native as N  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'RS'
N.require_gpu()
nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron(name))
freqs = [100e3, 500e3, 1e6, 2e6, 4e6]
amps = np.logspace(np.log10(10e3), np.log10(600e3), 20); PRFs = np.logspace(1, 3, 10); DCs = np.linspace(0.05, 1.0, 10)
for f in freqs:
    nbls._sonicModel(f, 1.)
groups = {f: [(AcousticDrive(f, float(a)), PulsedProtocol(100e-3, 50e-3, float(prf), float(dc)))
              for a in amps for prf in PRFs for dc in DCs] for f in freqs}
for rep in range(3):
    T = {}
    t0 = time.perf_counter()
    packed = {f: nbls._packConfigs(groups[f]) for f in freqs}
    T['pack'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    o = N.default_opts(write_traces=0)
    batches = [nbls._sonicModel(f, 1.)[0].prepare(*packed[f], nbls.initialConditionsSonic(), o) for f in freqs]
    T['prepare'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    for b in batches:
        b.launch(to_host=False)
    T['launch'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    kms = [b.sync() for b in batches]
    T['sync'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    res = [b.fetch(traces=False) for b in batches]
    T['fetch'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    for b in batches:
        b.close()
    T['close'] = time.perf_counter() - t0
    print(name, 'rep', rep, {k: round(v * 1e3, 2) for k, v in T.items()}, 'kernel ms', [round(k, 1) for k in kms], flush=True)
t0 = time.perf_counter()
nbls.runSonicBatches([(f, 1., groups[f], None) for f in freqs], traces=False)
print('runSonicBatches', round((time.perf_counter() - t0) * 1e3, 1), 'ms')
