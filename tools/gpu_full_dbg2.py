''' Development (GPU box): full-model runs of every neuron: status, steps, time, first NaN row. '''
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron
from pysonic_amd import _native
orig = _native.full_batch_run
last = {}
def spy(*a, **k):
    r = orig(*a, **k); last['nsteps'] = r[3]; return r
_native.full_batch_run = spy
A = float(sys.argv[1]) if len(sys.argv) > 1 else 120e3
for name in ['RS', 'FS', 'LTS', 'RE', 'TC', 'STN']:
    nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron(name))
    t0 = time.time()
    frames, status, ms = nbls.runFullBatch([(AcousticDrive(500e3, A), PulsedProtocol(4e-6, 1e-6), 1.)])
    v = frames[0].values
    bad = np.where(np.isnan(v[:, 2]))[0]
    print(name, 'status', status, 'steps', last['nsteps'], f'kernel {ms:.0f} ms', 'first nan row', bad[:1], flush=True)
