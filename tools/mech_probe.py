''' Development (GPU box): where the time of BASELINE config 3 goes -- the RS lookup grid of one radius, one
    frequency at a time, and the 20 kHz slice by amplitude band. Appends to gpurun_out/mech_probe.txt. '''
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, getPointNeuron
from pysonic_amd import _native as N
N.require_gpu()
pn = getPointNeuron('RS')
freqs = np.array([20., 100., 500., 1e3, 2e3, 3e3, 4e3]) * 1e3
amps = np.insert(np.logspace(np.log10(100.), np.log10(600e3), 50), 0, 0.)
charges = np.arange(pn.Qbounds[0], pn.Qbounds[1] + 1e-5, 1e-5)
os.makedirs('gpurun_out', exist_ok=True)
log = open('gpurun_out/mech_probe.txt', 'a')
def say(*a):
    print(*a, flush=True); print(*a, file=log, flush=True)
for a in (32e-9, 16e-9, 64e-9):
    nbls = NeuronalBilayerSonophore(a, pn)
    for f in freqs:
        A, Q = [x.ravel() for x in np.meshgrid(amps, charges, indexing='ij')]
        eff, ncyc, st, ms = nbls.runMechBatch(np.full(A.size, f), A, Q, [1.0])
        say(f'a={a*1e9:.0f} nm f={f*1e-3:.0f} kHz: {A.size} cells {ms:.0f} ms, cycles mean {ncyc.mean():.2f}, 11-cycle cells {int((ncyc == 11).sum())}')
    if a == 32e-9:
        for lo, hi in ((0, 10), (10, 30), (30, 40), (40, 46), (46, 51)):
            A, Q = [x.ravel() for x in np.meshgrid(amps[lo:hi], charges, indexing='ij')]
            eff, ncyc, st, ms = nbls.runMechBatch(np.full(A.size, 20e3), A, Q, [1.0])
            say(f'   20 kHz, A[{lo}:{hi}] = {amps[lo]:.0f} .. {amps[hi-1]:.0f} Pa: {A.size} cells {ms:.0f} ms')
