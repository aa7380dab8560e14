''' Development: BASELINE config 5 scaled -- 256 full RS configurations (16 A x 16 DC), f = 500 kHz,
    PRF = 10 kHz, tstim = 0.4 ms (+0.1 ms offset) '''
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron
tstim = float(sys.argv[1]) if len(sys.argv) > 1 else 0.4e-3
nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron('RS'))
amps = np.logspace(np.log10(10e3), np.log10(600e3), 16)
DCs = np.linspace(0.1, 1.0, 16)
cfgs = [(AcousticDrive(500e3, float(a)), PulsedProtocol(tstim, tstim / 4, 10e3, float(dc)), 1.) for a in amps for dc in DCs]
t0 = time.perf_counter()
frames, status, ms = nbls.runFullBatch(cfgs)
el = time.perf_counter() - t0
rows = sum(len(f) for f in frames)
print(f'256 full configs x {tstim*1.25e3:.2f} ms: wall {el:.2f} s, kernel {ms:.0f} ms, rows {rows} ({rows*10*8/1e9:.2f} GB), bad status {np.count_nonzero(status)}')
print('Qm range', min(f['Qm'].min() for f in frames), max(f['Qm'].max() for f in frames))
