''' Development (GPU box): when and where each configuration of the saturated sweep ran (PYSONIC_AMD_DIAG=4) '''
import sys, os
os.environ['PYSONIC_AMD_DIAG'] = '4'
os.environ['PYSONIC_AMD_WPS'] = sys.argv[1] if len(sys.argv) > 1 else '2'
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron
from pysonic_amd import _native as N
nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron('RS'))
model, _ = nbls._sonicModel(500e3, 1.)
amps = np.logspace(np.log10(10e3), np.log10(600e3), 256); DCs = np.linspace(0.05, 1.0, 256)
cfgs = [(AcousticDrive(500e3, float(a)), PulsedProtocol(100e-3, 0., 100., float(dc))) for a in amps for dc in DCs]
b = model.prepare(*nbls._packConfigs(cfgs), nbls.initialConditionsSonic(), N.default_opts(write_traces=0))
b.launch(); b.sync(); b.launch(); ms = b.sync()
_, met, st = b.fetch(traces=False)
t0 = met[:, N.M_NCAPPED + 3].min()
beg, end = (met[:, 15] - t0) * 1e-5, (met[:, 11] - t0) * 1e-5      # ms (100 MHz clock)
slot = met[:, 13].astype(np.int64); wave = slot // 16
steps = met[:, 0]
print(f'kernel {ms:.2f} ms; last end {end.max():.2f} ms')
nw = wave.max() + 1
wend = np.zeros(nw); wbusy = np.zeros(nw); wn = np.zeros(nw, int); wsteps = np.zeros(nw)
np.maximum.at(wend, wave, end); np.add.at(wbusy, wave, end - beg); np.add.at(wn, wave, 1); np.add.at(wsteps, wave, steps)
print(f'{nw} wavefronts: end time p10 {np.percentile(wend,10):.1f} p50 {np.percentile(wend,50):.1f} p90 {np.percentile(wend,90):.1f} max {wend.max():.1f} ms; '
      f'quad occupancy (sum of config lifetimes / 16 / wavefront end): mean {np.mean(wbusy / 16 / wend):.2f}')
us = (end - beg) * 1e3 / steps
print(f'us per step of a configuration: p10 {np.percentile(us,10):.2f} p50 {np.percentile(us,50):.2f} p90 {np.percentile(us,90):.2f}')
i = np.argsort(-end)[:8]
for k in i:
    print(f'  cfg {k}: A {cfgs[k][0].A/1e3:.0f} kPa DC {cfgs[k][1].DC:.2f} steps {steps[k]:.0f} began {beg[k]:.2f} ended {end[k]:.2f} ms ({us[k]:.2f} us/step) wave {wave[k]} static {beg[k] < 0.5}')
late = end > 0.8 * end.max()
print(f'configurations ending in the last 20 % of the launch: {late.sum()}, in {np.unique(wave[late]).size} wavefronts; of those static {np.sum(beg[late] < 0.5)}')
# how the time is spent: histogram of active wavefront count over time
ts = np.linspace(0, end.max(), 21)
print('time ms   wavefronts alive   configurations running')
for t in ts[:-1]:
    print(f'  {t:5.1f}   {np.sum(wend > t):6d}   {np.sum((beg <= t) & (end > t)):6d}')
