''' Development (GPU box): where the time of Batch(nbls.simulate, queue).run(mpi=True) goes for the 4096-cell
    map of bench.py: schedule, kernel, fetch (fresh / pre-touched host buffer), frames. '''
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron, Batch
from pysonic_amd import _native as N
from pysonic_amd.core.timeseries import TimeSeries
import logging
from pysonic_amd.utils import logger
logger.setLevel(logging.WARNING)
N.require_gpu()
pn = getPointNeuron('RS'); nbls = NeuronalBilayerSonophore(32e-9, pn)
amps = np.logspace(np.log10(10e3), np.log10(600e3), 64); DCs = np.linspace(0.05, 1.0, 64)
cfgs = [(AcousticDrive(500e3, float(a)), PulsedProtocol(100e-3, 0., 100., float(dc))) for a in amps for dc in DCs]
T = {}
def tic(): return time.perf_counter()
model, lkp = nbls._sonicModel(500e3, 1.)
t0 = tic(); packed = nbls._packConfigs(cfgs); T['pack (python)'] = tic() - t0
y0 = nbls.initialConditionsSonic()
t0 = tic(); b = model.prepare(*packed, y0); T['prepare (C: schedule, levels, upload)'] = tic() - t0
t0 = tic(); b.launch(); ms = b.sync(); T['launch + sync'] = tic() - t0
t0 = tic(); tr, met, st = b.fetch(); T['fetch into np.empty'] = tic() - t0
t0 = tic(); tr2, met, st = b.fetch(); T['fetch into np.empty again'] = tic() - t0
buf = np.empty_like(tr); buf[:] = 0
t0 = tic(); N.check(N.load().sonic_batch_fetch(b._h, N._ptr(buf), N._ptr(met), N._ptr(st, N._ip))); T['fetch into a touched buffer'] = tic() - t0
names = ['Qm'] + pn.statesNames() + ['Vm']
ro = b.row_off
t0 = tic()
frames = [TimeSeries.from_block(tr[ro[i]:ro[i + 1]], names, nan_columns=('Z', 'ng')) for i in range(len(cfgs))]
T['4096 x TimeSeries.from_block'] = tic() - t0
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
t0 = tic(); out = Batch(nbls.simulate, [[d, p, 1., 'sonic', None] for d, p in cfgs]).run(mpi=True); T['Batch.run total, FIRST call (log level WARNING, profiled)'] = tic() - t0
pr.disable(); sio = io.StringIO(); pstats.Stats(pr, stream=sio).sort_stats('tottime').print_stats(14); first_profile = sio.getvalue()
del out
t0 = tic(); out = Batch(nbls.simulate, [[d, p, 1., 'sonic', None] for d, p in cfgs]).run(mpi=True); T['Batch.run total, second call (log level WARNING)'] = tic() - t0
del out
logger.setLevel(logging.INFO)
t0 = tic(); out = Batch(nbls.simulate, [[d, p, 1., 'sonic', None] for d, p in cfgs]).run(mpi=True); T['Batch.run total (log level INFO)'] = tic() - t0
os.makedirs('gpurun_out', exist_ok=True)
with open('gpurun_out/e2e_probe.txt', 'w') as fh:
    for k, v in T.items():
        print(f'{k:45s} {v * 1e3:9.1f} ms', file=fh)
    print(f'kernel {ms:.1f} ms, traces {tr.nbytes / 1e6:.0f} MB', file=fh)
    print(first_profile[:3500], file=fh)
print(open('gpurun_out/e2e_probe.txt').read())
