''' Development (GPU box): where the time of Batch(nbls.simulate, queue).run(mpi=True) goes for the 4096-cell
    map of bench.py: pack, prepare, pipelined launch (kernels + copies), results; cProfile of one sweep. '''
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron, Batch
from pysonic_amd import _native as N
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
y0 = nbls.initialConditionsSonic()
for rep in range(3):
    for chunks in (0, 2, 3):
        t0 = tic(); packed = nbls._packConfigs(cfgs); t1 = tic()
        b = model.prepare(*packed, y0, N.default_opts(chunks=chunks)); t2 = tic()
        b.launch(to_host=True); ms = b.sync(); t3 = tic()
        _, met, st = b.fetch(traces=False); t4 = tic()
        blk = b.host_traces
        extra = ''
        if b.n_chunks:
            k, d = b.chunk_times()
            extra = ' chunk kernel ms ' + ' '.join(f'{x:.1f}' for x in k) + ' | done at ' + ' '.join(f'{x:.1f}' for x in d)
        b.close(); t5 = tic()
        if rep:
            print(f'chunks {chunks:2d}: pack {1e3*(t1-t0):.1f} prepare {1e3*(t2-t1):.1f} launch+kernels+copies {1e3*(t3-t2):.1f} '
                  f'(kernel span {ms:.1f}) metrics {1e3*(t4-t3):.1f} close {1e3*(t5-t4):.1f} total {1e3*(t5-t0):.1f} ms{extra}', flush=True)
        del blk
queue = [[d, p, 1., 'sonic', None] for d, p in cfgs]
for rep in range(3):
    t0 = tic(); out = Batch(nbls.simulate, queue).run(mpi=True, loglevel=logging.WARNING); t1 = tic()
    last = out[-1][0].shape; t2 = tic()
    print(f'Batch.run {1e3*(t1-t0):.1f} ms ({len(queue)/(t1-t0):.0f} configs/s), first frame access {1e3*(t2-t1):.2f} ms', flush=True)
    if rep == 2:
        t0 = tic(); frames = [o[0] for o in out]; t1 = tic()
        print(f'all {len(frames)} frames built in {1e3*(t1-t0):.1f} ms; Qm[-1] of last {frames[-1]["Qm"].values[-1]:.6e}')
    del out
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
out = Batch(nbls.simulate, queue).run(mpi=True, loglevel=logging.WARNING)
pr.disable(); sio = io.StringIO(); pstats.Stats(pr, stream=sio).sort_stats('tottime').print_stats(16)
print(sio.getvalue()[:4000])
