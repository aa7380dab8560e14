''' Development (GPU box): method='hybrid' (cooperative kernel) against method='full' on single RS
    configurations of BASELINE config 5 (f = 500 kHz, PRF = 1 kHz, 1 ms + 0.25 ms): kernel ms, step attempts,
    dense periods. Appends to gpurun_out/hybrid_probe.txt as it goes. '''
import sys, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pysonic_amd import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, getPointNeuron
from pysonic_amd import _native as N
N.require_gpu()
pn = getPointNeuron('RS'); nbls = NeuronalBilayerSonophore(32e-9, pn)
os.makedirs('gpurun_out', exist_ok=True)
log = open('gpurun_out/hybrid_probe.txt', 'a')
for A, dc in ((600e3, 1.0), (600e3, 0.1), (100e3, 0.5)):
    cfgs = [(AcousticDrive(500e3, A), PulsedProtocol(1e-3, 0.25e-3, 1e3, dc))]
    Aa, tstop, _, ev_t, ev_x, ev_off = nbls._packConfigs(cfgs)
    args = ('RS', pn.device_params(), nbls.device_params(), [500e3], Aa, [1.], tstop, ev_t, ev_x, ev_off,
            nbls.initialConditionsSonic())
    out = {}
    tr, ro, st, ns, nc, ms = N.hybrid_batch_run(*args, N.full_default_opts(kernel=2))
    out['hybrid_coop'] = {'ms': round(ms, 1), 'steps': int(ns[0]), 'dense_periods': int(nc[0]), 'status': int(st[0])}
    print(A, dc, json.dumps(out), file=log, flush=True)
    tr, ro, st, ns, ms = N.full_batch_run(*args, N.full_default_opts(kernel=2))
    out['full_coop'] = {'ms': round(ms, 1), 'steps': int(ns[0]), 'status': int(st[0])}
    print(A, dc, json.dumps(out), file=log, flush=True)
    print(A, dc, json.dumps(out), flush=True)
