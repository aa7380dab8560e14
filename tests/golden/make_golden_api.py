# -*- coding: utf-8 -*-
''' Host-API golden values captured from the REFERENCE: textual descriptions, file codes, queue
    orders and event schedules (PySONIC/core/stimobj.py, drives.py, protocols.py, batches.py,
    nbls.py:118-130,447-476). Output: tests/golden/golden_api.json (build container only). '''
import os
import sys
import json
import logging
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402
_refimport.setup()

from PySONIC.neurons import getPointNeuron  # noqa: E402
from PySONIC.core import (NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, Batch)  # noqa
from PySONIC.utils import logger, si_format  # noqa: E402

logger.setLevel(logging.ERROR)
out = {}
drives = [(500e3, 100e3), (20e3, 0.), (4e6, 599999.9), (500e3, None), (1.5e6, 12345.678)]
out['drives'] = []
for f, A in drives:
    d = AcousticDrive(f, A)
    out['drives'].append({'args': [f, A], 'repr': repr(d), 'desc': d.desc,
                          'filecodes': d.filecodes, 'dt': d.dt, 'T': d.periodicity})
pps = [(100e-3, 50e-3, 100., 1.), (100e-3, 50e-3, 100., 0.5), (100e-3, 0., 100., 0.05),
       (1., 0.1, 10., 0.33), (20e-3, 10e-3, 1000., 0.3), (0.15, 0.02, 100., 0.95)]
out['protocols'] = []
for args in pps:
    pp = PulsedProtocol(*args)
    ev = pp.stimEvents()
    out['protocols'].append({'args': list(args), 'repr': repr(pp), 'desc': pp.desc,
                             'filecodes': pp.filecodes, 'tstop': pp.tstop,
                             'nature': pp.nature, 'npulses': pp.npulses,
                             'ev_t': [float(e[0]) for e in ev], 'ev_x': [float(e[1]) for e in ev]})
nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron('RS'))
out['nbls_repr'] = repr(nbls)
out['filecodes'] = []
for (f, A) in drives[:3]:
    for args in pps[:3]:
        out['filecodes'].append(nbls.filecode(AcousticDrive(f, A), PulsedProtocol(*args), 1., 'sonic', None))
out['filecode_fs'] = nbls.filecode(AcousticDrive(500e3, 100e3), PulsedProtocol(*pps[1]), 0.5, 'sonic', None)
q = NeuronalBilayerSonophore.simQueue(
    [20e3, 500e3], [50e3, 100e3, 300e3], [0.1], [0.05], [10., 100.], [0.25, 0.5, 1.0], [1.],
    ['sonic'], None)
out['simQueue'] = [[repr(x[0]), repr(x[1]), x[2], x[3], x[4]] for x in q]
out['createQueue2'] = Batch.createQueue([1., 2., 3.], [10., 20.])
out['createQueue3'] = Batch.createQueue([1., 2.], [10., 20., 30.], [100., 200.])
out['ppQueue'] = [repr(p) for p in PulsedProtocol.createQueue([0.1, 0.2], [0.05, 0.1], [10., 100.], [0.5, 1.0])]
out['si_format'] = [[x, p, si_format(x, p, '')] for x in (0., 1e-9, 32e-9, 0.05, 1., 999.9, 1e3, 5e5, 1.234e7)
                    for p in (0, 2)]
# threshold-search histories of the reference's Thresholder on synthetic step functions
from PySONIC.threshold import threshold  # noqa: E402
out['thresholds'] = []
cases = [dict(xbounds=(0., 6e5), x0=1e4, rel_eps_thr=1e0, eps_thr=1e2, precheck=True),      # ASTIM
         dict(xbounds=(0., 1e5), x0=1e0, rel_eps_thr=1e-2, eps_thr=None, precheck=False)]    # ESTIM
for ic, kw in enumerate(cases):
    for thr in [3., 20., 5e3, 1e4, 52345.678, 99999., 3e5, 5.9e5, 7e5]:
        xh, eh = threshold(lambda x, thr=thr: bool(x >= thr), kw['xbounds'], x0=kw['x0'],
                           rel_eps_thr=kw['rel_eps_thr'], eps_thr=kw['eps_thr'],
                           precheck=kw['precheck'], output_history=True)
        res = threshold(lambda x, thr=thr: bool(x >= thr), kw['xbounds'], x0=kw['x0'],
                        rel_eps_thr=kw['rel_eps_thr'], eps_thr=kw['eps_thr'],
                        precheck=kw['precheck'])
        out['thresholds'].append({'case': ic, 'thr': thr, 'x_history': [float(v) for v in xh],
                                  'result': None if np.isnan(res) else float(res)})
with open(os.path.join(HERE, 'golden_api.json'), 'w') as fh:
    json.dump(out, fh, indent=1)
print('ok', len(out['simQueue']))
