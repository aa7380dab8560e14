# -*- coding: utf-8 -*-
''' The REFERENCE's activation-map sweep in miniature: getActivationMap('FR', root, RS, 32 nm, fs = 1,
    500 kHz, tstim = 40 ms, PRF = 100 Hz, 3 amplitudes x 2 duty cycles).run() (PySONIC/plt/actmap.py:19-159,
    plt/xymap.py:22-205, core/batches.py:186-375), with the shipped RS table injected as the lookup.
    Captured: the log-file name, its text (header + rows as the reference leaves them), the returned
    (n_DC x n_A) matrix, and what a LogBatch does on a second run (nothing).

    Output: tests/golden/golden_actmap.json        (build container only)
'''
import os
import sys
import json
import logging
import tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402
_refimport.setup()

from PySONIC.neurons import getPointNeuron  # noqa: E402
from PySONIC.core import EffectiveVariablesLookup  # noqa: E402
from PySONIC.plt import getActivationMap  # noqa: E402
from PySONIC.utils import logger  # noqa: E402

if __name__ == '__main__':
    logger.setLevel(logging.ERROR)
    d = np.load(os.path.join(os.path.dirname(os.path.dirname(HERE)), 'pysonic_amd', 'lookups', 'tables_RS_32nm_500kHz.npz'))
    keys = [str(k) for k in d['keys']]
    lkp = EffectiveVariablesLookup({'A': d['A'], 'Q': d['Q']}, {k: d[f'tab_{k}'] for k in keys})
    amps = np.array([40e3, 120e3, 400e3])
    DCs = np.array([0.3, 1.0])
    with tempfile.TemporaryDirectory() as root:
        m = getActivationMap('FR', root, getPointNeuron('RS'), 32e-9, 1., 500e3, 40e-3, 100., amps, DCs)
        m.nbls.getLookup2D = lambda f, fs: lkp
        corecode = m.corecode()         # before the run: it depends on the drive / protocol state the cells mutate
        ret = m.run(mpi=False)          # (XYMap.run returns nothing; the matrix comes from getOutput)
        out = m.getOutput()
        text = open(m.fpath).read()
        files = sorted(os.listdir(root))
        m.run(mpi=False)
        assert open(m.fpath).read() == text
        res = {'amps': amps.tolist(), 'DCs': DCs.tolist(), 'tstim': 40e-3, 'PRF': 100., 'filename': os.path.basename(m.fpath),
               'log_text': text, 'output': [[None if v != v else float(v) for v in row] for row in np.asarray(out, dtype=float)], 'files': files,
               'inputs': [list(map(float, x)) for x in m.inputs], 'corecode': corecode, 'inputscode': m.inputscode,
               'run_returns': repr(ret)}
    with open(os.path.join(HERE, 'golden_actmap.json'), 'w') as fh:
        json.dump(res, fh, indent=1)
    print(res['filename']); print(text); print(files)
