# -*- coding: utf-8 -*-
''' Generate 2-D (A x Q) effective-variable lookup tables with the REFERENCE's own
    NeuronalBilayerSonophore.computeEffVars (PySONIC/core/nbls.py:153-222), driven through the
    reference's Batch (PySONIC/core/batches.py:135-153) exactly as scripts/run_lookups.py:99-148.

    The shipped PySONIC/lookups/*.pkl are git-LFS pointer stubs in this checkout, so these tables
    are the only way to run the reference's `sonic` method here, and they pin A4/A5/A7 of
    SURVEY.md section 8.

    Usage (build container only):  python tests/golden/make_golden_tables.py RS [FS LTS ...]
    Output: pysonic_amd/lookups/tables_<neuron>_32nm_500kHz.npz (shipped as package data: the
    upstream .pkl lookups are not redistributable from this checkout, see pysonic_amd/lookups/README.md)
        refs:   A (51,) Pa, Q (nQ,) C/m2
        tables: V, alpha*, beta* each (51, nQ);  tcomp (51, nQ) kept for information
'''
import os
import sys
import logging
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402
_refimport.setup()

from PySONIC.neurons import getPointNeuron  # noqa: E402
from PySONIC.core import NeuronalBilayerSonophore, AcousticDrive, Batch  # noqa: E402
from PySONIC.utils import logger  # noqa: E402
from PySONIC.constants import DQ_LOOKUP  # noqa: E402

A_RADIUS = 32e-9
FREQ = 500e3
LOOKUPS = os.path.join(os.path.dirname(os.path.dirname(HERE)), 'pysonic_amd', 'lookups')


def default_amps():
    # scripts/run_lookups.py:186 -- np.insert(np.logspace(log10(0.1), log10(600), 50), 0, 0) kPa
    return np.insert(np.logspace(np.log10(0.1), np.log10(600.), 50), 0, 0.0) * 1e3


def default_charges(pneuron):
    # scripts/run_lookups.py:195-199
    Qmin, Qmax = pneuron.Qbounds
    return np.arange(Qmin, Qmax + DQ_LOOKUP, DQ_LOOKUP)


def _cells(args):
    ''' worker for the neurons the reference's own Batch cannot ship to its workers (the passive
        neuron is a class local to its factory function): same computeEffVars, cell by cell '''
    name, cells = args
    logger.setLevel(logging.ERROR)
    nbls = NeuronalBilayerSonophore(A_RADIUS, getPointNeuron(name))
    return [nbls.computeEffVars(AcousticDrive(FREQ, float(A)), 1., float(Q)) for A, Q in cells]


def main(names):
    logger.setLevel(logging.WARNING)
    for name in names:
        pneuron = getPointNeuron(name)
        nbls = NeuronalBilayerSonophore(A_RADIUS, pneuron)
        Aref = default_amps()
        Qref = default_charges(pneuron)
        queue = [[AcousticDrive(FREQ, float(A)), 1., float(Q)] for A in Aref for Q in Qref]
        if name.startswith('pas'):
            import multiprocessing as mp
            cells = [(q[0].A, q[2]) for q in queue]
            chunks = [(name, cells[i::8]) for i in range(8)]
            with mp.get_context('fork').Pool(8) as pool:
                parts = pool.map(_cells, chunks)
            out = [None] * len(cells)
            for i, part in enumerate(parts):
                out[i::8] = part
        else:
            out = Batch(nbls.computeEffVars, queue)(mpi=True, loglevel=logging.ERROR)
        keys = list(out[0][0][0].keys())
        tables = {k: np.array([o[0][0][k] for o in out]).reshape(Aref.size, Qref.size)
                  for k in keys}
        tcomp = np.array([o[1] for o in out]).reshape(Aref.size, Qref.size)
        fpath = os.path.join(LOOKUPS, f'tables_{name}_32nm_500kHz.npz')
        np.savez_compressed(
            fpath, A=Aref, Q=Qref, keys=np.array(keys), a=A_RADIUS, f=FREQ,
            tcomp=tcomp, **{f'tab_{k}': v for k, v in tables.items()})
        print(f'{name}: wrote {fpath} ({len(keys)} tables x {Aref.size} x {Qref.size}), '
              f'sum tcomp = {tcomp.sum():.1f} s', flush=True)


if __name__ == '__main__':
    main(sys.argv[1:] or ['RS'])
