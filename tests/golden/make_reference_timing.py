# -*- coding: utf-8 -*-
''' How fast the REFERENCE itself runs the BASELINE workloads on the cores of the build container
    (SURVEY.md section 8(d)): its own `Batch(func, queue).run(mpi=True)`
    process pool (PySONIC/core/batches.py:135-153) on
      (a) BASELINE config 1 (one CW simulation, RS sonic, 100 ms + 50 ms),
      (b) a fixed 64-cell slice of the 64 x 64 (A x DC) activation map of config 2 (every 8th amplitude
          and duty cycle, PRF 100 Hz, tstim 100 ms, toffset 0),
      (c) 96 cells of the lookup generation of config 3 (RS, a = 32 nm, f = 500 kHz, 8 amplitudes x
          12 charges of the run_lookups.py grid), `computeEffVars` (nbls.py:153-222).
    Timed with the pool's start-up inside the clock (that is what a user of the reference pays), and
    serially for (a). The Python reference never leaves this container: only these numbers do.

    Output: tests/golden/reference_timing.json   (build container only)
'''
import os
import sys
import json
import time
import logging
import platform
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refimport  # noqa: E402
_refimport.setup()

from PySONIC.neurons import getPointNeuron  # noqa: E402
from PySONIC.core import NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, Batch  # noqa: E402
from PySONIC.utils import logger  # noqa: E402


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


if __name__ == '__main__':
    logger.setLevel(logging.ERROR)
    pn = getPointNeuron('RS')
    nbls = NeuronalBilayerSonophore(32e-9, pn)
    # the reference's shipped lookup .pkl files are LFS stubs here: the 2-D table made by its own
    # computeEffVars (make_golden_tables.py) is injected through getLookup2D, as for the sonic goldens
    from PySONIC.core import EffectiveVariablesLookup
    d = np.load(os.path.join(os.path.dirname(os.path.dirname(HERE)), 'pysonic_amd', 'lookups',
                             'tables_RS_32nm_500kHz.npz'))
    lkp = EffectiveVariablesLookup({'A': d['A'], 'Q': d['Q']},
                                   {str(k): d[f'tab_{k}'] for k in d['keys']})
    nbls.getLookup2D = lambda f, fs: lkp
    out = {'cores': os.cpu_count(), 'cpu': cpu_model(), 'python': platform.python_version(),
           'numpy': np.__version__, 'date': time.strftime('%Y-%m-%d')}
    import scipy
    out['scipy'] = scipy.__version__

    # (a) config 1
    drive, pp = AcousticDrive(500e3, 100e3), PulsedProtocol(100e-3, 50e-3)
    t0 = time.perf_counter()
    data, meta = nbls.simulate(drive, pp)
    wall = time.perf_counter() - t0
    out['config1'] = {'wall_s': wall, 'tcomp_s': meta['tcomp'], 'rows': int(data.shape[0]),
                      'configs_per_s_per_core': 1. / wall}
    print('config 1', out['config1'], flush=True)

    # (b) 64 cells of the config-2 map
    amps = np.logspace(np.log10(10e3), np.log10(600e3), 64)[3::8]
    DCs = np.linspace(0.05, 1.0, 64)[3::8]
    queue = [[AcousticDrive(500e3, float(A)), PulsedProtocol(100e-3, 0., 100., float(DC))]
             for A in amps for DC in DCs]
    t0 = time.perf_counter()
    res = Batch(nbls.simulate, queue).run(mpi=True, loglevel=logging.ERROR)
    wall = time.perf_counter() - t0
    tcomps = [m['tcomp'] for _, m in res]
    out['config2_slice'] = {'cells': len(queue), 'grid': '64 x 64 map, amplitudes and duty cycles [3::8]',
                            'wall_s': wall, 'configs_per_s': len(queue) / wall,
                            'tcomp_sum_s': float(np.sum(tcomps)), 'tcomp_max_s': float(np.max(tcomps))}
    print('config 2 slice', out['config2_slice'], flush=True)

    # (c) 96 cells of config 3
    amps3 = np.insert(np.logspace(np.log10(100.), np.log10(600e3), 50), 0, 0.)[::7][:8]
    charges = np.arange(pn.Qbounds[0], pn.Qbounds[1] + 1e-5, 1e-5)[::14][:12]
    queue = [[AcousticDrive(500e3, float(A)), 1., float(Q)] for A in amps3 for Q in charges]
    t0 = time.perf_counter()
    res = Batch(nbls.computeEffVars, queue).run(mpi=True, loglevel=logging.ERROR)
    wall = time.perf_counter() - t0
    tcomps = [r[1] for r in res]
    out['config3_slice'] = {'cells': len(queue), 'grid': 'run_lookups.py grid, amplitudes [::7][:8] x charges [::14][:12]',
                            'wall_s': wall, 'cells_per_s': len(queue) / wall,
                            'tcomp_sum_s': float(np.sum(tcomps)), 'tcomp_max_s': float(np.max(tcomps))}
    print('config 3 slice', out['config3_slice'], flush=True)

    with open(os.path.join(HERE, 'reference_timing.json'), 'w') as fh:
        json.dump(out, fh, indent=1)
