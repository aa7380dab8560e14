# -*- coding: utf-8 -*-
''' Activation maps without plotting: the sweep logic of PySONIC/plt/actmap.py:19-127
    (ActivationMap.compute + FiringRateMap.xfunc) executed as ONE metrics-only launch.

    The reference runs one simulation per (DC, A) cell through LogBatch, writes a pickle per cell,
    detects spikes on the saved trace and stores mean(1 / ISI). Here every cell is a lane of the
    SONIC kernel, spikes are detected on the device while the rows are produced
    (csrc/sonic_integrator.hpp: SpikeTracker) and no trace ever reaches HBM; cells that raise a
    spike-detection flag are re-run with traces and analysed with the reference's host procedure.
'''
import numpy as np

from .core.drives import AcousticDrive
from .core.protocols import PulsedProtocol
from .postpro import detectSpikes
from . import _native as N


def firingRates(metrics):
    ''' mean(1 / ISI) per configuration from device metric rows; NaN below 2 spikes
        (FiringRateMap.xfunc, plt/actmap.py:119-127) '''
    n = metrics[:, N.M_NSPIKES]
    with np.errstate(invalid='ignore', divide='ignore'):
        fr = metrics[:, N.M_SUMINVISI] / (n - 1)
    return np.where(n > 1, fr, np.nan)


def computeFiringRateMap(nbls, f, amps, DCs, PRF=100., tstim=100e-3, toffset=0., fs=1.):
    ''' Firing-rate map of shape (len(DCs), len(amps)) like ActivationMap's output
        (x = duty cycle, y = amplitude; plt/actmap.py:29-34, plt/xymap.py).

        :return: (FR map in Hz, nspikes map) '''
    amps = np.asarray(amps, dtype=float)
    DCs = np.asarray(DCs, dtype=float)
    configs = [(AcousticDrive(f, float(A)), PulsedProtocol(tstim, toffset, PRF, float(DC)))
               for DC in DCs for A in amps]
    _, metrics, status, _ = nbls.runSonicBatch(f, fs, configs, traces=False)
    fr = firingRates(metrics)
    nspk = metrics[:, N.M_NSPIKES].copy()
    flagged = np.where((metrics[:, N.M_SPKFLAGS] != 0) & (status == 0))[0]
    if flagged.size:
        rows, _, _, _ = nbls.runSonicBatch(f, fs, [configs[i] for i in flagged], traces=True)
        for k, i in enumerate(flagged):
            data = nbls._toTimeSeries(rows[k])
            isp, _ = detectSpikes(data)
            nspk[i] = isp.size
            fr[i] = np.mean(1 / np.diff(data['t'].values[isp])) if isp.size > 1 else np.nan
    fr[status != 0] = np.nan
    return fr.reshape(DCs.size, amps.size), nspk.reshape(DCs.size, amps.size)
