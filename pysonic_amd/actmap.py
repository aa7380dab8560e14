# -*- coding: utf-8 -*-
''' Activation maps without plotting: the sweep logic of PySONIC/plt/actmap.py:19-127
    (ActivationMap.compute + FiringRateMap.xfunc) executed as ONE metrics-only launch.

    The reference runs one simulation per (DC, A) cell through LogBatch, writes a pickle per cell,
    detects spikes on the saved trace and stores mean(1 / ISI). Here every cell is a lane of the
    SONIC kernel, spikes are detected on the device while the rows are produced
    (csrc/sonic_integrator.hpp: SpikeTracker) and no trace ever reaches HBM; cells that raise a
    spike-detection flag are re-run with traces and analysed with the reference's host procedure.
'''
import abc

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
    fr, nspk = firingRatesOf(nbls, f, fs, configs)
    return fr.reshape(DCs.size, amps.size), nspk.reshape(DCs.size, amps.size)


def firingRatesOf(nbls, f, fs, configs):
    ''' (firing rates, spike counts) of a list of (drive, pp) configurations sharing (f, fs): one
        metrics-only launch; configurations whose on-device spike detection raised a flag are re-run with
        traces and analysed with the reference's host procedure; failed configurations give NaN '''
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
    return fr, nspk


# -------------------------------------------------------------------------------------------------------
# The caller of BASELINE config 2: scripts/run_actmaps.py -> getActivationMap(zkey, root, pneuron, a, fs, f,
# tstim, PRF, amps, DCs).run(mpi=True) (PySONIC/plt/actmap.py:19-159, plt/xymap.py:22-205, on top of
# LogBatch). Same constructor arguments, log-file name and format, (n_DC x n_A) output; no rendering.
# -------------------------------------------------------------------------------------------------------
from itertools import product

from .core.batches import LogBatch, Batch
from .core.nbls import NeuronalBilayerSonophore
from .utils import rangecode, isIterable, logger


class XYMap(LogBatch):
    ''' a LogBatch over the pairs of two vectors, x slowest; its output is the (n_x, n_y) matrix '''

    xkey = xunit = ykey = yunit = zkey = zunit = None
    xfactor = yfactor = zfactor = 1.

    def __init__(self, root, xvec, yvec):
        self.xvec, self.yvec = self._vector('x', xvec), self._vector('y', yvec)
        super().__init__([list(pair) for pair in product(self.xvec, self.yvec)], root=root)

    @staticmethod
    def _vector(name, value):
        if not isIterable(value):
            raise ValueError(f'{name} vector must be an iterable')
        value = np.asarray(value)
        if value.ndim > 1:
            raise ValueError(f'{name} vector must be one-dimensional')
        return value

    @property
    def in_key(self):
        return self.xkey

    @property
    def unit(self):
        return self.xunit

    @property
    def out_keys(self):
        return [f'{self.zkey} ({self.zunit})']

    @property
    def in_labels(self):
        return [f'{self.xkey} ({self.xunit})', f'{self.ykey} ({self.yunit})']

    @property
    def inputscode(self):
        return '_'.join([rangecode(self.xvec, self.xkey, self.xunit), rangecode(self.yvec, self.ykey, self.yunit)])

    def getOutput(self):
        return np.reshape(super().getOutput(), (self.xvec.size, self.yvec.size))

    # (the reference ends run() by writing a sorted copy of the log to filepath() -- re-evaluated with the
    # drive and protocol of the LAST cell, i.e. into a second, differently named file, xymap.py:202-204;
    # that copy is not produced here: the log itself is complete and getOutput() sorts on read)


class ActivationMap(XYMap):
    ''' response of a neuron over duty cycle (x, %) x amplitude (y, kPa) of a pulsed sonication '''

    xkey, xfactor, xunit = 'Duty cycle', 1e2, '%'
    ykey, yfactor, yunit = 'Amplitude', 1e-3, 'kPa'

    def __init__(self, root, pneuron, a, fs, f, tstim, PRF, amps, DCs):
        self.nbls = NeuronalBilayerSonophore(a, pneuron)
        self.drive = AcousticDrive(f, None)
        self.pp = PulsedProtocol(tstim, 0., PRF, .5)
        self.fs = fs
        super().__init__(root, np.asarray(DCs) * self.xfactor, np.asarray(amps) * self.yfactor)

    @property
    def sim_args(self):
        return [self.drive, self.pp, self.fs, 'sonic', None]

    def corecode(self):
        codes = self.nbls.filecodes(*self.sim_args)
        codes.pop('nature', None)
        codes.pop('DC', None)
        return '_'.join(v for v in codes.values() if v is not None)

    def _configure(self, x):
        self.pp.DC = x[0] / self.xfactor
        self.drive.A = x[1] / self.yfactor

    def compute(self, x):
        ''' one cell: simulate (or reload the saved output of) this duty cycle and amplitude and reduce it '''
        self._configure(x)
        data, _ = self.nbls.getOutput(*self.sim_args, outputdir=self.root)
        return self.xfunc(data)

    @abc.abstractmethod
    def xfunc(self, data):
        ''' scalar response of one simulation output '''

    def thresholdCurve(self, mpi=False):
        ''' threshold amplitude (Pa) at every duty cycle of the map: the queue of
            ActivationMap.addThresholdCurve (plt/actmap.py:69-78) run through Batch / the titration log '''
        queue = [[self.drive, PulsedProtocol(self.pp.tstim, self.pp.toffset, self.pp.PRF, DC / self.xfactor),
                  self.fs, 'sonic', None] for DC in self.xvec]
        return np.array(Batch(self.nbls.titrate, queue).run(mpi=mpi, loglevel=logger.level))


class FiringRateMap(ActivationMap):

    zkey, zunit, zfactor, suffix = 'Firing rate', 'Hz', 1e0, 'FRmap'

    def xfunc(self, data):
        ''' mean of the inverse inter-spike intervals, NaN below two spikes (plt/actmap.py:119-127) '''
        ispikes, _ = detectSpikes(data)
        if ispikes.size > 1:
            return np.mean(1 / np.diff(data['t'].values[ispikes]))
        return np.nan

    def computeMany(self, inputs):
        ''' all missing cells in ONE metrics-only launch, spikes detected on the device (see
            computeFiringRateMap); no per-cell output file is written on this path '''
        configs = [(AcousticDrive(self.drive.f, x[1] / self.yfactor),
                    PulsedProtocol(self.pp.tstim, self.pp.toffset, self.pp.PRF, x[0] / self.xfactor)) for x in inputs]
        fr, _ = firingRatesOf(self.nbls, self.drive.f, self.fs, configs)
        return list(fr)


class CalciumMap(ActivationMap):

    zkey, zunit, zfactor, suffix = '[Ca2+]i', 'uM', 1e6, 'Camap'

    def xfunc(self, data):
        return np.mean(data['Cai'].values * self.zfactor)

    def computeMany(self, inputs):
        ''' one launch with traces for all missing cells '''
        queue = []
        for x in inputs:
            queue.append([AcousticDrive(self.drive.f, x[1] / self.yfactor),
                          PulsedProtocol(self.pp.tstim, self.pp.toffset, self.pp.PRF, x[0] / self.xfactor),
                          self.fs, 'sonic', None])
        # (straight to the batched implementation, not through Batch.run: under a process group LogBatch.run
        # has already split the cells over the ranks)
        return [self.xfunc(data) for data, _ in self.nbls._batched_simulate([(q, {}) for q in queue])]


map_classes = {'FR': FiringRateMap, 'Cai': CalciumMap}


def getActivationMap(key, *args, **kwargs):
    if key not in map_classes:
        raise ValueError(f'{key} is not a valid map type')
    return map_classes[key](*args, **kwargs)
