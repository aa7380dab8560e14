# -*- coding: utf-8 -*-
''' Time protocols -- API of PySONIC/core/protocols.py (TimeProtocol 17-125, CustomProtocol
    127-221, PulsedProtocol 224-411). A protocol contributes the sorted (t, x) event list and
    tstop of a configuration: the kernel's schedule input. '''
import abc

import numpy as np

from .stimobj import StimObject, Param
from .batches import Batch


class TimeProtocol(StimObject):

    @property
    @abc.abstractmethod
    def nature(self):
        raise NotImplementedError

    @abc.abstractmethod
    def stimEvents(self):
        ''' Sorted list of (time, modulation factor) transitions. '''
        raise NotImplementedError

    @property
    @abc.abstractmethod
    def tstop(self):
        raise NotImplementedError

    def stimProfile(self):
        pts = [(0., 0)]
        for e in self.stimEvents():
            pts.append((e[0], pts[-1][1]))
            pts.append(e)
        if pts[-1][0] < self.tstop:
            pts.append((self.tstop, pts[-1][1]))
        t, x = zip(*pts)
        return np.array(t), np.array(x)


class CustomProtocol(TimeProtocol):
    ''' Arbitrary (tevents, xevents) sequence. '''

    def __init__(self, tevents, xevents, tstop, modfactor=1.):
        self.tevents = tevents
        self.xevents = xevents
        self.tstop = tstop
        self.modfactor = modfactor

    @property
    def tevents(self):
        return self._tevents

    @tevents.setter
    def tevents(self, value):
        value = np.asarray(value, dtype=float)
        if value.min() < 0.:
            raise ValueError('Invalid time events (must be positive or null)')
        self._tevents = value

    @property
    def xevents(self):
        return self._xevents

    @xevents.setter
    def xevents(self, value):
        self._xevents = np.asarray(value, dtype=float)

    @property
    def tstop(self):
        return self._tstop

    @tstop.setter
    def tstop(self, value):
        value = self.checkFloat('tstop', value)
        if value < self.tevents.max():
            raise ValueError('stopping time must be greater than largest event time')
        self._tstop = value

    def copy(self):
        return self.__class__(self.tevents, self.xevents, self.tstop, modfactor=self.modfactor)

    @staticmethod
    def inputs():
        return {
            'tevents': {'desc': 'events times', 'label': 't_{events}', 'unit': 's',
                        'precision': 2},
            'xevents': {'desc': 'events modulation factors', 'label': 'x_{events}',
                        'precision': 2},
            'tstop': {'desc': 'stopping time', 'label': 't_{stop}', 'unit': 's', 'precision': 0},
        }

    @property
    def nature(self):
        return 'custom'

    def stimEvents(self):
        return sorted(zip(self.tevents, self.xevents * self.modfactor), key=lambda e: e[0])


class PulsedProtocol(TimeProtocol):
    ''' tstim of (optionally pulsed: PRF, DC) stimulus followed by toffset. '''

    tstim = Param('checkPositiveOrNull')
    toffset = Param('checkPositiveOrNull')
    DC = Param(bounds=(0., 1.))
    # a pulsed protocol needs at least one full period inside the stimulus
    PRF = Param('checkPositiveOrNull',
                bounds=lambda self: (1 / self.tstim, np.inf) if self.DC < 1. else None)
    tstart = Param('checkPositiveOrNull')

    def __init__(self, tstim, toffset, PRF=100., DC=1., tstart=0., modfactor=1.):
        self.tstim = tstim
        self.toffset = toffset
        self.DC = DC
        self.PRF = PRF
        self.tstart = tstart
        self.modfactor = modfactor

    def copy(self):
        return self.__class__(self.tstim, self.toffset, PRF=self.PRF, DC=self.DC,
                              tstart=self.tstart)

    @property
    def tstop(self):
        return self.tstim + self.toffset + self.tstart

    def pdict(self, **kwargs):
        d = super().pdict(**kwargs)
        if 'toffset' in d and self.toffset == 0.:
            del d['toffset']
        if self.isCW:
            del d['PRF']
            del d['DC']
        if self.tstart == 0.:
            del d['tstart']
        return d

    @property
    def T_ON(self):
        return self.DC / self.PRF

    @property
    def T_OFF(self):
        return (1 - self.DC) / self.PRF

    @property
    def npulses(self):
        return int(np.round(self.tstim * self.PRF))

    @property
    def isCW(self):
        return self.DC == 1.

    @property
    def nature(self):
        return 'CW' if self.isCW else 'PW'

    @staticmethod
    def inputs():
        return {
            'tstim': {'desc': 'stimulus duration', 'label': 't_{stim}', 'unit': 's',
                      'factor': 1e0, 'precision': 0},
            'toffset': {'desc': 'offset duration', 'label': 't_{offset}', 'unit': 's',
                        'factor': 1e0, 'precision': 0},
            'PRF': {'desc': 'pulse repetition frequency', 'label': 'PRF', 'unit': 'Hz',
                    'factor': 1e0, 'precision': 2},
            'DC': {'desc': 'duty cycle', 'label': 'DC', 'unit': '%', 'factor': 1e2,
                   'precision': 1, 'minfigs': 2},
            'tstart': {'desc': 'stimulus start time', 'label': 't_{start}', 'unit': 's',
                       'precision': 0},
        }

    def tOFFON(self):
        if self.isCW:
            return np.array([self.tstart])
        return np.arange(self.npulses) / self.PRF + self.tstart

    def tONOFF(self):
        if self.isCW:
            return np.array([self.tstart + self.tstim])
        return (np.arange(self.npulses) + self.DC) / self.PRF + self.tstart

    def stimEvents(self):
        on = [(t, self.modfactor) for t in self.tOFFON()]
        off = [(t, 0.) for t in self.tONOFF()]
        return sorted(on + off, key=lambda e: e[0])

    @classmethod
    def createQueue(cls, durations, offsets, PRFs, DCs):
        ''' All (tstim, toffset, PRF, DC) combinations; CW protocols are not repeated across the
            PRF sweep (one entry at min(PRFs)). '''
        DCs = np.array(DCs)
        queue = []
        if 1.0 in DCs:
            queue += Batch.createQueue(durations, offsets, min(PRFs), 1.0)
        if np.any(DCs != 1.0):
            queue += Batch.createQueue(durations, offsets, PRFs, DCs[DCs != 1.0])
        return [cls(*item) for item in queue]


class BurstProtocol(PulsedProtocol):
    ''' nbursts bursts of (optionally pulsed) stimulus, repeated at the burst repetition frequency
        BRF: a PulsedProtocol of duration tburst and offset 1 / BRF - tburst tiled nbursts times
        (reference: PySONIC/core/protocols.py:414-518). '''

    def __init__(self, tburst, PRF=100., DC=1., BRF=None, nbursts=1, tstart=0., modfactor=1.):
        if BRF is None:
            BRF = 1 / (2 * tburst)
        self.checkBounded('BRF', BRF, (0, 1 / tburst))
        super().__init__(tburst, 1 / BRF - tburst, PRF=PRF, DC=DC, tstart=tstart,
                         modfactor=modfactor)
        self.BRF = BRF
        self.nbursts = nbursts

    def copy(self):
        return self.__class__(self.tburst, PRF=self.PRF, DC=self.DC, BRF=self.BRF,
                              nbursts=self.nbursts)

    @property
    def tburst(self):
        return self.tstim

    @property
    def tstop(self):
        return self.nbursts / self.BRF

    BRF = Param('checkPositiveOrNull', bounds=lambda self: (0, 1 / self.tburst))

    @staticmethod
    def inputs():
        d = PulsedProtocol.inputs()
        for k in ['tstim', 'toffset']:
            del d[k]
        return {
            'tburst': {'desc': 'burst duration', 'label': 't_{burst}', 'unit': 's',
                       'factor': 1e0, 'precision': 0},
            **d,
            'BRF': {'desc': 'burst repetition frequency', 'label': 'BRF', 'unit': 'Hz',
                    'precision': 1},
            'nbursts': {'desc': 'number of bursts', 'label': 'n_{bursts}'},
        }

    def _tile(self, t_one_burst):
        ''' event times of one burst repeated at every burst onset '''
        return np.ravel(np.array([t_one_burst + i / self.BRF for i in range(self.nbursts)]))

    def tOFFON(self):
        return self._tile(super().tOFFON())

    def tONOFF(self):
        return self._tile(super().tONOFF())

    @classmethod
    def createQueue(cls, durations, PRFs, DCs, BRFs, nbursts):
        ''' All (tburst, PRF, DC, BRF, nbursts) combinations, CW bursts not repeated across PRFs. '''
        base = [[p.tstim, p.PRF, p.DC] for p in PulsedProtocol.createQueue(durations, [0.], PRFs, DCs)]
        return [cls(*item, BRF, nb) for item in base for nb in nbursts for BRF in BRFs]


class BalancedPulsedProtocol(PulsedProtocol):
    ''' Charge-balanced pulses: tpulse at +modfactor followed by tpulse / xratio at
        -modfactor * xratio (reference: PySONIC/core/protocols.py:521-611). '''

    def __init__(self, tpulse, xratio, toffset, tstim=None, PRF=100, tstart=0., modfactor=1.):
        self.tpulse = tpulse
        self.xratio = xratio
        if tstim is None:
            tstim = self.ttotal
            PRF = 1 / tstim
        super().__init__(tstim, toffset, PRF=PRF, DC=self.tpulse * PRF, tstart=tstart,
                         modfactor=modfactor)

    tpulse = Param('checkPositiveOrNull')
    xratio = Param(bounds=(0., 1.))
    # between one pulse per stimulus and back-to-back (pulse + reversal) periods
    PRF = Param('checkPositiveOrNull',
                bounds=lambda self: (1 / self.tstim, 1 / self.ttotal)
                if self.tstim != self.ttotal else None)

    @property
    def treversal(self):
        return self.tpulse / self.xratio

    @property
    def ttotal(self):
        return self.tpulse + self.treversal

    def copy(self):
        return self.__class__(self.tpulse, self.xratio, self.toffset, tstim=self.tstim,
                              PRF=self.PRF)

    @staticmethod
    def inputs():
        d = PulsedProtocol.inputs()
        del d['DC']
        return {
            'tpulse': {'desc': 'pulse width', 'label': 't_{pulse}', 'unit': 's', 'factor': 1e0,
                       'precision': 2},
            'xratio': {'desc': 'balance amplitude factor', 'label': 'x_{ratio}', 'factor': 1e2,
                       'unit': '%', 'precision': 1},
            **d,
        }

    def tRev(self):
        return self.tOFFON() + self.tpulse

    def tONOFF(self):
        return self.tOFFON() + self.ttotal

    def stimEvents(self):
        events = [(t, self.modfactor) for t in self.tOFFON()]
        events += [(t, -self.modfactor * self.xratio) for t in self.tRev()]
        events += [(t, 0) for t in self.tONOFF()]
        return sorted(events, key=lambda e: e[0])


def getPulseTrainProtocol(PD, npulses, PRF):
    ''' npulses pulses of duration PD at PRF, the first one ending at 1 / PRF
        (reference: PySONIC/core/protocols.py:614-626). '''
    tstart = 1 / PRF - PD
    return PulsedProtocol(npulses / PRF + tstart, 0., PRF=PRF, DC=PD * PRF, tstart=tstart)
