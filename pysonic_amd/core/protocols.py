# -*- coding: utf-8 -*-
''' Time protocols -- API of PySONIC/core/protocols.py (TimeProtocol 17-125, CustomProtocol
    127-221, PulsedProtocol 224-411). A protocol contributes the sorted (t, x) event list and
    tstop of a configuration: the kernel's schedule input. '''
import abc

import numpy as np

from .stimobj import StimObject
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

    def __init__(self, tstim, toffset, PRF=100., DC=1., tstart=0., modfactor=1.):
        self.tstim = tstim
        self.toffset = toffset
        self.DC = DC
        self.PRF = PRF
        self.tstart = tstart
        self.modfactor = modfactor

    @property
    def tstim(self):
        return self._tstim

    @tstim.setter
    def tstim(self, value):
        value = self.checkFloat('tstim', value)
        self.checkPositiveOrNull('tstim', value)
        self._tstim = value

    @property
    def toffset(self):
        return self._toffset

    @toffset.setter
    def toffset(self, value):
        value = self.checkFloat('toffset', value)
        self.checkPositiveOrNull('toffset', value)
        self._toffset = value

    @property
    def DC(self):
        return self._DC

    @DC.setter
    def DC(self, value):
        value = self.checkFloat('DC', value)
        self.checkBounded('DC', value, (0., 1.))
        self._DC = value

    @property
    def PRF(self):
        return self._PRF

    @PRF.setter
    def PRF(self, value):
        value = self.checkFloat('PRF', value)
        self.checkPositiveOrNull('PRF', value)
        if self.DC < 1.:
            self.checkBounded('PRF', value, (1 / self.tstim, np.inf))
        self._PRF = value

    @property
    def tstart(self):
        return self._tstart

    @tstart.setter
    def tstart(self, value):
        value = self.checkFloat('tstart', value)
        self.checkPositiveOrNull('tstart', value)
        self._tstart = value

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
