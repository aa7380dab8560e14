# -*- coding: utf-8 -*-
''' Drive objects -- API of PySONIC/core/drives.py:16-304 (Drive, XDrive, ElectricDrive,
    AcousticDrive). Host-side only: a drive contributes (f, A, phi) to a configuration. '''
import abc

import numpy as np

from .stimobj import StimObject, Param
from .batches import Batch
from ..constants import (NPC_DENSE, NPC_SPARSE, ESTIM_AMP_INITIAL, ESTIM_REL_CONV_THR,
                         ESTIM_AMP_UPPER_BOUND, ASTIM_AMP_INITIAL, ASTIM_REL_CONV_THR,
                         ASTIM_ABS_CONV_THR)


class Drive(StimObject):

    @abc.abstractmethod
    def compute(self, t):
        raise NotImplementedError

    @classmethod
    def createQueue(cls, *args):
        if len(args) == 1:
            return [cls(item) for item in args[0]]
        return [cls(*item) for item in Batch.createQueue(*args)]

    @property
    def is_searchable(self):
        return False


class XDrive(Drive):
    ''' Drive with one titratable input (xvar). '''
    xvar_initial = None
    xvar_rel_thr = None
    xvar_thr = None
    xvar_precheck = False

    def updatedX(self, value):
        other = self.copy()
        other.xvar = value
        return other

    @property
    def is_searchable(self):
        return True

    @property
    def is_resolved(self):
        return self.xvar is not None

    def nullCopy(self):
        return self.copy().updatedX(0.)


class ElectricDrive(XDrive):
    xkey = 'I'
    xvar_initial = ESTIM_AMP_INITIAL
    xvar_rel_thr = ESTIM_REL_CONV_THR
    xvar_range = (0., ESTIM_AMP_UPPER_BOUND)

    I = Param(optional=True)

    def __init__(self, I):
        self.I = I

    @property
    def xvar(self):
        return self.I

    @xvar.setter
    def xvar(self, value):
        self.I = value

    def copy(self):
        return self.__class__(self.I)

    @staticmethod
    def inputs():
        return {'I': {'desc': 'current density amplitude', 'label': 'I', 'unit': 'A/m2',
                      'factor': 1e-3, 'precision': 1}}

    def compute(self, t):
        return self.I


class AcousticDrive(XDrive):
    ''' Sinusoidal pressure drive A sin(2 pi f t - phi). '''
    xkey = 'A'
    xvar_initial = ASTIM_AMP_INITIAL
    xvar_rel_thr = ASTIM_REL_CONV_THR
    xvar_thr = ASTIM_ABS_CONV_THR
    xvar_precheck = True

    f = Param('checkStrictlyPositive')
    A = Param('checkPositiveOrNull', optional=True)
    phi = Param()

    def __init__(self, f, A=None, phi=np.pi):
        self.f = f
        self.A = A
        self.phi = phi

    def pdict(self, **kwargs):
        d = super().pdict(**kwargs)
        if self.phi == np.pi:
            del d['phi']
        return d

    @property
    def xvar(self):
        return self.A

    @xvar.setter
    def xvar(self, value):
        self.A = value

    def copy(self):
        return self.__class__(self.f, self.A, phi=self.phi)

    @staticmethod
    def inputs():
        return {
            'f': {'desc': 'US drive frequency', 'label': 'f', 'unit': 'Hz', 'precision': 0},
            'A': {'desc': 'US pressure amplitude', 'label': 'A', 'unit': 'Pa', 'precision': 2},
            'phi': {'desc': 'US drive phase', 'label': '\\Phi', 'unit': 'rad', 'precision': 2},
        }

    @property
    def dt(self):
        return 1 / (NPC_DENSE * self.f)

    @property
    def dt_sparse(self):
        return 1 / (NPC_SPARSE * self.f)

    @property
    def periodicity(self):
        return 1. / self.f

    @property
    def nPerCycle(self):
        return NPC_DENSE

    @property
    def modulationFrequency(self):
        return self.f

    def compute(self, t):
        return self.A * np.sin(2 * np.pi * self.f * t - self.phi)
