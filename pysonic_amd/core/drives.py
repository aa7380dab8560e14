# -*- coding: utf-8 -*-
''' The acoustic drive of a configuration: what the kernels receive as (f, A, phi).

    Contract kept from PySONIC/core/drives.py:16-304 for the acoustic path: the class names `Drive` /
    `XDrive` (isinstance checks, NeuronalBilayerSonophore.checkInputs) and `AcousticDrive(f, A=None,
    phi=pi)` with its descriptions, file codes, queue builder and the members the titration and the
    solvers read (an amplitude left at None marks a drive whose threshold amplitude is to be found).
    The reference's electric drive belongs to its E-STIM path and is not part of this package. '''
import abc

import numpy as np

from .stimobj import StimObject, Param
from .batches import Batch
from ..constants import NPC_DENSE, NPC_SPARSE, ASTIM_AMP_INITIAL, ASTIM_REL_CONV_THR, ASTIM_ABS_CONV_THR


class Drive(StimObject):
    ''' something that can be evaluated in time and enumerated into a queue '''

    is_searchable = False

    @abc.abstractmethod
    def compute(self, t):
        ''' value of the drive at time t '''

    @classmethod
    def createQueue(cls, *sweeps):
        ''' one drive per combination of the sweeps (first sweep slowest), or per item of a single list '''
        if len(sweeps) == 1:
            return [cls(item) for item in sweeps[0]]
        return [cls(*combo) for combo in Batch.createQueue(*sweeps)]


class XDrive(Drive):
    ''' a drive with ONE input a threshold search may vary (`xvar`, named by `xkey`) '''

    is_searchable = True
    xkey = None
    xvar_initial = xvar_rel_thr = xvar_thr = None      # search start, relative / absolute convergence
    xvar_precheck = False                              # try the upper bound first

    @property
    def xvar(self):
        return getattr(self, self.xkey)

    @xvar.setter
    def xvar(self, value):
        setattr(self, self.xkey, value)

    @property
    def is_resolved(self):
        return self.xvar is not None

    def updatedX(self, value):
        twin = self.copy()
        twin.xvar = value
        return twin

    def nullCopy(self):
        return self.updatedX(0.)


class AcousticDrive(XDrive):
    ''' continuous-wave pressure  A sin(2 pi f t - phi)  (drives.py:191-304) '''

    xkey = 'A'
    xvar_initial, xvar_rel_thr, xvar_thr = ASTIM_AMP_INITIAL, ASTIM_REL_CONV_THR, ASTIM_ABS_CONV_THR
    xvar_precheck = True

    f = Param('checkStrictlyPositive')
    A = Param('checkPositiveOrNull', optional=True)
    phi = Param()

    _INPUTS = {
        'f': {'desc': 'US drive frequency', 'label': 'f', 'unit': 'Hz', 'precision': 0},
        'A': {'desc': 'US pressure amplitude', 'label': 'A', 'unit': 'Pa', 'precision': 2},
        'phi': {'desc': 'US drive phase', 'label': '\\Phi', 'unit': 'rad', 'precision': 2},
    }

    def __init__(self, f, A=None, phi=np.pi):
        self.f, self.A, self.phi = f, A, phi

    @staticmethod
    def inputs():
        return AcousticDrive._INPUTS

    def copy(self):
        return type(self)(self.f, self.A, phi=self.phi)

    def pdict(self, **kwargs):
        ''' the default phase is left out of descriptions and file codes '''
        d = super().pdict(**kwargs)
        if self.phi == np.pi:
            d.pop('phi')
        return d

    def compute(self, t):
        return self.A * np.sin(2 * np.pi * self.f * t - self.phi)

    # time scales the solvers take from the drive (drives.py:276-295)
    periodicity = property(lambda self: 1. / self.f)
    dt = property(lambda self: 1. / (NPC_DENSE * self.f))
    dt_sparse = property(lambda self: 1. / (NPC_SPARSE * self.f))
    nPerCycle = property(lambda self: NPC_DENSE)
    modulationFrequency = property(lambda self: self.f)
