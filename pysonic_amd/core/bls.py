# -*- coding: utf-8 -*-
''' BilayerSonophore: geometry, constants and cached intermolecular-pressure parameters of the
    bilayer sonophore model -- host-side subset of PySONIC/core/bls.py:80-828 needed by the
    acoustic path (the mechanical ODE itself runs on the device / in the oracle).

    The Lennard-Jones fit of the average intermolecular pressure (bls.py:410-470) is NOT redone:
    `Delta_eq` and `LJ_approx` are read from data/bls_lookups.json, the same cache format the
    reference keeps next to bls.py (bls.py:44-77).
'''
import json
import os

import numpy as np

from .model import Model
from .drives import Drive
from ..constants import Rg, CHARGE_RANGE
from ..utils import isIterable, si_format

_PM_CACHE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                         'data', 'bls_lookups.json')


class BilayerSonophore(Model):

    # biomechanical / biophysical parameters (bls.py:88-110)
    T = 309.15
    delta0 = 2.0e-9
    Delta_ = 1.4e-9
    pDelta = 1.0e5
    m = 5.0
    n = 3.3
    rhoL = 1075.0
    muL = 7.0e-4
    muS = 0.035
    kA = 0.24
    alpha = 7.56
    C0 = 0.62
    kH = 1.613e5
    P0 = 1.0e5
    Dgl = 3.68e-9
    xi = 0.5e-9
    c = 1515.0
    epsilon0 = 8.854e-12
    epsilonR = 1.0
    rel_Zmin = -0.49

    tscale = 'us'
    simkey = 'MECH'

    def __init__(self, a, Cm0, Qm0, embedding_depth=0.0):
        if a <= 0.:
            raise ValueError('Sonophore radius must be positive')
        if Cm0 <= 0.:
            raise ValueError('Resting membrane capacitance must be positive')
        if embedding_depth < 0.:
            raise ValueError('Embedding depth cannot be negative')
        self.Cm0 = Cm0
        self.Qm0 = Qm0
        self.a = a
        self.d = embedding_depth
        self.S0 = np.pi * self.a**2
        self.kA_tissue = 0.
        self.computePMparams()
        self.V0 = np.pi * self.Delta * self.a**2
        self.ng0 = self.gasPa2mol(self.P0, self.V0)

    def copy(self):
        return self.__class__(self.a, self.Cm0, self.Qm0, embedding_depth=self.d)

    def __repr__(self):
        s = f'{self.__class__.__name__}({self.a * 1e9:.1f} nm'
        if self.d > 0.:
            s += f', d={si_format(self.d, precision=1)}m'
        return f'{s})'

    @property
    def meta(self):
        return {'a': self.a, 'd': self.d, 'Cm0': self.Cm0, 'Qm0': self.Qm0}

    def filecodes(self, drive, Qm, PmCompMethod='predict'):
        if isIterable(Qm):
            Qm_code = f'{Qm.min() * 1e5:.1f}nCcm2_{Qm.max() * 1e5:.1f}nCcm2_{Qm.size}'
        else:
            Qm_code = f'{Qm * 1e5:.1f}nCcm2'
        return {'simkey': self.simkey, 'a': f'{self.a * 1e9:.0f}nm', **drive.filecodes,
                'Qm': Qm_code}

    def computePMparams(self):
        ''' Load Delta_eq and the LJ approximation for (a, Qm0) from the JSON cache. '''
        akey, Qkey = f'{self.a * 1e9:.1f}', f'{self.Qm0 * 1e5:.2f}'
        with open(_PM_CACHE) as fh:
            cache = json.load(fh)
        try:
            entry = cache[akey][Qkey]
        except KeyError:
            raise NotImplementedError(
                f'no cached intermolecular-pressure parameters for a = {akey} nm, Qm0 = {Qkey} '
                f'nC/cm2 in {_PM_CACHE} (the Lennard-Jones fit itself is out of scope)')
        self.LJ_approx = entry['LJ_approx']
        self.Delta = entry['Delta_eq']

    @property
    def Zmin(self):
        return self.rel_Zmin * self.Delta

    def curvrad(self, Z):
        return np.inf if Z == 0.0 else (self.a**2 + Z**2) / (2 * Z)

    def surface(self, Z):
        return np.pi * (self.a**2 + Z**2)

    def volume(self, Z):
        return np.pi * self.a**2 * self.Delta * (1 + (Z / (3 * self.Delta) * (3 + Z**2 / self.a**2)))

    def capacitance(self, Z):
        ''' Parallel-plate capacitance at average inter-leaflet distance (bls.py:334-345). '''
        if Z == 0.0:
            return self.Cm0
        Z2 = (self.a**2 - Z**2 - Z * self.Delta) / (2 * Z)
        return self.Cm0 * self.Delta / self.a**2 * (Z + Z2 * np.log((2 * Z + self.Delta) / self.Delta))

    def v_capacitance(self, Z):
        return np.array(list(map(self.capacitance, Z)))

    def PMavgpred(self, Z):
        x0, C = self.LJ_approx['x0'], self.LJ_approx['C']
        r = x0 / (2 * Z + self.Delta)
        return C * (np.power(r, self.LJ_approx['nrep']) - np.power(r, self.LJ_approx['nattr']))

    @classmethod
    def gasmol2Pa(cls, ng, V):
        return ng * Rg * cls.T / V

    @classmethod
    def gasPa2mol(cls, P, V):
        return P * V / (Rg * cls.T)

    def setTissueModulus(self, drive):
        self.kA_tissue = 2 * (self.alpha * drive.modulationFrequency) * self.d

    def device_params(self):
        ''' Parameter vector of the mechanical model for the native library. '''
        LJ = self.LJ_approx
        return np.array([self.a, self.Cm0, self.Delta, LJ['x0'], LJ['C'], LJ['nrep'], LJ['nattr'],
                         self.kA_tissue, self.ng0])

    @staticmethod
    def checkInputs(drive, Qm, Pm_comp_method=None):
        if not isinstance(drive, Drive):
            raise TypeError('Invalid "drive" parameter (must be an "Drive" object)')
        if not (isinstance(Qm, float) or isIterable(Qm)):
            raise TypeError('Invalid "Qm" parameter (must be a scalar or T-periodic vector)')
        Qmin, Qmax = CHARGE_RANGE
        if np.min(Qm) < Qmin or np.max(Qm) > Qmax:
            raise ValueError(f'Invalid applied charge: {Qm * 1e5} nC/cm2 '
                             f'(must be within [{Qmin * 1e5}, {Qmax * 1e5}] interval')
