# -*- coding: utf-8 -*-
''' Subthalamic nucleus neuron (Otsuka et al. 2004; Tarnaud et al. 2018).
    Parameters and kinetics as in PySONIC/neurons/stn.py:14-430. '''
import numpy as np
from scipy.optimize import brentq

from ..core.pneuron import PointNeuron
from ..constants import FARADAY, Z_Ca
from .cortical import _inf_tau_rates


def _xinf(var, theta, k):
    return 1 / (1 + np.exp((var - theta) / k))


def _taux1(Vm, theta, sigma, tau0, tau1):
    return tau0 + tau1 / (1 + np.exp(-(Vm - theta) / sigma))


def _taux2(Vm, theta1, theta2, sigma1, sigma2, tau0, tau1):
    return tau0 + tau1 / (np.exp(-(Vm - theta1) / sigma1) + np.exp(-(Vm - theta2) / sigma2))


class OtsukaSTN(PointNeuron):
    ''' Sub-thalamic nucleus neuron '''
    name = 'STN'
    native_id = 5
    Cm0 = 1e-2
    Vm0 = -58.0
    Cai0 = 5e-9
    ENa = 60.0
    EK = -90.0
    ELeak = -60.0
    gNabar = 490.0
    gLeak = 3.5
    gKdbar = 570.0
    gCaTbar = 50.0
    gCaLbar = 150.0
    gAbar = 50.0
    gKCabar = 10.0
    Cao = 2e-3
    taur_Cai = 0.5e-3
    tau_d2 = 130e-3
    tau_r = 2e-3
    thetax_d2, kx_d2 = 0.1e-6, 0.02e-6
    thetax_r, kx_r = 0.17e-6, -0.08e-6
    area = 2.86e-9

    # gate: (theta_x, k_x, tau form, tau parameters) -- stn.py:54-136
    _gates = {
        'a': (-45, -14.7, 1, (-40, -0.5, 1e-3, 1e-3)),
        'b': (-90, 7.5, 2, (-60, -40, -30, 10, 0e-3, 200e-3)),
        'c': (-30.6, -5, 2, (-27, -50, -20, 15, 45e-3, 10e-3)),
        'd1': (-60, 7.5, 2, (-40, -20, -15, 20, 400e-3, 500e-3)),
        'm': (-40, -8, 1, (-53, -0.7, 0.2e-3, 3e-3)),
        'h': (-45.5, 6.4, 2, (-50, -50, -15, 16, 0e-3, 24.5e-3)),
        'n': (-41, -14, 2, (-40, -40, -40, 50, 0e-3, 11e-3)),
        'p': (-56, -6.7, 2, (-27, -102, -10, 15, 5e-3, 0.33e-3)),
        'q': (-85, 5.8, 2, (-50, -50, -15, 16, 0e-3, 400e-3)),
    }

    states = {'m': 'iNa activation gate', 'h': 'iNa inactivation gate', 'n': 'iKd gate',
              'a': 'iA activation gate', 'b': 'iA inactivation gate',
              'p': 'iCaT activation gate', 'q': 'iCaT inactivation gate',
              'c': 'iCaL activation gate', 'd1': 'iCaL inactivation gate 1',
              'd2': 'iCaL inactivation gate 2', 'r': 'iCaK gate',
              'Cai': 'submembrane Calcium concentration (M)'}
    rates = ['alphaa', 'betaa', 'alphab', 'betab', 'alphac', 'betac', 'alphad1', 'betad1',
             'alpham', 'betam', 'alphah', 'betah', 'alphan', 'betan', 'alphap', 'betap',
             'alphaq', 'betaq']

    @classmethod
    def xinf(cls, g, Vm):
        th, k, _, _ = cls._gates[g]
        return _xinf(Vm, th, k)

    @classmethod
    def taux(cls, g, Vm):
        _, _, form, prm = cls._gates[g]
        return _taux1(Vm, *prm) if form == 1 else _taux2(Vm, *prm)

    @classmethod
    def d2inf(cls, Cai):
        return _xinf(Cai, cls.thetax_d2, cls.kx_d2)

    @classmethod
    def rinf(cls, Cai):
        return _xinf(Cai, cls.thetax_r, cls.kx_r)

    @classmethod
    def iNa(cls, m, h, Vm):
        return cls.gNabar * m**3 * h * (Vm - cls.ENa)

    @classmethod
    def iKd(cls, n, Vm):
        return cls.gKdbar * n**4 * (Vm - cls.EK)

    @classmethod
    def iA(cls, a, b, Vm):
        return cls.gAbar * a**2 * b * (Vm - cls.EK)

    @classmethod
    def iCaT(cls, p, q, Vm, Cai):
        return cls.gCaTbar * p**2 * q * (Vm - cls.nernst(Z_Ca, Cai, cls.Cao, cls.T))

    @classmethod
    def iCaL(cls, c, d1, d2, Vm, Cai):
        return cls.gCaLbar * c**2 * d1 * d2 * (Vm - cls.nernst(Z_Ca, Cai, cls.Cao, cls.T))

    @classmethod
    def iKCa(cls, r, Vm):
        return cls.gKCabar * r**2 * (Vm - cls.EK)

    @classmethod
    def iLeak(cls, Vm):
        return cls.gLeak * (Vm - cls.ELeak)

    @classmethod
    def getEffectiveDepth(cls, Cai, Vm):
        ''' Depth making Cai0 an equilibrium at rest (stn.py:198-207). '''
        iCaT = cls.iCaT(cls.xinf('p', Vm), cls.xinf('q', Vm), Vm, Cai)
        iCaL = cls.iCaL(cls.xinf('c', Vm), cls.xinf('d1', Vm), cls.d2inf(Cai), Vm, Cai)
        return -(iCaT + iCaL) / (Z_Ca * FARADAY * Cai / cls.taur_Cai) * 1e-6

    @classmethod
    def derCai(cls, p, q, c, d1, d2, Cai, Vm):
        iCa_tot = cls.iCaT(p, q, Vm, Cai) + cls.iCaL(c, d1, d2, Vm, Cai)
        return -cls.current_to_molar_rate_Ca * iCa_tot - Cai / cls.taur_Cai

    @classmethod
    def effRates(cls):
        d = {}
        for g in ['a', 'b', 'c', 'd1', 'm', 'h', 'n', 'p', 'q']:
            al, be = _inf_tau_rates(lambda Vm, g=g: cls.xinf(g, Vm),
                                    lambda Vm, g=g: cls.taux(g, Vm))
            d[f'alpha{g}'], d[f'beta{g}'] = al, be
        return d

    @classmethod
    def derStates(cls):
        d = {g: (lambda Vm, x, g=g: (cls.xinf(g, Vm) - x[g]) / cls.taux(g, Vm))
             for g in cls._gates}
        d['d2'] = lambda Vm, x: (cls.d2inf(x['Cai']) - x['d2']) / cls.tau_d2
        d['r'] = lambda Vm, x: (cls.rinf(x['Cai']) - x['r']) / cls.tau_r
        d['Cai'] = lambda Vm, x: cls.derCai(x['p'], x['q'], x['c'], x['d1'], x['d2'],
                                            x['Cai'], Vm)
        return d

    @classmethod
    def Caiinf(cls, p, q, c, d1, Vm):
        ''' Root of dCai/dt around Cai0 (utils.findModifiedEq, utils.py:659-682). '''
        return brentq(lambda Cai: cls.derCai(p, q, c, d1, cls.d2inf(Cai), Cai, Vm),
                      cls.Cai0 * 1e-4, cls.Cai0 * 1e3, xtol=1e-16)

    @classmethod
    def steadyStates(cls):
        d = {g: (lambda Vm, g=g: cls.xinf(g, Vm)) for g in cls._gates}
        d['Cai'] = lambda Vm: cls.Caiinf(d['p'](Vm), d['q'](Vm), d['c'](Vm), d['d1'](Vm), Vm)
        d['d2'] = lambda Vm: cls.d2inf(d['Cai'](Vm))
        d['r'] = lambda Vm: cls.rinf(d['Cai'](Vm))
        return d

    @classmethod
    def currents(cls):
        return {
            'iNa': lambda Vm, x: cls.iNa(x['m'], x['h'], Vm),
            'iKd': lambda Vm, x: cls.iKd(x['n'], Vm),
            'iA': lambda Vm, x: cls.iA(x['a'], x['b'], Vm),
            'iCaT': lambda Vm, x: cls.iCaT(x['p'], x['q'], Vm, x['Cai']),
            'iCaL': lambda Vm, x: cls.iCaL(x['c'], x['d1'], x['d2'], Vm, x['Cai']),
            'iKCa': lambda Vm, x: cls.iKCa(x['r'], Vm),
            'iLeak': lambda Vm, _: cls.iLeak(Vm),
        }

    @classmethod
    def titrationFunc(cls, *args, **kwargs):
        return cls.isSilenced(*args, **kwargs)

    @classmethod
    def device_params(cls):
        nernst_factor = 8.31342 * cls.T / (Z_Ca * FARADAY) * 1e3   # mV
        return np.array([cls.gNabar, cls.ENa, cls.gKdbar, cls.EK, cls.gAbar, cls.gCaTbar,
                         cls.gCaLbar, cls.gKCabar, cls.gLeak, cls.ELeak, cls.Cao, nernst_factor,
                         cls.taur_Cai, cls.current_to_molar_rate_Ca, cls.tau_d2, cls.thetax_d2,
                         cls.kx_d2, cls.tau_r, cls.thetax_r, cls.kx_r])


OtsukaSTN.deff = OtsukaSTN.getEffectiveDepth(OtsukaSTN.Cai0, OtsukaSTN.Vm0)
OtsukaSTN.current_to_molar_rate_Ca = PointNeuron.currentToConcentrationRate(Z_Ca, OtsukaSTN.deff)
