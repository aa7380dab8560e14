# -*- coding: utf-8 -*-
''' Cortical point neurons (Pospischil et al. 2008): regular spiking, fast spiking,
    low-threshold spiking. Parameters and kinetics as in PySONIC/neurons/cortical.py:12-303. '''
import numpy as np

from ..core.pneuron import PointNeuron


from ._kinetics import SodiumPotassiumKinetics, inf_tau_rates as _inf_tau_rates


class Cortical(SodiumPotassiumKinetics, PointNeuron):
    ''' m, h, n from the shared kinetics + the slow non-inactivating potassium gate p '''

    @staticmethod
    def pinf(Vm):
        return 1.0 / (1 + np.exp(-(Vm + 35) / 10))

    @classmethod
    def taup(cls, Vm):
        return cls.TauMax / (3.3 * np.exp((Vm + 35) / 20) + np.exp(-(Vm + 35) / 20))

    @classmethod
    def effRates(cls):
        ap, bp = _inf_tau_rates(cls.pinf, cls.taup)
        return {**cls._mhn_rates(), 'alphap': ap, 'betap': bp}

    @classmethod
    def derStates(cls):
        return {**cls._mhn_derivatives(),
                'p': lambda Vm, x: (cls.pinf(Vm) - x['p']) / cls.taup(Vm)}

    @classmethod
    def steadyStates(cls):
        return {**cls._mhn_steady_states(), 'p': lambda Vm: cls.pinf(Vm)}

    @classmethod
    def iM(cls, p, Vm):
        return cls.gMbar * p * (Vm - cls.EK)

    @classmethod
    def currents(cls):
        return {
            'iNa': lambda Vm, x: cls.iNa(x['m'], x['h'], Vm),
            'iKd': lambda Vm, x: cls.iKd(x['n'], Vm),
            'iM': lambda Vm, x: cls.iM(x['p'], Vm),
            'iLeak': lambda Vm, _: cls.iLeak(Vm),
        }

    @classmethod
    def device_params(cls):
        return np.array([cls.gNabar, cls.ENa, cls.gKdbar, cls.EK, cls.gMbar, cls.gLeak,
                         cls.ELeak])


class CorticalRS(Cortical):
    ''' Cortical regular spiking neuron '''
    name = 'RS'
    native_id = 0
    Vm0 = -71.9
    ELeak = -70.3
    gNabar = 560.0
    gKdbar = 60.0
    gMbar = 0.75
    gLeak = 0.205
    VT = -56.2
    TauMax = 0.608
    area = 11.84e-9
    states = {'m': 'iNa activation gate', 'h': 'iNa inactivation gate', 'n': 'iKd gate',
              'p': 'iM gate'}
    rates = ['alpham', 'betam', 'alphah', 'betah', 'alphan', 'betan', 'alphap', 'betap']


class CorticalFS(Cortical):
    ''' Cortical fast-spiking neuron '''
    name = 'FS'
    native_id = 1
    Vm0 = -71.4
    ELeak = -70.4
    gNabar = 580.0
    gKdbar = 39.0
    gMbar = 0.787
    gLeak = 0.38
    VT = -57.9
    TauMax = 0.502
    area = 10.17e-9
    states = {'m': 'iNa activation gate', 'h': 'iNa inactivation gate', 'n': 'iKd gate',
              'p': 'iM gate'}
    rates = ['alpham', 'betam', 'alphah', 'betah', 'alphan', 'betan', 'alphap', 'betap']


class CorticalLTS(Cortical):
    ''' Cortical low-threshold spiking neuron '''
    name = 'LTS'
    native_id = 2
    Vm0 = -54.0
    ELeak = -50.0
    gNabar = 500.0
    gKdbar = 40.0
    gMbar = 0.28
    gCaTbar = 4.0
    gLeak = 0.19
    VT = -50.0
    TauMax = 4.0
    Vx = -7.0
    area = 25.00e-9
    states = {'m': 'iNa activation gate', 'h': 'iNa inactivation gate', 'n': 'iKd gate',
              'p': 'iM gate', 's': 'iCaT activation gate', 'u': 'iCaT inactivation gate'}
    rates = ['alpham', 'betam', 'alphah', 'betah', 'alphan', 'betan', 'alphap', 'betap',
             'alphas', 'betas', 'alphau', 'betau']

    @classmethod
    def sinf(cls, Vm):
        return 1.0 / (1.0 + np.exp(-(Vm + cls.Vx + 57.0) / 6.2))

    @classmethod
    def taus(cls, Vm):
        x = np.exp(-(Vm + cls.Vx + 132.0) / 16.7) + np.exp((Vm + cls.Vx + 16.8) / 18.2)
        return 1.0 / 3.7 * (0.612 + 1.0 / x) * 1e-3

    @classmethod
    def uinf(cls, Vm):
        return 1.0 / (1.0 + np.exp((Vm + cls.Vx + 81.0) / 4.0))

    @classmethod
    def tauu(cls, Vm):
        if Vm + cls.Vx < -80.0:
            return 1.0 / 3.7 * np.exp((Vm + cls.Vx + 467.0) / 66.6) * 1e-3
        return 1.0 / 3.7 * (np.exp(-(Vm + cls.Vx + 22) / 10.5) + 28.0) * 1e-3

    @classmethod
    def effRates(cls):
        a_s, b_s = _inf_tau_rates(cls.sinf, cls.taus)
        a_u, b_u = _inf_tau_rates(cls.uinf, cls.tauu)
        return {**super().effRates(), 'alphas': a_s, 'betas': b_s, 'alphau': a_u, 'betau': b_u}

    @classmethod
    def derStates(cls):
        return {**super().derStates(),
                's': lambda Vm, x: (cls.sinf(Vm) - x['s']) / cls.taus(Vm),
                'u': lambda Vm, x: (cls.uinf(Vm) - x['u']) / cls.tauu(Vm)}

    @classmethod
    def steadyStates(cls):
        return {**super().steadyStates(), 's': lambda Vm: cls.sinf(Vm),
                'u': lambda Vm: cls.uinf(Vm)}

    @classmethod
    def iCaT(cls, s, u, Vm):
        return cls.gCaTbar * s**2 * u * (Vm - cls.ECa)

    @classmethod
    def currents(cls):
        return {**super().currents(), 'iCaT': lambda Vm, x: cls.iCaT(x['s'], x['u'], Vm)}

    @classmethod
    def device_params(cls):
        return np.array([cls.gNabar, cls.ENa, cls.gKdbar, cls.EK, cls.gMbar, cls.gLeak,
                         cls.ELeak, cls.gCaTbar, cls.ECa])


class CorticalIB(Cortical):
    ''' Cortical intrinsically bursting neuron (PySONIC/neurons/cortical.py:307-400): the regular
        spiking set of currents plus a high-threshold (L-type) calcium current with alpha / beta
        gates q, r. On the device it shares the six-gate cortical model with LTS (same current
        form g x1^2 x2 (Vm - ECa)). '''
    name = 'IB'
    native_id = 6
    Vm0 = -71.4
    ELeak = -70.0
    gNabar = 500.0
    gKdbar = 50.0
    gMbar = 0.3
    gCaLbar = 1.0
    gLeak = 0.1
    VT = -56.2
    TauMax = 0.608
    area = 28.95e-9
    states = {'m': 'iNa activation gate', 'h': 'iNa inactivation gate', 'n': 'iKd gate',
              'p': 'iM gate', 'q': 'iCaL activation gate', 'r': 'iCaL inactivation gate'}
    rates = ['alpham', 'betam', 'alphah', 'betah', 'alphan', 'betan', 'alphap', 'betap',
             'alphaq', 'betaq', 'alphar', 'betar']

    @classmethod
    def alphaq(cls, Vm):
        return 0.055 * cls.vtrap(-(Vm + 27), 3.8) * 1e3

    @staticmethod
    def betaq(Vm):
        return 0.94 * np.exp(-(Vm + 75) / 17) * 1e3

    @staticmethod
    def alphar(Vm):
        return 0.000457 * np.exp(-(Vm + 13) / 50) * 1e3

    @staticmethod
    def betar(Vm):
        return 0.0065 / (np.exp(-(Vm + 15) / 28) + 1) * 1e3

    @classmethod
    def effRates(cls):
        return {**super().effRates(), 'alphaq': cls.alphaq, 'betaq': cls.betaq,
                'alphar': cls.alphar, 'betar': cls.betar}

    @classmethod
    def derStates(cls):
        gate = lambda a, b, k: (lambda Vm, x: a(Vm) * (1 - x[k]) - b(Vm) * x[k])   # noqa: E731
        return {**super().derStates(), 'q': gate(cls.alphaq, cls.betaq, 'q'),
                'r': gate(cls.alphar, cls.betar, 'r')}

    @classmethod
    def steadyStates(cls):
        return {**super().steadyStates(),
                'q': lambda Vm: cls.alphaq(Vm) / (cls.alphaq(Vm) + cls.betaq(Vm)),
                'r': lambda Vm: cls.alphar(Vm) / (cls.alphar(Vm) + cls.betar(Vm))}

    @classmethod
    def iCaL(cls, q, r, Vm):
        return cls.gCaLbar * q**2 * r * (Vm - cls.ECa)

    @classmethod
    def currents(cls):
        return {**super().currents(), 'iCaL': lambda Vm, x: cls.iCaL(x['q'], x['r'], Vm)}

    @classmethod
    def device_params(cls):
        return np.array([cls.gNabar, cls.ENa, cls.gKdbar, cls.EK, cls.gMbar, cls.gLeak,
                         cls.ELeak, cls.gCaLbar, cls.ECa])
