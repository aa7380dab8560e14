# -*- coding: utf-8 -*-
''' Thalamic point neurons: reticular (RE) and thalamo-cortical (TC).
    Parameters and kinetics as in PySONIC/neurons/thalamic.py:12-366. '''
import numpy as np

from ..core.pneuron import PointNeuron
from ..constants import Z_Ca
from ._kinetics import SodiumPotassiumKinetics, inf_tau_rates as _inf_tau_rates


class Thalamic(SodiumPotassiumKinetics, PointNeuron):
    ''' m, h, n from the shared kinetics + the T-type calcium gates s, u of the subclass '''

    @classmethod
    def effRates(cls):
        a_s, b_s = _inf_tau_rates(cls.sinf, cls.taus)
        a_u, b_u = _inf_tau_rates(cls.uinf, cls.tauu)
        return {**cls._mhn_rates(), 'alphas': a_s, 'betas': b_s, 'alphau': a_u, 'betau': b_u}

    @classmethod
    def derStates(cls):
        return {**cls._mhn_derivatives(),
                's': lambda Vm, x: (cls.sinf(Vm) - x['s']) / cls.taus(Vm),
                'u': lambda Vm, x: (cls.uinf(Vm) - x['u']) / cls.tauu(Vm)}

    @classmethod
    def steadyStates(cls):
        return {**cls._mhn_steady_states(), 's': lambda Vm: cls.sinf(Vm), 'u': lambda Vm: cls.uinf(Vm)}

    @classmethod
    def iCaT(cls, s, u, Vm):
        return cls.gCaTbar * s**2 * u * (Vm - cls.ECa)

    @classmethod
    def currents(cls):
        return {
            'iNa': lambda Vm, x: cls.iNa(x['m'], x['h'], Vm),
            'iKd': lambda Vm, x: cls.iKd(x['n'], Vm),
            'iCaT': lambda Vm, x: cls.iCaT(x['s'], x['u'], Vm),
            'iLeak': lambda Vm, _: cls.iLeak(Vm),
        }


class ThalamicRE(Thalamic):
    ''' Thalamic reticular neuron '''
    name = 'RE'
    native_id = 3
    Vm0 = -89.5
    ELeak = -90.0
    gNabar = 2000.0
    gKdbar = 200.0
    gCaTbar = 30.0
    gLeak = 0.5
    VT = -67.0
    area = 14.00e-9
    states = {'m': 'iNa activation gate', 'h': 'iNa inactivation gate', 'n': 'iKd gate',
              's': 'iCaT activation gate', 'u': 'iCaT inactivation gate'}
    rates = ['alpham', 'betam', 'alphah', 'betah', 'alphan', 'betan', 'alphas', 'betas',
             'alphau', 'betau']

    @staticmethod
    def sinf(Vm):
        return 1.0 / (1.0 + np.exp(-(Vm + 52.0) / 7.4))

    @staticmethod
    def taus(Vm):
        return (1 + 0.33 / (np.exp((Vm + 27.0) / 10.0) + np.exp(-(Vm + 102.0) / 15.0))) * 1e-3

    @staticmethod
    def uinf(Vm):
        return 1.0 / (1.0 + np.exp((Vm + 80.0) / 5.0))

    @staticmethod
    def tauu(Vm):
        return (28.3 + 0.33 / (np.exp((Vm + 48.0) / 4.0) + np.exp(-(Vm + 407.0) / 50.0))) * 1e-3

    @classmethod
    def device_params(cls):
        return np.array([cls.gNabar, cls.ENa, cls.gKdbar, cls.EK, cls.gCaTbar, cls.ECa,
                         cls.gLeak, cls.ELeak])


class ThalamoCortical(Thalamic):
    ''' Thalamo-cortical neuron '''
    name = 'TC'
    native_id = 4
    Vm0 = -61.93
    EH = -40.0
    ELeak = -70.0
    gNabar = 900.0
    gKdbar = 100.0
    gCaTbar = 20.0
    gKLeak = 0.138
    gHbar = 0.175
    gLeak = 0.1
    VT = -52.0
    Vx = 0.0
    taur_Cai = 5e-3
    Cai_min = 50e-9
    deff = 100e-9
    nCa = 4
    k1 = 2.5e22
    k2 = 0.4
    k3 = 100.0
    k4 = 1.0
    area = 29.00e-9
    states = {'m': 'iNa activation gate', 'h': 'iNa inactivation gate', 'n': 'iKd gate',
              's': 'iCaT activation gate', 'u': 'iCaT inactivation gate',
              'Cai': 'submembrane Ca2+ concentration (M)',
              'P0': 'proportion of unbound iH regulating factor',
              'O': 'iH gate open state', 'C': 'iH gate closed state'}
    rates = ['alpham', 'betam', 'alphah', 'betah', 'alphan', 'betan', 'alphas', 'betas',
             'alphau', 'betau', 'alphao', 'betao']
    current_to_molar_rate_Ca = PointNeuron.currentToConcentrationRate(Z_Ca, deff)

    @staticmethod
    def OL(O, C):
        return 1 - O - C

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
        return 1 / 3.7 * (np.exp(-(Vm + cls.Vx + 22) / 10.5) + 28.0) * 1e-3

    @staticmethod
    def oinf(Vm):
        return 1.0 / (1.0 + np.exp((Vm + 75.0) / 5.5))

    @staticmethod
    def tauo(Vm):
        return 1 / (np.exp(-14.59 - 0.086 * Vm) + np.exp(-1.87 + 0.0701 * Vm)) * 1e-3

    @classmethod
    def alphao(cls, Vm):
        return cls.oinf(Vm) / cls.tauo(Vm)

    @classmethod
    def betao(cls, Vm):
        return (1 - cls.oinf(Vm)) / cls.tauo(Vm)

    @classmethod
    def effRates(cls):
        return {**super().effRates(), 'alphao': cls.alphao, 'betao': cls.betao}

    @classmethod
    def derStates(cls):
        return {**super().derStates(),
                'Cai': lambda Vm, x: ((cls.Cai_min - x['Cai']) / cls.taur_Cai -
                                      cls.current_to_molar_rate_Ca *
                                      cls.iCaT(x['s'], x['u'], Vm)),
                'P0': lambda _, x: cls.k2 * (1 - x['P0']) - cls.k1 * x['P0'] * x['Cai']**cls.nCa,
                'O': lambda Vm, x: (cls.alphao(Vm) * x['C'] - cls.betao(Vm) * x['O'] -
                                    cls.k3 * x['O'] * (1 - x['P0']) +
                                    cls.k4 * (1 - x['O'] - x['C'])),
                'C': lambda Vm, x: cls.betao(Vm) * x['O'] - cls.alphao(Vm) * x['C']}

    @classmethod
    def steadyStates(cls):
        d = super().steadyStates()
        d['Cai'] = lambda Vm: (cls.Cai_min - cls.taur_Cai * cls.current_to_molar_rate_Ca *
                               cls.iCaT(cls.sinf(Vm), cls.uinf(Vm), Vm))
        d['P0'] = lambda Vm: cls.k2 / (cls.k2 + cls.k1 * d['Cai'](Vm)**cls.nCa)
        d['O'] = lambda Vm: (cls.k4 / (cls.k3 * (1 - d['P0'](Vm)) +
                                       cls.k4 * (1 + cls.betao(Vm) / cls.alphao(Vm))))
        d['C'] = lambda Vm: cls.betao(Vm) / cls.alphao(Vm) * d['O'](Vm)
        return d

    @classmethod
    def _quasiSteadyOthers(cls, g):
        ''' the calcium / regulation states of the steady state above, on a lookup: gates from the
            effective rates, the potential of the T-current from the effective potential lkp['V'] '''
        q = {}
        q['Cai'] = lambda lkp: (cls.Cai_min - cls.taur_Cai * cls.current_to_molar_rate_Ca *
                                cls.iCaT(g['s'](lkp), g['u'](lkp), lkp['V']))
        q['P0'] = lambda lkp: cls.k2 / (cls.k2 + cls.k1 * q['Cai'](lkp)**cls.nCa)
        q['O'] = lambda lkp: (cls.k4 / (cls.k3 * (1 - q['P0'](lkp)) +
                                        cls.k4 * (1 + lkp['betao'] / lkp['alphao'])))
        q['C'] = lambda lkp: lkp['betao'] / lkp['alphao'] * q['O'](lkp)
        return q

    @classmethod
    def iKLeak(cls, Vm):
        return cls.gKLeak * (Vm - cls.EK)

    @classmethod
    def iH(cls, O, C, Vm):
        return cls.gHbar * (O + 2 * cls.OL(O, C)) * (Vm - cls.EH)

    @classmethod
    def currents(cls):
        return {**super().currents(),
                'iKLeak': lambda Vm, x: cls.iKLeak(Vm),
                'iH': lambda Vm, x: cls.iH(x['O'], x['C'], Vm)}

    @classmethod
    def device_params(cls):
        return np.array([cls.gNabar, cls.ENa, cls.gKdbar, cls.EK, cls.gCaTbar, cls.ECa,
                         cls.gLeak, cls.ELeak, cls.gKLeak, cls.gHbar, cls.EH, cls.taur_Cai,
                         cls.Cai_min, cls.current_to_molar_rate_Ca, cls.k1, cls.k2, cls.k3,
                         cls.k4, float(cls.nCa)])
