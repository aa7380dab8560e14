# -*- coding: utf-8 -*-
''' Kinetics shared by the cortical and thalamic point neurons (Pospischil et al. 2008; the
    reference defines them twice, PySONIC/neurons/cortical.py:36-70 and thalamic.py:36-70):
    sodium (m, h) and delayed-rectifier potassium (n) rate constants relative to the spike
    threshold adjustment VT of the class, in 1/s. Expressions are kept term for term: the
    golden neuron tests compare values bit for bit. '''
import numpy as np


def inf_tau_rates(xinf, taux):
    ''' alpha = xinf / tau, beta = (1 - xinf) / tau (translators.py:317-320) '''
    return (lambda Vm: xinf(Vm) / taux(Vm)), (lambda Vm: (1 - xinf(Vm)) / taux(Vm))


class SodiumPotassiumKinetics:
    ''' mixin: needs `VT` and PointNeuron.vtrap on the class '''

    ENa = 50.0     # mV
    EK = -90.0
    ECa = 120.0
    Cm0 = 1e-2     # F/m2

    @classmethod
    def alpham(cls, Vm):
        return 0.32 * cls.vtrap(13 - (Vm - cls.VT), 4) * 1e3

    @classmethod
    def betam(cls, Vm):
        return 0.28 * cls.vtrap((Vm - cls.VT) - 40, 5) * 1e3

    @classmethod
    def alphah(cls, Vm):
        return 0.128 * np.exp(-((Vm - cls.VT) - 17) / 18) * 1e3

    @classmethod
    def betah(cls, Vm):
        return 4 / (1 + np.exp(-((Vm - cls.VT) - 40) / 5)) * 1e3

    @classmethod
    def alphan(cls, Vm):
        return 0.032 * cls.vtrap(15 - (Vm - cls.VT), 5) * 1e3

    @classmethod
    def betan(cls, Vm):
        return 0.5 * np.exp(-((Vm - cls.VT) - 10) / 40) * 1e3

    @classmethod
    def _mhn_rates(cls):
        return {f'{ab}{x}': getattr(cls, f'{ab}{x}') for x in 'mhn' for ab in ('alpha', 'beta')}

    @classmethod
    def _mhn_derivatives(cls):
        def gate(x):
            a, b = getattr(cls, f'alpha{x}'), getattr(cls, f'beta{x}')
            return lambda Vm, s: a(Vm) * (1 - s[x]) - b(Vm) * s[x]
        return {x: gate(x) for x in 'mhn'}

    @classmethod
    def _mhn_steady_states(cls):
        def ss(x):
            a, b = getattr(cls, f'alpha{x}'), getattr(cls, f'beta{x}')
            return lambda Vm: a(Vm) / (a(Vm) + b(Vm))
        return {x: ss(x) for x in 'mhn'}

    @classmethod
    def iNa(cls, m, h, Vm):
        return cls.gNabar * m**3 * h * (Vm - cls.ENa)

    @classmethod
    def iKd(cls, n, Vm):
        return cls.gKdbar * n**4 * (Vm - cls.EK)

    @classmethod
    def iLeak(cls, Vm):
        return cls.gLeak * (Vm - cls.ELeak)
