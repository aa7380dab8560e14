# -*- coding: utf-8 -*-
''' Axon membrane models whose states are all alpha / beta voltage gates and whose currents are
    products of gate powers: Hodgkin-Huxley segment (PySONIC/neurons/hh.py:10-129), Sweeney node
    (sweeney.py:10-106), MRG node (mrg.py:10-172), Sundt segment (sundt.py:12-182), and the
    Frankenhaeuser-Huxley node (fh.py:11-159) whose currents use Goldman-Hodgkin-Katz driving forces. They run on the device through the data-driven
    gated model (csrc/sonic_models.hpp: GatedModel) with a faster output step than the cortical
    and thalamic neurons. '''
import numpy as np

from ..core.pneuron import PointNeuron
from ..constants import FARADAY, Rg


class AlphaBetaNeuron(PointNeuron):
    ''' Point neuron defined by its alpha<x> / beta<x> class methods and by `conductances`:
        current name -> (maximal conductance attribute, reversal attribute, {gate: exponent}).
        Everything the PointNeuron API needs is derived from those two. '''
    conductances = {}
    ghk_currents = {}      # current name -> (intracellular, extracellular concentration attributes)
    dt_factor = 1.0

    @classmethod
    def effRates(cls):
        out = {}
        for x in cls.states:
            out[f'alpha{x}'] = getattr(cls, f'alpha{x}')
            out[f'beta{x}'] = getattr(cls, f'beta{x}')
        return out

    @classmethod
    def derStates(cls):
        def der(x):
            a, b = getattr(cls, f'alpha{x}'), getattr(cls, f'beta{x}')
            return lambda Vm, s: a(Vm) * (1 - s[x]) - b(Vm) * s[x]
        return {x: der(x) for x in cls.states}

    @classmethod
    def steadyStates(cls):
        def ss(x):
            a, b = getattr(cls, f'alpha{x}'), getattr(cls, f'beta{x}')
            return lambda Vm: a(Vm) / (a(Vm) + b(Vm))
        return {x: ss(x) for x in cls.states}

    @classmethod
    def currents(cls):
        def gated(iname, gname, ename, powers):
            def i(Vm, s):
                g = getattr(cls, gname)
                for x, e in powers.items():
                    g = g * s[x]**e
                if iname in cls.ghk_currents:
                    Cin, Cout = [getattr(cls, k) for k in cls.ghk_currents[iname]]
                    return g * cls.ghkDrive(Vm, 1, Cin, Cout, cls.T)
                return g * (Vm - getattr(cls, ename))
            return i
        out = {k: gated(k, *spec) for k, spec in cls.conductances.items()}
        out['iLeak'] = lambda Vm, _: cls.gLeak * (Vm - cls.ELeak)
        return out

    def chooseTimeStep(self):
        return super().chooseTimeStep() * self.dt_factor

    @classmethod
    def device_params(cls):
        ''' parameter block of GatedModel<n_states>: gLeak, ELeak, g[4], E[4], ghk[4], Cin[4],
            Cout[4], exponents[4][n] '''
        names = list(cls.states)
        g, E, ghk, Cin, Cout = [np.zeros(4) for _ in range(5)]
        expo = np.zeros((4, len(names)))
        for c, (iname, (gname, ename, powers)) in enumerate(cls.conductances.items()):
            g[c] = getattr(cls, gname)
            if iname in cls.ghk_currents:
                ghk[c] = 1.
                Cin[c], Cout[c] = [getattr(cls, k) for k in cls.ghk_currents[iname]]
            else:
                E[c] = getattr(cls, ename)
            for x, e in powers.items():
                expo[c, names.index(x)] = e
        return np.concatenate(([cls.gLeak, cls.ELeak], g, E, ghk, Cin, Cout, expo.ravel()))


class HodgkinHuxleySegment(AlphaBetaNeuron):
    ''' Unmyelinated giant squid axon segment (Hodgkin & Huxley 1952), rates scaled to 36 C '''
    name = 'HHseg'
    native_id = 7
    Cm0 = 1e-2
    Vm0 = -65.0
    ENa, EK, ELeak = 50.0, -77.0, -54.3
    gNabar, gKdbar, gLeak = 1200.0, 360.0, 3.0
    celsius_HH = 6.3
    q10 = 3**((PointNeuron.celsius - celsius_HH) / 10.)
    states = {'m': 'iNa activation gate', 'h': 'iNa inactivation gate', 'n': 'iKd gate'}
    rates = ['alpham', 'betam', 'alphah', 'betah', 'alphan', 'betan']
    conductances = {'iNa': ('gNabar', 'ENa', {'m': 3, 'h': 1}), 'iKd': ('gKdbar', 'EK', {'n': 4})}
    dt_factor = 1e-1

    @classmethod
    def alpham(cls, Vm):
        return cls.q10 * 0.1 * cls.vtrap(-(Vm + 40), 10) * 1e3

    @classmethod
    def betam(cls, Vm):
        return cls.q10 * 4 * np.exp(-(Vm + 65) / 18) * 1e3

    @classmethod
    def alphah(cls, Vm):
        return cls.q10 * 0.07 * np.exp(-(Vm + 65) / 20) * 1e3

    @classmethod
    def betah(cls, Vm):
        return cls.q10 * 1.0 / (np.exp(-(Vm + 35) / 10) + 1) * 1e3

    @classmethod
    def alphan(cls, Vm):
        return cls.q10 * 0.01 * cls.vtrap(-(Vm + 55), 10) * 1e3

    @classmethod
    def betan(cls, Vm):
        return cls.q10 * 0.125 * np.exp(-(Vm + 65) / 80) * 1e3


class SweeneyNode(AlphaBetaNeuron):
    ''' Mammalian (rabbit) myelinated motor fiber node (Sweeney et al. 1987, Basser & Roth 1991) '''
    name = 'SWnode'
    native_id = 8
    Cm0 = 2.5e-2
    Vm0 = -80.0
    ENa, ELeak = 35.64, -80.01
    gNabar, gLeak = 1445e1, 128e1
    states = {'m': 'iNa activation gate', 'h': 'iNa inactivation gate'}
    rates = ['alpham', 'betam', 'alphah', 'betah']
    conductances = {'iNa': ('gNabar', 'ENa', {'m': 2, 'h': 1})}
    dt_factor = 1e-2

    @classmethod
    def alpham(cls, Vm):
        return (126 + 0.363 * Vm) / (1 + np.exp(-(Vm + 49) / 5.3)) * 1e3

    @classmethod
    def betam(cls, Vm):
        return cls.alpham(Vm) / (np.exp((Vm + 56.2) / 4.17))

    @classmethod
    def betah(cls, Vm):
        return 15.6 / (1 + np.exp(-(Vm + 56) / 10)) * 1e3

    @classmethod
    def alphah(cls, Vm):
        return cls.betah(Vm) / np.exp((Vm + 74.5) / 5)


class MRGNode(AlphaBetaNeuron):
    ''' Mammalian myelinated fiber node (McIntyre, Richardson & Grill 2002) '''
    name = 'MRGnode'
    native_id = 9
    Cm0 = 2e-2
    Vm0 = -80.0
    ENa, EK, ELeak = 50.0, -90.0, -90.0
    gNafbar, gNapbar, gKsbar, gLeak = 3e4, 100.0, 800.0, 70.0
    celsius_Schwarz, celsius_Ks = 20.0, 36.0
    mhshift, vtraub = 3.0, -80.0
    q10_mp = 2.2**((PointNeuron.celsius - celsius_Schwarz) / 10)
    q10_h = 2.9**((PointNeuron.celsius - celsius_Schwarz) / 10)
    q10_s = 3.0**((PointNeuron.celsius - celsius_Ks) / 10)
    states = {'m': 'iNaf activation gate', 'h': 'iNaf inactivation gate',
              'p': 'iNap activation gate', 's': 'iKs activation gate'}
    rates = ['alpham', 'betam', 'alphah', 'betah', 'alphap', 'betap', 'alphas', 'betas']
    conductances = {'iNaf': ('gNafbar', 'ENa', {'m': 3, 'h': 1}), 'iNap': ('gNapbar', 'ENa', {'p': 3}),
                    'iKs': ('gKsbar', 'EK', {'s': 1})}
    dt_factor = 1e-2

    @classmethod
    def alpham(cls, Vm):
        return cls.q10_mp * 1.86 * cls.vtrap(-(Vm + cls.mhshift + 18.4), 10.3) * 1e3

    @classmethod
    def betam(cls, Vm):
        return cls.q10_mp * 0.086 * cls.vtrap(Vm + cls.mhshift + 22.7, 9.16) * 1e3

    @classmethod
    def alphah(cls, Vm):
        return cls.q10_h * 0.062 * cls.vtrap(Vm + cls.mhshift + 111.0, 11.0) * 1e3

    @classmethod
    def betah(cls, Vm):
        return cls.q10_h * 2.3 / (1 + np.exp(-(Vm + cls.mhshift + 28.8) / 13.4)) * 1e3

    @classmethod
    def alphap(cls, Vm):
        return cls.q10_mp * 0.01 * cls.vtrap(-(Vm + 27.), 10.2) * 1e3

    @classmethod
    def betap(cls, Vm):
        return cls.q10_mp * 0.00025 * cls.vtrap(Vm + 34., 10.) * 1e3

    @classmethod
    def alphas(cls, Vm):
        return cls.q10_s * 0.3 / (1 + np.exp(-(Vm - cls.vtraub - 27.) / 5.)) * 1e3

    @classmethod
    def betas(cls, Vm):
        return cls.q10_s * 0.03 / (1 + np.exp(-(Vm - cls.vtraub + 10.) / 1.)) * 1e3


class SundtSegment(AlphaBetaNeuron):
    ''' Unmyelinated C-fiber segment (Sundt et al. 2015): Traub-type sodium gates, Borg-Graham
        potassium gates; the leakage reversal potential balances the other currents at rest
        (sundt.py:55-67). '''
    name = 'SUseg'
    native_id = 10
    Cm0 = 1e-2
    Vm0 = -60.0
    ENa, EK = 55.0, -90.0
    gNabar, gKdbar, gLeak = 400.0, 400.0, 1.0
    Vrest_Traub, mshift, hshift = -65.0, -6.0, 6.0
    q10_Traub = 3**((PointNeuron.celsius - 30.0) / 10)
    q10_BG = 3**((PointNeuron.celsius - 30.0) / 10)
    states = {'m': 'iNa activation gate', 'h': 'iNa inactivation gate',
              'n': 'iKd activation gate', 'l': 'iKd inactivation gate'}
    rates = ['alpham', 'betam', 'alphah', 'betah', 'alphan', 'betan', 'alphal', 'betal']
    conductances = {'iNa': ('gNabar', 'ENa', {'m': 3, 'h': 1}), 'iKd': ('gKdbar', 'EK', {'n': 3, 'l': 1})}
    dt_factor = 1e-2

    @classmethod
    def _xBG(cls, Vref, Vm):
        return (Vm - Vref) * FARADAY / (Rg * cls.T) * 1e-3

    @classmethod
    def _aBG(cls, a0, zeta, gamma, Vref, Vm):
        return a0 * np.exp(-zeta * gamma * cls._xBG(Vref, Vm))

    @classmethod
    def _bBG(cls, b0, zeta, gamma, Vref, Vm):
        return b0 * np.exp(zeta * (1 - gamma) * cls._xBG(Vref, Vm))

    @classmethod
    def alpham(cls, Vm):
        return cls.q10_Traub * 0.32 * cls.vtrap(13.1 - (Vm - cls.Vrest_Traub + cls.mshift), 4) * 1e3

    @classmethod
    def betam(cls, Vm):
        return cls.q10_Traub * 0.28 * cls.vtrap((Vm - cls.Vrest_Traub + cls.mshift) - 40.1, 5) * 1e3

    @classmethod
    def alphah(cls, Vm):
        return cls.q10_Traub * 0.128 * np.exp((17.0 - (Vm - cls.Vrest_Traub + cls.hshift)) / 18) * 1e3

    @classmethod
    def betah(cls, Vm):
        return cls.q10_Traub * 4 / (1 + np.exp((40.0 - (Vm - cls.Vrest_Traub + cls.hshift)) / 5)) * 1e3

    @classmethod
    def alphan(cls, Vm):
        return cls.q10_BG * cls._aBG(0.03, -5, 0.4, -32., Vm) * 1e3

    @classmethod
    def betan(cls, Vm):
        return cls.q10_BG * cls._bBG(0.03, -5, 0.4, -32., Vm) * 1e3

    @classmethod
    def alphal(cls, Vm):
        return cls.q10_BG * cls._aBG(0.001, 2, 1., -61., Vm) * 1e3

    @classmethod
    def betal(cls, Vm):
        return cls.q10_BG * cls._bBG(0.001, 2, 1., -61., Vm) * 1e3

    @staticmethod
    def getNSpikes(data):
        from ..postpro import detectSpikes
        return detectSpikes(data, mph=-8.0e-5)[0].size


def _sundt_leak_reversal(cls):
    ''' ELeak such that the net current vanishes at rest (sundt.py:57-66) '''
    ss = {k: f(cls.Vm0) for k, f in cls.steadyStates().items()}
    cls.ELeak = 0.
    inet = sum(f(cls.Vm0, ss) for k, f in cls.currents().items() if k != 'iLeak')
    return cls.Vm0 + inet / cls.gLeak


SundtSegment.ELeak = _sundt_leak_reversal(SundtSegment)


class FrankenhaeuserHuxleyNode(AlphaBetaNeuron):
    ''' Xenopus myelinated fiber node (Frankenhaeuser & Huxley 1964): permeabilities and
        Goldman-Hodgkin-Katz driving forces for the sodium, potassium and non-specific currents '''
    name = 'FHnode'
    native_id = 11
    Cm0 = 2e-2
    Vm0 = -70.0
    ELeak, gLeak = -69.974, 300.3
    pNabar, pKbar, pPbar = 8e-5, 1.2e-5, .54e-5
    Nai, Nao, Ki, Ko = 13.74e-3, 114.5e-3, 120e-3, 2.5e-3
    q10 = 3**((PointNeuron.celsius - 20.0) / 10)
    states = {'m': 'iNa activation gate', 'h': 'iNa inactivation gate', 'n': 'iKd gate', 'p': 'iP gate'}
    rates = ['alpham', 'betam', 'alphah', 'betah', 'alphan', 'betan', 'alphap', 'betap']
    conductances = {'iNa': ('pNabar', None, {'m': 2, 'h': 1}), 'iKd': ('pKbar', None, {'n': 2}),
                    'iP': ('pPbar', None, {'p': 2})}
    ghk_currents = {'iNa': ('Nai', 'Nao'), 'iKd': ('Ki', 'Ko'), 'iP': ('Nai', 'Nao')}
    dt_factor = 1e-1

    @staticmethod
    def efun(x):
        return x / (np.exp(x) - 1)

    @classmethod
    def ghkDrive(cls, Vm, Z_ion, Cion_in, Cion_out, T):
        ''' electrochemical driving force of one ion species (pneuron.py:361-375), mC/m3 '''
        x = Z_ion * FARADAY * Vm / (Rg * T) * 1e-3
        return FARADAY * (Cion_in * cls.efun(-x) - Cion_out * cls.efun(x)) * 1e6

    @classmethod
    def alpham(cls, Vm):
        return cls.q10 * 0.36 * cls.vtrap(22. - (Vm - cls.Vm0), 3.) * 1e3

    @classmethod
    def betam(cls, Vm):
        return cls.q10 * 0.4 * cls.vtrap(Vm - cls.Vm0 - 13., 20.) * 1e3

    @classmethod
    def alphah(cls, Vm):
        return cls.q10 * 0.1 * cls.vtrap(Vm - cls.Vm0 + 10.0, 6.) * 1e3

    @classmethod
    def betah(cls, Vm):
        return cls.q10 * 4.5 / (np.exp((45. - (Vm - cls.Vm0)) / 10.) + 1) * 1e3

    @classmethod
    def alphan(cls, Vm):
        return cls.q10 * 0.02 * cls.vtrap(35. - (Vm - cls.Vm0), 10.0) * 1e3

    @classmethod
    def betan(cls, Vm):
        return cls.q10 * 0.05 * cls.vtrap(Vm - cls.Vm0 - 10., 10.) * 1e3

    @classmethod
    def alphap(cls, Vm):
        return cls.q10 * 0.006 * cls.vtrap(40. - (Vm - cls.Vm0), 10.0) * 1e3

    @classmethod
    def betap(cls, Vm):
        return cls.q10 * 0.09 * cls.vtrap(Vm - cls.Vm0 + 25., 20.) * 1e3
