# -*- coding: utf-8 -*-
''' PointNeuron base class -- API of PySONIC/core/pneuron.py:22-601 restricted to what the
    acoustic (SONIC / NICE) path needs on the host: identity, resting values, state and rate
    lists, steady states (initial conditions), charge bounds, time step, and numpy versions of
    the rate / current functions (used to build lookups and for post-processing).

    Neurons are described as DATA + plain functions (pysonic_amd/neurons/*.py); there is no source
    rewriting (the reference's translators.py): the set of effective rates of each neuron
    (`rates`) and its device parameter vector are stated explicitly.
'''
import abc

import numpy as np

from ..constants import (FARADAY, Rg, CELSIUS_2_KELVIN, DT_EFFECTIVE, TMIN_STABILIZATION,
                         QSS_Q_DIV_THR)


class PointNeuron(metaclass=abc.ABCMeta):

    tscale = 'ms'
    simkey = 'ESTIM'
    celsius = 36.0
    T = celsius + CELSIUS_2_KELVIN

    # subclasses define: name, Cm0, Vm0, states (dict name -> description), rates (list),
    # native_id (include/pysonic_amd.h) and the functions below

    def __repr__(self):
        return self.__class__.__name__

    def copy(self):
        return self.__class__()

    def __eq__(self, other):
        return isinstance(other, PointNeuron) and self.name == other.name

    def __hash__(self):
        return hash(self.name)

    @property
    def Qm0(self):
        return self.Cm0 * self.Vm0 * 1e-3   # C/m2

    @property
    def meta(self):
        return {'neuron': self.name}

    @classmethod
    def statesNames(cls):
        return list(cls.states.keys())

    @property
    def Qbounds(self):
        ''' Physiological charge range used for lookups (pneuron.py:423-426). '''
        return np.array([np.round(self.Vm0 - 35.0), 50.0]) * self.Cm0 * 1e-3

    def chooseTimeStep(self):
        return DT_EFFECTIVE

    @property
    def is_passive(self):
        return False

    # ---- kinetics -------------------------------------------------------------------------
    @classmethod
    @abc.abstractmethod
    def effRates(cls):
        ''' {rate name: function(Vm)} in lookup-table order. '''
        raise NotImplementedError

    @classmethod
    @abc.abstractmethod
    def steadyStates(cls):
        ''' {state: function(Vm)} '''
        raise NotImplementedError

    @classmethod
    @abc.abstractmethod
    def derStates(cls):
        ''' {state: function(Vm, states dict)} true (non-effective) derivatives '''
        raise NotImplementedError

    @classmethod
    @abc.abstractmethod
    def currents(cls):
        ''' {current: function(Vm, states dict)} in mA/m2 '''
        raise NotImplementedError

    @classmethod
    @abc.abstractmethod
    def device_params(cls):
        ''' Parameter vector handed to the native library (order: csrc/sonic_models.hpp). '''
        raise NotImplementedError

    @classmethod
    def iNet(cls, Vm, states):
        return sum([cfunc(Vm, states) for cfunc in cls.currents().values()])

    @classmethod
    def getEffRates(cls, Vm):
        ''' Cycle-averaged rates for a potential vector (pneuron.py:268-271). '''
        return {k: np.mean(np.vectorize(v)(Vm)) for k, v in cls.effRates().items()}

    @classmethod
    def getSteadyStates(cls, Vm):
        return np.array([cls.steadyStates()[k](Vm) for k in cls.statesNames()])

    @classmethod
    def quasiSteadyStates(cls):
        ''' {state: f(lkp)}: the steady state of every state evaluated on EFFECTIVE (cycle-averaged) rate
            constants instead of a membrane potential -- what the reference obtains by rewriting its
            steadyStates lambdas (translators.py:374-388). A voltage-gated state is alpha / (alpha + beta) of
            its lookup entries; neurons with other states override `_quasiSteadyOthers`. '''
        d = {}
        for k in cls.statesNames():
            if f'alpha{k}' in cls.rates:
                d[k] = lambda lkp, k=k: lkp[f'alpha{k}'] / (lkp[f'alpha{k}'] + lkp[f'beta{k}'])
        d.update(cls._quasiSteadyOthers(d))
        missing = [k for k in cls.statesNames() if k not in d]
        if missing:
            raise NotImplementedError(f'quasi-steady states of {missing} ({cls.name}) are not defined on lookups')
        return {k: d[k] for k in cls.statesNames()}

    @classmethod
    def _quasiSteadyOthers(cls, gates):
        ''' quasi-steady states of the states that are not voltage-gated, given those of the gates '''
        return {}

    @classmethod
    def getCurrentsNames(cls):
        return list(cls.currents().keys())

    @classmethod
    def isVoltageGated(cls, state):
        return f'alpha{state.lower()}' in cls.rates

    # ---- shared rate-function building blocks (pneuron.py:328-413) --------------------------
    @staticmethod
    def currentToConcentrationRate(z_ion, depth):
        return 1e-6 / (z_ion * depth * FARADAY)

    @staticmethod
    def nernst(z_ion, Cion_in, Cion_out, T):
        return (Rg * T) / (z_ion * FARADAY) * np.log(Cion_out / Cion_in) * 1e3

    @staticmethod
    def vtrap(x, y):
        return x / (np.exp(x / y) - 1)

    # ---- output analysis (pneuron.py:544-594) ------------------------------------------------
    @staticmethod
    def getNSpikes(data):
        from ..postpro import detectSpikes
        return detectSpikes(data)[0].size

    @staticmethod
    def getStabilizationValue(data):
        t, Qm = [data[key].values for key in ['t', 'Qm']]
        if t.max() <= TMIN_STABILIZATION:
            raise ValueError('solution length is too short to assess stabilization')
        Qm = Qm[t > TMIN_STABILIZATION]
        return Qm[-1] if np.ptp(Qm) < QSS_Q_DIV_THR else np.nan

    @classmethod
    def isExcited(cls, data):
        return cls.getNSpikes(data) > 0

    @classmethod
    def isSilenced(cls, data):
        return not np.isnan(cls.getStabilizationValue(data))

    @classmethod
    def titrationFunc(cls, *args, **kwargs):
        return cls.isExcited(*args, **kwargs)
