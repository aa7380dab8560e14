# -*- coding: utf-8 -*-
from .batches import Batch
from .model import Model
from .stimobj import StimObject
from .drives import Drive, XDrive, ElectricDrive, AcousticDrive
from .protocols import (TimeProtocol, CustomProtocol, PulsedProtocol, BurstProtocol,
                        BalancedPulsedProtocol, getPulseTrainProtocol)
from .timeseries import TimeSeries
from .lookups import Lookup, EffectiveVariablesLookup, EffectiveVariablesDict
from .pneuron import PointNeuron
from .bls import BilayerSonophore
from .nbls import NeuronalBilayerSonophore, DrivenNeuronalBilayerSonophore

__all__ = ['Batch', 'Model', 'StimObject', 'Drive', 'XDrive', 'ElectricDrive', 'AcousticDrive',
           'TimeProtocol', 'CustomProtocol', 'PulsedProtocol', 'BurstProtocol',
           'BalancedPulsedProtocol', 'getPulseTrainProtocol', 'TimeSeries', 'Lookup',
           'EffectiveVariablesLookup', 'EffectiveVariablesDict', 'PointNeuron',
           'BilayerSonophore', 'NeuronalBilayerSonophore', 'DrivenNeuronalBilayerSonophore']
