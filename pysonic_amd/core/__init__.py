# -*- coding: utf-8 -*-
from .batches import Batch, LogBatch
from .model import Model
from .stimobj import StimObject
from .drives import Drive, XDrive, AcousticDrive
from .protocols import (TimeProtocol, CustomProtocol, PulsedProtocol, BurstProtocol,
                        BalancedPulsedProtocol, getPulseTrainProtocol)
from .timeseries import TimeSeries
from .lookups import Lookup, EffectiveVariablesLookup, EffectiveVariablesDict
from .pneuron import PointNeuron
from .bls import BilayerSonophore
from .nbls import NeuronalBilayerSonophore, DrivenNeuronalBilayerSonophore

__all__ = ['Batch', 'LogBatch', 'Model', 'StimObject', 'Drive', 'XDrive', 'AcousticDrive',
           'TimeProtocol', 'CustomProtocol', 'PulsedProtocol', 'BurstProtocol',
           'BalancedPulsedProtocol', 'getPulseTrainProtocol', 'TimeSeries', 'Lookup',
           'EffectiveVariablesLookup', 'EffectiveVariablesDict', 'PointNeuron',
           'BilayerSonophore', 'NeuronalBilayerSonophore', 'DrivenNeuronalBilayerSonophore']
