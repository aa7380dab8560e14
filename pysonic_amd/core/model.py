# -*- coding: utf-8 -*-
''' Abstract base of the simulable models. The reference wraps simulate() in a stack of decorators
    (PySONIC/core/model.py:110-215: input check, description log, meta-data, titration of an
    unresolved drive, spike count log); here those steps are part of the batched entry point
    (NeuronalBilayerSonophore._batched_simulate), which applies them to a whole queue at once.
    What remains is the queue decoration for file output and the save / reload helpers. '''
import abc
from functools import wraps

from .batches import Batch
from ..utils import logger, filecode, simAndSave, loadData


class Model(metaclass=abc.ABCMeta):

    @property
    @abc.abstractmethod
    def tscale(self):
        ''' relevant temporal scale of the model ('ms', 'us') '''

    @property
    @abc.abstractmethod
    def simkey(self):
        ''' keyword of the simulation type in file codes ('ASTIM', ...) '''

    @abc.abstractmethod
    def copy(self):
        ''' an independent object with the same parameters '''

    @abc.abstractmethod
    def filecodes(self, *args):
        ''' ordered dictionary of the file-code fragments of one simulation '''

    def filecode(self, *args):
        return filecode(self, *args)

    @staticmethod
    def checkOutputDir(queuefunc):
        ''' Queue builders accept outputdir= / overwrite=: with an output directory every item
            becomes (args, {'overwrite': ..., 'outputdir': ...}) for simAndSave
            (model.py:85-108); without one, long queues get a warning. '''
        @wraps(queuefunc)
        def wrapper(self, *args, **kwargs):
            queue = queuefunc(self, *args, **kwargs)
            outputdir = kwargs.get('outputdir')
            if outputdir is None:
                if len(queue) > 5:
                    logger.warning('Running more than 5 simulations without file saving')
                return queue
            extra = {'overwrite': kwargs.get('overwrite', True), 'outputdir': outputdir}
            out = []
            for item in queue:
                pos, kw = Batch.resolve(item)
                out.append((pos, {**kw, **extra}))
            return out
        return wrapper

    def simAndSave(self, *args, **kwargs):
        return simAndSave(self, *args, **kwargs)

    def getOutput(self, *args, **kwargs):
        ''' run (or find on disk) and load '''
        return loadData(self.simAndSave(*args, overwrite=False, **kwargs))
