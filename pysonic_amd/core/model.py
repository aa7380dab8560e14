# -*- coding: utf-8 -*-
''' Abstract Model + the decorator stack wrapped around simulate() -- semantics of
    PySONIC/core/model.py:20-228 (checkSimParams, logDesc, addMeta, checkTitrate, logNSpikes,
    checkOutputDir, simAndSave, getOutput). '''
import abc
from functools import wraps

import numpy as np

from .batches import Batch
from ..utils import (logger, timer, getMeta, alignWithMethodDef, filecode, simAndSave, loadData,
                     si_format)


class Model(metaclass=abc.ABCMeta):

    @property
    @abc.abstractmethod
    def tscale(self):
        raise NotImplementedError

    @property
    @abc.abstractmethod
    def simkey(self):
        raise NotImplementedError

    @abc.abstractmethod
    def __repr__(self):
        raise NotImplementedError

    @abc.abstractmethod
    def copy(self):
        raise NotImplementedError

    @abc.abstractmethod
    def filecodes(self, *args):
        raise NotImplementedError

    def filecode(self, *args):
        return filecode(self, *args)

    @staticmethod
    def checkOutputDir(queuefunc):
        ''' Add outputdir / overwrite keyword arguments to every queue item when an output
            directory is given (model.py:85-108). '''
        @wraps(queuefunc)
        def wrapper(self, *args, **kwargs):
            outputdir = kwargs.get('outputdir')
            queue = queuefunc(self, *args, **kwargs)
            if outputdir is not None:
                overwrite = kwargs.get('overwrite', True)
                for i, params in enumerate(queue):
                    pos, kw = Batch.resolve(params)
                    kw = dict(kw)
                    kw['overwrite'] = overwrite
                    kw['outputdir'] = outputdir
                    queue[i] = (pos, kw)
            elif len(queue) > 5:
                logger.warning('Running more than 5 simulations without file saving')
            return queue
        return wrapper

    @staticmethod
    def addMeta(simfunc):
        @wraps(simfunc)
        def wrapper(self, *args, **kwargs):
            data, tcomp = timer(simfunc)(self, *args, **kwargs)
            logger.debug('completed in %ss', si_format(tcomp, 1))
            meta = getMeta(self, simfunc, *args, **kwargs)
            meta['tcomp'] = tcomp
            return data, meta
        return wrapper

    @staticmethod
    def logNSpikes(simfunc):
        @wraps(simfunc)
        def wrapper(self, *args, **kwargs):
            out = simfunc(self, *args, **kwargs)
            if out is None:
                return None
            data, meta = out
            nspikes = self.getNSpikes(data)
            logger.debug(f'{nspikes} spike{"s" if nspikes != 1 else ""} detected')
            return data, meta
        return wrapper

    @staticmethod
    def checkSimParams(simfunc):
        @wraps(simfunc)
        def wrapper(self, *args, **kwargs):
            args, kwargs = alignWithMethodDef(simfunc, args, kwargs)
            self.checkInputs(*args, *list(kwargs.values()))
            return simfunc(self, *args, **kwargs)
        return wrapper

    @staticmethod
    def logDesc(simfunc):
        @wraps(simfunc)
        def wrapper(self, *args, **kwargs):
            args, kwargs = alignWithMethodDef(simfunc, args, kwargs)
            logger.info(self.desc(getMeta(self, simfunc, *args, **kwargs)))
            return simfunc(self, *args, **kwargs)
        return wrapper

    def titrate(self, *args, **kwargs):
        raise NotImplementedError('titration (threshold search) is not part of this round')

    @staticmethod
    def checkTitrate(simfunc):
        ''' Resolve an unresolved (A is None) drive by titration before simulating; return None
            if no threshold is found (model.py:187-215). '''
        @wraps(simfunc)
        def wrapper(self, *args, **kwargs):
            drive, *other_args = args
            if drive.is_searchable and not drive.is_resolved:
                xthr = self.titrate(*args)
                if np.isnan(xthr):
                    logger.error(f'Could not find threshold {drive.inputs()[drive.xkey]["desc"]}')
                    return None
                args = (drive.updatedX(xthr), *other_args)
            return simfunc(self, *args, **kwargs)
        return wrapper

    def simAndSave(self, *args, **kwargs):
        return simAndSave(self, *args, **kwargs)

    def getOutput(self, *args, **kwargs):
        fpath = self.simAndSave(*args, overwrite=False, **kwargs)
        return loadData(fpath)
