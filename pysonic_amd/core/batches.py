# -*- coding: utf-8 -*-
''' Batch: run one function over a queue of argument lists -- API of
    PySONIC/core/batches.py:70-183.

    The reference executes the queue serially or, with mpi=True, on a single-node `multiprocess`
    pool (one configuration per worker process). Here `mpi=True` means "use the accelerator":
    when `func` is a bound method of a pysonic_amd model that has a batched device
    implementation (NeuronalBilayerSonophore.simulate with method='sonic', .computeEffVars, ...),
    the WHOLE queue is executed by one kernel launch per model and results are returned in
    queue order, exactly like Batch.get re-orders worker outputs (batches.py:118-128).
    With mpi=False the queue is looped over on the host (each call is still a batch of one on the
    GPU: there is no CPU integrator in this package).
'''
import logging
import time

import numpy as np

from ..utils import logger, getTimeStr


class Batch:

    def __init__(self, func, queue):
        self.func = func
        self.queue = queue

    def __call__(self, *args, **kwargs):
        return self.run(*args, **kwargs)

    @staticmethod
    def resolve(params):
        ''' queue item -> (args, kwargs): items are [args] or ([args], {kwargs}) '''
        if isinstance(params, tuple):
            args, kwargs = params
        else:
            args, kwargs = params, {}
        return args, kwargs

    def _batched_impl(self):
        ''' Device-batched counterpart of self.func, if its owner provides one. '''
        owner = getattr(self.func, '__self__', None)
        name = getattr(self.func, '__name__', None)
        if owner is None or name is None:
            return None
        return getattr(owner, f'_batched_{name}', None)

    def run(self, mpi=False, loglevel=logging.INFO):
        s = 'en' if mpi else 'dis'
        logger.info(f'Starting {len(self.queue)}-job(s) batch (accelerator batching {s}abled)')
        t0 = time.perf_counter()
        impl = self._batched_impl() if mpi else None
        if impl is not None:
            calls = [self.resolve(p) for p in self.queue]
            # the reference's workers run at `loglevel` (batches.py:62-66), and the level decides
            # whether a detailed simulation is integrated with progress-log events (nbls.py:345-346)
            previous = logger.level
            logger.setLevel(loglevel)
            try:
                outputs = impl(calls)
            finally:
                logger.setLevel(previous)
        else:
            outputs = []
            for params in self.queue:
                args, kwargs = self.resolve(params)
                outputs.append(self.func(*args, **kwargs))
        logger.info(f'Batch completed in {getTimeStr(time.perf_counter() - t0)} s')
        return outputs

    @staticmethod
    def createQueue(*dims):
        ''' Cartesian product of the input sweeps as a list of lists, first dimension slowest
            (same order as batches.py:155-171). '''
        ndims = len(dims)
        dims_in = [dims[1], dims[0]] if ndims > 1 else [dims[0]]
        inds_out = [1, 0] if ndims > 1 else [0]
        if ndims > 2:
            dims_in += list(dims[2:])
            inds_out += list(range(2, ndims))
        queue = np.stack(np.meshgrid(*dims_in), -1).reshape(-1, ndims)
        return queue[:, inds_out].tolist()

    @staticmethod
    def printQueue(queue, nmax=20):
        if len(queue) <= nmax:
            for x in queue:
                print(x)
        else:
            for x in queue[:nmax // 2]:
                print(x)
            print(f'... {len(queue) - nmax} more entries ...')
            for x in queue[-nmax // 2:]:
                print(x)
