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
import abc
import csv
import logging
import os
import time

import numpy as np
import pandas as pd

from ..utils import logger, getTimeStr, isIterable, rangecode


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

    def run(self, mpi=False, loglevel=logging.INFO, gather=None):
        ''' :param gather: under an initialised torch.distributed group (one process per GPU; EVERY rank must
                make this call, with the same queue) the queue is split over the ranks. True: every rank gets
                every result (all_gather_object: small results -- thresholds, effective variables, file
                paths). False: a rank keeps only the results it computed, the other entries of the returned
                list are None. Default: False for `simulate` (DataFrames of traces: 0.5 GB per 4096-cell
                map), True otherwise. Sweeps that need a reduction of every simulation on every rank go
                through parallel.run_sharded / the map classes (metric rows, ONE all-gather). '''
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
                # under an initialised torch.distributed group (one process per GPU) the queue is
                # split over the ranks -- the analogue of the reference's worker pool
                # (batches.py:86-153) -- and every rank gets the full result list back, in queue order
                from ..parallel import run_sharded_objects, _group
                if _group(None)[2] > 1:
                    owner = getattr(self.func, '__self__', None)
                    costs = owner._queueCosts(calls) if hasattr(owner, '_queueCosts') else None
                    if gather is None:
                        gather = getattr(self.func, '__name__', '') != 'simulate'
                    # (dealt over the ranks by estimated cost, not cut into blocks: parallel.dealt_shards)
                    outputs = run_sharded_objects(lambda idx: impl([calls[i] for i in idx]), len(calls), costs,
                                                  gather=bool(gather), dealt=True)
                else:
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


class LogBatch(metaclass=abc.ABCMeta):
    ''' A sweep whose input -> output pairs live in a delimited log file (API and file format of
        PySONIC/core/batches.py:186-375): one header line `<in_label> <out_keys...>`, one line per
        evaluated input, appended as results arrive, so an interrupted sweep resumes where it stopped
        and a finished one is just read back.

        Subclasses define `in_key`, `unit`, `out_keys`, `suffix`, `corecode()` and `compute(x)`.
        `run(mpi=True)` hands the inputs that are not logged yet to `computeMany(inputs)` -- one batched
        evaluation (one kernel launch for the maps of pysonic_amd.actmap) where the reference starts
        a worker process per input; the default `computeMany` just loops over `compute`. '''

    delimiter = '\t'
    rtol = 1e-9
    atol = 1e-16

    def __init__(self, inputs, root='.'):
        self.inputs = inputs
        self.root = root
        self.fpath = self.filepath()

    @property
    def root(self):
        return self._root

    @root.setter
    def root(self, value):
        if not os.path.isdir(value):
            raise ValueError(f'{value} is not a valid directory')
        self._root = value

    # ---- what a subclass declares ------------------------------------------------------------
    @property
    @abc.abstractmethod
    def in_key(self):
        ''' name of the input '''

    @property
    @abc.abstractmethod
    def unit(self):
        ''' unit of the input '''

    @property
    @abc.abstractmethod
    def out_keys(self):
        ''' names of the outputs '''

    @property
    @abc.abstractmethod
    def suffix(self):
        ''' file-name suffix '''

    @abc.abstractmethod
    def corecode(self):
        ''' file-name fragment describing everything but the inputs '''

    @abc.abstractmethod
    def compute(self, x):
        ''' output(s) of one input '''

    def computeMany(self, inputs):
        return [self.compute(x) for x in inputs]

    # ---- file naming -------------------------------------------------------------------------
    @property
    def in_label(self):
        return f'{self.in_key} ({self.unit})'

    @property
    def in_labels(self):
        return [self.in_label]

    @property
    def inputscode(self):
        return rangecode(self.inputs, self.in_key, self.unit)

    def filecode(self):
        return f'{self.corecode()}_{self.inputscode}_{self.suffix}_results'

    def filename(self):
        return f'{self.filecode()}.csv'

    def filepath(self):
        return os.path.join(self.root, self.filename())

    # ---- the log -----------------------------------------------------------------------------
    def createLogFile(self):
        if not os.path.isfile(self.fpath):
            self._append([*self.in_labels, *self.out_keys], mode='w')

    def _append(self, row, mode='a'):
        from ..utils import file_lock
        with open(self.fpath, mode, newline='') as fh:
            with file_lock(fh):          # concurrent writers: as the reference's logCache does (utils.py:486-491)
                csv.writer(fh, delimiter=self.delimiter).writerow(row)

    def writeEntry(self, entry):
        self._append(entry)

    def getLogData(self):
        ''' the log as a DataFrame sorted by input '''
        return pd.read_csv(self.fpath, sep=self.delimiter).sort_values(self.in_labels)

    def getInput(self):
        v = self.getLogData()[self.in_labels].values
        return v[:, 0] if len(self.in_labels) == 1 else v

    def getSerializedOutput(self):
        data = self.getLogData()
        if len(self.out_keys) == 1:
            return data[self.out_keys[0]].values
        return pd.DataFrame({k: data[k].values for k in self.out_keys})

    def getOutput(self):
        return self.getSerializedOutput()

    def isFinished(self):
        return os.path.isfile(self.fpath) and len(self.getLogData()) == len(self.inputs)

    def _matches(self, logged, entry):
        ''' indexes of the logged inputs equal to `entry` within (rtol, atol) '''
        logged = np.asarray(logged, dtype=float)
        entry = np.atleast_1d(np.asarray(entry, dtype=float))
        if logged.size == 0:
            return np.zeros(0, dtype=int)
        logged = logged.reshape(len(logged), -1)
        return np.where(np.all(np.isclose(logged, entry, rtol=self.rtol, atol=self.atol), axis=1))[0]

    def getEntryIndex(self, entry):
        logged = self.getInput()
        if len(logged) == 0:
            raise ValueError('no entries in batch')
        imatches = self._matches(logged, entry)
        if imatches.size == 0:
            raise ValueError(f'{entry} entry not found in batch log')
        if imatches.size > 1:
            raise ValueError(f'duplicate {entry} entry found in batch log')
        return int(imatches[0])

    def getEntryOutput(self, entry):
        out = self.getSerializedOutput()
        i = self.getEntryIndex(entry)
        return out.iloc[i] if isinstance(out, pd.DataFrame) else out[i]

    def isEntry(self, value):
        return self._matches(self.getInput(), value).size > 0

    @staticmethod
    def _entry(x, out):
        return [*(x if isIterable(x) else [x]), *(out if isIterable(out) else [out])]

    def computeAndLog(self, x):
        ''' compute and log one input unless it is logged already; returns the new entry or None '''
        if self.isEntry(x):
            return None
        entry = self._entry(x, self.compute(x))
        self.writeEntry(entry)
        return entry

    def run(self, mpi=False):
        ''' evaluate every input that is not in the log yet and return the outputs of the whole batch.
            Under an initialised torch.distributed group (one process per GPU; every rank makes this call)
            the missing inputs are split over the ranks, their outputs all-gathered as rows (RCCL when the
            backend is nccl: parallel.run_sharded), and rank 0 alone writes the log: every rank returns the
            same output, the file holds each entry once. '''
        from ..parallel import _group, run_sharded, barrier
        dist, rank, world = _group(None)
        if rank == 0:
            self.createLogFile()
        barrier(dist)
        logged = self.getInput()
        todo = [x for x in self.inputs if self._matches(logged, x).size == 0]
        if todo and world > 1:
            many = self.computeMany if mpi else (lambda xs: [self.compute(x) for x in xs])
            nout = len(self.out_keys)
            # (dealt, not cut into blocks: the cost of an input follows its place in the sweep)
            rows = run_sharded(lambda idx: np.asarray(many([todo[i] for i in idx]), dtype=float).reshape(len(idx), nout),
                               len(todo), dist=dist, dealt=True)
            if rank == 0:
                for x, out in zip(todo, rows):
                    self.writeEntry(self._entry(x, out if nout > 1 else out[0]))
            barrier(dist)
        elif todo:
            if mpi:
                for x, out in zip(todo, self.computeMany(todo)):
                    self.writeEntry(self._entry(x, out))
            else:
                for x in todo:
                    self.writeEntry(self._entry(x, self.compute(x)))
        else:
            logger.debug('all entries already present')
        return self.getOutput()
