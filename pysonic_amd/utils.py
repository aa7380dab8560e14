# -*- coding: utf-8 -*-
''' Small host-side helpers with the semantics of the reference's PySONIC/utils.py that the hot
    path's callers rely on: isWithin (utils.py:321-348), timer (408-417), filecode (727-752),
    simAndSave (755-825), getMeta (872-884), loadData (283-290), si_format (149-160). '''
import csv
import logging
import math
import os
import pickle
import time
from functools import wraps
from inspect import signature

import numpy as np
import pandas as pd

logger = logging.getLogger('PySONIC')
if not logger.handlers:
    _h = logging.StreamHandler()
    _h.setFormatter(logging.Formatter('%(asctime)s %(message)s', datefmt='%d/%m/%Y %H:%M:%S:'))
    logger.addHandler(_h)
    # level left unset, as in the reference (utils.py:58-79): effectively WARNING until a script
    # or Batch.run(mpi=True, loglevel=...) lowers it -- the level decides whether the detailed model
    # is integrated with progress-log events (nbls.py:345-346)

LOOKUP_DIR = os.environ.get(
    'PYSONIC_AMD_LOOKUP_DIR',
    os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lookups'))

_SI_PREFIXES = [(-24, 'y'), (-21, 'z'), (-18, 'a'), (-15, 'f'), (-12, 'p'), (-9, 'n'), (-6, 'u'),
                (-3, 'm'), (0, ''), (3, 'k'), (6, 'M'), (9, 'G'), (12, 'T'), (15, 'P'), (18, 'E'),
                (21, 'Z'), (24, 'Y')]
_SI_FACTORS = np.array([np.power(10., e) for e, _ in _SI_PREFIXES])


def getSIpair(x):
    ''' (factor, prefix) with the largest SI factor <= |x|  (utils.py:131-146). '''
    if x == 0:
        return 1e0, ''
    ix = int(np.searchsorted(_SI_FACTORS, np.abs(x))) - 1
    if ix + 1 < _SI_FACTORS.size and np.abs(x) == _SI_FACTORS[ix + 1]:
        ix += 1
    ix = max(ix, 0)
    return _SI_FACTORS[ix], _SI_PREFIXES[ix][1]


def si_format(x, precision=0, space=' '):
    ''' Format with SI prefix, e.g. 5e5 -> "500 k" (utils.py:149-160). '''
    if isinstance(x, (list, tuple)) or (isinstance(x, np.ndarray) and x.ndim == 1):
        return [si_format(float(v), precision, space) for v in x]
    factor, prefix = getSIpair(x)
    return f'{x / factor:.{precision}f}{space}{prefix}'


def isIterable(x):
    return isinstance(x, (list, tuple, np.ndarray, pd.Index, pd.MultiIndex, pd.Series))


def isWithin(name, val, bounds, rel_tol=1e-9, raise_warning=True):
    ''' Return val if inside bounds, the bound if within rel_tol of it, else raise ValueError. '''
    if isIterable(val):
        return np.array([isWithin(name, v, bounds, rel_tol, raise_warning) for v in val])
    lo, hi = bounds
    if lo <= val <= hi:
        return val
    if val < lo and math.isclose(val, lo, rel_tol=rel_tol):
        if raise_warning:
            logger.warning('Rounding %s value (%s) to interval lower bound (%s)', name, val, lo)
        return lo
    if val > hi and math.isclose(val, hi, rel_tol=rel_tol):
        if raise_warning:
            logger.warning('Rounding %s value (%s) to interval upper bound (%s)', name, val, hi)
        return hi
    raise ValueError(f'{name} value ({val}) out of [{lo}, {hi}] interval')


def rmse(x1, x2, axis=None):
    return np.sqrt(((x1 - x2) ** 2).mean(axis=axis))


def timer(func):
    ''' Decorator returning (value, wall time in s). '''
    @wraps(func)
    def wrapper(*args, **kwargs):
        t0 = time.perf_counter()
        value = func(*args, **kwargs)
        return value, time.perf_counter() - t0
    return wrapper


def getMeta(model, simfunc, *args, **kwargs):
    ''' {'simkey', 'model': model.meta, <simulate() arguments incl. defaults>} '''
    bound = signature(simfunc).bind(model, *args, **kwargs)
    bound.apply_defaults()
    meta = {'simkey': model.simkey}
    for k, v in bound.arguments.items():
        meta['model' if k == 'self' else k] = v.meta if k == 'self' else v
    return meta


def alignWithMethodDef(method, args, kwargs):
    ''' Split call arguments into the method's positional part and a complete kwargs dict. '''
    params = list(signature(method).parameters.values())[1:]   # drop self
    pos = [p for p in params if p.default is p.empty]
    kw = {p.name: p.default for p in params if p.default is not p.empty}
    new_args = tuple(args[:len(pos)])
    for name, val in zip(list(kw.keys()), args[len(pos):]):
        kw[name] = val
    kw.update(kwargs)
    return new_args, kw


def filecode(model, *args):
    ''' File code from model inputs or from a meta dictionary. '''
    if len(args) == 1 and isinstance(args[0], dict):
        meta = args[0].copy()
        if meta['simkey'] == 'ASTIM' and 'fs' not in meta:
            meta['fs'] = meta['model']['fs']
            meta['method'] = meta['model']['method']
            meta['qss_vars'] = None
        for k in ['simkey', 'model', 'tcomp', 'dt', 'atol']:
            meta.pop(k, None)
        args = list(meta.values())
    else:
        args = list(args)
    for i, a in enumerate(args):
        if isIterable(a):
            args[i] = ''.join(str(x) for x in a)
    return '_'.join(x for x in model.filecodes(*args).values() if x is not None)


def loadData(fpath, frequency=1):
    ''' Load (data, meta) from a simulation pickle written by simAndSave. '''
    with open(fpath, 'rb') as fh:
        frame = pickle.load(fh)
    data = frame['data'].iloc[::frequency]
    return data, frame['meta']


def simAndSave(model, *args, **kwargs):
    ''' Simulate and pickle {'meta', 'data'} to <outputdir>/<filecode>.pkl; skip if the file
        exists and overwrite is False. Returns the file path (None if titration failed). '''
    outputdir = kwargs.pop('outputdir', '.')
    overwrite = kwargs.pop('overwrite', True)
    full_output = kwargs.pop('full_output', True)
    data, meta = None, None
    drive, *other_args = args
    if drive.is_searchable and not drive.is_resolved:
        out = model.simulate(*args, **kwargs)
        if out is None:
            logger.warning('returning None')
            return None
        data, meta = out
        args = (meta['drive'], *other_args)
    fname = f'{model.filecode(*args)}.pkl'
    fpath = os.path.join(outputdir, fname)
    exists = os.path.isfile(fpath)
    if exists and not overwrite:
        logger.warning(f'File "{fname}" already present in directory "{outputdir}" -> preserving')
        return fpath
    if data is None:
        data, meta = model.simulate(*args, **kwargs)
    if not full_output:
        data.dumpOutputsOtherThan(['Qm', 'Vm'])
    if exists:
        logger.warning(f'File "{fname}" already present in directory "{outputdir}" -> overwriting')
    with open(fpath, 'wb') as fh:
        pickle.dump({'meta': meta, 'data': data}, fh)
    return fpath


def getTimeStr(seconds):
    if seconds < 60:
        return f'{seconds:.2f}'
    m, s = divmod(seconds, 60)
    h, m = divmod(m, 60)
    return f'{int(h)}:{int(m):02d}:{s:05.2f}'


def expandRange(xmin, xmax, exp_factor=2):
    if exp_factor < 1:
        raise ValueError('expansion factor must be superior or equal to 1')
    xptp = xmax - xmin
    xmid = (xmin + xmax) / 2
    xdev = xptp * exp_factor / 2
    return (xmid - xdev, xmid + xdev)


# ---- file-backed memoisation of expensive scalar results (utils.py:391-394, 419-497) -----------------
def funcSig(name, args, kwargs):
    ''' 'name(repr(arg), ..., key=repr(value), ...)': the key format of the reference's log caches '''
    return f'{name}({", ".join([repr(a) for a in args] + [f"{k}={v!r}" for k, v in kwargs.items()])})'


def methodCallSignature(method, args, kwargs):
    ''' Key of one call of a bound method, arguments aligned with the method's definition: the positional
        parameters (owner first, as `self`) as positional reprs, every keyword parameter with its value or
        default. Equals the reference's logCache key for the same call, so a cache file written by the
        reference (PySONIC/core/astim_titrations.log) is valid here and vice versa. '''
    pos, kw = alignWithMethodDef(getattr(method, '__func__', method), args, kwargs)
    return funcSig(method.__name__, (method.__self__,) + tuple(pos), kw)


class file_lock:
    ''' advisory exclusive lock on an open file for the duration of a `with` block (no-op where fcntl is missing) '''

    def __init__(self, fh):
        self.fh = fh

    def __enter__(self):
        try:
            import fcntl
            fcntl.flock(self.fh.fileno(), fcntl.LOCK_EX)
        except (ImportError, OSError):
            pass
        return self.fh

    def __exit__(self, *exc):
        try:
            import fcntl
            self.fh.flush()
            fcntl.flock(self.fh.fileno(), fcntl.LOCK_UN)
        except (ImportError, OSError):
            pass
        return False


class LogCache:
    ''' signature -> value pairs in a delimited text file, one entry per line, appended as they come '''

    def __init__(self, fpath, delimiter='\t', out_type=float):
        self.fpath, self.delimiter, self.out_type = fpath, delimiter, out_type
        self._mem, self._mtime = {}, None

    def _load(self):
        if not os.path.isfile(self.fpath):
            self._mem, self._mtime = {}, None
            return
        mtime = os.path.getmtime(self.fpath)
        if mtime == self._mtime:
            return
        mem = {}
        with open(self.fpath, 'r', newline='') as fh:
            for row in csv.reader(fh, delimiter=self.delimiter):
                if len(row) >= 2 and row[0] not in mem:       # the first entry wins, like the reference's scan
                    mem[row[0]] = row[1]
        self._mem, self._mtime = mem, mtime

    def get(self, sig):
        ''' cached value or None '''
        self._load()
        v = self._mem.get(sig)
        return None if v is None else self.out_type(v)

    def put(self, sig, value):
        ''' append one entry; writers in other processes (the ranks of a process group, parallel sessions) are
            kept apart by an advisory lock on the file, as the reference's lockfile.FileLock does '''
        os.makedirs(os.path.dirname(os.path.abspath(self.fpath)), exist_ok=True)
        with open(self.fpath, 'a', newline='') as fh:
            with file_lock(fh):
                csv.writer(fh, delimiter=self.delimiter).writerow([sig, str(value)])
        self._mtime = None


def rangecode(x, label, unit):
    ''' 'Label_min<unit>-max<unit>_n': the fragment of a batch file name describing an input vector '''
    x = np.asarray(x)
    lo, hi = si_format([x.min(), x.max()], 1, space='')
    return f'{label.replace(" ", "_")}{lo}{unit}-{hi}{unit}_{x.size}'
