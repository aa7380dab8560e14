# -*- coding: utf-8 -*-
''' N-dimensional lookup container -- API and on-disk format of PySONIC/core/lookups.py
    (Lookup 19-398, EffectiveVariablesLookup 401-460, EffectiveVariablesDict 463-521).

    On-disk format (interchangeable with upstream): pickle({'refs': {name: 1-D array},
    'tables': {name: N-D array}}) with axes in refs order (lookups.py:381-392).
    Projection = linear scipy interp1d along one axis with utils.isWithin range snapping
    (lookups.py:230-271); 1-D evaluation = np.interp with NaN outside (lookups.py:309-322).
    The native library re-implements exactly these two operations for the device tables
    (csrc/sonic_lib.hip: build_level_records).
'''
import os
import pickle
import re

from collections.abc import MutableMapping

import numpy as np
from scipy.interpolate import interp1d

from ..utils import isWithin, isIterable


class Lookup(MutableMapping):
    ''' N-dimensional lookup: `refs` {input name: 1-D grid} and `tables` {output name: N-D array on
        that grid}. Behaves as a mapping over its tables; arithmetic acts table-wise. '''

    interp_choices = ('linear', 'quadratic', 'cubic', 'poly1', 'poly2', 'poly3')

    def __init__(self, refs, tables, interp_method='linear', extrapolate=False):
        self.refs, self.tables = refs, tables
        self.interp_method, self.extrapolate = interp_method, extrapolate
        bad = [k for k, v in tables.items() if v.shape != self.dims]
        if bad:
            raise ValueError(f'{bad[0]} Table dimensions {tables[bad[0]].shape} does not match '
                             f'references {self.dims}')
        if self.ndims == 0 and isinstance(next(iter(tables.values())), np.ndarray):
            self.tables = {k: v.item(0) for k, v in tables.items()}     # fully projected: scalars
        if self.ndims == 1:
            (self.refkey, self.ref), = self.refs.items()
            self.refbounds = (self.ref.min(), self.ref.max())

    def __repr__(self):
        ref_str = ', '.join(f'{k}: {n}' for k, n in zip(self.inputs, self.dims))
        return f'{self.__class__.__name__}{self.ndims}D({ref_str})[{", ".join(self.outputs)}]'

    # ---- mapping over the tables (keys / values / items / pop come with MutableMapping) ----
    def __getitem__(self, key):
        return self.tables[key]

    def __setitem__(self, key, value):
        self.tables[key] = value

    def __delitem__(self, key):
        del self.tables[key]

    def __iter__(self):
        return iter(self.tables.keys())

    def __len__(self):
        return len(self.tables.keys())

    __hash__ = object.__hash__
    __eq__ = object.__eq__

    def refitems(self):
        return self.refs.items()

    def rename(self, key1, key2):
        self.tables[key2] = self.tables.pop(key1)

    dims = property(lambda self: tuple(x.size for x in self.refs.values()))
    ndims = property(lambda self: len(self.refs))
    inputs = property(lambda self: list(self.refs.keys()))
    outputs = property(lambda self: list(self.keys()))
    kwattrs = property(lambda self: {'interp_method': self.interp_method,
                                     'extrapolate': self.extrapolate})

    def __setattr__(self, name, value):
        # validated options (lookups.py:122-141 of the reference)
        if name == 'interp_method':
            if value not in self.interp_choices:
                raise ValueError(f'interpolation method must be one of {self.interp_choices}')
            if value.startswith('poly') and self.ndims > 1:
                raise ValueError('polynomial interpolation only available for 1D lookups')
        elif name == 'extrapolate' and not isinstance(value, bool):
            raise ValueError('extrapolate: expected boolean')
        object.__setattr__(self, name, value)

    def checkAgainst(self, other):
        ''' same inputs, same grids, same outputs -- or ValueError '''
        if self.inputs != other.inputs:
            raise ValueError('Differing lookups (references names do not match)')
        if self.dims != other.dims:
            raise ValueError(f'Differing lookup dimensions ({self.dims} - {other.dims})')
        for k, v in self.refitems():
            if (other.refs[k] != v).any():
                raise ValueError(f'Differing {k} lookup reference')
        if self.outputs != other.outputs:
            raise ValueError('Differing lookups (table names do not match)')

    def operate(self, other, op):
        ''' table-wise binary operation with a compatible lookup or a number '''
        if isinstance(other, self.__class__):
            self.checkAgainst(other)
            rhs = other.__getitem__
        elif isinstance(other, (int, float)):
            rhs = lambda k, x=float(other): x       # noqa: E731
        else:
            raise ValueError(f'Cannot {op} {self.__class__} object with {type(other)} variable')
        return self.__class__(self.refs, {k: getattr(v, op)(rhs(k)) for k, v in self.items()},
                              **self.kwattrs)

    def squeeze(self):
        new_tables = {k: v.squeeze() for k, v in self.items()}
        new_refs = {k: v for k, v in self.refitems() if v.size > 1}
        return self.__class__(new_refs, new_tables, **self.kwattrs)

    def getAxisIndex(self, key):
        assert key in self.inputs, f'Unkown input dimension: {key}'
        return self.inputs.index(key)

    def copy(self):
        return self.__class__(self.refs, self.tables, **self.kwattrs)

    def getInterpolator(self, ref_key, table_key, axis=-1):
        if self.interp_method.startswith('poly'):
            return np.poly1d(np.polyfit(self.refs[ref_key], self.tables[table_key],
                                        int(self.interp_method[-1])))
        fill_value = 'extrapolate' if self.extrapolate else np.nan
        return interp1d(self.refs[ref_key], self.tables[table_key], axis=axis,
                        kind=self.interp_method, assume_sorted=True, fill_value=fill_value)

    def _bracket(self, key, at):
        ''' grid interval of every abscissa: indices (lo, hi) and their grid values '''
        grid = self.refs[key]
        hi = np.clip(np.searchsorted(grid, at), 1, grid.size - 1).astype(int)
        lo = hi - 1
        return lo, hi, grid[lo], grid[hi]

    def _lerpTable(self, table, axis, at, bracket):
        ''' `table` evaluated at the abscissae `at` (1-D) along `axis`, the interpolated axis first:
            y_lo + (y_hi - y_lo) / (x_hi - x_lo) * (at - x_lo), the arithmetic of scipy's linear interp1d
            (lookups.py:224-228 of the reference builds one such object per table and call; here the
            interval search is shared by all tables). A 1-D table without extrapolation goes through
            np.interp, as it does inside scipy. '''
        if table.ndim == 1 and not self.extrapolate:
            return np.interp(at, self.refs[self.inputs[axis]], table)
        lo, hi, x_lo, x_hi = bracket
        y = np.moveaxis(table, axis, 0)
        flat = y.reshape(y.shape[0], -1)
        y_lo, y_hi = flat[lo], flat[hi]
        out = (y_hi - y_lo) / (x_hi - x_lo)[:, None] * (at - x_lo)[:, None] + y_lo
        return out.reshape(at.shape + y.shape[1:])

    def project(self, key, value):
        ''' New lookup with tables interpolated at value(s) along dimension `key` (a scalar drops the
            dimension, an array replaces its grid) -- lookups.py:230-271. '''
        scalar = not isIterable(value)
        at = value if scalar else np.asarray(value)
        grid = self.refs[key]
        if not self.extrapolate:
            at = isWithin(key, at, (grid.min(), grid.max()))
        axis = self.getAxisIndex(key)
        if grid.size == 1:
            tables = {k: v.mean(axis=axis) for k, v in self.items()}     # degenerate axis: its mean
        elif self.interp_method == 'linear':
            pts = np.atleast_1d(np.asarray(at, dtype=float))
            bracket = self._bracket(key, pts)
            tables = {}
            for k, v in self.items():
                cut = self._lerpTable(np.asarray(v), axis, pts, bracket)
                tables[k] = cut[0] if scalar else np.moveaxis(cut, 0, axis)
        else:
            tables = {k: self.getInterpolator(key, k, axis=axis)(at) for k in self.keys()}
        if scalar:
            refs = {k: v for k, v in self.refitems() if k != key}
        else:
            refs = {k: (at if k == key else v) for k, v in self.refitems()}
        return self.__class__(refs, tables, **self.kwattrs)

    def projectN(self, projections):
        lkp = self.copy()
        for k, v in projections.items():
            lkp = lkp.project(k, v)
        return lkp

    def move(self, key, index):
        if index == -1:
            index = self.ndims - 1
        iref = self.getAxisIndex(key)
        for k in self.keys():
            self.tables[k] = np.moveaxis(self.tables[k], iref, index)
        names = list(self.refs.keys())
        names.insert(index, names.pop(iref))
        self.refs = {k: self.refs[k] for k in names}

    def interpVar1D(self, ref_value, var_key):
        assert self.ndims == 1, 'Cannot interpolate multi-dimensional object'
        if isinstance(ref_value, float):
            isWithin(self.inputs[0], ref_value, self.refbounds)
        return np.interp(ref_value, self.ref, self.tables[var_key], left=np.nan, right=np.nan)

    def interpolate1D(self, value):
        return {k: self.interpVar1D(value, k) for k in self.outputs}

    def tile(self, ref_name, ref_values):
        tables = {k: np.array([v for _ in range(ref_values.size)]) for k, v in self.items()}
        refs = {**{ref_name: ref_values}, **self.refs}
        return self.__class__(refs, tables, **self.kwattrs)

    def reduce(self, rfunc, ref_name):
        iaxis = self.getAxisIndex(ref_name)
        refs = {k: v for k, v in self.refitems() if k != ref_name}
        tables = {k: rfunc(v, axis=iaxis) for k, v in self.items()}
        return self.__class__(refs, tables, **self.kwattrs)

    def toDict(self):
        return {'refs': {k: v.tolist() for k, v in self.refs.items()},
                'tables': {k: v.tolist() for k, v in self.tables.items()}}

    @classmethod
    def fromDict(cls, d):
        return cls({k: np.array(v) for k, v in d['refs'].items()},
                   {k: np.array(v) for k, v in d['tables'].items()})

    def toPickle(self, fpath):
        tables = self.tables.d if isinstance(self.tables, EffectiveVariablesDict) else self.tables
        with open(fpath, 'wb') as fh:
            pickle.dump({'refs': self.refs, 'tables': tables}, fh)

    @classmethod
    def fromPickle(cls, fpath):
        cls.checkForExistence(fpath)
        with open(fpath, 'rb') as fh:
            d = _LookupUnpickler(fh).load()
        tables = d['tables']
        if isinstance(tables, EffectiveVariablesDict):
            tables = tables.d
        return cls(d['refs'], tables)

    @staticmethod
    def checkForExistence(fpath):
        if not os.path.isfile(fpath):
            raise FileNotFoundError(f'Missing lookup file: "{fpath}"')


for _op in ('__add__', '__sub__', '__mul__', '__truediv__'):
    setattr(Lookup, _op, (lambda op: lambda self, other: self.operate(other, op))(_op))


class _LookupUnpickler(pickle.Unpickler):
    ''' Upstream lookup files written by EffectiveVariablesLookup.toPickle embed an instance of
        PySONIC.core.lookups.EffectiveVariablesDict: map it onto this module's class so that
        upstream .pkl files load without PySONIC installed. '''

    def find_class(self, module, name):
        if module.startswith('PySONIC') and name == 'EffectiveVariablesDict':
            return EffectiveVariablesDict
        return super().find_class(module, name)


class EffectiveVariablesDict:
    ''' dict wrapper deriving tau<x> = 1/(alpha<x>+beta<x>) and <x>inf = alpha<x>*tau<x> on the
        fly for keys that are not stored (lookups.py:463-521). '''

    _suffix = '[A-Za-z0-9_]+'
    xinf_pattern = re.compile(f'^({_suffix})inf$')
    taux_pattern = re.compile(f'^tau({_suffix})$')

    def __init__(self, d):
        self.d = d

    def __repr__(self):
        return self.__class__.__name__ + '(' + ', '.join(self.d.keys()) + ')'

    def items(self):
        return self.d.items()

    def keys(self):
        return self.d.keys()

    def values(self):
        return self.d.values()

    def __contains__(self, key):
        return key in self.d

    def taux(self, x):
        return 1 / (self.d[f'alpha{x}'] + self.d[f'beta{x}'])

    def xinf(self, x):
        return self.d[f'alpha{x}'] * self.taux(x)

    def __getitem__(self, key):
        if key in self.d:
            return self.d[key]
        m = self.taux_pattern.match(key)
        if m is not None:
            return self.taux(m.group(1))
        m = self.xinf_pattern.match(key)
        if m is not None:
            return self.xinf(m.group(1))
        raise KeyError(key)

    def __setitem__(self, key, value):
        self.d[key] = value

    def __delitem__(self, key):
        del self.d[key]

    def pop(self, key):
        return self.d.pop(key)


class EffectiveVariablesLookup(Lookup):

    def __init__(self, refs, tables, **kwargs):
        if not isinstance(tables, EffectiveVariablesDict):
            tables = EffectiveVariablesDict(tables)
        super().__init__(refs, tables, **kwargs)

    def interpolate1D(self, value):
        return EffectiveVariablesDict(super().interpolate1D(value))

    def projectOff(self):
        ''' Zero-amplitude lookup reduced to the charge dimension. '''
        lkp0 = self.project('A', 0.)
        Qaxis = lkp0.getAxisIndex('Q')
        for k, v in lkp0.items():
            lkp0.tables[k] = np.moveaxis(v, Qaxis, -1)
        for _ in range(lkp0.ndims - 1):
            for k, v in lkp0.items():
                lkp0.tables[k] = v[0]
        lkp0.refs = {'Q': lkp0.refs['Q']}
        return lkp0

    def projectDC(self, amps=None, DC=1.):
        ''' Duty-cycle-averaged lookup: DC * ON + (1 - DC) * OFF. '''
        if amps is None:
            amps = self.refs['A']
        elif not isIterable(amps):
            amps = np.array([amps])
        lkp0 = self.project('A', 0.)
        lkps_ON = self.project('A', amps)
        A_axis = lkps_ON.getAxisIndex('A')
        lkps_ON.move('A', 0)
        lkps_OFF = lkp0.tile('A', lkps_ON.refs['A'])
        lkp = lkps_ON * DC + lkps_OFF * (1 - DC)
        lkp.move('A', A_axis)
        return lkp
