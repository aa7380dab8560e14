# -*- coding: utf-8 -*-
''' Base class of stimulus objects (drives, protocols): parameter validation, textual description
    and file codes. API of PySONIC/core/stimobj.py:14-128. '''
import abc

from ..utils import isIterable, si_format


class Param:
    ''' Validated float parameter of a stimulus object, declared once at class level:

            f = Param('checkStrictlyPositive')
            A = Param('checkPositiveOrNull', optional=True)
            DC = Param(bounds=(0., 1.))

        Assignment converts ints to float (TypeError otherwise, as StimObject.checkFloat), runs the
        named StimObject checks and the bounds check, and stores the value under `_<name>`.
        `bounds` may be a function of the object for bounds that depend on other parameters.
        The exception types and messages are those of the reference's per-attribute setters
        (PySONIC/core/drives.py, protocols.py). '''

    def __init__(self, *checks, optional=False, bounds=None):
        self.checks, self.optional, self.bounds = checks, optional, bounds

    def __set_name__(self, owner, name):
        self.name, self.slot = name, '_' + name

    def __get__(self, obj, objtype=None):
        return self if obj is None else getattr(obj, self.slot)

    def __set__(self, obj, value):
        if not (value is None and self.optional):
            value = obj.checkFloat(self.name, value)
            for check in self.checks:
                getattr(obj, check)(self.name, value)
            if self.bounds is not None:
                bounds = self.bounds(obj) if callable(self.bounds) else self.bounds
                if bounds is not None:
                    obj.checkBounded(self.name, value, bounds)
        setattr(obj, self.slot, value)


class StimObject(metaclass=abc.ABCMeta):

    _slug_pairs = [('/', '_per_'), (',', '_'), ('(', ''), (')', ''), (' ', '')]

    @abc.abstractmethod
    def copy(self):
        raise NotImplementedError

    @staticmethod
    @abc.abstractmethod
    def inputs():
        ''' {param: {'desc', 'label', 'unit', 'factor', 'precision', ...}} '''
        raise NotImplementedError

    def xformat(self, x, factor, precision, minfigs, strict_nfigs=False):
        if isIterable(x):
            items = [self.xformat(v, factor, precision, minfigs, strict_nfigs=strict_nfigs)
                     for v in x]
            return f'({", ".join(items)})'
        if isinstance(x, str):
            return x
        xf = si_format(x * factor, precision=precision, space='')
        if strict_nfigs and minfigs is not None:
            nfigs = len(xf.split('.')[0])
            if nfigs < minfigs:
                xf = '0' * (minfigs - nfigs) + xf
        return xf

    # formatted parameter values, by (class, parameter, value, strict_nfigs): a sweep describes the same
    # few hundred values thousands of times (log line and meta of every configuration of a queue)
    _param_strings = {}

    def paramStr(self, k, **kwargs):
        val = getattr(self, k)
        if val is None:
            return None
        key = None
        if isinstance(val, float):
            key = (type(self), k, val, kwargs.get('strict_nfigs', False))
            hit = StimObject._param_strings.get(key)
            if hit is not None:
                return hit
        info = self.inputs()[k]
        xf = self.xformat(val, info.get('factor', 1.), info.get('precision', 0),
                          info.get('minfigs', None), **kwargs)
        out = f"{xf}{info.get('unit', '')}"
        if key is not None and len(StimObject._param_strings) < 100000:
            StimObject._param_strings[key] = out
        return out

    def pdict(self, sf='{key}={value}', **kwargs):
        d = {k: self.paramStr(k, **kwargs) for k in self.inputs().keys()}
        return {k: sf.format(key=k, value=v) for k, v in d.items() if v is not None}

    @property
    def meta(self):
        return {k: getattr(self, k) for k in self.inputs().keys()}

    def __eq__(self, other):
        if not isinstance(other, self.__class__):
            return False
        return all(getattr(self, k) == getattr(other, k) for k in self.inputs().keys())

    def __hash__(self):
        return hash((self.__class__.__name__,) + tuple(getattr(self, k) for k in self.inputs()))

    def __repr__(self):
        return f'{self.__class__.__name__}({", ".join(self.pdict().values())})'

    @property
    def desc(self):
        return ', '.join(self.pdict(sf='{key} = {value}').values())

    def slugify(self, s):
        for a, b in self._slug_pairs:
            s = s.replace(a, b)
        return s

    @property
    def filecodes(self):
        d = self.pdict(sf='{key}_{value}', strict_nfigs=True)
        return {k: self.slugify(v) for k, v in d.items()}

    # ---- validators (same exception types / conditions as the reference) ----
    def checkInt(self, key, value):
        if not isinstance(value, int):
            raise TypeError(f'Invalid {self.inputs()[key]["desc"]} (must be an integer)')
        return value

    def checkFloat(self, key, value):
        if isinstance(value, int):
            value = float(value)
        if not isinstance(value, float):
            raise TypeError(f'Invalid {self.inputs()[key]["desc"]} (must be float typed)')
        return value

    def checkStrictlyPositive(self, key, value):
        if value <= 0:
            raise ValueError(f'Invalid {key} (must be strictly positive)')

    def checkPositiveOrNull(self, key, value):
        if value < 0:
            raise ValueError(f'Invalid {key} (must be positive or null)')

    def checkBounded(self, key, value, bounds):
        if value < bounds[0] or value > bounds[1]:
            d = self.inputs()[key]
            f, u = d.get('factor', 1), d['unit']
            raise ValueError(f'Invalid {d["desc"]}: {value * f} {u} '
                             f'(must be within [{bounds[0] * f}; {bounds[1] * f}] {u})')
