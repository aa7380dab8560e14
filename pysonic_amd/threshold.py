# -*- coding: utf-8 -*-
''' Threshold search (titration) -- the procedure of PySONIC/threshold.py:25-363 restated as a
    COROUTINE so that many searches advance in lock-step, one batched GPU launch per round.

    The reference's Thresholder calls feval(x) -- one full simulation -- at every iteration of a
    sequential search: evaluate x0, optional pre-check at a bound, factor-2 bracketing
    ("preCondition"), bisection until |ub - lb| <= 2 min(rel_eps * lb, eps), a final check at a
    bound if the outcome never changed, and a refinement step so that the returned value is above
    threshold. `threshold_search` yields the next value to evaluate and receives the boolean
    outcome through send(); the sequence of evaluated values for given outcomes is exactly the
    reference's. `titrate_many` drives any number of such searches: every round gathers the
    pending amplitudes of all unfinished searches, simulates them in ONE launch
    (metrics-only: the excitation predicate is the device spike count) and feeds the outcomes back.
'''
import math

import numpy as np

from .utils import logger, isWithin


class OutOfBoundsError(Exception):
    def __init__(self, bounds):
        super().__init__(
            f'No threshold found within the [{bounds[0]:.2e} - {bounds[1]:.2e}] interval')


class MaxNIterations(Exception):
    def __init__(self, max_nit, history):
        super().__init__(f'Maximum number of iterations ({max_nit}) reached, history = {history}')


def _start_point(bounds, x=0.5, scale='lin'):
    ''' Thresholder.getStartPoint (threshold.py:218-233) '''
    if scale == 'log':
        bounds = np.log10(bounds)
    x0 = (1 - x) * bounds[0] + x * bounds[1]
    if scale == 'log':
        x0 = np.power(10., x0)
    return x0


def threshold_search(xbounds, x0=None, eps_thr=None, rel_eps_thr=1e-2, max_nit=50,
                     precheck=False, fbound=2, history=None):
    ''' Generator form of Thresholder.run (threshold.py:288-303).

        Usage:  gen = threshold_search(...); x = next(gen); loop: x = gen.send(is_above(x))
        until StopIteration, whose .value is the threshold (nan if none was found).
        `history`, if given, is a list that receives the (x, outcome) pairs.
    '''
    # ---- constructor logic (threshold.py:33-165), in the reference's setter order ----
    if len(xbounds) != 2:
        raise ValueError('xbounds must be an iterbale of size 2')
    if xbounds[0] >= xbounds[1]:
        raise ValueError('lower bound must be smaller than upper bound')
    fixed_lb, fixed_ub = xbounds
    rel_eps_thr = isWithin('rel_eps_thr', rel_eps_thr, (0., 1.))
    if eps_thr is None:
        eps_thr = np.inf
    if not isinstance(max_nit, int) or max_nit < 1:
        raise ValueError('max_nit must be an integer greater than 0')
    if fbound is not None:
        if fbound <= 1:
            raise ValueError('bounding factor must be greater than 1')
        if fixed_lb == 0.:
            fixed_lb = eps_thr / 2 if eps_thr < np.inf else math.sqrt(np.finfo(float).eps)
        if fixed_ub / fixed_lb <= 2 * fbound:
            raise ValueError('search interval too narrow for factor bounding')
    bounds = (fixed_lb, fixed_ub)
    if x0 is None:
        x0 = _start_point(bounds, x=0.5, scale='log')
    if x0 == 0.:
        x0 = _start_point(bounds, x=0.5, scale='lin')

    xs, evals = [], []

    def evaluate(x):
        ''' eval(): feval, range check (ValueError propagates), iteration budget '''
        xs.append(x)
        above = yield x
        evals.append(bool(above))
        if history is not None:
            history.append((x, bool(above)))
        isWithin('x', x, bounds, raise_warning=False)
        if len(xs) >= max_nit:
            raise MaxNIterations(max_nit, xs)

    def check_at_bound(lb, ub):
        last_eval = evals[-1]
        yield from evaluate(lb if last_eval else ub)
        if evals[-1] == last_eval:
            raise OutOfBoundsError(bounds)

    lb, ub = bounds
    x = x0
    try:
        yield from evaluate(x)
        if precheck:
            yield from check_at_bound(lb, ub)
            x = xs[-1]
            lb, ub = bounds
        if fbound is not None:
            # preCondition (threshold.py:247-271)
            if x * fbound == ub or lb * fbound == x:
                fbound *= 0.99
            while lb < x / fbound or ub > x * fbound:
                if evals[-1]:
                    ub = x
                    x = ub / fbound
                else:
                    lb = x
                    x = fbound * lb
                if lb >= ub:
                    raise OutOfBoundsError(bounds)
                yield from evaluate(x)
            x = (ub + lb) / 2
            yield from evaluate(x)
        # binSearch (threshold.py:273-283)
        while not (np.abs(ub - lb) <= 2 * min(rel_eps_thr * lb, eps_thr)):
            if evals[-1]:
                ub = x
            else:
                lb = x
            x = (ub + lb) / 2
            yield from evaluate(x)
        if len(set(evals)) <= 1:
            yield from check_at_bound(lb, ub)
            x = xs[-1]
        # refine (threshold.py:285-296)
        if not evals[-1]:
            # `self.lb, self.x = self.x, self.midpoint`: the midpoint is taken BEFORE lb moves
            lb, x = x, (ub + lb) / 2
            yield from evaluate(x)
            if not evals[-1]:
                x = ub
                yield from evaluate(x)
    except (OutOfBoundsError, MaxNIterations) as err:
        logger.error(err)
        return np.nan
    return xs[-1]


def titrate_many(evaluate_round, searches):
    ''' Drive several threshold searches in lock-step.

        :param evaluate_round: function(list of (search index, x)) -> list of booleans
        :param searches: list of generators from threshold_search
        :return: list of thresholds (nan where none was found), number of rounds
    '''
    results = [None] * len(searches)
    pending = {}
    for i, gen in enumerate(searches):
        try:
            pending[i] = next(gen)
        except StopIteration as stop:
            results[i] = stop.value
    nrounds = 0
    while pending:
        items = sorted(pending.items())
        outcomes = evaluate_round(items)
        nrounds += 1
        for (i, _), above in zip(items, outcomes):
            try:
                pending[i] = searches[i].send(bool(above))
            except StopIteration as stop:
                results[i] = stop.value
                del pending[i]
    return results, nrounds
