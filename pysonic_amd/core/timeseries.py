# -*- coding: utf-8 -*-
''' TimeSeries: the output DataFrame type -- API of PySONIC/core/timeseries.py:16-146.
    Column contract: 't', 'stimstate', then the solution variables. '''
import numpy as np
import pandas as pd
from scipy.interpolate import interp1d


class TimeSeries(pd.DataFrame):

    time_key = 't'
    stim_key = 'stimstate'

    def __init__(self, t, stim, dout):
        super().__init__(data={self.time_key: t, self.stim_key: stim, **dout})

    @property
    def time(self):
        return self[self.time_key].values

    @property
    def tbounds(self):
        return self.time.min(), self.time.max()

    @property
    def stim(self):
        return self[self.stim_key].values

    @property
    def inputs(self):
        return [self.time_key, self.stim_key]

    @property
    def outputs(self):
        return list(set(self.columns.values) - set(self.inputs))

    def addColumn(self, key, arr, preceding_key=None):
        ''' Append a column. As in the reference (timeseries.py:47-55) the new column ends up
            LAST whatever preceding_key says: this fixes the observable column order
            t, stimstate, Qm, states..., Vm, Z, ng of sonic outputs. '''
        self[key] = arr

    def interpCol(self, t, k):
        kind = 'nearest' if k == self.stim_key else 'linear'
        return interp1d(self.time, self[k].values, kind=kind)(t)

    def interpolate(self, t):
        stim = self.interpCol(t, self.stim_key)
        outputs = {k: self.interpCol(t, k) for k in self.outputs}
        return self.__class__(t, stim, outputs)

    def resample(self, dt):
        tmin, tmax = self.tbounds
        n = int((tmax - tmin) / dt) + 1
        return self.interpolate(np.linspace(tmin, tmax, n))

    def bound(self, tbounds):
        tmin, tmax = tbounds
        return self[np.logical_and(self.time >= tmin, self.time <= tmax)].reset_index(drop=True)

    def dump(self, keys):
        for k in keys:
            del self[k]

    def dumpOutputsOtherThan(self, storekeys):
        self.dump([k for k in self.outputs if k not in storekeys])

    def sampleEvery(self, frequency):
        return self.__class__(self.time[::frequency], self.stim[::frequency],
                              {k: self[k][::frequency] for k in self.outputs})
