# -*- coding: utf-8 -*-
''' Spike detection and spiking metrics on simulation outputs -- semantics of
    PySONIC/postpro.py:96-144 (helpers), 175-284 (find_tpeaks / detectSpikes), 301-320
    (computeFRProfile) and 323-411 (computeSpikingMetrics). These are the "spike-metric
    reductions" gathered across GPUs in multi-GPU sweeps. '''
import numpy as np
import pandas as pd
from scipy.signal import find_peaks, peak_prominences

from .constants import DT_MAX_REL_TOL, SPIKE_MIN_DT, SPIKE_MIN_QAMP, SPIKE_MIN_QPROM
from .utils import isIterable, loadData


def computeTimeStep(t):
    ''' Mean time step of a regular time vector (zero increments ignored); ValueError if the
        relative spread of the increments exceeds DT_MAX_REL_TOL. '''
    dt = np.diff(t)
    dt = dt[dt != 0]
    rel_dt_var = (dt.max() - dt.min()) / dt.min()
    if rel_dt_var > DT_MAX_REL_TOL:
        raise ValueError(f'irregular time step (rel. variance = {rel_dt_var:.2e})')
    return np.mean(dt)


def resample(t, y, dt):
    n = int(np.ptp(t) / dt) + 1
    ts = np.linspace(t.min(), t.max(), n)
    return ts, np.interp(ts, t, y)


def resolveIndexes(indexes, y, choice='max'):
    ''' Round fractional indexes to the neighbour with max / min signal value. '''
    if indexes.size == 0:
        return indexes
    icomp = np.array([np.floor(indexes), np.ceil(indexes)]).astype(int).T
    ycomp = np.array([y[i] for i in icomp])
    pick = {'min': np.argmin, 'max': np.argmax}[choice](ycomp, axis=1)
    return np.array([pair[pick[i]] for i, pair in enumerate(icomp)])


def _time2samples(x, dt, nsamples):
    if isIterable(x) and len(x) == 2:
        return tuple(_time2samples(v, dt, nsamples) for v in x)
    if isIterable(x) and len(x) == nsamples:
        return np.array([_time2samples(v, dt, nsamples) for v in x])
    if x is None:
        return None
    return int(np.ceil(x / dt))


def find_tpeaks(t, y, **kwargs):
    ''' scipy.signal.find_peaks with time-based criteria on a possibly irregular time grid:
        leading duplicate-time samples are dropped, an irregular grid is linearly resampled at
        max(min(diff t), 1e-7) s, prominences are recomputed with wlen = 5 * min(width), and
        index outputs are mapped back onto the original rows. '''
    ipad = 0
    while t[ipad + 1] == t[ipad]:
        ipad += 1
    if ipad > 0:
        t, y = t[ipad:], y[ipad:]
    try:
        dt = computeTimeStep(t)
        t_raw = y_raw = indexes_raw = None
    except ValueError:
        new_dt = max(np.diff(t).min(), 1e-7)
        t_raw, y_raw = t.copy(), y.copy()
        indexes_raw = np.arange(t_raw.size)
        t, y = resample(t, y, new_dt)
        dt = computeTimeStep(t)
    for key in ['distance', 'width', 'wlen', 'plateau_size']:
        if key in kwargs:
            kwargs[key] = _time2samples(kwargs[key], dt, t.size)
    kwargs.setdefault('width', 1)
    ipeaks, pps = find_peaks(y, **kwargs)
    if len(ipeaks) > 0:
        wlen = 5 * min(pps['widths'])
        pps['prominences'], pps['left_bases'], pps['right_bases'] = peak_prominences(
            y, ipeaks, wlen=wlen)
    if t_raw is not None:
        ipeaks_raw = np.interp(t[ipeaks], t_raw, indexes_raw, left=np.nan, right=np.nan)
        ipeaks = resolveIndexes(ipeaks_raw, y_raw, choice='max')
        for key in ['left_bases', 'right_bases']:
            if key in pps:
                ibase_raw = np.interp(t[pps[key]], t_raw, indexes_raw, left=np.nan, right=np.nan)
                pps[key] = resolveIndexes(ibase_raw, y_raw, choice='min')
        for key in ['left_ips', 'right_ips']:
            if key in pps:
                pps[key] = np.interp(dt * pps[key], t_raw, indexes_raw, left=np.nan, right=np.nan)
    if ipad > 0:
        ipeaks = ipeaks + ipad
        for key in ['left_bases', 'right_bases', 'left_ips', 'right_ips']:
            if key in pps:
                pps[key] = pps[key] + ipad
    if 'widths' in pps:
        pps['widths'] = np.array(pps['widths']) * dt
    return ipeaks, pps


def detectSpikes(data, key='Qm', mpt=SPIKE_MIN_DT, mph=SPIKE_MIN_QAMP, mpp=SPIKE_MIN_QPROM):
    ''' Spike row indexes + properties from peaks of data[key] (height >= mph,
        prominence >= mpp, separation >= mpt). '''
    if key not in data:
        raise ValueError(f'{key} vector not available in dataframe')
    return find_tpeaks(data['t'].values, data[key].values,
                       height=mph, distance=mpt, prominence=mpp)


def computeFRProfile(data):
    ispikes, _ = detectSpikes(data)
    if len(ispikes) == 0:
        return np.ones(len(data)) * np.nan
    t = data['t'].values
    sr = 1 / np.diff(t[ispikes])
    if len(sr) == 0:
        return np.ones(t.size) * np.nan
    return np.interp(t, t[ispikes][:-1], sr, left=np.nan, right=np.nan)


METRIC_KEYS = ['latencies (ms)', 'mean firing rates (Hz)', 'std firing rates (Hz)',
               'mean spike amplitudes (nC/cm2)', 'std spike amplitudes (nC/cm2)',
               'mean spike widths (ms)', 'std spike widths (ms)']


def spikingMetricsRow(data, tstim):
    ''' One row of computeSpikingMetrics for a single (data, tstim). '''
    t = data['t'].values
    ispikes, props = detectSpikes(data)
    widths, prominences = props['widths'], props['prominences']
    if ispikes.size > 0:
        latency = t[ispikes[0]]
        prior = ispikes[t[ispikes] < tstim]
    else:
        latency = np.nan
        prior = np.array([])
    if prior.size > 0:
        w_prior, p_prior = widths[:prior.size], prominences[:prior.size]
    else:
        w_prior = p_prior = np.array([np.nan])
    FRs = 1 / np.diff(t[prior]) if prior.size > 1 else np.array([np.nan])
    return [latency * 1e3, np.mean(FRs), np.std(FRs), np.mean(p_prior) * 1e5,
            np.std(p_prior) * 1e5, np.mean(w_prior) * 1e3, np.std(w_prior) * 1e3]


def computeSpikingMetrics(outputs):
    ''' DataFrame of latency / firing-rate / amplitude / width statistics, one row per output
        ((data, meta) tuples or file paths). '''
    rows = []
    for output in outputs:
        data, meta = loadData(output) if isinstance(output, str) else output
        rows.append(spikingMetricsRow(data, meta['pp'].tstim))
    return pd.DataFrame(rows, columns=METRIC_KEYS)
