# -*- coding: utf-8 -*-
''' Host-side mirror of the reference's Python API (no GPU needed): descriptions, file codes,
    queue orders, event schedules, lookups, validation errors -- against values captured from the
    reference (tests/golden/golden_api.json, golden_neurons.npz). '''
import json
import os
import pickle

import numpy as np
import pytest

from conftest import GOLDEN, NEURONS, load_golden
import pysonic_amd as ps
from pysonic_amd import (NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, Batch,
                         EffectiveVariablesLookup, getPointNeuron, TimeSeries)
from pysonic_amd.utils import si_format, isWithin
from pysonic_amd.postpro import detectSpikes, computeSpikingMetrics


@pytest.fixture(scope='module')
def api():
    with open(os.path.join(GOLDEN, 'golden_api.json')) as fh:
        return json.load(fh)


def test_drives(api):
    for e in api['drives']:
        d = AcousticDrive(*e['args'])
        assert repr(d) == e['repr'] and d.desc == e['desc'] and d.filecodes == e['filecodes']
        assert d.dt == e['dt'] and d.periodicity == e['T']
    with pytest.raises(ValueError):
        AcousticDrive(-1., 1e3)
    with pytest.raises(ValueError):
        AcousticDrive(500e3, -1.)
    with pytest.raises(TypeError):
        AcousticDrive('500e3', 1e3)
    d = AcousticDrive(500e3)
    assert d.is_searchable and not d.is_resolved and d.updatedX(5e4).A == 5e4
    assert AcousticDrive(500e3, 1e5).compute(0.5e-6) == pytest.approx(1e5 * np.sin(0.5 * np.pi - np.pi))


def test_protocols(api):
    for e in api['protocols']:
        p = PulsedProtocol(*e['args'])
        ev = p.stimEvents()
        assert repr(p) == e['repr'] and p.desc == e['desc'] and p.filecodes == e['filecodes']
        assert p.tstop == e['tstop'] and p.nature == e['nature'] and p.npulses == e['npulses']
        assert [float(x[0]) for x in ev] == e['ev_t'] and [float(x[1]) for x in ev] == e['ev_x']
    with pytest.raises(ValueError):
        PulsedProtocol(0.1, 0.05, 100., 1.5)       # DC out of [0, 1]
    with pytest.raises(ValueError):
        PulsedProtocol(0.1, 0.05, 5., 0.5)         # PRF < 1 / tstim
    with pytest.raises(ValueError):
        PulsedProtocol(-0.1, 0.05)
    assert [repr(p) for p in PulsedProtocol.createQueue(
        [0.1, 0.2], [0.05, 0.1], [10., 100.], [0.5, 1.0])] == api['ppQueue']


def test_queues_and_filecodes(api):
    nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron('RS'))
    assert repr(nbls) == api['nbls_repr']
    codes = []
    for e in api['drives'][:3]:
        for pe in api['protocols'][:3]:
            codes.append(nbls.filecode(AcousticDrive(*e['args']), PulsedProtocol(*pe['args']),
                                       1., 'sonic', None))
    assert codes == api['filecodes']
    assert nbls.filecode(AcousticDrive(500e3, 100e3), PulsedProtocol(*api['protocols'][1]['args']),
                         0.5, 'sonic', None) == api['filecode_fs']
    q = NeuronalBilayerSonophore.simQueue(
        [20e3, 500e3], [50e3, 100e3, 300e3], [0.1], [0.05], [10., 100.], [0.25, 0.5, 1.0], [1.],
        ['sonic'], None)
    assert [[repr(x[0]), repr(x[1]), x[2], x[3], x[4]] for x in q] == api['simQueue']
    assert Batch.createQueue([1., 2., 3.], [10., 20.]) == api['createQueue2']
    assert Batch.createQueue([1., 2.], [10., 20., 30.], [100., 200.]) == api['createQueue3']
    qd = NeuronalBilayerSonophore.simQueue([500e3], [1e5], [0.1], [0.05], [100.], [1.0], [1.],
                                           ['sonic'], None, outputdir='/tmp/x', overwrite=False)
    assert qd[0][1] == {'overwrite': False, 'outputdir': '/tmp/x'}


def test_si_format_and_iswithin(api):
    assert all(si_format(x, p, '') == s for x, p, s in api['si_format'])
    assert isWithin('A', 600000.0, (0., 599999.9999999997)) == 599999.9999999997
    with pytest.raises(ValueError):
        isWithin('A', 600001.0, (0., 6e5))


def test_neuron_definitions():
    for n in NEURONS + ['IB', 'HHseg', 'SWnode', 'MRGnode', 'SUseg', 'FHnode']:
        g = load_golden('golden_neurons.npz' if n in NEURONS else f'golden_{n}.npz')
        pn = getPointNeuron(n)
        assert pn.name == n and list(g[f'{n}_states']) == pn.statesNames()
        assert list(g[f'{n}_rates']) == pn.rates == list(pn.effRates().keys())
        assert pn.Qm0 == float(g[f'{n}_Qm0']) and pn.Vm0 == float(g[f'{n}_Vm0'])
        np.testing.assert_array_equal(pn.Qbounds, g[f'{n}_Qbounds'])
        np.testing.assert_array_equal(pn.getSteadyStates(pn.Vm0), g[f'{n}_y0'])
        for pt, inet, ders in zip(g[f'{n}_pts'], g[f'{n}_iNet'], g[f'{n}_ders']):
            x = dict(zip(pn.statesNames(), pt[1:]))
            assert pn.iNet(pt[0], x) == pytest.approx(inet, rel=1e-13)
            d = [float(pn.derStates()[k](pt[0], x)) for k in pn.statesNames()]
            np.testing.assert_allclose(d, ders, rtol=1e-12)
    with pytest.raises(ValueError):
        getPointNeuron('nope')


def test_lookup_project_is_scipy_linear_interp1d():
    ''' Lookup.project along any axis, scalar and array abscissae, N-D and 1-D tables: the bits of the scipy
        interp1d objects the reference builds per table (lookups.py:224-228, 230-271) '''
    from scipy.interpolate import interp1d
    rng = np.random.default_rng(5)
    refs = {'a': np.array([16e-9, 32e-9, 64e-9]), 'A': np.concatenate([[0.], np.logspace(2, 5.78, 9)]),
            'Q': np.linspace(-1e-3, 5e-4, 11)}
    dims = tuple(v.size for v in refs.values())
    tables = {k: rng.normal(size=dims) * 10.0**rng.integers(-3, 6) for k in ['V', 'alpham', 'betah']}
    lkp = EffectiveVariablesLookup(refs, tables)
    cases = [('a', 32e-9), ('a', 41.3e-9), ('A', 0.), ('A', 6e5), ('A', 1234.5), ('Q', -3.3e-4),
             ('A', np.array([0., 17., 4.2e3, 6e5])), ('Q', np.array([-1e-3, 2.5e-5, 5e-4])), ('a', np.array([20e-9]))]
    for key, at in cases:
        got = lkp.project(key, at)
        axis = list(refs).index(key)
        for k in tables:
            ref = interp1d(refs[key], tables[k], axis=axis, kind='linear', assume_sorted=True, fill_value=np.nan)(at)
            assert np.array_equal(got[k], ref), (key, at, k)
        assert got.inputs == ([r for r in refs if r != key] if np.ndim(at) == 0 else list(refs))
    l1 = lkp.project('a', 32e-9).project('A', 5e4)          # 1-D tables: scipy hands these to np.interp
    for at in (-2.2e-4, np.array([-1e-3, 0., 5e-4])):
        for k in tables:
            ref = interp1d(refs['Q'], l1[k], kind='linear', assume_sorted=True, fill_value=np.nan)(at)
            assert np.array_equal(np.asarray(l1.project('Q', at)[k]), ref)


def test_lookup_container(tmp_path):
    rng = np.random.default_rng(0)
    refs = {'a': np.array([16e-9, 32e-9]), 'f': np.array([20e3, 500e3, 4e6]),
            'A': np.linspace(0, 6e5, 5), 'Q': np.linspace(-1e-3, 5e-4, 7), 'fs': np.array([1.])}
    dims = tuple(v.size for v in refs.values())
    tables = {k: rng.uniform(1, 2, size=dims) for k in ['V', 'alpham', 'betam']}
    lkp = EffectiveVariablesLookup(refs, {k: v.copy() for k, v in tables.items()})
    l2 = lkp.projectN({'a': 32e-9, 'f': 500e3, 'fs': 1.})
    assert l2.inputs == ['A', 'Q'] and l2.dims == (5, 7)
    np.testing.assert_allclose(l2['V'], tables['V'][1, 1, :, :, 0])
    l1 = l2.project('A', 1.5e5)
    assert l1.ndims == 1
    np.testing.assert_allclose(l1['V'], tables['V'][1, 1, 1, :, 0], rtol=1e-14)
    q = -2.5e-4
    got = l1.interpolate1D(q)
    assert got['V'] == np.interp(q, refs['Q'], l1['V'])
    assert got['taum'] == 1 / (got['alpham'] + got['betam'])
    assert got['minf'] == got['alpham'] * got['taum']
    assert np.isnan(l1.interpVar1D(np.array([1.]), 'V'))[0]
    with pytest.raises(ValueError):
        l2.project('A', 7e5)
    # DC-averaging and arithmetic
    ldc = l2.projectDC(DC=0.3)
    np.testing.assert_allclose(ldc['V'][2], 0.3 * l2['V'][2] + 0.7 * l2['V'][0])
    np.testing.assert_allclose((l2 * 2.0)['V'], 2 * l2['V'])
    # upstream on-disk format: pickle({'refs', 'tables'})
    fpath = os.path.join(tmp_path, 'lkp.pkl')
    lkp.toPickle(fpath)
    with open(fpath, 'rb') as fh:
        raw = pickle.load(fh)
    assert set(raw.keys()) == {'refs', 'tables'} and isinstance(raw['tables'], dict)
    back = EffectiveVariablesLookup.fromPickle(fpath)
    np.testing.assert_array_equal(back['V'], tables['V'])
    with pytest.raises(FileNotFoundError):
        EffectiveVariablesLookup.fromPickle(os.path.join(tmp_path, 'missing.pkl'))


def test_packaged_lookups_and_nbls_inputs():
    for n in NEURONS:
        pn = getPointNeuron(n)
        nbls = NeuronalBilayerSonophore(32e-9, pn)
        lkp = nbls.getLookup2D(500e3, 1.)
        assert lkp.inputs == ['A', 'Q'] and lkp.outputs == ['V'] + pn.rates
        assert lkp.refs['A'].size == 51 and lkp.refs['A'][0] == 0.
        Qmin, Qmax = pn.Qbounds
        assert lkp.refs['Q'][0] == pytest.approx(Qmin) and lkp.refs['Q'][-1] == pytest.approx(Qmax)
    nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron('RS'))
    assert nbls.Delta == 1.2553492695740507e-9
    assert nbls.meta == {'neuron': 'RS', 'a': 32e-9, 'd': 0.}
    d, pp = AcousticDrive(500e3, 1e5), PulsedProtocol(0.1, 0.05)
    with pytest.raises(TypeError):
        nbls.checkInputs(d, pp, 1, 'sonic', None)          # fs must be float
    with pytest.raises(ValueError):
        nbls.checkInputs(d, pp, 1., 'euler', None)
    with pytest.raises(ValueError):
        nbls.checkInputs(d, pp, 1., 'sonic', ['zz'])
    with pytest.raises(TypeError):
        nbls.checkInputs('drive', pp, 1., 'sonic', None)
    with pytest.raises(ValueError):
        nbls.getLookup2D(10e6, 1.)                         # f outside every lookup range
    y0 = nbls.initialConditionsSonic()
    assert y0[0] == -7.19e-4 * (1 + 0) or y0[0] == pytest.approx(-7.19e-4)
    assert y0[1] == 4.509061287474801e-4 and y0[4] == 2.4363594315277837e-2
    A, tstop, dt, ev_t, ev_x, ev_off = nbls._packConfigs([(d, pp), (d, PulsedProtocol(0.1, 0., 100., .5))])
    assert list(ev_off) == [0, 2, 22] and list(dt) == [5e-5, 5e-5] and list(tstop) == [0.1 + 0.05 + 0., 0.1]


def test_postpro_on_reference_outputs():
    ''' detectSpikes / spiking metrics of the host layer on the reference's own outputs '''
    g = load_golden('golden_sonic_RS.npz')
    cols = [str(c) for c in g['columns']]
    for i in range(len(g['configs'])):
        ref = g[f'c{i}_default']
        data = TimeSeries(ref[:, 0], ref[:, 1], {k: ref[:, 2 + j] for j, k in enumerate(cols[2:])})
        assert list(data.columns) == cols
        isp, props = detectSpikes(data)
        np.testing.assert_array_equal(isp, g[f'c{i}_spikes'])
        np.testing.assert_allclose(props['widths'], g[f'c{i}_widths'], rtol=1e-12)
    ref = g['c0_default']
    data = TimeSeries(ref[:, 0], ref[:, 1], {k: ref[:, 2 + j] for j, k in enumerate(cols[2:])})
    m = computeSpikingMetrics([(data, {'pp': PulsedProtocol(0.1, 0.05)})])
    isp = g['c0_spikes']
    prior = isp[ref[isp, 0] < 0.1]
    assert m.shape == (1, 7)
    assert m['mean firing rates (Hz)'][0] == pytest.approx(np.mean(1 / np.diff(ref[prior, 0])))
    assert m['latencies (ms)'][0] == pytest.approx(ref[isp[0], 0] * 1e3)


def test_timeseries_from_block_routes_agree(monkeypatch):
    ''' TimeSeries.from_block builds the frame over the device's row block through pandas' single-block route
        when this pandas has it (probed at import) and through the public constructor otherwise: same frame,
        no copy of the block either way, and the frame behaves like any other (slicing, new columns, pickling,
        the reference's accessors) '''
    import pickle
    from pysonic_amd.core import timeseries as ts
    g = load_golden('golden_sonic_RS.npz')
    cols = [str(c) for c in g['columns']]
    block = np.ascontiguousarray(g['c0_default'], dtype=float)
    frames = {}
    for fast in ([True, False] if ts._FAST_FRAMES else [False]):
        monkeypatch.setattr(ts, '_FAST_FRAMES', fast)
        f = TimeSeries.from_block(block, cols[2:])
        assert type(f) is TimeSeries and list(f.columns) == cols and f.shape == block.shape
        assert np.shares_memory(f.values, block)
        np.testing.assert_array_equal(f.values, block)
        np.testing.assert_array_equal(f.time, block[:, 0])
        assert f.outputs == cols[2:] and type(f.iloc[10:20]) is TimeSeries and f.iloc[10:20].shape == (10, len(cols))
        h = f.copy(); h['extra'] = 1.
        assert h.shape == (block.shape[0], len(cols) + 1) and 'extra' not in f.columns
        assert pickle.loads(pickle.dumps(f)).equals(f)
        isp, _ = detectSpikes(f)
        np.testing.assert_array_equal(isp, g['c0_spikes'])
        frames[fast] = f
    if len(frames) == 2:
        assert frames[True].equals(frames[False])
    # a block that is not contiguous (columns cut off wider rows) takes the public constructor
    wide = np.zeros((block.shape[0], block.shape[1] + 3)); wide[:, :block.shape[1]] = block
    f = TimeSeries.from_block(wide[:, :block.shape[1]], cols[2:])
    np.testing.assert_array_equal(f.values, block)


def test_batch_serial_and_resolve():
    calls = []

    class Dummy:
        def f(self, a, b=2):
            calls.append((a, b))
            return a * b
    d = Dummy()
    out = Batch(d.f, [[1], ([2], {'b': 5}), [3, 4]]).run(mpi=False)
    assert out == [2, 10, 12] and calls == [(1, 2), (2, 5), (3, 4)]
    # mpi=True without a batched implementation falls back to the loop
    assert Batch(d.f, [[1], [2]])(mpi=True) == [2, 4]


def test_threshold_search_matches_reference_histories(api):
    ''' coroutine Thresholder vs the evaluation sequences of the reference's Thresholder.run
        on synthetic step functions (ASTIM and ESTIM parameter sets, incl. no-threshold cases) '''
    import logging
    from pysonic_amd.threshold import threshold_search, titrate_many
    logging.getLogger('PySONIC').setLevel(logging.CRITICAL)
    cases = [dict(xbounds=(0., 6e5), x0=1e4, rel_eps_thr=1e0, eps_thr=1e2, precheck=True),
             dict(xbounds=(0., 1e5), x0=1e0, rel_eps_thr=1e-2, eps_thr=None, precheck=False)]
    gens, hists, thrs = [], [], []
    for e in api['thresholds']:
        kw = cases[e['case']]
        h = []
        gens.append(threshold_search(kw['xbounds'], x0=kw['x0'], eps_thr=kw['eps_thr'],
                                     rel_eps_thr=kw['rel_eps_thr'], precheck=kw['precheck'],
                                     history=h))
        hists.append(h)
        thrs.append(e['thr'])
    # all searches advance in lock-step, as on the GPU
    results, nrounds = titrate_many(lambda items: [x >= thrs[i] for i, x in items], gens)
    assert nrounds == max(len(h) for h in hists)
    for e, h, res in zip(api['thresholds'], hists, results):
        assert [x for x, _ in h] == [v for v in e['x_history'] if not np.isnan(v)]
        assert (e['result'] is None and np.isnan(res)) or res == e['result']
    logging.getLogger('PySONIC').setLevel(logging.INFO)
    with pytest.raises(ValueError):
        next(threshold_search((1., 0.5)))
    with pytest.raises(ValueError):
        next(threshold_search((1., 3.)))          # too narrow for factor bounding


def test_burst_balanced_and_train_protocols():
    ''' BurstProtocol, BalancedPulsedProtocol, getPulseTrainProtocol against values captured from
        the reference (tests/golden/make_golden_protocols.py): event schedules bit-exact '''
    from pysonic_amd import BurstProtocol, BalancedPulsedProtocol, getPulseTrainProtocol
    with open(os.path.join(GOLDEN, 'golden_protocols.json')) as fh:
        g = json.load(fh)

    def check(pp, e):
        ev = pp.stimEvents()
        assert repr(pp) == e['repr'] and pp.desc == e['desc'] and pp.filecodes == e['filecodes']
        assert pp.tstop == e['tstop']
        assert [float(x[0]) for x in ev] == e['ev_t'] and [float(x[1]) for x in ev] == e['ev_x']

    for e in g['burst']:
        pp = BurstProtocol(**e['kwargs'])
        check(pp, e)
        assert repr(pp.copy()) == e['copy']
    for e in g['balanced']:
        pp = BalancedPulsedProtocol(*e['args'], **e['kwargs'])
        check(pp, e)
        assert pp.treversal == e['treversal'] and pp.ttotal == e['ttotal']
        assert pp.DC == e['DC'] and pp.PRF == e['PRF']
    for e in g['train']:
        pp = getPulseTrainProtocol(*e['args'])
        check(pp, e)
        assert pp.tstart == e['tstart'] and pp.tstim == e['tstim'] and pp.DC == e['DC']
    assert [repr(p) for p in BurstProtocol.createQueue(
        [10e-3, 20e-3], [100., 1e3], [0.5, 1.0], [10., 20.], [2, 3])] == g['burstQueue']
    for name, fn in [('BRF too high', lambda: BurstProtocol(20e-3, BRF=60.)),
                     ('xratio > 1', lambda: BalancedPulsedProtocol(1e-3, 1.5, 0.)),
                     ('negative tpulse', lambda: BalancedPulsedProtocol(-1e-3, 0.5, 0.))]:
        assert g['errors'][name] == 'ValueError'
        with pytest.raises(ValueError):
            fn()


@pytest.mark.parametrize('name', ['RS', 'LTS', 'TC'])
def test_quasi_steady_states(name):
    ''' NeuronalBilayerSonophore.getQuasiSteadyStates (nbls.py:573-603) against the reference on the same
        tables (golden_qss_states.npz): duty-cycle-averaged effective potential and the quasi-steady
        state of every state, over amplitude x charge, default / subset / projected charges, squeezed '''
    from pysonic_amd import NeuronalBilayerSonophore, getPointNeuron
    g = np.load(os.path.join(GOLDEN, 'golden_qss_states.npz'))
    nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron(name))
    cases = [dict(amps=None, charges=None, DC=1.0), dict(amps=np.array([20e3, 150e3, 480e3]), charges=None, DC=0.35),
             dict(amps=np.array([50e3, 300e3]), charges=np.array([-80e-5, -60.25e-5, 0., 12e-5]), DC=0.8)]
    for i, case in enumerate(cases):
        lkp, QSS = nbls.getQuasiSteadyStates(500e3, **case)
        assert list(lkp.refs.keys()) == [str(k) for k in g[f'{name}_c{i}_refs']]
        for k, v in lkp.refs.items():
            np.testing.assert_allclose(v, g[f'{name}_c{i}_ref_{k}'], rtol=1e-14)
        np.testing.assert_allclose(lkp['V'], g[f'{name}_c{i}_V'], rtol=1e-12, atol=1e-12)
        assert list(QSS.tables.keys()) == [str(k) for k in g[f'{name}_c{i}_qsskeys']] == nbls.pneuron.statesNames()
        for k, v in QSS.tables.items():
            np.testing.assert_allclose(v, g[f'{name}_c{i}_qss_{k}'], rtol=1e-10, atol=1e-300, err_msg=k)
    lkp, QSS = nbls.getQuasiSteadyStates(500e3, amps=100e3, charges=-65e-5, DC=0.5, squeeze_output=True)
    assert np.shape(lkp['V']) == () and float(lkp['V']) == pytest.approx(float(g[f'{name}_sq_V']), rel=1e-12)
    np.testing.assert_allclose([float(QSS[k]) for k in QSS.tables], g[f'{name}_sq_qss'], rtol=1e-10)
    with pytest.raises(NotImplementedError):
        getPointNeuron('STN').quasiSteadyStates()      # Cai steady state = a scalar root search, as in the reference


def test_log_batch_and_activation_map_files(tmp_path):
    ''' LogBatch (batches.py:186-375) with a stub computation: file name, header, append-as-you-go,
        resume after an interruption, entry lookup; and the activation map's names / inputs / header
        against the reference's own run (golden_actmap.json). No simulation here (GPU test below). '''
    from pysonic_amd.core.batches import LogBatch
    from pysonic_amd.actmap import getActivationMap, FiringRateMap

    class Squares(LogBatch):
        in_key, unit, out_keys, suffix = 'x', 'mm', ['sq', 'cube'], 'pow'
        ncalls = 0

        def corecode(self):
            return 'stub'

        def compute(self, x):
            Squares.ncalls += 1
            if x > 3.5 and getattr(self, 'fail', False):
                raise RuntimeError('interrupted')
            return [x**2, x**3]

    b = Squares(np.array([1., 2., 3., 4., 5.]), root=str(tmp_path))
    assert os.path.basename(b.fpath) == 'stub_x1.0mm-5.0mm_5_pow_results.csv'
    b.fail = True
    with pytest.raises(RuntimeError):
        b.run()
    assert not b.isFinished() and open(b.fpath).read().splitlines()[0] == 'x (mm)\tsq\tcube'
    assert list(b.getInput()) == [1., 2., 3.] and b.isEntry(2. * (1 + 1e-12)) and not b.isEntry(4.)
    b.fail = False
    n0 = Squares.ncalls
    out = b.run()
    assert Squares.ncalls == n0 + 2 and b.isFinished()                      # only the missing inputs
    assert list(out['sq']) == [1., 4., 9., 16., 25.] and b.getEntryOutput(3.)['cube'] == 27.
    assert b.run(mpi=True) is not None and Squares.ncalls == n0 + 2        # nothing left to do
    with pytest.raises(ValueError):
        b.getEntryIndex(7.)
    with pytest.raises(ValueError):
        Squares(np.array([1.]), root=str(tmp_path / 'missing'))

    g = json.load(open(os.path.join(GOLDEN, 'golden_actmap.json')))
    m = getActivationMap('FR', str(tmp_path), getPointNeuron('RS'), 32e-9, 1., 500e3, g['tstim'], g['PRF'],
                         np.array(g['amps']), np.array(g['DCs']))
    assert isinstance(m, FiringRateMap)
    assert os.path.basename(m.fpath) == g['filename'] and m.corecode() == g['corecode'] and m.inputscode == g['inputscode']
    assert [list(map(float, x)) for x in m.inputs] == g['inputs']
    m.createLogFile()
    assert open(m.fpath).read().splitlines()[0] == g['log_text'].splitlines()[0]
    with pytest.raises(ValueError):
        getActivationMap('nope', str(tmp_path), getPointNeuron('RS'), 32e-9, 1., 500e3, 0.1, 100., [1e5], [0.5])


def test_titration_log_cache(tmp_path, monkeypatch):
    ''' nbls.titrate behind the log cache (nbls.py:559, utils.py:457-497): the keys are the reference's call
        signatures, so a slice of its own astim_titrations.log serves as a warm cache; misses are computed
        together (stubbed here: no GPU) and appended '''
    import shutil
    from pysonic_amd.utils import LogCache, methodCallSignature
    log = tmp_path / 'titrations.log'
    shutil.copy(os.path.join(GOLDEN, 'titration_slice.tsv'), log)
    nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron('RS'))
    nbls.titration_cache = LogCache(str(log))
    computed = []
    monkeypatch.setattr(nbls, '_titrate_uncached', lambda calls: computed.append(len(calls)) or [123.5] * len(calls))
    rows = [l.rstrip('\n').split('\t') for l in open(log)]
    drive = AcousticDrive(500e3)
    hit = PulsedProtocol(1., 0., 10., 0.09)
    assert methodCallSignature(nbls.titrate, (drive, hit), {}) == rows[1][0]
    assert nbls.titrate(drive, hit) == float(rows[1][1]) and not computed
    assert np.isnan(nbls.titrate(drive, PulsedProtocol(1., 0., 10., 0.05)))          # a logged 'nan'
    miss = PulsedProtocol(0.5, 0., 10., 0.09)
    out = Batch(nbls.titrate, [[drive, hit], [drive, miss], [drive, miss, 1.]]).run(mpi=True)
    assert out[0] == float(rows[1][1]) and out[1] == out[2] == 123.5 and computed == [2]
    assert nbls.titrate(drive, miss) == 123.5 and computed == [2]                   # now logged
    assert open(log).read().splitlines()[-1].endswith('\t123.5')
    nbls.titration_cache = None
    assert nbls.titrate(drive, hit) == 123.5 and computed == [2, 1]


def test_pack_configs_memo_equals_per_config_events():
    ''' the CSR event arrays of a queue (include/pysonic_amd.h) with the per-protocol memo of _packConfigs
        equal the ones built protocol by protocol from stimEvents() (solvers.py:441-443), for repeated,
        burst and array-valued (unhashable) protocols alike '''
    from pysonic_amd import (NeuronalBilayerSonophore, AcousticDrive, PulsedProtocol, BurstProtocol,
                             CustomProtocol, getPointNeuron)
    nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron('RS'))
    pps = [PulsedProtocol(50e-3, 10e-3, 100., 0.5), PulsedProtocol(50e-3, 10e-3, 100., 0.5),
           PulsedProtocol(50e-3, 10e-3), BurstProtocol(10e-3, 100., 0.5, 5., 3),
           CustomProtocol([0., 10e-3, 20e-3], [1., 0.5, 0.], 30e-3), PulsedProtocol(50e-3, 10e-3, 100., 0.5)]
    cfgs = [(AcousticDrive(500e3, 1e3 * (i + 1)), pp) for i, pp in enumerate(pps)]
    A, tstop, dt, ev_t, ev_x, ev_off = nbls._packConfigs(cfgs)
    assert list(A) == [1e3 * (i + 1) for i in range(len(pps))] and np.all(dt == nbls.pneuron.chooseTimeStep())
    assert ev_off.dtype == np.int64 and ev_off[0] == 0 and ev_off[-1] == ev_t.size == ev_x.size
    for i, pp in enumerate(pps):
        ev = sorted(pp.stimEvents(), key=lambda e: e[0])
        np.testing.assert_array_equal(ev_t[ev_off[i]:ev_off[i + 1]], [e[0] for e in ev])
        np.testing.assert_array_equal(ev_x[ev_off[i]:ev_off[i + 1]], [e[1] for e in ev])
        assert tstop[i] == pp.tstop
