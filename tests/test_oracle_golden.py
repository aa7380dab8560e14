# -*- coding: utf-8 -*-
''' The oracle (oracle/oracle.py + sonic_oracle.c) against golden vectors captured from the
    reference itself (tests/golden/make_golden_*.py). CPU only.

    Two kinds of comparison:
      * "tight": both sides integrate with rtol=1e-12 -> both converge to the exact solution of
        the same equations, so any restatement error shows up undiluted (bar: 1e-12 .. 1e-9);
      * "default": scipy odeint defaults on both sides. LSODA's adaptive path amplifies 1-ulp
        differences between numpy's SIMD exp/pow and libm's to the tolerance level, so the bar
        is the reference's own default-vs-tight distance, not round-off.
'''
import ctypes
import os

import numpy as np
import pytest

from conftest import GOLDEN, NEURONS, load_tables, load_golden, rms
from oracle import oracle as O

TIGHT = dict(rtol=1e-12, atol=1e-15, mxstep=100000)


def test_neuron_definitions():
    ''' states / rates order, y0, Qm0, Qbounds, rate functions, iNet, true derivatives '''
    L = O.lib()
    for n in NEURONS + ['IB']:
        g = load_golden('golden_IB.npz' if n == 'IB' else 'golden_neurons.npz')
        nid = O.NEURON_IDS[n]
        assert list(g[f'{n}_states']) == O.STATES[n]
        assert list(g[f'{n}_rates']) == O.RATES[n]
        assert L.orc_nstates(nid) == len(O.STATES[n]) and L.orc_nrates(nid) == len(O.RATES[n])
        assert O.neuron_Qm0(n) == float(g[f'{n}_Qm0'])
        np.testing.assert_array_equal(O.neuron_Qbounds(n), g[f'{n}_Qbounds'])
        np.testing.assert_allclose(O.steady_states(n), g[f'{n}_y0'], rtol=1e-12)
        r = O.rates(n, g['Vsamples'])
        ref = g[f'{n}_ratevals']
        for i, k in enumerate(O.RATES[n]):
            ok = np.isfinite(ref[i])
            np.testing.assert_allclose(r[k][ok], ref[i][ok], rtol=1e-12, err_msg=f'{n} {k}')
        for pt, inet, ders in zip(g[f'{n}_pts'], g[f'{n}_iNet'], g[f'{n}_ders']):
            x = np.ascontiguousarray(pt[1:])
            assert L.orc_iNet(nid, float(pt[0]), O._ptr(x)) == pytest.approx(inet, rel=1e-13)
            y = np.concatenate(([pt[0] * 1e-2 * 1e-3], x))   # Qm = Cm0 * Vm * 1e-3
            dy = np.empty_like(y)
            L.orc_hh_rhs(nid, y.ctypes.data, 1e-2, dy.ctypes.data)
            np.testing.assert_allclose(dy[1:], ders, rtol=2e-11, err_msg=n)
            assert dy[0] == pytest.approx(-inet * 1e-3, rel=2e-11)


def test_interp_matches_numpy():
    rng = np.random.default_rng(0)
    xp = np.arange(-107e-5, 50e-5 + 1e-5, 1e-5)
    fp = rng.normal(size=xp.size)
    L = O.lib()
    xs = np.concatenate([rng.uniform(xp[0], xp[-1], 2000), xp, [xp[0] - 1e-9, xp[-1] + 1e-9]])
    mine = np.array([L.orc_interp(float(x), O._ptr(xp), O._ptr(fp), xp.size) for x in xs])
    ref = np.interp(xs, xp, fp, left=np.nan, right=np.nan)
    np.testing.assert_array_equal(mine, ref)


@pytest.mark.parametrize('icfg', [0, 1, 2, 6, 9, 11])
def test_sonic_RS_default(icfg):
    ''' full pipeline at scipy default tolerances: exact grid, LSODA-noise-level states '''
    A, Q, keys, tables = load_tables('RS')
    g = load_golden('golden_sonic_RS.npz')
    cols = list(g['columns'])
    assert cols == ['t', 'stimstate', 'Qm', 'm', 'h', 'n', 'p', 'Vm', 'Z', 'ng']
    amp, tstim, toffset, PRF, DC = g['configs'][icfg]
    out = O.sim_sonic('RS', A, Q, tables, amp, *O.pulsed_events(tstim, toffset, PRF, DC))
    ref = g[f'c{icfg}_default']
    np.testing.assert_array_equal(out['t'], ref[:, 0])
    np.testing.assert_array_equal(out['stimstate'], ref[:, 1])
    # the reference's own default-vs-tight distance on these configs is 2e-8 .. 1.4e-7 C/m2
    assert rms(out['Qm'], ref[:, 2]) < 2e-8
    for k in ['m', 'h', 'n', 'p']:
        assert rms(out[k], ref[:, cols.index(k)]) < 5e-5
    assert np.nanmax(np.abs(out['Vm'] - ref[:, cols.index('Vm')])) < 0.05   # mV
    assert np.all(np.isnan(ref[:, cols.index('Z')])) and np.all(np.isnan(ref[:, cols.index('ng')]))


@pytest.mark.parametrize('icfg', [0, 2, 6])
def test_sonic_RS_tight(icfg):
    ''' converged solutions agree to round-off: pins the restatement '''
    A, Q, keys, tables = load_tables('RS')
    g = load_golden('golden_sonic_RS.npz')
    amp, tstim, toffset, PRF, DC = g['configs'][icfg]
    out = O.sim_sonic('RS', A, Q, tables, amp, *O.pulsed_events(tstim, toffset, PRF, DC),
                      odeint_kwargs=TIGHT)
    ref = g[f'c{icfg}_tight']
    assert rms(out['Qm'], ref[:, 0]) < 1e-12
    for i, k in enumerate(['m', 'h', 'n', 'p']):
        assert rms(out[k], ref[:, 1 + i]) < 1e-9


@pytest.mark.parametrize('name', ['FS', 'LTS', 'RE', 'TC', 'STN', 'IB'])
def test_sonic_other_neurons_tight(name):
    fpath = os.path.join(GOLDEN, f'golden_sonic_{name}.npz')
    if not os.path.isfile(fpath):
        pytest.skip(f'{fpath} not generated yet')
    A, Q, keys, tables = load_tables(name)
    assert keys == ['V'] + O.RATES[name]
    g = np.load(fpath)
    cols = list(g['columns'])
    assert cols == ['t', 'stimstate', 'Qm'] + O.STATES[name] + ['Vm', 'Z', 'ng']
    icfg = 1
    amp, tstim, toffset, PRF, DC = g['configs'][icfg]
    ev, tstop = O.pulsed_events(tstim, toffset, PRF, DC)
    out = O.sim_sonic(name, A, Q, tables, amp, ev, tstop, odeint_kwargs=TIGHT)
    ref = g[f'c{icfg}_tight']
    # bursting neurons (LTS, TC) amplify round-off: two oracle runs at rtol 1e-12 / 1e-13 differ
    # by ~1e-10 C/m2 RMS on LTS, so the bar is 5e-10 rather than the 1e-12 reached on RS
    assert rms(out['Qm'], ref[:, 0]) < 5e-10
    for i, k in enumerate(O.STATES[name]):
        scale = max(np.abs(ref[:, 1 + i]).max(), 1e-30)
        assert rms(out[k], ref[:, 1 + i]) / scale < 1e-6, k
    out_d = O.sim_sonic(name, A, Q, tables, amp, ev, tstop)
    refd = g[f'c{icfg}_default']
    np.testing.assert_array_equal(out_d['t'], refd[:, 0])
    np.testing.assert_array_equal(out_d['stimstate'], refd[:, 1])
    # default tolerances: some configurations are ill-conditioned (the reference's own default
    # and rtol=1e-12 runs differ by up to 2e-4 C/m2 RMS on them), so the bar scales with that
    spread = rms(refd[:, 2], ref[:, 0])
    assert rms(out_d['Qm'], refd[:, 2]) < max(1e-7, 3 * spread)


def test_detect_spikes_matches_reference():
    g = load_golden('golden_sonic_RS.npz')
    cols = list(g['columns'])
    for i in range(len(g['configs'])):
        ref = g[f'c{i}_default']
        isp, props = O.detect_spikes(ref[:, 0], ref[:, cols.index('Qm')])
        np.testing.assert_array_equal(isp, g[f'c{i}_spikes'])
        np.testing.assert_allclose(props['widths'], g[f'c{i}_widths'], rtol=1e-12)
        np.testing.assert_allclose(props['prominences'], g[f'c{i}_prominences'], rtol=1e-12)


def _bls(name='RS'):
    pm = O.load_pm_params(os.path.join(GOLDEN, 'bls_params.json'), 32e-9, O.neuron_Qm0(name))
    return O.bls_params(32e-9, 1e-2, O.neuron_Qm0(name), pm)


def test_known_answers_survey():
    ''' SURVEY.md section 8 A4/A5 probe values (RS, 32 nm, 500 kHz, 100 kPa, fs = 1) '''
    p = _bls()
    assert p.Delta == 1.2553492695740507e-9 and p.LJ_nrep == 3.9147721975981384
    ev = O.compute_eff_vars('RS', p, 500e3, 100e3, -71.9e-5)
    known = {'V': -136.7874, 'alpham': 14.4579, 'betam': 33764.8657, 'alphah': 10509286.3933,
             'betah': 0.1113, 'alphan': 3.1652, 'betan': 35427.8347, 'alphap': 0.2137,
             'betap': 46970.6717}
    for k, v in known.items():
        assert ev[k] == pytest.approx(v, rel=2e-5, abs=6e-5), k   # SURVEY rounds to 4 decimals
    ev0 = O.compute_eff_vars('RS', p, 500e3, 100e3, 0.)
    assert ev0['V'] == 0. and ev0['alpham'] == pytest.approx(13824.282, rel=2e-5)


def test_mech_cycles_and_effvars():
    g = load_golden('golden_mech.npz')
    p = _bls()
    f = float(g['f'])
    keys = [str(k) for k in g['keys']]
    tight = dict(rtol=1e-12, atol=np.array([1e-12, 1e-21, 1e-34]), mxstep=1000000)
    for i in [0, 1, 5, 9, 13, 15, 21]:
        A, Q = g['pairs'][i]
        # default tolerances: same number of cycles except where LSODA noise decides convergence
        data, ncycles, conv = O.sim_cycles(p, f, A, Q)
        if A == 0.:
            assert ncycles == 11 and not conv          # 0/0 convergence ratio quirk
            assert data['t'].size == int(g[f'p{i}_default_nrows']) == 10991
        # tight tolerances: cycle count, last-cycle traces and effective variables match
        data, ncycles, conv = O.sim_cycles(p, f, A, Q, odeint_kwargs=tight)
        assert data['t'].size == int(g[f'p{i}_tight_nrows'])
        np.testing.assert_allclose(data['Z'][-1000:], g[f'p{i}_tight_Z'], rtol=1e-7, atol=1e-18)
        np.testing.assert_allclose(data['ng'][-1000:], g[f'p{i}_tight_ng'], rtol=1e-9)
        ev = O.compute_eff_vars('RS', p, f, A, Q, odeint_kwargs=tight)
        ref = g[f'p{i}_tight_eff']
        mine = np.array([ev[k] for k in keys])
        ok = np.isfinite(ref)
        np.testing.assert_allclose(mine[ok], ref[ok], rtol=5e-8, atol=1e-12)
        # default: within the reference's own default-vs-tight spread
        evd = O.compute_eff_vars('RS', p, f, A, Q)
        mined = np.array([evd[k] for k in keys])
        spread = np.abs(g[f'p{i}_default_eff'][ok] - ref[ok])
        assert np.all(np.abs(mined[ok] - ref[ok]) <= 20 * spread + 1e-5 * np.abs(ref[ok]) + 1e-12)


def test_effvars_with_charge_overtones():
    ''' computeEffVars with Qm_overtones (nbls.py:169-201): the oracle against the reference, both
        at tight tolerances; effective potential, its Fourier coefficients and the rates '''
    g = load_golden('golden_overtones.npz')
    tight = dict(rtol=1e-12, atol=np.array([1e-12, 1e-21, 1e-34]), mxstep=1000000)
    for i in range(int(g['ncases'])):
        f, A, Q0 = g[f'c{i}_in']
        ov = [tuple(x) for x in g[f'c{i}_ov']]
        cols = [str(c) for c in g[f'c{i}_cols']]
        p = _bls()            # no embedding depth: the tissue modulus does not depend on f
        for j, fs in enumerate(g[f'c{i}_fs']):
            ev = O.compute_eff_vars('RS', p, f, A, Q0, fs=fs, odeint_kwargs=tight, Qm_overtones=ov)
            assert list(ev.keys()) == cols
            mine = np.array([ev[k] for k in cols])
            np.testing.assert_allclose(mine, g[f'c{i}_tight'][j], rtol=2e-7, atol=1e-9), (i, j)


def test_effvars_IB():
    ''' computeEffVars for the intrinsically bursting neuron (cortical.py:307-400) '''
    g = load_golden('golden_IB.npz')
    p = _bls('IB')
    keys = [str(k) for k in g['keys']]
    tight = dict(rtol=1e-12, atol=np.array([1e-12, 1e-21, 1e-34]), mxstep=1000000)
    for i, (A, Q) in enumerate(g['pairs']):
        ev = O.compute_eff_vars('IB', p, float(g['f']), A, Q, odeint_kwargs=tight)
        ref = g[f'p{i}_tight_eff']
        ok = np.isfinite(ref)
        np.testing.assert_allclose(np.array([ev[k] for k in keys])[ok], ref[ok], rtol=5e-8, atol=1e-12)


def test_lookup_cells_against_reference_tables():
    ''' a few cells of the shipped tables (made by the reference's computeEffVars) '''
    A, Q, keys, tables = load_tables('RS')
    p = _bls()
    for ia, iq in [(0, 35), (25, 107), (50, 0), (40, 157)]:
        ev = O.compute_eff_vars('RS', p, 500e3, float(A[ia]), float(Q[iq]))
        for k, name in enumerate(keys):
            assert ev[name] == pytest.approx(tables[k, ia, iq], rel=1e-4, abs=1e-9), (name, ia, iq)


def test_full_RS():
    fpath = os.path.join(GOLDEN, 'golden_full_RS.npz')
    if not os.path.isfile(fpath):
        pytest.skip('golden_full_RS.npz not generated yet')
    g = np.load(fpath)
    cols = list(g['columns'])
    assert cols == ['t', 'stimstate', 'Z', 'ng', 'Qm', 'm', 'h', 'n', 'p', 'Vm']
    p = _bls()
    ev, tstop = O.pulsed_events(20e-6, 4e-6)
    out = O.sim_full('RS', p, 500e3, 100e3, ev, tstop)
    ref = g['default']
    assert out['t'].size == ref.shape[0] == 2400
    np.testing.assert_allclose(out['t'], ref[:, 0], rtol=0, atol=1e-18)
    np.testing.assert_array_equal(out['stimstate'], ref[:, 1])
    assert rms(out['Z'], ref[:, cols.index('Z')]) < 1e-13          # m (Z ~ 1e-9)
    assert rms(out['Qm'], ref[:, cols.index('Qm')]) < 1e-11
    assert rms(out['Vm'], ref[:, cols.index('Vm')]) < 1e-2          # mV (Vm swings ~ 500 mV)


@pytest.mark.parametrize('icfg', [0, 1])
def test_sonic_RS_burst_protocol_tight(icfg):
    ''' the oracle under the reference's BurstProtocol schedules (golden_sonic_burst_RS.npz) '''
    import json
    from pysonic_amd import BurstProtocol
    A, Q, keys, tables = load_tables('RS')
    g = load_golden('golden_sonic_burst_RS.npz')
    pp = BurstProtocol(**json.loads(str(g['kwargs']))[icfg])
    out = O.sim_sonic('RS', A, Q, tables, float(g['A'][icfg]),
                      [(float(t), float(x)) for t, x in pp.stimEvents()], pp.tstop,
                      odeint_kwargs=TIGHT)
    ref, dflt = g[f'c{icfg}_tight'], g[f'c{icfg}_default']
    np.testing.assert_array_equal(out['t'], dflt[:, 0])
    np.testing.assert_array_equal(out['stimstate'], dflt[:, 1])
    assert rms(out['Qm'], ref[:, 0]) < 1e-12


def test_hybrid_RS():
    ''' the oracle's restatement of HybridSolver against the reference's hybrid run
        (golden_hybrid_RS.npz, configuration 0: 1.2 ms CW + 0.4 ms offset = 4 update intervals):
        time grid and stimulus state bit-exact, variables within the reference's own
        default-vs-tight spread (its sparse phases run dop853 at rtol 1e-6) '''
    g = load_golden('golden_hybrid_RS.npz')
    cols = [str(c) for c in g['c0_columns']]
    assert cols == ['t', 'stimstate', 'Z', 'ng', 'Qm', 'm', 'h', 'n', 'p', 'Vm']
    A, tstim, toffset, PRF, DC = g['configs'][0]
    ev, tstop = O.pulsed_events(tstim, toffset, PRF, DC)
    out = O.sim_hybrid('RS', _bls(), 500e3, A, ev, tstop)
    ref, tight, dec = g['c0_default'], g['c0_tight'], int(g['decimation'])
    assert out['t'].size == int(g['c0_nrows'])
    np.testing.assert_array_equal(out['t'], np.linspace(*g['c0_t_first_last'], out['t'].size))
    np.testing.assert_array_equal(out['stimstate'], g['c0_stimstate'].astype(float))
    for i, k in enumerate(cols):
        if i < 2:
            continue
        spread = rms(ref[:, i], tight[:, i])
        assert rms(out[k][::dec], ref[:, i]) <= 3 * spread, k


@pytest.mark.parametrize('icfg', [0, 1, 2, 3])
def test_sonic_qss_vars_tight(icfg):
    ''' quasi-steady-state variables (qss_vars of NBLS.simulate): the oracle against the reference's
        rtol=1e-12 run, including the QSS columns (interpolated nodal x_inf) and the column order '''
    import json
    g = load_golden('golden_sonic_qss.npz')
    name, amp, tstim, toffset, PRF, DC, qss = json.loads(str(g['configs']))[icfg]
    A, Q, keys, tables = load_tables(name)
    out = O.sim_sonic(name, A, Q, tables, amp, *O.pulsed_events(tstim, toffset, PRF, DC),
                      odeint_kwargs=TIGHT, qss_vars=qss)
    cols = [str(c) for c in g[f'c{icfg}_columns']]
    states = O.STATES[name]
    assert cols == ['t', 'stimstate', 'Qm'] + [k for k in states if k not in qss] + ['Vm'] + qss + ['Z', 'ng']
    ref = g[f'c{icfg}_tight']
    np.testing.assert_array_equal(out['t'], ref[:, 0])
    for k in ['Qm'] + states:
        scale = max(np.abs(ref[:, cols.index(k)]).max(), 1e-30)
        assert rms(out[k], ref[:, cols.index(k)]) / scale < 2e-8, k


@pytest.mark.parametrize('a, f', [(16e-9, 20e3), (64e-9, 1e6), (16e-9, 4e6), (64e-9, 100e3)])
def test_mech_other_radii_and_frequencies(a, f):
    ''' computeEffVars of the reference at the other radii / frequencies of BASELINE config 3
        (golden_mech_axes.npz): cycle counts and effective variables of the converged runs '''
    g = load_golden('golden_mech_axes.npz')
    keys = [str(k) for k in g['keys']]
    pm = O.load_pm_params(os.path.join(GOLDEN, 'bls_params.json'), a, O.neuron_Qm0('RS'))
    p = O.bls_params(a, 1e-2, O.neuron_Qm0('RS'), pm)
    tight = dict(rtol=1e-12, atol=np.array([1e-12, 1e-21, 1e-34]), mxstep=1000000)
    idx = [i for i, c in enumerate(g['cells']) if c[0] == a and c[1] == f and c[2] <= 600e3]
    assert len(idx) == 7
    for i in idx[1:5]:
        _, _, A, Q = g['cells'][i]
        data, ncycles, conv = O.sim_cycles(p, f, A, Q, odeint_kwargs=tight)
        assert data['t'].size == int(g[f'c{i}_tight_nrows'])
        ev = O.compute_eff_vars('RS', p, f, A, Q, odeint_kwargs=tight)
        ref = g[f'c{i}_tight_eff']
        mine = np.array([ev[k] for k in keys])
        ok = np.isfinite(ref)
        np.testing.assert_allclose(mine[ok], ref[ok], rtol=2e-7, atol=1e-12)


@pytest.mark.parametrize('name', ['RS', 'LTS'])
def test_sonic_second_frequency_tight(name):
    ''' sonic at a second frequency (RS 100 kHz, LTS 2 MHz): the reference fed with the committed
        device-made tables (golden_sonic_freq.npz) against the oracle fed with the same tables '''
    g = load_golden('golden_sonic_freq.npz')
    f = float(g[f'{name}_f'])
    d = np.load(os.path.join(GOLDEN, f'devtables_{name}_32nm_{f * 1e-3:.0f}kHz.npz'))
    keys = [str(k) for k in d['keys']]
    assert keys == ['V'] + O.RATES[name]
    tables = np.array([d[f'tab_{k}'] for k in keys])
    for icfg in (0, 2):
        amp, tstim, toffset, PRF, DC = g[f'{name}_configs'][icfg]
        ev, tstop = O.pulsed_events(tstim, toffset, PRF, DC)
        out = O.sim_sonic(name, d['A'], d['Q'], tables, amp, ev, tstop, odeint_kwargs=TIGHT)
        ref, refd = g[f'{name}_c{icfg}_tight'], g[f'{name}_c{icfg}_default']
        np.testing.assert_array_equal(out['t'], refd[:, 0])
        np.testing.assert_array_equal(out['stimstate'], refd[:, 1])
        assert rms(out['Qm'], ref[:, 0]) < 5e-10


def test_sonic_STN_high_amplitude_tight():
    ''' OtsukaSTN at 484 and 600 kPa (golden_sonic_STN_range.npz): the reference completes, and so
        does the oracle, on the same rows '''
    g = load_golden('golden_sonic_STN_range.npz')
    A, Q, keys, tables = load_tables('STN')
    for icfg in (2, 3):
        amp, tstim, toffset, PRF, DC = g['configs'][icfg]
        assert not bool(g[f'c{icfg}_tight_raised'])
        ev, tstop = O.pulsed_events(tstim, toffset, PRF, DC)
        out = O.sim_sonic('STN', A, Q, tables, amp, ev, tstop, odeint_kwargs=TIGHT)
        np.testing.assert_array_equal(out['t'], g[f'c{icfg}_t'])
        assert rms(out['Qm'], g[f'c{icfg}_tight_Qm']) < 5e-9
