# -*- coding: utf-8 -*-
''' The cooperative integrator cores on the CPU: csrc/sonic_group.hpp, full_coop.hpp and hybrid_coop.hpp are
    written over an `Ops` backend, and tests/native/harness.cpp instantiates them with the array-emulation
    backends (16 / 8 values per "wavefront row" instead of DPP moves). Same arithmetic as on the device up to
    the order of the sums, so the box without a GPU checks the new kernels' numerics against the reference
    goldens too. (The HIP build of the same headers is what tests/test_gpu_*.py run through the C ABI.) '''
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT, GOLDEN, load_golden, load_tables, rms
from oracle import oracle as O

dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)


@pytest.fixture(scope='module')
def harness(tmp_path_factory):
    if os.environ.get('HARNESS_SO'):             # a prebuilt harness, e.g. an -fsanitize=address,undefined build
        return ctypes.CDLL(os.environ['HARNESS_SO'])
    cxx = shutil.which('g++')
    if cxx is None:
        pytest.skip('no host C++ compiler')
    so = os.path.join(tmp_path_factory.mktemp('harness'), 'libharness.so')
    subprocess.run([cxx, '-O2', '-shared', '-fPIC', '-o', so, os.path.join(ROOT, 'tests', 'native', 'harness.cpp')],
                   check=True, capture_output=True, timeout=600)
    return ctypes.CDLL(so)


def _schedule(events, tstop, dt):
    t0s, t1s, xs, ns, lv = [], [], [], [], []
    tnow, xcur = 0., 0.
    for te, xe in sorted(events, key=lambda e: e[0]) + [(tstop, None)]:
        t0s.append(tnow); t1s.append(te); xs.append(xcur); ns.append(O.get_nsamples(tnow, te, dt))
        lv.append(0 if xcur == 0. else 1)
        if xe is not None:
            xcur = xe
        tnow = te
    return (np.array(t0s), np.array(t1s), np.array(xs), np.array(ns, dtype=np.int32), np.array(lv, dtype=np.int32))


def _records(Aref, Qref, tables, amps):
    recs = np.empty((len(amps), Qref.size - 1, 2 + 2 * tables.shape[0]))
    for l, A in enumerate(amps):
        t1d = O.project_A(Aref, tables, A)
        recs[l, :, 0], recs[l, :, 1] = Qref[:-1], Qref[1:]
        recs[l, :, 2::2] = t1d[:, :-1].T
        recs[l, :, 3::2] = ((t1d[:, 1:] - t1d[:, :-1]) / (Qref[1:] - Qref[:-1])).T
    return np.ascontiguousarray(recs)


@pytest.mark.parametrize('name', ['LTS', 'RE', 'TC', 'STN', 'HHseg', 'FHnode'])      # fixed lane roles; roles derived from the parameter block
def test_group_core_against_lane_core_and_golden(harness, name):
    ''' sonic_group.hpp (one configuration per 16 lanes, emulated) on the first golden configuration of the
        neuron: same rows as the lane-per-configuration core to rounding amplified by the dynamics, and within
        the bar of tests/test_gpu_parity.py against the reference's converged run '''
    from pysonic_amd.neurons import getPointNeuron
    Aref, Qref, keys, tables = load_tables(name)
    g = load_golden(f'golden_sonic_{name}.npz')
    pn = getPointNeuron(name)
    P = np.ascontiguousarray(pn.device_params(), dtype=float)
    y0 = np.concatenate(([pn.Qm0], pn.getSteadyStates(pn.Vm0)))
    A, tstim, toffset, PRF, DC = g['configs'][0]
    recs = _records(Aref, Qref, tables, [0., float(A)])
    ev, tstop = O.pulsed_events(tstim, toffset, PRF, DC)
    t0s, t1s, xs, ns, lv = _schedule(ev, tstop, pn.chooseTimeStep())
    N = 1 + int(ns.sum())
    out = {}
    for kind, fn in (('lane', harness.harness_run), ('group', harness.harness_run_group)):
        rows = np.zeros((N, y0.size + 3)); nst, nrj = ctypes.c_int(), ctypes.c_int()
        st = fn(pn.native_id, P.ctypes.data_as(dp), recs.ctypes.data_as(dp), 2, Qref.size - 1,
                ctypes.c_double(Qref[0]), ctypes.c_double(Qref[-1]), ctypes.c_double(1 / 1e-5),
                t0s.ctypes.data_as(dp), t1s.ctypes.data_as(dp), xs.ctypes.data_as(dp), ns.ctypes.data_as(ip),
                lv.ctypes.data_as(ip), len(ns), y0.ctypes.data_as(dp), ctypes.c_double(1e-6), ctypes.c_double(1e-8),
                ctypes.c_double(1e-6), ctypes.c_double(1e-30), 10000000, rows.ctypes.data_as(dp),
                ctypes.byref(nst), ctypes.byref(nrj))
        assert st == 0
        out[kind] = (rows, nst.value)
    (rl, nl), (rg, ng) = out['lane'], out['group']
    ref, tight = g['c0_default'], g['c0_tight']
    assert rg.shape == (ref.shape[0], ref.shape[1] - 2)            # the reference adds the NaN Z / ng columns
    np.testing.assert_array_equal(rg[:, 0], ref[:, 0])                  # t, stimstate: bit-exact
    np.testing.assert_array_equal(rg[:, 1], ref[:, 1])
    spread = rms(ref[:, 2], tight[:, 0])
    assert rms(rg[:, 2], tight[:, 0]) <= max(3e-8, 2 * spread), (name, rms(rg[:, 2], tight[:, 0]), spread)
    assert rms(rg[:, 2], rl[:, 2]) <= max(3e-8, 2 * spread)
    assert abs(ng - nl) <= 0.02 * nl
    for j in range(2, rg.shape[1]):                                     # every column, against the lane core
        scale = max(np.abs(rl[:, j]).max(), 1e-30)
        assert rms(rg[:, j], rl[:, j]) <= 1e-4 * scale, (name, j)


@pytest.mark.parametrize('name', ['RS', 'FS'])
def test_coop_rhs_against_oracle_rhs(harness, name):
    ''' one evaluation of the octet-cooperative right-hand side (full_coop.hpp: one exponential and one logarithm
        for all lanes, pressure terms as per-lane linear forms, two sums in one butterfly) against the oracle's
        restatement of NeuronalBilayerSonophore.fullDerivatives (nbls.py:265-278), on states drawn over the range
        a 600 kPa run visits -- deflections from the clamp to 12 nm, potentials down to -500 mV -- and at
        Z = 0 exactly; the mechanical system alone (lookup generation) on the same states '''
    from pysonic_amd import NeuronalBilayerSonophore, getPointNeuron
    from test_oracle_golden import _bls
    pn = getPointNeuron(name); nbls = NeuronalBilayerSonophore(32e-9, pn)
    P = np.ascontiguousarray(pn.device_params()); B = np.ascontiguousarray(nbls.device_params())
    p = _bls(name)
    L = O.lib(); nid = O.NEURON_IDS[name]
    rng = np.random.default_rng(5 + len(name))
    f, phi = 500e3, np.pi
    worst = 0.
    for k in range(400):
        Z = 0. if k == 0 else float(rng.choice([rng.uniform(-0.6e-9, 0.5e-9), rng.uniform(0.5e-9, 12e-9)]))
        y = np.array([rng.uniform(-0.3, 0.3), Z, p.ng0 * rng.uniform(0.5, 1.5), rng.uniform(-80e-5, 40e-5),
                      rng.uniform(0, 1), rng.uniform(0, 1), rng.uniform(0, 1), rng.uniform(0, 1)])
        A, t, fs = float(rng.choice([0., 50e3, 600e3])), float(rng.uniform(0, 2e-6)), float(rng.choice([1., 0.75]))
        pac = A * np.sin(2 * np.pi * f * t - phi)
        ref = np.empty(8); cl = ctypes.c_int(0)
        L.orc_full_rhs(nid, ctypes.byref(p), ctypes.c_double(t), y.ctypes.data, ctypes.c_double(f), ctypes.c_double(A),
                       ctypes.c_double(phi), ctypes.c_double(fs), ref.ctypes.data, ctypes.byref(cl))
        out = np.empty(8)
        c = harness.harness_coop_rhs(pn.native_id, P.ctypes.data_as(dp), B.ctypes.data_as(dp), ctypes.c_double(fs),
                                     ctypes.c_double(pac), 1, y.ctypes.data_as(dp), out.ctypes.data_as(dp))
        assert c == cl.value
        # dU / dt is a sum of pressure terms that cancel to a few 1e-3 of the largest: bar relative to it
        scale = np.maximum(np.abs(ref), [1e-3 * np.abs(ref[0]) + 1e3, 0, 0, 0, 0, 0, 0, 0])
        err = np.abs(out - ref) / np.maximum(scale, 1e-300)
        assert np.all(err[1:] < 1e-10), (k, y, err)
        assert err[0] < 1e-9, (k, y, err)
        worst = max(worst, err[1:].max())
        mech = np.empty(8)
        harness.harness_coop_rhs(pn.native_id, P.ctypes.data_as(dp), B.ctypes.data_as(dp), ctypes.c_double(fs),
                                 ctypes.c_double(pac), 0, y.ctypes.data_as(dp), mech.ctypes.data_as(dp))
        np.testing.assert_allclose(mech[:3], out[:3], rtol=1e-12, atol=0)
        assert np.all(mech[3:] == 0)


def test_hybrid_coop_core_against_golden(harness):
    ''' hybrid_coop.hpp (one configuration per 8 lanes, emulated; dense periods on the 8(5,3) pair, sparse phase
        on RODAS4) on the reference's CW hybrid run: bars of tests/test_gpu_full.py::test_hybrid_golden '''
    from pysonic_amd import NeuronalBilayerSonophore, getPointNeuron
    g = load_golden('golden_hybrid_RS.npz')
    pn = getPointNeuron('RS'); nbls = NeuronalBilayerSonophore(32e-9, pn)
    P = np.ascontiguousarray(pn.device_params()); B = np.ascontiguousarray(nbls.device_params())
    y0 = np.ascontiguousarray(nbls.initialConditionsSonic())
    A, tstim, toff, PRF, DC = g['configs'][0]
    ev, tstop = O.pulsed_events(tstim, toff, PRF, DC)
    ev_t, ev_x = np.array([e[0] for e in ev]), np.array([e[1] for e in ev])
    M = O.get_nsamples(0., tstop, 1e-8)
    tr = np.zeros((M, 10)); st, nst, ncy = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    scratch = np.zeros(harness.harness_hybrid_scratch_doubles())
    harness.harness_hybrid_coop(0, P.ctypes.data_as(dp), B.ctypes.data_as(dp), ctypes.c_double(500e3),
                                ctypes.c_double(float(A)), ctypes.c_double(1.), ctypes.c_double(tstop),
                                ev_t.ctypes.data_as(dp), ev_x.ctypes.data_as(dp), len(ev), ctypes.c_longlong(M),
                                y0.ctypes.data_as(dp), ctypes.c_double(1e-7), 2000000000, tr.ctypes.data_as(dp),
                                scratch.ctypes.data_as(dp), ctypes.byref(st), ctypes.byref(nst), ctypes.byref(ncy))
    assert st.value == 0 and M == int(g['c0_nrows'])
    np.testing.assert_array_equal(tr[:, 1], g['c0_stimstate'].astype(float))
    ref, tight, dec = g['c0_default'], g['c0_tight'], int(g['decimation'])
    for i in range(2, 10):
        ptp, spread = np.ptp(tight[:, i]), rms(ref[:, i], tight[:, i])
        assert rms(tr[::dec, i], tight[:, i]) <= max(0.5 * spread, 1e-7 * ptp), i
    # one step per sparse step: a few hundred thousand attempts, not the millions of an explicit sparse phase
    assert 1e5 < nst.value < 2e5 and 150 < ncy.value < 400


@pytest.mark.parametrize('name,A', [('LTS', 100e3), ('STN', 100e3), ('MRGnode', 30e3)])
def test_hybrid_row_core_against_lane_core(harness, name, A):
    """ hybrid_row.hpp (one configuration per 16-lane row, emulated; dense periods on the 8(5,3) pair with dense output,
        sparse phases on RODAS4 over the membrane states) against hybrid_core.hpp (one per lane, 5(4) pair): 0.6 ms ON +
        0.15 ms OFF -- the same row grid, the same number of dense periods, every variable within 2e-6 of its range. """
    from pysonic_amd import NeuronalBilayerSonophore, getPointNeuron
    pn = getPointNeuron(name); nbls = NeuronalBilayerSonophore(32e-9, pn)
    P = np.ascontiguousarray(pn.device_params()); B = np.ascontiguousarray(nbls.device_params())
    y0 = np.ascontiguousarray(nbls.initialConditionsSonic())
    ev, tstop = O.pulsed_events(0.6e-3, 0.15e-3)
    ev_t, ev_x = np.array([e[0] for e in ev]), np.array([e[1] for e in ev])
    M = O.get_nsamples(0., tstop, 1e-8)
    out = {}
    for fn in ('harness_hybrid', 'harness_hybrid_row'):
        tr = np.zeros((M, y0.size + 5)); st, nst, ncy = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        scratch = np.zeros(harness.harness_hybrid_scratch_doubles())
        getattr(harness, fn)(pn.native_id, P.ctypes.data_as(dp), B.ctypes.data_as(dp), ctypes.c_double(500e3),
                             ctypes.c_double(A), ctypes.c_double(1.), ctypes.c_double(tstop), ev_t.ctypes.data_as(dp),
                             ev_x.ctypes.data_as(dp), len(ev), ctypes.c_longlong(M), y0.ctypes.data_as(dp),
                             ctypes.c_double(1e-8), 2000000000, tr.ctypes.data_as(dp), scratch.ctypes.data_as(dp),
                             ctypes.byref(st), ctypes.byref(nst), ctypes.byref(ncy))
        assert st.value == 0 and np.isfinite(tr).all(), fn
        out[fn] = (tr, nst.value, ncy.value)
    (a, na, ca), (b, nb, cb) = out['harness_hybrid'], out['harness_hybrid_row']
    np.testing.assert_array_equal(a[:, :2], b[:, :2])
    assert ca == cb and 4 <= cb < 0.9 * 0.75e-3 * 500e3           # dense AND sparse phases (STN: 241 of 375 periods)
    assert nb < 1.3 * na           # (about one step per output row on either pair: bound by stability, not by the tolerance)
    for j in range(2, a.shape[1]):
        ptp = max(np.ptp(a[:, j]), 1e-3 * np.abs(a[:, j]).max(), 1e-300)
        assert rms(a[:, j], b[:, j]) <= 2e-6 * ptp, (j, rms(a[:, j], b[:, j]) / ptp)


def test_mech_coop_core_against_lane_core_and_golden(harness):
    ''' mech_coop.hpp (one lookup cell per 8 lanes, emulated) on cells of golden_mech.npz spanning the amplitude
        range: the reference's cycle counts, every effective variable within 1e-6 (relative) of its converged
        run, as the one-cell-per-lane core; and finite at 5 MPa, far above the lookup grid (Vm down to -3 V) '''
    from pysonic_amd import NeuronalBilayerSonophore, getPointNeuron
    g = load_golden('golden_mech.npz')
    nbls = NeuronalBilayerSonophore(32e-9, getPointNeuron('RS'))
    P = np.ascontiguousarray(nbls.device_params())
    fs = np.array([1.0])

    def run(kind, P, f, A, Q):
        eff, st = np.zeros(9), ctypes.c_int()
        args = (0, P.ctypes.data_as(dp), ctypes.c_double(f), ctypes.c_double(A), ctypes.c_double(np.pi),
                ctypes.c_double(Q), fs.ctypes.data_as(dp), 1, ctypes.c_double(1e-9), 100000000)
        if kind == 'lane':
            zs, ngs = np.zeros(999), np.zeros(999)
            nc = harness.harness_mech(*args, zs.ctypes.data_as(dp), ngs.ctypes.data_as(dp), eff.ctypes.data_as(dp),
                                      ctypes.byref(st))
        else:
            sc = np.zeros(4 * 999)
            nc = harness.harness_mech_coop(*args, sc.ctypes.data_as(dp), eff.ctypes.data_as(dp), ctypes.byref(st))
        return nc, st.value, eff
    relerr = lambda x, r: np.max(np.abs(x - r) / np.maximum(np.abs(r), 1e-300))
    for i in (1, 5, 9, 13, 17, 19, 21):
        A, Q = g['pairs'][i]
        ncr = (int(g[f'p{i}_tight_nrows']) - 2) // 999
        (ncl, stl, el), (ncc, stc, ec) = run('lane', P, float(g['f']), A, Q), run('coop', P, float(g['f']), A, Q)
        assert ncl == ncc == ncr and stl == stc, (i, ncl, ncc, ncr, stl, stc)
        tight = g[f'p{i}_tight_eff']
        assert relerr(ec, tight) <= 1e-6 and relerr(el, tight) <= 1e-6, (i, relerr(ec, tight), relerr(el, tight))
    P64 = np.ascontiguousarray(NeuronalBilayerSonophore(64e-9, getPointNeuron('RS')).device_params())
    (ncl, stl, el), (ncc, stc, ec) = run('lane', P64, 500e3, 5e6, -107e-5), run('coop', P64, 500e3, 5e6, -107e-5)
    # (alpha_h overflows there in the reference's formula too: the same entries are infinite in both cores)
    ok = np.isfinite(el)
    assert ncl == ncc and np.array_equal(np.isfinite(ec), ok) and not np.any(np.isnan(ec))
    assert relerr(ec[ok], el[ok]) < 1e-4


@pytest.mark.parametrize('name', ['LTS', 'RE', 'TC', 'STN'])
def test_row_rhs_against_oracle_rhs(harness, name):
    ''' one evaluation of the row-cooperative right-hand side of the detailed model (full_row.hpp: every state one
        lane of a row of 16, the two rate constants of each gate from one generic per-lane form, the currents and
        Ca2+ machinery of the group kernel) against the oracle's restatement of
        NeuronalBilayerSonophore.fullDerivatives (nbls.py:265-278), on states drawn over the range a 600 kPa run
        visits, and at Z = 0 exactly '''
    from pysonic_amd import NeuronalBilayerSonophore, getPointNeuron
    from test_oracle_golden import _bls
    pn = getPointNeuron(name); nbls = NeuronalBilayerSonophore(32e-9, pn)
    P = np.ascontiguousarray(pn.device_params()); B = np.ascontiguousarray(nbls.device_params())
    p = _bls(name)
    L = O.lib(); nid = O.NEURON_IDS[name]
    ns = len(pn.statesNames())
    rng = np.random.default_rng(11 + len(name))
    f, phi = 500e3, np.pi
    y0 = np.asarray(nbls.initialConditionsSonic())[1:]
    for k in range(300):
        Z = 0. if k == 0 else float(rng.choice([rng.uniform(-0.6e-9, 0.5e-9), rng.uniform(0.5e-9, 12e-9)]))
        states = rng.uniform(0, 1, ns)
        for i, s in enumerate(pn.statesNames()):
            if s in ('Cai',):                       # concentrations around their resting value, not in [0, 1]
                states[i] = y0[i] * rng.uniform(0.5, 20.)
        y = np.concatenate([[rng.uniform(-0.3, 0.3), Z, p.ng0 * rng.uniform(0.5, 1.5), rng.uniform(-80e-5, 40e-5)], states])
        A, t, fs = float(rng.choice([0., 50e3, 600e3])), float(rng.uniform(0, 2e-6)), float(rng.choice([1., 0.75]))
        pac = A * np.sin(2 * np.pi * f * t - phi)
        ref = np.empty(4 + ns); cl = ctypes.c_int(0)
        L.orc_full_rhs(nid, ctypes.byref(p), ctypes.c_double(t), y.ctypes.data, ctypes.c_double(f), ctypes.c_double(A),
                       ctypes.c_double(phi), ctypes.c_double(fs), ref.ctypes.data, ctypes.byref(cl))
        out = np.full(4 + ns, np.nan)
        c = harness.harness_row_rhs(pn.native_id, P.ctypes.data_as(dp), B.ctypes.data_as(dp), ctypes.c_double(fs),
                                    ctypes.c_double(pac), y.ctypes.data_as(dp), out.ctypes.data_as(dp))
        assert c == cl.value, (k, c, cl.value)
        scale = np.abs(ref)
        scale[0] = max(scale[0], 1e-3 * abs(ref[0]) + 1e3)         # dU / dt: a sum of pressure terms that cancel
        scale[3] = max(scale[3], 1e-3)                             # dQ / dt: a sum of currents that cancel (A/m2)
        err = np.abs(out - ref) / np.maximum(scale, 1e-300)
        assert np.all(np.isfinite(out)), (k, y, out)
        assert np.all(err[1:] < 1e-9), (k, y, err, out, ref)
        assert err[0] < 1e-9, (k, y, err)


@pytest.mark.parametrize('name', ['LTS', 'TC', 'STN'])
def test_row_core_against_reference_golden(harness, name):
    ''' full_row.hpp emulated on the CPU (one configuration on a row of 16 lanes, 8(5,3) pair with the per-state
        guard, rtol 1e-8) on the reference's own run of the detailed model (120 kPa, 4 us + 1 us,
        tests/golden/golden_<neuron>.npz): the bars of tests/test_gpu_full.py::test_full_golden_other_neurons, in
        about a third of the lane core's steps; and a stiff configuration (STN at 450 kPa) is given up at once '''
    from pysonic_amd import NeuronalBilayerSonophore, getPointNeuron
    g = np.load(os.path.join(ROOT, 'tests', 'golden', f'golden_{name}.npz'), allow_pickle=True)
    cols = [str(c) for c in g['full_columns']]
    ref, tight = g['full_default'], g['full_tight']
    pn = getPointNeuron(name); nbls = NeuronalBilayerSonophore(32e-9, pn)
    ev, tstop = O.pulsed_events(4e-6, 1e-6)
    t0s, t1s, xs, ns, _ = _schedule(ev, tstop, 1 / (1000 * 500e3))
    M = O.get_nsamples(0., tstop, 1e-8)
    P = np.ascontiguousarray(pn.device_params()); B = np.ascontiguousarray(nbls.device_params())
    y0 = np.ascontiguousarray(nbls.initialConditionsSonic())
    ip = ctypes.POINTER(ctypes.c_int)

    def run(fn, A, rtol):
        tr = np.zeros((M, len(cols))); st = ctypes.c_int(); nst = ctypes.c_int()
        getattr(harness, fn)(pn.native_id, P.ctypes.data_as(dp), B.ctypes.data_as(dp), ctypes.c_double(500e3), ctypes.c_double(A),
                             ctypes.c_double(1.), ctypes.c_double(tstop), t0s.ctypes.data_as(dp), t1s.ctypes.data_as(dp),
                             xs.ctypes.data_as(dp), ns.ctypes.data_as(ip), len(ns), ctypes.c_longlong(M), y0.ctypes.data_as(dp),
                             ctypes.c_double(rtol), 50000000, tr.ctypes.data_as(dp), ctypes.byref(st), ctypes.byref(nst))
        return tr, st.value, nst.value
    tr, st, nrow = run('harness_full_row', 120e3, 1e-8)
    _, st_lane, nlane = run('harness_full', 120e3, 1e-8)
    assert st == 0 and st_lane == 0 and not np.isnan(tr).any()
    assert np.array_equal(tr[:, 0], ref[:, 0]) and np.array_equal(tr[:, 1], ref[:, 1])
    rms = lambda a, b: float(np.sqrt(np.mean((a - b)**2)))      # noqa: E731
    for i, k in enumerate(cols):
        if i < 2:
            continue
        spread, ptp = rms(ref[:, i], tight[:, i]), np.ptp(tight[:, i])
        assert rms(tr[:, i], tight[:, i]) <= max(3 * spread, 1e-6 * ptp, 1e-13 * np.abs(tight[:, i]).max()), k
    assert 2.5 * nrow < nlane, (nrow, nlane)
    if name == 'STN':
        _, st, nst = run('harness_full_row', 450e3, 1e-8)
        assert st & 64 and nst < 2000, (st, nst)


@pytest.mark.parametrize('name', ['LTS', 'TC', 'STN', 'SUseg', 'FHnode'])
def test_row_rosenbrock_linear_algebra_against_finite_differences(harness, name):
    """ W = I / (h gamma) - df/dy of the row Rosenbrock path (full_row.hpp: analytic Jacobian of row_rhs_jac, gates
        eliminated lane-wise, Schur complement of the extended core factorised across the lanes) against finite
        differences of the right-hand side: r = c0 k - (f(y + eps k) - f(y)) / eps, then W x = r must give k back --
        for every unit direction, from c0 = 1e9 (h = 4 ns) to 3e15 (h = 1.4 fs: the gate diagonal 1 / (c0 + r) must not
        be rounded against 1 -- the failure that stalled SUseg at 400 kPa). """
    from pysonic_amd import NeuronalBilayerSonophore, getPointNeuron
    pn = getPointNeuron(name); nbls = NeuronalBilayerSonophore(32e-9, pn)
    P = np.ascontiguousarray(pn.device_params()); B = np.ascontiguousarray(nbls.device_params())
    names = pn.statesNames(); ns = len(names)
    rng = np.random.default_rng(5)
    y0 = np.asarray(nbls.initialConditionsSonic())[1:]
    for trial in range(4):
        states = rng.uniform(0.01, 0.95, ns)
        for i, s_ in enumerate(names):
            if s_ == 'Cai':
                states[i] = y0[i] * rng.uniform(0.5, 5.)
        y = np.concatenate([[rng.uniform(-0.1, 0.1), rng.uniform(0.5e-9, 4e-9), 3.7e-22 * rng.uniform(0.8, 1.2),
                             rng.uniform(-70e-5, -20e-5)], states])
        sc = np.array([1., 1e-9, 1e-22, 1e-3] + [max(abs(v), 1e-6) for v in states])
        for c0 in (1e9, 1e12, 2.8e15):
            for j in range(4 + ns):
                k = np.zeros(4 + ns); k[j] = sc[j]
                x = np.zeros(4 + ns)
                rc = harness.harness_row_jac(pn.native_id, P.ctypes.data_as(dp), B.ctypes.data_as(dp), ctypes.c_double(1.0),
                                             ctypes.c_double(-2e5), y.ctypes.data_as(dp), k.ctypes.data_as(dp),
                                             ctypes.c_double(c0), ctypes.c_double(1e-6), x.ctypes.data_as(dp))
                assert rc == 0
                # (finite differences with a step of 1e-6 of the component's scale: second-order terms of ~1e-5)
                assert np.all(np.abs(x - k) / sc <= 2e-4), (trial, c0, j, (x - k) / sc)


@pytest.mark.parametrize('name,A,tstim,mode', [('HHseg', 100e3, 10e-6, 0), ('SWnode', 100e3, 10e-6, 0), ('MRGnode', 100e3, 10e-6, 0),
                                               ('FHnode', 300e3, 10e-6, 0), ('SUseg', 120e3, 5e-6, -2)])
def test_row_core_data_driven_neurons_against_lane_core(harness, name, A, tstim, mode, monkeypatch):
    """ the data-driven neurons on the row layout (full_row.hpp: row_gate_rate, ids 7 .. 11 -- the rate
        functions of mech_core.hpp's NeuronRates as per-lane data; GroupModel<GatedModel<N>>'s currents incl. the
        Goldman-Hodgkin-Katz force of FHnode) against the lane core (full_core.hpp) on one configuration: the same rows
        within 1e-6 of each variable's range, in less than half the step attempts. SUseg (Borg-Graham rates of
        1e10 / s) through the stiffness switch (stiff mode 1: explicit pair and RODAS4 in turns). """
    from pysonic_amd import NeuronalBilayerSonophore, getPointNeuron
    pn = getPointNeuron(name); nbls = NeuronalBilayerSonophore(32e-9, pn)
    ev, tstop = O.pulsed_events(tstim, tstim / 4)
    t0s, t1s, xs, ns, _ = _schedule(ev, tstop, 1 / (1000 * 500e3))
    M = O.get_nsamples(0., tstop, 1e-8)
    P = np.ascontiguousarray(pn.device_params()); B = np.ascontiguousarray(nbls.device_params())
    y0 = np.ascontiguousarray(nbls.initialConditionsSonic())
    ip = ctypes.POINTER(ctypes.c_int)
    monkeypatch.setenv('ROW_RTOL_STIFF', '3e-7')
    out = {}
    for fn, ms in (('harness_full', 50000000), ('harness_full_row', mode if mode else 50000000)):
        tr = np.zeros((M, y0.size + 5)); st = ctypes.c_int(); nst = ctypes.c_int()
        getattr(harness, fn)(pn.native_id, P.ctypes.data_as(dp), B.ctypes.data_as(dp), ctypes.c_double(500e3), ctypes.c_double(A),
                             ctypes.c_double(1.), ctypes.c_double(tstop), t0s.ctypes.data_as(dp), t1s.ctypes.data_as(dp),
                             xs.ctypes.data_as(dp), ns.ctypes.data_as(ip), len(ns), ctypes.c_longlong(M), y0.ctypes.data_as(dp),
                             ctypes.c_double(1e-8), ms, tr.ctypes.data_as(dp), ctypes.byref(st), ctypes.byref(nst))
        assert st.value == 0 and np.isfinite(tr).all(), (fn, st.value)
        out[fn] = (tr, nst.value)
    (a, na), (b, nb) = out['harness_full'], out['harness_full_row']
    np.testing.assert_array_equal(a[:, :2], b[:, :2])
    assert 2 * nb < na, (nb, na)
    for j in range(2, a.shape[1]):
        ptp = max(np.ptp(a[:, j]), 1e-3 * np.abs(a[:, j]).max(), 1e-300)
        assert rms(a[:, j], b[:, j]) <= 1e-6 * ptp, (j, rms(a[:, j], b[:, j]) / ptp)


@pytest.mark.parametrize('name,gfile', [('STN', 'golden_full_stiff.npz'), ('TC', 'golden_full_stiff2.npz')])
def test_row_stiff_path_against_reference_golden(harness, name, gfile, monkeypatch):
    ''' the Rosenbrock path of the row-cooperative detailed model (full_row.hpp: RODAS4 on the whole system, gates
        eliminated lane-wise, analytic Jacobian incl. d(rates)/dVm of the generic rate form) on the reference's runs of
        the configurations that turn stiff -- STN at 500 kPa, TC at 600 kPa, 4 us + 1 us
        (tests/golden/make_golden_full_pw.py stiff / stiff2): the explicit pair gives them up within a microsecond
        (DOP853's stiffness bookkeeping on the live gates' rates) and RODAS4 at 3e-7 -- in turns with the explicit pair
        wherever no gate is fast (row_switching_segment) -- finishes them within the golden
        bars of the reference's converged run, in less than half the steps the lane core takes at 1e-8 '''
    from pysonic_amd import NeuronalBilayerSonophore, getPointNeuron
    g = np.load(os.path.join(ROOT, 'tests', 'golden', gfile), allow_pickle=True)
    f, A, tstim, toffset, _, _ = [float(x) for x in g[f'{name}_cfg']]
    cols = [str(c) for c in g[f'{name}_columns']]
    ref, tight = g[f'{name}_default'], g[f'{name}_tight']
    pn = getPointNeuron(name); nbls = NeuronalBilayerSonophore(32e-9, pn)
    ev, tstop = O.pulsed_events(tstim, toffset)
    t0s, t1s, xs, ns, _ = _schedule(ev, tstop, 1 / (1000 * f))
    M = O.get_nsamples(0., tstop, 1e-8)
    P = np.ascontiguousarray(pn.device_params()); B = np.ascontiguousarray(nbls.device_params())
    y0 = np.ascontiguousarray(nbls.initialConditionsSonic())
    ip = ctypes.POINTER(ctypes.c_int)
    monkeypatch.setenv('ROW_RTOL_STIFF', '3e-7')

    def run(fn, mode=-2):
        tr = np.zeros((M, len(cols))); st = ctypes.c_int(); nst = ctypes.c_int()
        getattr(harness, fn)(pn.native_id, P.ctypes.data_as(dp), B.ctypes.data_as(dp), ctypes.c_double(f), ctypes.c_double(A),
                             ctypes.c_double(1.), ctypes.c_double(tstop), t0s.ctypes.data_as(dp), t1s.ctypes.data_as(dp),
                             xs.ctypes.data_as(dp), ns.ctypes.data_as(ip), len(ns), ctypes.c_longlong(M), y0.ctypes.data_as(dp),
                             ctypes.c_double(1e-8), mode, tr.ctypes.data_as(dp), ctypes.byref(st), ctypes.byref(nst))   # -2: stiff mode 1
        return tr, st.value, nst.value
    tr, st, nrow = run('harness_full_row')
    _, st_lane, nlane = run('harness_full')
    assert st == 0 and st_lane == 0 and not np.isnan(tr).any(), (st, st_lane)
    # the way back: outside the stiff part of the acoustic period the configuration returns to the explicit pair, and
    # takes fewer step attempts than on RODAS4 throughout (stiff mode 2)
    _, st2, nrodas = run('harness_full_row', -3)
    assert st2 == 0 and nrow < 0.7 * nrodas, (nrow, nrodas)
    rms = lambda a, b: float(np.sqrt(np.mean((a - b)**2)))      # noqa: E731
    for i, k in enumerate(cols):
        if i < 2:
            continue
        spread, ptp = rms(ref[:, i], tight[:, i]), np.ptp(tight[:, i])
        bar = max(3 * spread, 1e-6 * ptp, 1e-12 * np.abs(tight[:, i]).max())
        assert rms(tr[:, i], tight[:, i]) <= 0.3 * bar, (k, rms(tr[:, i], tight[:, i]) / bar)
    assert 2 * nrow < nlane, (nrow, nlane)
