// pysonic_amd/csrc/hybrid_row.hpp
//
// ROW-COOPERATIVE version of the hybrid integration (method='hybrid') for LTS / IB / RE / TC / STN: hybrid_coop.hpp
// on the layout, right-hand side and integrators of full_row.hpp -- one configuration per row of 16 lanes, every
// state one lane.
//
// Reference: NeuronalBilayerSonophore.__simHybrid + HybridSolver.solve (PySONIC/core/nbls.py:356-387,
// solvers.py:483-633); see hybrid_core.hpp for the scheme (dense periods until the deflection cycle closes, `bound`,
// sparse phase replaying the last cycle at 40 points per period with the capacitance frozen per sparse step, events,
// on-the-fly resampling to 10 ns). As in hybrid_coop.hpp:
//   * the ring of the last two dense periods holds U / Z / ng / t in four arrays written by the lanes of U, Z, ng,
//     Qm (12 .. 15), the 40-point cycle of the sparse phase likewise;
//   * the dense periods are integrated by row_integrate_segment (8(5,3) pair with the per-state guard);
//   * the membrane equations of the sparse phase are integrated by RODAS4 with the exact Jacobian
//     (row_membrane_rodas4) instead of the reference's explicit dop853.
// STIFF = false: a dense period that turns stiff (FULL_ST_STIFF, row_integrate_segment) ends the configuration with
// that status, and the host restarts it on the STIFF = true build of the kernel, whose dense periods alternate
// between the explicit pair and RODAS4 over the whole system (row_switching_segment; opts.stiff_mode 2: RODAS4
// throughout) -- two kernels for the reason full_row_config has its MODE: both integrators in one kernel spill a few
// hundred scalar registers, and the configurations that never turn stiff need not pay for them.
#pragma once
#include "full_row.hpp"
#include "hybrid_core.hpp"

namespace sonic {

template <class O, class M, bool STIFF>
SONIC_HD void hybrid_row_config(const HybridDev &D, const BLSParams &p, const typename M::Params &P,
                                const LaneSpec *glanes, const RowLaneSpec *rlanes, long long c, bool store)
{
    typedef typename O::V V;
    typedef GroupModel<M> GM;
    typedef RowModel<M> RM;
    static_assert(RM::LU == 12 && RM::LZ == 13 && RM::LNG == 14 && RM::LQ == 15, "store_mech4 / load_mech4: lanes 12 .. 15");
    constexpr int NCOL = M::NY + 5;
    const double f = D.f[c], fs = D.fs[c], tstop = D.tstop[c];
    const double w = 2.0 * bls::PI * f;
    const double T = 1.0 / f;
    const double dt = 1.0 / (MECH_NPC * f);
    const double dt_sparse = 1.0 / (HYB_NPC_SPARSE * f);
    const int max_steps = full_step_budget(D.opts, f, tstop);
    int status = 0, nsteps = 0, ncycles_total = 0;
    bool clamped = false;

    GroupConsts<O> C;
    O::load_consts(glanes, C);
    RowConsts<O> R;
    O::load_row_consts(rlanes, R);
    const V mech3 = O::add(O::add(R.r[RR_MU], R.r[RR_MZ]), R.r[RR_MNG]);       // 1 on the lanes of U, Z, ng
    const V nmech3 = O::sub(O::splat(1.0), mech3);

    double *ring = D.scratch + c * (long long)HYB_SCRATCH_DOUBLES;       // [4][HYB_RING]: U, Z, ng, t
    double *cyc = ring + 4 * HYB_RING;                                  // [4][HYB_NSPARSE_MAX]
    const double *ring_t = ring + 3 * HYB_RING;
    long long nring = 0;                     // dense rows pushed so far (ring index = nring % HYB_RING)
    long long nreg = 0;                      // trailing dense rows that are dt-regular (getCycle)

    // initial conditions: two rows at t = 0 (Z = 0, then the quasi-static deflection)
    const double Pac_dt = D.A[c] * sin(w * dt - D.phi);
    const double Zqs = bls_balancedefQS(p, p.ng0, D.y0[0], Pac_dt);
    if (!(Zqs == Zqs)) status |= 2;
    V y = O::init_gates(D.y0, C.colx);
    y = O::fma_(R.r[RR_MZ], O::splat(Zqs), y);
    y = O::fma_(R.r[RR_MNG], O::splat(p.ng0), y);
    y = O::fma_(R.r[RR_MQ], O::splat(D.y0[0]), y);
    if constexpr (GM::NC > 1) y = O::fma_(R.r[RR_MC1], O::splat(D.y0[GM::core_col(1) - 2]), y);
    if constexpr (GM::NC > 2) y = O::fma_(R.r[RR_MC2], O::splat(D.y0[GM::core_col(2) - 2]), y);
    if constexpr (GM::NC > 3) y = O::fma_(R.r[RR_MC3], O::splat(D.y0[GM::core_col(3) - 2]), y);
    if constexpr (GM::NC > 4) y = O::fma_(R.r[RR_MC4], O::splat(D.y0[GM::core_col(4) - 2]), y);

    const long long M_rows = D.row_off[c + 1] - D.row_off[c];
    double *rows = D.traces + D.row_off[c] * NCOL;
    const Linspace out = linspace_make(0.0, tstop, (int)M_rows);
    long long j = 0;
    double tau = linspace_at(out, 0);
    double tp = 0.0, xp = 0.0;
    V yp = y;

    // one row (ti, yi, stimulus state xs) of the solution: emit every output row <= ti (np.interp for the
    // variables; interp1d 'nearest' for the state: the left row up to and including the midpoint)
    auto consume = [&](double ti, V yi, double xs) SONIC_COOP_INLINE {
        while (j < M_rows && tau <= ti) {
            V r = yi;
            if (ti > tp) {
                const V wgt = O::splat((tau - tp) / (ti - tp));
                r = O::fma_(O::sub(yi, yp), wgt, yp);
            }
            const double Zr = O::template bcast<RM::LZ>(r), Qr = O::template bcast<RM::LQ>(r);
            const double Vm = Qr / (fs * bls_capacitance(p, Zr) + (1.0 - fs) * p.Cm0) * 1e3;
            if (store) O::store_full_row(rows + j * NCOL, R, NCOL, tau, (tau <= (tp + ti) / 2.0) ? xp : xs, r, Vm);
            j++;
            if (j < M_rows) tau = linspace_at(out, (int)j);
        }
        tp = ti;
        xp = xs;
        yp = yi;
    };
    // of the two initial rows at t = 0 only the second is seen by np.interp
    consume(0.0, y, 0.0);

    const long long e0 = D.ev_off[c];
    const int nev = (int)(D.ev_off[c + 1] - e0);
    int iev = 0;
    double t = 0.0, xref = 0.0, As = 0.0;    // event_params: drive amplitude 0 before the first event
    double h = 0.25 * dt;
    V K[16];
    int iasti = 0, nonsti = 0;
    bool stiff = STIFF && D.opts.stiff_mode == 2;     // (STIFF build: which integrator the dense periods are on)
    bool failed = false;
    auto event_t = [&](int i) { return i < nev ? D.ev_t[e0 + i] : tstop; };

    while (iev <= nev && !failed) {
        const double tevent = event_t(iev);
        const double tend = fmin(tevent, t + HYB_UPDATE_INTERVAL);
        const int nmax = (int)nearbyint((tend - t) / T);

        // ---------------- 1. dense periods ----------------
        bool bounded = false;
        if (nmax > 0) {
            if (nmax < 2) { status |= 16; failed = true; break; }   // the reference asserts nmin <= nmax
            int icount = 0;                    // the reference's loop counter `i`
            int ndone = 0;                     // periods integrated in this call
            while (true) {
                // one period: rows on np.linspace(t, t + T, 1000)[1:]
                V sse = O::splat(0.0), vmax = O::splat(-INFINITY), nvmin = O::splat(-INFINITY);
                auto dense = [&](double td, V yd) SONIC_COOP_INLINE {
                    // rows beyond tend are dropped by `bound` (solvers.py:129-139)
                    if (td <= tend) {
                        const long slot = (long)(nring % HYB_RING);
                        // periodic stability: this row against the row one period earlier
                        if (nring >= HYB_NPC) {
                            const V d = O::sub(yd, O::load_mech4(ring, HYB_RING, (long)((nring - HYB_NPC) % HYB_RING)));
                            sse = O::fma_(d, d, sse);
                        }
                        vmax = O::max_(vmax, yd);
                        nvmin = O::max_(nvmin, O::neg(yd));
                        // U, Z, ng from their lanes, t on the lane of Qm
                        O::store_mech4(ring, HYB_RING, slot, O::fma_(R.r[RR_MQ], O::splat(td), O::mul(mech3, yd)));
                        nring++;
                        nreg++;
                        consume(td, yd, xref);
                    } else {
                        bounded = true;
                    }
                };
                double t_stop = t;
                int i_stop = 1;
                int bad;
                if constexpr (STIFF)
                    bad = row_switching_segment<O, M>(p, P, C, R, fs, D.opts.qdrive, w, D.phi, D.opts.rtol, D.opts.rtol_stiff,
                                                      As, t, t + T, MECH_NPC, dt, y, K, h, nsteps, max_steps, clamped, iasti,
                                                      nonsti, stiff, D.opts.stiff_mode == 2 ? 2 : 1, dense);
                else
                    bad = row_integrate_segment<O, M>(p, P, C, R, fs, D.opts.qdrive, w, D.phi, D.opts.rtol, As, t, t + T,
                                                      MECH_NPC, dt, y, K, h, nsteps, max_steps, clamped, iasti, nonsti,
                                                      t_stop, i_stop, dense);
                if (bad) { status |= bad; failed = true; break; }
                t = t + T;
                ndone++;
                ncycles_total++;
                if (bounded) break;        // everything from here on would be dropped by `bound`
                if (ndone < 2) continue;                           // nmin = 2 periods first
                if (ndone == 2) icount = 1;
                // isPeriodicallyStable on the two periods just produced (solvers.py:317-330), Z and ng
                const V ptp = O::add(vmax, nvmin);
                const double rz = sqrt(O::template bcast<RM::LZ>(sse) / HYB_NPC) / O::template bcast<RM::LZ>(ptp);
                const double rn = sqrt(O::template bcast<RM::LNG>(sse) / HYB_NPC) / O::template bcast<RM::LNG>(ptp);
                const bool stable = rz < 1e-4 && rn < 1e-4;
                if (stable || !(icount < nmax)) break;
                icount++;
            }
            if (failed) break;
        }
        // the state after `bound`: the last row kept -- U, Z, ng from the ring, (Qm, states) = the last consumed row
        if (bounded) {
            const long lastslot = (long)((nring - 1) % HYB_RING);
            t = ring_t[lastslot];
            y = O::fma_(mech3, O::load_mech4(ring, HYB_RING, lastslot), O::mul(nmech3, yp));
        }

        // ---------------- 3. sparse phase ----------------
        if (t < tend) {
            // last period = the last 999 dt-regular rows (getCycle(-1), solvers.py:283-315)
            if (nreg < HYB_NPC || nring < HYB_NPC) { status |= 32; failed = true; break; }
            const long long first = nring - HYB_NPC;
            auto rt = [&](int k) { return ring_t[(int)((first + k) % HYB_RING)]; };
            const double tl0 = rt(0), tl1 = rt(HYB_NPC - 1);
            const long long ns_ll = (long long)nearbyint((tl1 - tl0) / dt_sparse);
            const int npc = (int)(ns_ll > 2 ? ns_ll : 2);
            if (npc > HYB_NSPARSE_MAX) { status |= 32; failed = true; break; }
            const Linspace sg = linspace_make(tl0, tl1, npc);
            int lo = 0;
            for (int k = 0; k < npc; k++) {                          // np.interp of U, Z, ng
                const double xq = linspace_at(sg, k);
                while (lo < HYB_NPC - 2 && rt(lo + 1) <= xq) lo++;
                const long a = (long)((first + lo) % HYB_RING), b = (long)((first + lo + 1) % HYB_RING);
                V v;
                if (xq >= tl1) {
                    v = O::load_mech4(ring, HYB_RING, (long)((first + HYB_NPC - 1) % HYB_RING));
                } else {
                    const V va = O::load_mech4(ring, HYB_RING, a), vb = O::load_mech4(ring, HYB_RING, b);
                    const double dx = xq - ring_t[a], den = ring_t[b] - ring_t[a];
                    v = O::fma_(O::mul(O::sub(vb, va), O::splat(1.0 / den)), O::splat(dx), va);
                }
                O::store_mech4(cyc, HYB_NSPARSE_MAX, k, v);
            }
            const int n = (int)ceil((tend - t) / dt_sparse);
            const Linspace ts = linspace_make(t, tend, n + 1);
            double tsol = t;
            double hs = dt_sparse;
            for (int i = 0; i < n && !failed; i++) {
                const double tt = linspace_at(ts, i + 1);
                const V prof = O::load_mech4(cyc, HYB_NSPARSE_MAX, i % npc);
                if (tt - tsol > HYB_MIN_SPARSE_DT) {
                    const double Zi = O::template bcast<RM::LZ>(prof);
                    const double Cm = fs * bls_capacitance(p, Zi) + (1.0 - fs) * p.Cm0;
                    const double kV = 1e3 / Cm;
                    // membrane equations at the frozen capacitance (solvers.py:590-633)
                    if (!row_membrane_rodas4<O, M>(P, C, R, D.opts.qdrive, kV, D.opts.rtol, tt - tsol, y, hs, nsteps, max_steps)) {
                        status |= 4;
                        failed = true;
                    }
                    tsol = tt;
                }
                // the row: U, Z, ng of the replayed cycle, (Qm, states) as integrated
                y = O::fma_(mech3, prof, O::mul(nmech3, y));
                consume(tt, y, xref);
            }
            t = tend;
            nreg = 0;                 // sparse rows break the dt-regular run
        }

        // ---------------- 4. event ----------------
        if (t == tevent) {
            if (iev < nev) {
                xref = D.ev_x[e0 + iev];
                As = D.A[c] * xref;                    // eventfunc: drive.xvar * x (nbls.py:367)
            }
            iev++;
        }
    }

    for (; j < M_rows; j++)             // rows not produced (failed configuration): NaN
        if (store) O::fill_full_row_nan(rows + j * NCOL, R, NCOL, linspace_at(out, (int)j));
    if (clamped) status |= 1;
    if (store && O::leader()) {
        D.status[c] = status;
        D.nsteps[c] = nsteps;
        D.ncycles[c] = ncycles_total;
    }
}

}  // namespace sonic
