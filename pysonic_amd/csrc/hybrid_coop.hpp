// pysonic_amd/csrc/hybrid_coop.hpp
//
// OCTET-COOPERATIVE version of the hybrid integration (method='hybrid') for the cortical RS / FS neurons:
// hybrid_core.hpp with the layout, right-hand side and Dormand-Prince 8(5,3) stepper of full_coop.hpp -- one
// configuration per 8 lanes, lane l owns component l of (U, Z, ng, Qm, m, h, n, p).
//
// Reference: NeuronalBilayerSonophore.__simHybrid + HybridSolver.solve (PySONIC/core/nbls.py:356-387,
// solvers.py:483-633); see hybrid_core.hpp for the scheme (dense periods until the deflection cycle closes,
// `bound`, sparse phase replaying the last cycle at 40 points per period with the capacitance frozen per
// sparse step, events, on-the-fly resampling to 10 ns). Differences from hybrid_core.hpp are of layout only:
//   * the ring of the last two dense periods holds, per row, U / Z / ng / t in four arrays written by lanes
//     0..3 (OctOps::store4), the 40-point cycle of the sparse phase likewise;
//   * the dense periods are integrated by coop_integrate_segment (8(5,3) pair, steps of at most two dense
//     intervals);
// and one of method: the membrane equations of the sparse phase are stiff at high pressure amplitudes and are
// integrated by RODAS4 instead of an explicit pair (coop_membrane_rodas4 below).
#pragma once
#include "full_coop.hpp"
#include "hybrid_core.hpp"

namespace sonic {

// scratch per configuration: 4 x ring (U, Z, ng, t) + 4 x sparse cycle (the 4th row unused)
constexpr int HYB_COOP_SCRATCH_DOUBLES = 4 * HYB_RING + 4 * HYB_NSPARSE_MAX;
static_assert(HYB_COOP_SCRATCH_DOUBLES <= HYB_SCRATCH_DOUBLES, "scratch of the hybrid kernels");

// ---- membrane equations at a frozen capacitance: RODAS4 on the octet -------------------------------------
// Sparse phase (solvers.py:590-633): y = (Qm, m, h, n, p) with Vm = Qm / Cm, Cm that of the replayed
// deflection. The reference integrates it with scipy's explicit dop853; at high pressure amplitudes the
// replayed deflection takes Vm to -460 mV within every acoustic period (600 kPa), where alpha_h = 0.128e3
// exp(-(v - 17) / 18) is 2e12 / s: an explicit pair then needs ~2000 steps per sparse step of 50 ns (measured:
// 25.7e6 step attempts for 1.25 ms, against 2.7e6 for the whole detailed model). The system has the arrow
// structure of the effective one (gates depend on themselves and on Qm only), so it is integrated like the
// SONIC kernels: RODAS4 (L-stable, stiffly accurate) with the exact Jacobian, the charge replicated, the
// gates on lanes 4..7, one all-reduce per stage -- one step per sparse step.

// rate constants of the eight lanes (0..3: beta_m beta_h beta_n beta_p, 4..7: alpha_*) and their derivatives
// with respect to Vm; see coop_consts for the generic form K num(u, e) / den(e), e = exp(u), u = (Vm - vc) vs
template <class O>
SONIC_HD void coop_rates(const CoopConsts<O> &C, typename O::V Vm, typename O::V &rate, typename O::V &drate)
{
    typedef typename O::V V;
    const V u = O::mul(O::sub(Vm, C.vc), C.vs);
    const V e = O::exp_(u);
    const V e2 = O::mul(e, e);
    V num = O::fma_(C.a1, u, C.a0);
    num = O::fma_(C.a3, e2, num);
    num = O::fma_(e, O::fma_(C.a4, e2, C.a2), num);
    V den = O::fma_(C.b1, e, C.b0);
    den = O::fma_(O::fma_(C.b3, e, C.b2), e2, den);
    // d/du: de/du = e
    V dnum = O::fma_(O::mul(O::splat(2.0), C.a3), e2, C.a1);
    dnum = O::fma_(e, O::fma_(O::mul(O::splat(3.0), C.a4), e2, C.a2), dnum);
    V dden = O::mul(C.b1, e);
    dden = O::fma_(O::fma_(O::mul(O::splat(3.0), C.b3), e, O::mul(O::splat(2.0), C.b2)), e2, dden);
    const V inv = O::div(O::splat(1.0), den);
    rate = O::mul(C.K, O::mul(num, inv));
    drate = O::mul(O::mul(C.K, C.vs), O::mul(O::sub(O::mul(dnum, den), O::mul(num, dden)), O::mul(inv, inv)));
}

template <class O>
struct MembraneEval {
    typename O::V fg, r, gpw, other, drive, a_, b_;
    double fq;
};

// right-hand side of (q, gates x on lanes 4..7) at Vm = q kV
template <class O, bool JAC>
SONIC_HD void coop_membrane_eval(const CoopConsts<O> &C, const CoopScalars<O> &S, double kV, double q,
                                 typename O::V x, MembraneEval<O> &R, typename O::V *drate_out)
{
    typedef typename O::V V;
    const V Vm = O::splat(q * kV);
    V rate, drate;
    if (JAC) {
        coop_rates<O>(C, Vm, rate, drate);
        *drate_out = drate;
    } else {
        rate = coop_rate<O>(C, Vm);
    }
    const V beta = O::shr4(rate);
    R.a_ = rate;
    R.b_ = beta;
    R.r = O::add(rate, beta);
    R.fg = O::template on_lanes<0x0F>(O::splat(0.0), O::sub(rate, O::mul(R.r, x)));
    const V xo = O::swap1(x);
    const V x2 = O::mul(x, x);
    const V pw = O::fma_(x2, O::fma_(C.c4, x2, O::mul(C.c3, x)), O::fma_(C.c1, x, C.c0));
    R.other = O::fma_(xo, C.c3, C.nc3);
    R.drive = O::sub(Vm, C.E);
    R.gpw = O::mul(C.G, pw);
    R.fq = O::first(O::allsum(O::mul(O::mul(R.gpw, R.other), R.drive))) + S.qdrive;
}

// Integrate (q, x) over an interval of length `span` with RODAS4 steps (error control as the dense phase:
// rtol relative to max(|y|, floor)). `hs` carries the step size. Returns false if the step budget runs out.
template <class O>
SONIC_HD bool coop_membrane_rodas4(const CoopConsts<O> &C, const CoopScalars<O> &S, double kV, double rtol,
                                   double span, double &q, typename O::V &x, double &hs, int &nsteps, int max_steps)
{
    using namespace rodas4;
    typedef typename O::V V;
    const V gmask = O::template on_lanes<0x0F>(O::splat(0.0), O::splat(1.0));   // 1 on the gate lanes
    double tcur = 0.0;
    hs = fmin(hs, span);
    while (tcur < span) {
        bool last = false;
        double h = hs;
        if (tcur + 1.0001 * h >= span) { h = span - tcur; last = true; }
        // f and the Jacobian at (q, x)
        MembraneEval<O> R0;
        V drate;
        coop_membrane_eval<O, true>(C, S, kV, q, x, R0, &drate);
        const V cond = O::mul(R0.gpw, R0.other);
        const double Jqq = kV * O::first(O::allsum(cond));
        const V x2 = O::mul(x, x);
        const V dpw = O::fma_(x2, O::fma_(O::mul(O::splat(4.0), C.c4), x, O::mul(O::splat(3.0), C.c3)), C.c1);
        const V own = O::mul(O::mul(O::mul(C.G, dpw), R0.other), R0.drive);
        const V jq = O::fma_(O::swap1(O::mul(R0.gpw, R0.drive)), C.c0, own);     // the h lane: d iNa / dh
        const V db = O::shr4(drate);
        const V Jgq = O::mul(O::mul(gmask, O::splat(kV)), O::sub(drate, O::mul(O::add(drate, db), x)));
        const double inv_h = 1.0 / h;
        const double c0 = inv_h * (1.0 / gamma);
        const V invd = O::div(gmask, O::add(O::splat(c0), R0.r));               // 0 off the gate lanes
        const V wq = O::mul(jq, invd);
        const double piv = 1.0 / (c0 - Jqq - O::first(O::allsum(O::mul(wq, Jgq))));
        double kq[6];
        V kg[6];
        auto solve = [&](int i, double fq_, const V &tsum, const V &rg) SONIC_COOP_INLINE {
            const double b = (fq_ + O::first(O::allsum(tsum))) * piv;
            kq[i] = b;
            kg[i] = O::mul(O::fma_(Jgq, O::splat(b), rg), invd);
        };
        double qt;
        V xt;
        auto stage = [&](int i, double cq, const V &cg) SONIC_COOP_INLINE {
            MembraneEval<O> R;
            coop_membrane_eval<O, false>(C, S, kV, qt, xt, R, nullptr);
            const V rg = O::add(R.fg, cg);
            // the current sum is already in R.fq: only the eliminated gates go through the butterfly
            solve(i, R.fq + cq, O::mul(wq, rg), rg);
        };
        solve(0, R0.fq, O::mul(wq, R0.fg), R0.fg);
        {
            const double g1 = c21 * inv_h;
            qt = q + a21 * kq[0];
            xt = O::fma_(O::splat(a21), kg[0], x);
            stage(1, g1 * kq[0], O::mul(O::splat(g1), kg[0]));
        }
        {
            const double g1 = c31 * inv_h, g2 = c32 * inv_h;
            qt = q + a31 * kq[0] + a32 * kq[1];
            xt = O::fma_(O::splat(a32), kg[1], O::fma_(O::splat(a31), kg[0], x));
            stage(2, g1 * kq[0] + g2 * kq[1], O::fma_(O::splat(g2), kg[1], O::mul(O::splat(g1), kg[0])));
        }
        {
            const double g1 = c41 * inv_h, g2 = c42 * inv_h, g3 = c43 * inv_h;
            qt = q + a41 * kq[0] + a42 * kq[1] + a43 * kq[2];
            xt = O::fma_(O::splat(a43), kg[2], O::fma_(O::splat(a42), kg[1], O::fma_(O::splat(a41), kg[0], x)));
            stage(3, g1 * kq[0] + g2 * kq[1] + g3 * kq[2],
                  O::fma_(O::splat(g3), kg[2], O::fma_(O::splat(g2), kg[1], O::mul(O::splat(g1), kg[0]))));
        }
        {
            const double g1 = c51 * inv_h, g2 = c52 * inv_h, g3 = c53 * inv_h, g4 = c54 * inv_h;
            qt = q + a51 * kq[0] + a52 * kq[1] + a53 * kq[2] + a54 * kq[3];
            xt = O::fma_(O::splat(a54), kg[3], O::fma_(O::splat(a53), kg[2],
                 O::fma_(O::splat(a52), kg[1], O::fma_(O::splat(a51), kg[0], x))));
            stage(4, g1 * kq[0] + g2 * kq[1] + g3 * kq[2] + g4 * kq[3],
                  O::fma_(O::splat(g4), kg[3], O::fma_(O::splat(g3), kg[2],
                  O::fma_(O::splat(g2), kg[1], O::mul(O::splat(g1), kg[0])))));
        }
        {
            const double g1 = c61 * inv_h, g2 = c62 * inv_h, g3 = c63 * inv_h, g4 = c64 * inv_h, g5 = c65 * inv_h;
            qt += kq[4];
            xt = O::add(xt, kg[4]);
            stage(5, g1 * kq[0] + g2 * kq[1] + g3 * kq[2] + g4 * kq[3] + g5 * kq[4],
                  O::fma_(O::splat(g5), kg[4], O::fma_(O::splat(g4), kg[3], O::fma_(O::splat(g3), kg[2],
                  O::fma_(O::splat(g2), kg[1], O::mul(O::splat(g1), kg[0]))))));
        }
        nsteps++;
        const double qnew = qt + kq[5];
        const V xnew = O::add(xt, kg[5]);
        // embedded error estimate = k6; RMS over (Qm, m, h, n, p) of err / (rtol max(|y|, |ynew|, floor))
        const V scg = O::mul(O::splat(rtol), O::max_(O::max_(O::abs_(x), O::abs_(xnew)), O::splat(FULL_FLOOR_Y)));
        const V eg = O::mul(gmask, O::div(kg[5], scg));
        const double eq = kq[5] / (rtol * fmax(fmax(fabs(q), fabs(qnew)), FULL_FLOOR_Y));
        const double en = sqrt((eq * eq + O::first(O::allsum(O::mul(eg, eg)))) * 0.2);
        double fac = 0.9 * O::fast_pow(fmax(en, 1e-10), -0.25);
        fac = fmin(6.0, fmax(0.2, fac));
        if (!(en == en)) fac = 0.2;
        if (en <= 1.0) {
            q = qnew;
            x = xnew;
            tcur = last ? span : tcur + h;
            hs = last ? fmax(h * fac, hs) : h * fac;
        } else {
            hs = h * fmin(fac, 1.0);
        }
        if (nsteps >= max_steps || !(hs > 1e-18)) return false;
    }
    return true;
}

template <class O>
SONIC_HD void hybrid_coop_config(const HybridDev &D, const BLSParams &p, const CorticalParams &P, int neuron,
                                 long long c, bool store)
{
    typedef typename O::V V;
    constexpr int NCOL = 10;                     // t stim Z ng Qm m h n p Vm
    const double f = D.f[c], fs = D.fs[c], tstop = D.tstop[c];
    const double w = 2.0 * bls::PI * f;
    const double T = 1.0 / f;
    const double dt = 1.0 / (MECH_NPC * f);
    const double dt_sparse = 1.0 / (HYB_NPC_SPARSE * f);
    const int max_steps = full_step_budget(D.opts, f, tstop);
    int status = 0, nsteps = 0, ncycles_total = 0;
    bool clamped = false;

    const CoopConsts<O> C = coop_consts<O>(p, P, neuron, D.opts.qdrive);
    const CoopScalars<O> S = coop_scalars<O>(p, fs, D.opts.qdrive);

    double *ring = D.scratch + c * (long long)HYB_SCRATCH_DOUBLES;       // [4][HYB_RING]: U, Z, ng, t
    double *cyc = ring + 4 * HYB_RING;                                  // [4][HYB_NSPARSE_MAX]
    const double *ring_t = ring + 3 * HYB_RING;
    long long nring = 0;                     // dense rows pushed so far (ring index = nring % HYB_RING)
    long long nreg = 0;                      // trailing dense rows that are dt-regular (getCycle)

    // initial conditions: two rows at t = 0 (Z = 0, then the quasi-static deflection)
    const double Pac_dt = D.A[c] * sin(w * dt - D.phi);
    const double Zqs = bls_balancedefQS(p, p.ng0, D.y0[0], Pac_dt);
    if (!(Zqs == Zqs)) status |= 2;
    V y = O::roles(0.0, Zqs, p.ng0, D.y0[0], D.y0[1], D.y0[2], D.y0[3], D.y0[4]);

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
            const double Zr = O::first(O::template bcast<1>(r)), Qr = O::first(O::template bcast<3>(r));
            const double Vm = Qr / (fs * bls_capacitance(p, Zr) + (1.0 - fs) * p.Cm0) * 1e3;
            if (store) O::store_row(rows + j * NCOL, tau, (tau <= (tp + ti) / 2.0) ? xp : xs, r, Vm);
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
                            const V d = O::sub(yd, O::load4(ring, HYB_RING, (long)((nring - HYB_NPC) % HYB_RING)));
                            sse = O::fma_(d, d, sse);
                        }
                        vmax = O::max_(vmax, yd);
                        nvmin = O::max_(nvmin, O::neg(yd));
                        O::store4(ring, HYB_RING, slot, O::template on_lane<3>(O::splat(td), yd));
                        nring++;
                        nreg++;
                        consume(td, yd, xref);
                    } else {
                        bounded = true;
                    }
                };
                const int bad = coop_integrate_segment<O, 8>(C, S, w, D.phi, D.opts.rtol, As, t, t + T, MECH_NPC, dt, y,
                                                             K, h, nsteps, max_steps, clamped, dense);
                if (bad) { status |= bad; failed = true; break; }
                t = t + T;
                ndone++;
                ncycles_total++;
                if (bounded) break;        // everything from here on would be dropped by `bound`
                if (ndone < 2) continue;                           // nmin = 2 periods first
                if (ndone == 2) icount = 1;
                // isPeriodicallyStable on the two periods just produced (solvers.py:317-330), Z and ng
                const V ptp = O::add(vmax, nvmin);
                const double rz = sqrt(O::first(O::template bcast<1>(sse)) / HYB_NPC) / O::first(O::template bcast<1>(ptp));
                const double rn = sqrt(O::first(O::template bcast<2>(sse)) / HYB_NPC) / O::first(O::template bcast<2>(ptp));
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
            y = O::template on_lanes<7>(O::load4(ring, HYB_RING, lastslot), yp);
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
                    v = O::load4(ring, HYB_RING, (long)((first + HYB_NPC - 1) % HYB_RING));
                } else {
                    const V va = O::load4(ring, HYB_RING, a), vb = O::load4(ring, HYB_RING, b);
                    const double dx = xq - ring_t[a], den = ring_t[b] - ring_t[a];
                    v = O::fma_(O::div(O::sub(vb, va), O::splat(den)), O::splat(dx), va);
                }
                O::store4(cyc, HYB_NSPARSE_MAX, k, v);
            }
            const int n = (int)ceil((tend - t) / dt_sparse);
            const Linspace ts = linspace_make(t, tend, n + 1);
            double tsol = t;
            double hs = dt_sparse;
            for (int i = 0; i < n && !failed; i++) {
                const double tt = linspace_at(ts, i + 1);
                const V prof = O::load4(cyc, HYB_NSPARSE_MAX, i % npc);
                if (tt - tsol > HYB_MIN_SPARSE_DT) {
                    const double Zi = O::first(O::template bcast<1>(prof));
                    const double Cm = fs * bls_capacitance(p, Zi) + (1.0 - fs) * p.Cm0;
                    const double kV = 1e3 / Cm;
                    // membrane equations at the frozen capacitance (solvers.py:590-633)
                    double qm = O::first(O::template bcast<3>(y));
                    V xg = O::template on_lanes<0x0F>(O::splat(0.0), y);
                    if (!coop_membrane_rodas4<O>(C, S, kV, D.opts.rtol, tt - tsol, qm, xg, hs, nsteps, max_steps)) {
                        status |= 4;
                        failed = true;
                    }
                    y = O::template on_lane<3>(O::splat(qm), O::template on_lanes<0x0F>(y, xg));
                    tsol = tt;
                }
                // the row: U, Z, ng of the replayed cycle, (Qm, states) as integrated
                y = O::template on_lanes<7>(prof, y);
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
        if (store) O::fill_row_nan(rows + j * NCOL, linspace_at(out, (int)j));
    if (clamped) status |= 1;
    if (store && O::leader()) {
        D.status[c] = status;
        D.nsteps[c] = nsteps;
        D.ncycles[c] = ncycles_total;
    }
}

}  // namespace sonic
