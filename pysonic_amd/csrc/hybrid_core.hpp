// pysonic_amd/csrc/hybrid_core.hpp -- per-configuration HYBRID integration of the detailed NICE
// model: NeuronalBilayerSonophore.__simHybrid + HybridSolver.solve
// (PySONIC/core/nbls.py:356-387, solvers.py:483-633). Shared by the HIP kernel (full_lib.hip) and
// the CPU test harness (tests/native/, development only).
//
// Per interval [t, tend], tend = min(next event, t + HYBRID_UPDATE_INTERVAL):
//   1. dense phase -- the full system [U, Z, ng | Qm, states] is integrated for whole acoustic
//      periods T, each sampled on np.linspace(t_c, t_c + T, 1000)[1:] (999 rows), two periods and
//      then more until rmse(last, previous) / ptp(last) < 1e-4 for Z and ng, at most
//      nmax = round((tend - t) / T) iterations of the loop counter (PeriodicSolver.solve,
//      solvers.py:336-365: the counter starts at 1, so nmax + 1 periods can be run);
//   2. rows beyond tend are dropped (`bound`);
//   3. sparse phase, if t < tend -- the last 999 dense rows are resampled on
//      np.linspace(first, last, 40); U, Z, ng replay that profile with period 40 while
//      (Qm, states) advance with the capacitance frozen at Cm(Z_i) per sparse step
//      (the reference: scipy dop853, rtol 1e-6; here RODAS4 at the dense tolerance: the system is stiff
//      at high pressure amplitudes, see membrane_rodas4);
//   4. the event is fired if the interval ended on it.
// The solution (dense + sparse rows, irregular in time) is resampled to CLASSIC_TARGET_DT with
// np.interp for the variables and 'nearest' for the stimulus state, on the fly.
#pragma once
#include "full_core.hpp"

namespace sonic {

constexpr int HYB_NPC = 999;          // dense rows per period (np.linspace(.., 1000)[1:])
constexpr int HYB_NSPARSE_MAX = 64;   // >= rows of the resampled period (40 for the reference's constants)
constexpr double HYB_UPDATE_INTERVAL = 5e-4;   // constants.py:41
constexpr double HYB_MIN_SPARSE_DT = 1e-12;    // constants.py:40
constexpr double HYB_NPC_SPARSE = 40.0;        // constants.py:39
// ring of the last two periods of dense rows, per configuration: t, U, Z, ng
constexpr int HYB_RING = 2 * HYB_NPC;
// (the octet-cooperative variant keeps a fourth, unused row in the sparse cycle: hybrid_coop.hpp)
constexpr int HYB_SCRATCH_DOUBLES = 4 * HYB_RING + 4 * HYB_NSPARSE_MAX;

struct HybridDev {
    const double *f, *A, *fs, *tstop;        // [n]
    const double *ev_t, *ev_x;               // events (CSR by ev_off), sorted by time
    const long long *ev_off, *row_off;
    const double *y0;                        // [1 + NS] reference order
    double *traces;                          // [rows][NS + 6]
    double *scratch;                         // [n][HYB_SCRATCH_DOUBLES]
    int *status, *nsteps, *ncycles;          // per configuration: status bits, step attempts, dense periods
    long long n;
    double phi;
    FullOpts opts;
    const long long *sel = nullptr;          // lane kernel: the configurations to integrate (n of them); null: 0 .. n - 1
};

// d/dt of the membrane state (Qm, states) at a frozen capacitance: the sparse phase of the hybrid
// scheme (solvers.py:590-633)
template <class M, int NEURON>
SONIC_HD void membrane_rhs(const typename M::Params &P, double Cm, const double *y, double *dy)
{
    double lk[M::NT], dlk[M::NT];
    lk[0] = qdiv(y[0], Cm) * 1e3;
    NeuronRates<NEURON>::eval(lk[0], lk + 1);
#pragma unroll
    for (int k = 0; k < M::NT; k++) dlk[k] = 0.0;
    M::template eval<false>(P, lk, dlk, y, dy, nullptr);
}

// The same system advanced over `span` by RODAS4 with the Jacobian of Model::eval (sonic_integrator.hpp:
// factor_W / solve_W; the rate functions' derivatives by central differences of 1 uV). The reference integrates
// the sparse phase with scipy's explicit dop853, and so did this kernel with its 5(4) pair -- but with the
// capacitance of a replayed deflection of several nanometres Vm = Qm / Cm reaches -460 mV in every acoustic
// period (RS, 600 kPa), where gate rate constants are ~1e12 / s and an explicit pair needs ~1e3 steps per
// sparse step of 50 ns (profiles/r02h_hybrid_probe.txt). `hs` carries the step size; false = step budget spent.
template <class M, int NEURON>
SONIC_HD bool membrane_rodas4(const typename M::Params &P, double Cm, double *y, double span, double &hs,
                              double rtol, int &nsteps, int max_steps)
{
    using namespace rodas4;
    constexpr int NY = M::NY, NT = M::NT;
    const double kV = qdiv(1e3, Cm);
    auto lines = [&](double q, double *lk) {
        lk[0] = q * kV;
        NeuronRates<NEURON>::eval(lk[0], lk + 1);
    };
    double zero[NT];
#pragma unroll
    for (int k = 0; k < NT; k++) zero[k] = 0.0;
    double tcur = 0.0;
    hs = fmin(hs, span);
    while (tcur < span) {
        bool last = false;
        double h = hs;
        if (tcur + 1.0001 * h >= span) { h = span - tcur; last = true; }
        double lk[NT], dlk[NT], lp[NT], lm[NT], k[6][NY], yt[NY], f[NY];
        Jac<M::NC, M::NG> J;
        lines(y[0], lk);
        {
            const double dv = 1e-3;                                  // mV
            lp[0] = lk[0] + dv; lm[0] = lk[0] - dv;
            NeuronRates<NEURON>::eval(lp[0], lp + 1);
            NeuronRates<NEURON>::eval(lm[0], lm + 1);
            dlk[0] = kV;
#pragma unroll
            for (int i = 1; i < NT; i++) dlk[i] = (lp[i] - lm[i]) * (0.5 / dv) * kV;
        }
        M::template eval<true>(P, lk, dlk, y, k[0], &J);
        const double inv_h = 1.0 / h;
        WFactor<M> F;
        factor_W<M>(J, inv_h * (1.0 / gamma), F);
        solve_W<M>(J, F, k[0]);
        // stage s >= 1: k_s = W^-1 (f(Y_s) + sum_j c_sj / h k_j)
        auto stage = [&](int s_, const double *cc) {
            lines(yt[0], lk);
            M::template eval<false>(P, lk, zero, yt, f, nullptr);
#pragma unroll
            for (int i = 0; i < NY; i++) {
                double acc = f[i];
                for (int j = 0; j < s_; j++) acc += cc[j] * inv_h * k[j][i];
                k[s_][i] = acc;
            }
            solve_W<M>(J, F, k[s_]);
        };
#pragma unroll
        for (int i = 0; i < NY; i++) yt[i] = y[i] + a21 * k[0][i];
        { const double cc[1] = {c21}; stage(1, cc); }
#pragma unroll
        for (int i = 0; i < NY; i++) yt[i] = y[i] + a31 * k[0][i] + a32 * k[1][i];
        { const double cc[2] = {c31, c32}; stage(2, cc); }
#pragma unroll
        for (int i = 0; i < NY; i++) yt[i] = y[i] + a41 * k[0][i] + a42 * k[1][i] + a43 * k[2][i];
        { const double cc[3] = {c41, c42, c43}; stage(3, cc); }
#pragma unroll
        for (int i = 0; i < NY; i++) yt[i] = y[i] + a51 * k[0][i] + a52 * k[1][i] + a53 * k[2][i] + a54 * k[3][i];
        { const double cc[4] = {c51, c52, c53, c54}; stage(4, cc); }
#pragma unroll
        for (int i = 0; i < NY; i++) yt[i] += k[4][i];
        { const double cc[5] = {c61, c62, c63, c64, c65}; stage(5, cc); }
        nsteps++;
        // embedded error estimate = k6; RMS of err / (rtol max(|y|, |ynew|, 1e-6)) as the explicit pair had it
        double e2 = 0.0;
#pragma unroll
        for (int i = 0; i < NY; i++) {
            const double yn = yt[i] + k[5][i];
            const double e = k[5][i] / (rtol * fmax(fmax(fabs(y[i]), fabs(yn)), 1e-6));
            e2 += e * e;
        }
        const double en = sqrt(e2 * (1.0 / NY));
        double fac = 0.9 * fast_exp(-0.25 * fast_log(fmax(en, 1e-10)));
        fac = fmin(6.0, fmax(0.2, fac));
        if (!(en == en)) fac = 0.2;
        if (en <= 1.0) {
#pragma unroll
            for (int i = 0; i < NY; i++) y[i] = yt[i] + k[5][i];
            tcur = last ? span : tcur + h;
            hs = last ? fmax(h * fac, hs) : h * fac;
        } else {
            hs = h * fmin(fac, 1.0);
        }
        if (nsteps >= max_steps || !(hs > 1e-18)) return false;
    }
    return true;
}

template <class M, int NEURON>
SONIC_HD void hybrid_config(const HybridDev &D, const BLSParams &p, const typename M::Params &P,
                            long long c)
{
    constexpr int NY = M::NY, N = 3 + NY, NCOL = NY + 5;   // t stim Z ng Qm states Vm
    const double f = D.f[c], fs = D.fs[c], tstop = D.tstop[c];
    const double T = 1.0 / f;
    const double dt = 1.0 / (MECH_NPC * f);
    const double dt_sparse = 1.0 / (HYB_NPC_SPARSE * f);
    int status = 0, nsteps = 0, ncycles_total = 0;
    const int max_steps = full_step_budget(D.opts, f, tstop);
    bool clamped = false, trial_clamped = false;   // see bls_rhs: kept for accepted steps only

    double *ring_t = D.scratch + c * (long long)HYB_SCRATCH_DOUBLES;
    double *ring_u = ring_t + HYB_RING, *ring_z = ring_u + HYB_RING, *ring_n = ring_z + HYB_RING;
    double *sp_u = ring_n + HYB_RING, *sp_z = sp_u + HYB_NSPARSE_MAX, *sp_n = sp_z + HYB_NSPARSE_MAX;
    long long nring = 0;                     // dense rows pushed so far (ring index = nring % HYB_RING)
    long long nreg = 0;                      // trailing dense rows that are dt-regular (getCycle)

    // initial conditions: two rows at t = 0 (Z = 0, then the quasi-static deflection)
    double y[N];
    {
        const double Pac_dt = D.A[c] * sin(2.0 * bls::PI * f * dt - D.phi);
        const double Zqs = bls_balancedefQS(p, p.ng0, D.y0[0], Pac_dt);
        if (!(Zqs == Zqs)) status |= 2;
        y[0] = 0.0; y[1] = Zqs; y[2] = p.ng0;
#pragma unroll
        for (int i = 0; i < NY; i++) y[3 + M::out_perm(i)] = D.y0[i];
    }

    const long long M_rows = D.row_off[c + 1] - D.row_off[c];
    double *rows = D.traces + D.row_off[c] * NCOL;
    const Linspace out = linspace_make(0.0, tstop, (int)M_rows);
    long long j = 0;
    double tau = linspace_at(out, 0);
    double tp = 0.0, xp = 0.0, yp[N];
#pragma unroll
    for (int i = 0; i < N; i++) yp[i] = y[i];

    // one row (ti, yi, stimulus state xs) of the solution: emit every output row <= ti
    // (np.interp for the variables; interp1d 'nearest' for the state: the left row up to and
    // including the midpoint, solvers.py:184-191)
    auto consume = [&](double ti, const double *yi, double xs) {
        while (j < M_rows && tau <= ti) {
            double r[N];
            if (ti > tp) {
                const double w = (tau - tp) / (ti - tp);
#pragma unroll
                for (int i = 0; i < N; i++) r[i] = (yi[i] - yp[i]) * w + yp[i];
            } else {
#pragma unroll
                for (int i = 0; i < N; i++) r[i] = yi[i];
            }
            double *o = rows + j * NCOL;
            o[0] = tau;
            o[1] = (tau <= (tp + ti) / 2.0) ? xp : xs;
            o[2] = r[1];
            o[3] = r[2];
#pragma unroll
            for (int i = 0; i < NY; i++) o[4 + i] = r[3 + M::out_perm(i)];
            o[4 + NY] = r[3] / (fs * bls_capacitance(p, r[1]) + (1.0 - fs) * p.Cm0) * 1e3;
            j++;
            if (j < M_rows) tau = linspace_at(out, (int)j);
        }
        tp = ti;
        xp = xs;
#pragma unroll
        for (int i = 0; i < N; i++) yp[i] = yi[i];
    };
    // of the two initial rows at t = 0 only the second (quasi-static deflection) is seen by
    // np.interp, which picks the last duplicate
    consume(0.0, y, 0.0);

    const double floor_[4] = {1e-6, 1e-13, 1e-25, 1e-6};
    const long long e0 = D.ev_off[c];
    const int nev = (int)(D.ev_off[c + 1] - e0);
    int iev = 0;
    double t = 0.0, xref = 0.0;
    MechDrive drv{2.0 * bls::PI * f, 0.0, D.phi};      // event_params: drive amplitude 0 before the first event
    double h = 0.25 * dt;
    double k1[N], k7[N], ynew[N], err[N], r4[N];
    bool failed = false;

    // event list + the terminal (tstop, none)
    auto event_t = [&](int i) { return i < nev ? D.ev_t[e0 + i] : tstop; };

    while (iev <= nev && !failed) {
        const double tevent = event_t(iev);
        const double tend = fmin(tevent, t + HYB_UPDATE_INTERVAL);
        const int nmax = (int)nearbyint((tend - t) / T);

        // ---------------- 1. dense periods ----------------
        bool bounded = false;
        if (nmax > 0) {
            if (nmax < 2) { status |= 16; failed = true; break; }   // the reference asserts nmin <= nmax
            auto F = [&](double tt, const double *yy, double *dy) {
                full_rhs<M, NEURON>(p, P, drv, fs, tt, yy, dy, trial_clamped);
                dy[3] += D.opts.qdrive;
            };
            int icount = 0;                    // the reference's loop counter `i`
            int ndone = 0;                     // periods integrated in this call
            bool stable = false;
            while (true) {
                // one period: rows on np.linspace(t, t + T, 1000)[1:]
                const double t0c = t, t1c = t + T;
                const Linspace grid = linspace_make(t0c, t1c, MECH_NPC);
                double sse_z = 0.0, sse_n = 0.0, zmin = INFINITY, zmax = -INFINITY,
                       nmin_ = INFINITY, nmax_ = -INFINITY;
                int i_d = 1;
                double td = linspace_at(grid, i_d);
                double tc = t0c;
                F(tc, y, k1);
                h = fmin(h, T);
                while (i_d < MECH_NPC) {
                    bool last = false;
                    if (tc + 1.0001 * h >= t1c) { h = t1c - tc; last = true; }
                    trial_clamped = false;
                    dopri5_step<N>(F, tc, y, k1, h, ynew, k7, err, r4);
                    nsteps++;
                    double e2 = 0.0;
#pragma unroll
                    for (int i = 0; i < N; i++) {
                        const double fl = floor_[i < 3 ? i : 3];
                        const double sc = D.opts.rtol * fmax(fmax(fabs(y[i]), fabs(ynew[i])), fl);
                        const double e = err[i] / sc;
                        e2 += e * e;
                    }
                    const double en = sqrt(e2 * (1.0 / N));
                    double fac = 0.9 * fast_exp(-0.2 * fast_log(fmax(en, 1e-10)));
                    fac = fmin(5.0, fmax(0.2, fac));
                    if (!(en == en)) fac = 0.2;
                    if (en <= 1.0) {
                        clamped = clamped || trial_clamped;
                        const double tnew = last ? t1c : tc + h;
                        while (i_d < MECH_NPC && (last || td <= tnew)) {
                            double yd[N];
                            if (td >= tnew) {
#pragma unroll
                                for (int i = 0; i < N; i++) yd[i] = ynew[i];
                            } else {
                                const double sg = (td - tc) / h;
#pragma unroll
                                for (int i = 0; i < N; i++)
                                    yd[i] = dopri5_dense(y[i], ynew[i], k1[i], k7[i], r4[i], h, sg);
                            }
                            // rows beyond tend are dropped by `bound` (solvers.py:129-139)
                            if (td <= tend) {
                                const int slot = (int)(nring % HYB_RING);
                                // periodic stability: this row against the row one period earlier
                                if (nring >= HYB_NPC) {
                                    const int prev = (int)((nring - HYB_NPC) % HYB_RING);
                                    const double dz = yd[1] - ring_z[prev], dn = yd[2] - ring_n[prev];
                                    sse_z += dz * dz;
                                    sse_n += dn * dn;
                                }
                                zmin = fmin(zmin, yd[1]); zmax = fmax(zmax, yd[1]);
                                nmin_ = fmin(nmin_, yd[2]); nmax_ = fmax(nmax_, yd[2]);
                                ring_t[slot] = td; ring_u[slot] = yd[0];
                                ring_z[slot] = yd[1]; ring_n[slot] = yd[2];
                                nring++;
                                nreg++;
                                consume(td, yd, xref);
                            } else {
                                bounded = true;
                            }
                            i_d++;
                            if (i_d < MECH_NPC) td = linspace_at(grid, i_d);
                        }
#pragma unroll
                        for (int i = 0; i < N; i++) { y[i] = ynew[i]; k1[i] = k7[i]; }
                        tc = tnew;
                        h *= fac;
                    } else {
                        h *= fmin(fac, 1.0);
                    }
                    if (nsteps >= max_steps || !(h > 1e-18)) { status |= 4; failed = true; break; }
                }
                if (failed) break;
                t = t1c;
                ndone++;
                ncycles_total++;
                if (bounded) break;        // everything from here on would be dropped by `bound`
                if (ndone < 2) continue;                           // nmin = 2 periods first
                if (ndone == 2) icount = 1;
                // isPeriodicallyStable on the two periods just produced (solvers.py:317-330)
                const double rz = sqrt(sse_z / HYB_NPC) / (zmax - zmin);
                const double rn = sqrt(sse_n / HYB_NPC) / (nmax_ - nmin_);
                stable = rz < 1e-4 && rn < 1e-4;
                if (stable || !(icount < nmax)) break;
                icount++;
            }
            if (failed) break;
        }
        // the state after `bound`: the last row kept
        if (bounded) {
            const int lastslot = (int)((nring - 1) % HYB_RING);
            t = ring_t[lastslot];
            // y of that row: U, Z, ng from the ring, (Qm, states) = the last consumed row
            y[0] = ring_u[lastslot]; y[1] = ring_z[lastslot]; y[2] = ring_n[lastslot];
#pragma unroll
            for (int i = 3; i < N; i++) y[i] = yp[i];
        }

        // ---------------- 3. sparse phase ----------------
        if (t < tend) {
            // last period = the last 999 dt-regular rows (getCycle(-1), solvers.py:283-315)
            if (nreg < HYB_NPC || nring < HYB_NPC) { status |= 32; failed = true; break; }
            const long long first = nring - HYB_NPC;
            auto rt = [&](int k) { return ring_t[(int)((first + k) % HYB_RING)]; };
            const double tl0 = rt(0), tl1 = rt(HYB_NPC - 1);
            long long ns_ll = (long long)nearbyint((tl1 - tl0) / dt_sparse);
            int npc = (int)(ns_ll > 2 ? ns_ll : 2);
            if (npc > HYB_NSPARSE_MAX) { status |= 32; failed = true; break; }
            const Linspace sg = linspace_make(tl0, tl1, npc);
            int lo = 0;
            for (int k = 0; k < npc; k++) {                          // np.interp of U, Z, ng
                const double xq = linspace_at(sg, k);
                while (lo < HYB_NPC - 2 && rt(lo + 1) <= xq) lo++;
                const int a = (int)((first + lo) % HYB_RING), b = (int)((first + lo + 1) % HYB_RING);
                if (xq >= tl1) {
                    const int e = (int)((first + HYB_NPC - 1) % HYB_RING);
                    sp_u[k] = ring_u[e]; sp_z[k] = ring_z[e]; sp_n[k] = ring_n[e];
                } else {
                    const double dx = xq - ring_t[a], den = ring_t[b] - ring_t[a];
                    sp_u[k] = (ring_u[b] - ring_u[a]) / den * dx + ring_u[a];
                    sp_z[k] = (ring_z[b] - ring_z[a]) / den * dx + ring_z[a];
                    sp_n[k] = (ring_n[b] - ring_n[a]) / den * dx + ring_n[a];
                }
            }
            const int n = (int)ceil((tend - t) / dt_sparse);
            const Linspace ts = linspace_make(t, tend, n + 1);
            double ys[NY];
#pragma unroll
            for (int i = 0; i < NY; i++) ys[i] = y[3 + i];
            double tsol = t;
            double hs = dt_sparse;
            for (int i = 0; i < n && !failed; i++) {
                const double tt = linspace_at(ts, i + 1);
                if (tt - tsol > HYB_MIN_SPARSE_DT) {
                    const double Cm = fs * bls_capacitance(p, sp_z[i % npc]) + (1.0 - fs) * p.Cm0;
                    if (!membrane_rodas4<M, NEURON>(P, Cm, ys, tt - tsol, hs, D.opts.rtol, nsteps, max_steps)) {
                        status |= 4;
                        failed = true;
                    }
                    tsol = tt;
                }
                double yrow[N];
                yrow[0] = sp_u[i % npc]; yrow[1] = sp_z[i % npc]; yrow[2] = sp_n[i % npc];
#pragma unroll
                for (int k = 0; k < NY; k++) yrow[3 + k] = ys[k];
                consume(tt, yrow, xref);
#pragma unroll
                for (int k = 0; k < N; k++) y[k] = yrow[k];
            }
            t = tend;
            nreg = 0;                 // sparse rows break the dt-regular run
        }

        // ---------------- 4. event ----------------
        if (t == tevent) {
            if (iev < nev) {
                xref = D.ev_x[e0 + iev];
                drv.A = D.A[c] * xref;                  // eventfunc: drive.xvar * x (nbls.py:367)
            }
            iev++;
        }
    }

    for (; j < M_rows; j++) {           // rows not produced (failed configuration): NaN
        double *o = rows + j * NCOL;
        o[0] = linspace_at(out, (int)j);
        for (int i = 1; i < NCOL; i++) o[i] = NAN;
    }
    if (clamped) status |= 1;
    D.status[c] = status;
    D.nsteps[c] = nsteps;
    D.ncycles[c] = ncycles_total;
}

}  // namespace sonic
