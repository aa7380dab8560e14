// pysonic_amd/csrc/full_core.hpp -- per-configuration integration of the detailed NICE model
// (see full_lib.hip for the references into PySONIC). Shared by the HIP kernel and the CPU test
// harness (tests/native/, development only).
#pragma once
#include "mech_core.hpp"
#include "sonic_integrator.hpp"   // Schedule, Linspace

namespace sonic {

struct FullOpts {
    double rtol;
    int max_steps;     // > 0: step budget per configuration; 0: proportional to the dense grid (full_step_budget)
    double qdrive;     // Idrive 1e-3 of DrivenNeuronalBilayerSonophore.fullDerivatives (nbls.py:712-715)
};

// Step budget of one configuration when the caller sets none: FULL_STEPS_PER_POINT attempts per
// point of the reference's dense grid (1000 per acoustic period). The six BASELINE neurons take
// ~6 (RS 1.5e4 steps per 5 us at 500 kHz), SUseg with its 1e10 /s Borg-Graham rates ~100: a
// configuration that crawls fails within ~60x its normal time with status 4 instead
// of looking like a hang.
constexpr double FULL_STEPS_PER_POINT = 400.0;
SONIC_HD int full_step_budget(const FullOpts &o, double f, double tstop)
{
    if (o.max_steps > 0) return o.max_steps;
    const double b = FULL_STEPS_PER_POINT * MECH_NPC * f * tstop + 1e5;
    return b < 2e9 ? (int)b : 2000000000;
}

// d/dt of y = [U, Z, ng | model state (NY)]
template <class M, int NEURON>
SONIC_HD void full_rhs(const BLSParams &p, const typename M::Params &P, const MechDrive &d,
                       double fs, double t, const double *y, double *dy, bool &clamped)
{
    constexpr int NR = NeuronRates<NEURON>::NR;
    static_assert(1 + NR == M::NT, "rate list does not match the model's tables");
    bls_rhs(p, d, t, y, y[3], dy, clamped);
    // deflection-dependent capacitance and potential (nbls.py:148-151, 276-277; pneuron.py:498)
    const double Cm = fs * bls_capacitance(p, y[1]) + (1.0 - fs) * p.Cm0;
    double lk[M::NT], dlk[M::NT];
    lk[0] = qdiv(y[3], Cm) * 1e3;
    NeuronRates<NEURON>::eval(lk[0], lk + 1);
#pragma unroll
    for (int k = 0; k < M::NT; k++) dlk[k] = 0.0;
    M::template eval<false>(P, lk, dlk, y + 3, dy + 3, nullptr);
}

struct FullDev {
    const double *f, *A, *fs, *tstop;       // [n]
    const double *seg_t0, *seg_t1, *seg_x;   // dense-grid segments (CSR by seg_off)
    const int *seg_n;
    const long long *seg_off, *row_off;
    const double *y0;                        // [1 + NS] reference order
    double *traces;                          // [rows][NS + 6]
    int *status, *nsteps;
    long long n;
    double phi;
    FullOpts opts;
};

template <class M, int NEURON>
SONIC_HD void full_config(const FullDev &D, const BLSParams &p, const typename M::Params &P,
                          long long c)
{
    constexpr int NY = M::NY, N = 3 + NY, NCOL = NY + 5;   // t stim Z ng Qm states Vm
    const double f = D.f[c], fs = D.fs[c];
    const MechDrive d{2.0 * bls::PI * f, 0.0, D.phi};
    const double dt = 1.0 / (MECH_NPC * f);
    int status = 0;
    const int max_steps = full_step_budget(D.opts, f, D.tstop[c]);
    bool clamped = false, trial_clamped = false;   // see bls_rhs: kept for accepted steps only

    // initial conditions (nbls.py:321-329, bls.py:720-747): Z = quasi-static deflection at the
    // full amplitude's Pac(t = dt); the first of the two t = 0 rows (Z = 0) is never seen by the
    // resampling (np.interp picks the last duplicate)
    double y[N];
    {
        const double Pac_dt = D.A[c] * sin(d.w * dt - D.phi);
        const double Zqs = bls_balancedefQS(p, p.ng0, D.y0[0], Pac_dt);
        if (!(Zqs == Zqs)) status |= 2;
        y[0] = 0.0; y[1] = Zqs; y[2] = p.ng0;
#pragma unroll
        for (int i = 0; i < NY; i++) y[3 + M::out_perm(i)] = D.y0[i];
    }

    const long long s0 = D.seg_off[c];
    const int nseg = (int)(D.seg_off[c + 1] - s0);
    const long long M_rows = D.row_off[c + 1] - D.row_off[c];
    double *rows = D.traces + D.row_off[c] * NCOL;
    const Linspace out = linspace_make(0.0, D.tstop[c], (int)M_rows);   // resampled time grid
    long long j = 0;                         // next output row
    double tau = linspace_at(out, 0);

    double tp = 0.0, yp[N];                  // previous dense sample
#pragma unroll
    for (int i = 0; i < N; i++) yp[i] = y[i];
    int nsteps = 0;
#ifndef FULL_FLOOR_U
#define FULL_FLOOR_U 1e-6
#endif
#ifndef FULL_FLOOR_Z
#define FULL_FLOOR_Z 1e-13
#endif
#ifndef FULL_FLOOR_Y
#define FULL_FLOOR_Y 1e-6
#endif
    const double floor_[4] = {FULL_FLOOR_U, FULL_FLOOR_Z, 1e-25, FULL_FLOOR_Y};

    // consume one dense sample (ti, yi) of stimulus state xs: emit every output row <= ti
    auto consume = [&](double ti, const double *yi, double xs) {
        while (j < M_rows && tau <= ti) {
            double r[N];
            if (ti > tp) {
                const double w = (tau - tp) / (ti - tp);
#pragma unroll
                for (int i = 0; i < N; i++) r[i] = (yi[i] - yp[i]) * w + yp[i];   // np.interp
            } else {
#pragma unroll
                for (int i = 0; i < N; i++) r[i] = yi[i];
            }
            double *o = rows + j * NCOL;
            o[0] = tau;
            o[1] = (j == 0) ? 0.0 : xs;
            o[2] = r[1];
            o[3] = r[2];
#pragma unroll
            for (int i = 0; i < NY; i++) o[4 + i] = r[3 + M::out_perm(i)];
            // Vm from the RESAMPLED Qm and Z (nbls.py:317-319, 349-351)
            o[4 + NY] = r[3] / (fs * bls_capacitance(p, r[1]) + (1.0 - fs) * p.Cm0) * 1e3;
            j++;
            if (j < M_rows) tau = linspace_at(out, (int)j);
        }
        tp = ti;
#pragma unroll
        for (int i = 0; i < N; i++) yp[i] = yi[i];
    };

    double k1[N], k7[N], ynew[N], err[N], r4[N];
    double h = 0.25 * dt;
    for (int s = 0; s < nseg && !(status & 6); s++) {
        const double t0 = D.seg_t0[s0 + s], t1 = D.seg_t1[s0 + s], xs = D.seg_x[s0 + s];
        const int ns = D.seg_n[s0 + s];
        const Linspace grid = linspace_make(t0, t1, ns);
        MechDrive ds = d;
        ds.A = D.A[c] * xs;                       // eventfunc: drive.xvar * x (nbls.py:337)
        auto F = [&](double t, const double *yy, double *dy) {
            full_rhs<M, NEURON>(p, P, ds, fs, t, yy, dy, trial_clamped);
            dy[3] += D.opts.qdrive;
        };
        consume(t0, y, xs);                       // first dense row of the segment (duplicate)
        if (!(t1 > t0)) { consume(t1, y, xs); continue; }
        double t = t0;
        int i_d = 1;                              // next dense point of this segment
        double td = linspace_at(grid, i_d);
        F(t, y, k1);                              // the drive amplitude changed: no FSAL reuse
        h = fmin(h, t1 - t0);
        while (i_d < ns) {
            bool last = false;
            if (t + 1.0001 * h >= t1) { h = t1 - t; last = true; }
            trial_clamped = false;
            dopri5_step<N>(F, t, y, k1, h, ynew, k7, err, r4);
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
#if defined(FULL_TRACE) && defined(__HIP_DEVICE_COMPILE__)
            if ((nsteps % 20000) == 1 && (threadIdx.x & 63) == 0) {
                int im = 0; double em = 0.0;
                for (int i = 0; i < N; i++) {
                    const double fl = floor_[i < 3 ? i : 3];
                    const double e = fabs(err[i]) / (D.opts.rtol * fmax(fmax(fabs(y[i]), fabs(ynew[i])), fl));
                    if (e > em || !(e == e)) { em = e; im = i; }
                }
                printf("step %d t %.6e h %.3e en %.3e worst %d (e %.3e err %.3e y %.6e k1 %.3e)\n", nsteps, t, h, en, im,
                       em, err[im], y[im], k1[im]);
            }
#endif
            double fac = 0.9 * fast_exp(-0.2 * fast_log(fmax(en, 1e-10)));
            fac = fmin(5.0, fmax(0.2, fac));
            if (!(en == en)) fac = 0.2;
            if (en <= 1.0) {
                clamped = clamped || trial_clamped;
                const double tnew = last ? t1 : t + h;
                while (i_d < ns && (last || td <= tnew)) {
                    double yd[N];
                    if (td >= tnew) {
#pragma unroll
                        for (int i = 0; i < N; i++) yd[i] = ynew[i];
                    } else {
                        const double sg = (td - t) / h;
#pragma unroll
                        for (int i = 0; i < N; i++)
                            yd[i] = dopri5_dense(y[i], ynew[i], k1[i], k7[i], r4[i], h, sg);
                    }
                    consume(td, yd, xs);
                    i_d++;
                    if (i_d < ns) td = linspace_at(grid, i_d);
                }
#pragma unroll
                for (int i = 0; i < N; i++) { y[i] = ynew[i]; k1[i] = k7[i]; }
                t = tnew;
                h *= fac;
            } else {
                h *= fmin(fac, 1.0);
            }
            if (nsteps >= max_steps || !(h > 1e-18)) {
                status |= 4;
                break;
            }
        }
    }
    // rows not produced (failed configuration): NaN
    for (; j < M_rows; j++) {
        double *o = rows + j * NCOL;
        o[0] = linspace_at(out, (int)j);
        for (int i = 1; i < NCOL; i++) o[i] = NAN;
    }
    if (clamped) status |= 1;
    D.status[c] = status;
    D.nsteps[c] = nsteps;
}

}  // namespace sonic
