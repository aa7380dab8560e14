// pysonic_amd/csrc/mech_core.hpp
//
// Bilayer-sonophore mechanics on the device: the 3-ODE system (U, Z, ng) of
// BilayerSonophore.derivatives (PySONIC/core/bls.py:681-718 and the pressure / geometry terms it
// calls: 286-319, 472-491, 508-526, 596-655), the true voltage-dependent rate functions of the six
// BASELINE neurons (PySONIC/neurons/cortical.py:36-66,254-272; thalamic.py:31-53,164-179,289-323;
// stn.py:209-338) and the explicit adaptive Dormand-Prince integrators: the 5(4) pair with its 4th-order
// continuous extension and the 8(5,3) pair with its 7th-order one (coefficients: dop853_coeffs.hpp).
//
// Used by  mech_cycles_kernel  (NeuronalBilayerSonophore.computeEffVars, nbls.py:153-222 =
// simCycles + PeriodicSolver, bls.py:749-789, solvers.py:224-365)  and by the `full` kernel.
//
// Why explicit: the reference's LSODA spends most of these runs in its BDF mode (the Lennard-Jones
// and gas-pressure terms are stiff near the turning points), but an implicit scheme needs a 3x3
// (8x8 in the `full` system) Newton solve per step and per lane; at the accuracy asked here
// (rtol 1e-9, the reference's converged runs are the target) a high-order explicit pair with
// step-size control costs fewer right-hand sides: 8(5,3) takes ~1/3 of the evaluations of 5(4)
// (DESIGN.md section 5), and mech_cell steps with it.
#pragma once
#include <math.h>
#include "sonic_models.hpp"
#include "dop853_coeffs.hpp"

namespace sonic {

// constants of the model (bls.py:88-110, constants.py:13)
namespace bls {
constexpr double T = 309.15, delta0 = 2.0e-9, rhoL = 1075.0, muL = 7.0e-4, muS = 0.035, kA = 0.24,
                 C0 = 0.62, kH = 1.613e5, P0 = 1.0e5, Dgl = 3.68e-9, xi = 0.5e-9,
                 epsilon0 = 8.854e-12, epsilonR = 1.0, rel_Zmin = -0.49, Rg = 8.31342;
constexpr double PI = 3.14159265358979323846;
}  // namespace bls

// per-sonophore parameters = BilayerSonophore.device_params() (pysonic_amd/core/bls.py)
struct BLSParams {
    double a, Cm0, Delta, LJ_x0, LJ_C, LJ_nrep, LJ_nattr, kA_tissue, ng0;
};

SONIC_HD double bls_volume(const BLSParams &p, double Z)
{
    const double a2 = p.a * p.a;
    return bls::PI * a2 * p.Delta * (1.0 + (qdiv(Z, 3.0 * p.Delta) * (3.0 + qdiv(Z * Z, a2))));
}

SONIC_HD double bls_PMavgpred(const BLSParams &p, double Z)
{
    const double r = qdiv(p.LJ_x0, 2.0 * Z + p.Delta);
    const double lr = fast_log(r);                 // r > 0: Z is at least Zmin = -0.49 Delta wherever this is called
    return p.LJ_C * (fast_exp(p.LJ_nrep * lr) - fast_exp(p.LJ_nattr * lr));
}

SONIC_HD double bls_Pelec(const BLSParams &p, double Z, double Qm)
{
    const double a2 = p.a * p.a;
    const double relS = qdiv(a2, a2 + Z * Z);
    return -relS * Qm * Qm * (1.0 / (2.0 * bls::epsilon0 * bls::epsilonR));
}

// bls.py:334-345
SONIC_HD double bls_capacitance(const BLSParams &p, double Z)
{
    // branch-free (this sits inside the right-hand sides): the formula is evaluated at a harmless
    // Z when Z = 0 and the reference's special case selected afterwards
    const double Zs = Z == 0.0 ? p.Delta : Z;
    const double a2 = p.a * p.a;
    const double Z2 = qdiv(a2 - Zs * Zs - Zs * p.Delta, 2.0 * Zs);
    const double w = qdiv(2.0 * Zs + p.Delta, p.Delta);
    const double lw = w > 0.0 ? fast_log(w) : NAN;     // (a deflection below -Delta / 2: NaN, as the reference's log)
    const double Cm = qdiv(p.Cm0 * p.Delta, a2) * (Zs + Z2 * lw);
    return Z == 0.0 ? p.Cm0 : Cm;
}

// net quasi-steady pressure (bls.py:538-553), used for the initial deflection
SONIC_HD double bls_PtotQS(const BLSParams &p, double Z, double ng, double Qm, double Pac)
{
    return bls_PMavgpred(p, Z) + qdiv(ng * bls::Rg * bls::T, bls_volume(p, Z)) - bls::P0 - Pac +
           bls_Pelec(p, Z, Qm);
}

// Root of PtotQS(Z) on [Zmin, a] (balancedefQS, bls.py:555-573: brentq, xtol = 1e-16).
// Bisection to the last bit; returns NaN if the pressure does not change sign (ValueError in the
// reference).
SONIC_HD double bls_balancedefQS(const BLSParams &p, double ng, double Qm, double Pac)
{
    double lo = bls::rel_Zmin * p.Delta, hi = p.a;
    const double flo = bls_PtotQS(p, lo, ng, Qm, Pac), fhi = bls_PtotQS(p, hi, ng, Qm, Pac);
    if (!(flo > 0.0 && 0.0 > fhi)) return NAN;
    for (int it = 0; it < 200; it++) {
        const double mid = 0.5 * (lo + hi);
        if (!(mid > lo && mid < hi)) break;
        if (bls_PtotQS(p, mid, ng, Qm, Pac) > 0.0) lo = mid; else hi = mid;
    }
    return 0.5 * (lo + hi);
}

// dy/dt of y = (U, Z, ng) (bls.py:681-718). `clamped` is set if Z had to be clamped at Zmin. The
// integrators pass a per-step flag and keep it only for ACCEPTED steps: a rejected trial step (the
// first step of a 20 kHz cell starts at h = 50 ns, far above the ns scale of the leaflet dynamics)
// may overshoot below Zmin without the solution ever getting there, and the reference's warning
// (bls.py:694-696) is not issued on such cells either (tests/golden/golden_mech_axes.npz).
struct MechDrive {
    double w;      // 2 pi f
    double A;      // Pa
    double phi;    // rad
};

SONIC_HD void bls_rhs(const BLSParams &p, const MechDrive &d, double t, const double *y,
                      double Qm, double *dy, bool &clamped)
{
    const double U = y[0], ng = y[2];
    double Z = y[1];
    const double Zmin = bls::rel_Zmin * p.Delta;
    clamped = clamped || Z < Zmin;
    Z = Z < Zmin ? Zmin : Z;
    const double a2 = p.a * p.a;
    const double is = fast_rcp(a2 + Z * Z);
    const double invR = 2.0 * Z * is;                   // 1 / curvrad (0 at Z = 0)
    const double ainvR = fabs(invR);
    const double Pg = qdiv(ng * bls::Rg * bls::T, bls_volume(p, Z));
    const double Pm = bls_PMavgpred(p, Z);
    const double Pac = d.A * fast_sin(d.w * t - d.phi);
    const double Pv = -12.0 * U * bls::delta0 * bls::muS * invR * invR - 4.0 * U * bls::muL * ainvR;
    const double za = qdiv(Z, p.a);
    const double strain = za * za;
    const double PE = -(bls::kA + p.kA_tissue) * strain * invR;
    const double Pel = -(a2 * is) * Qm * Qm * (1.0 / (2.0 * bls::epsilon0 * bls::epsilonR));     // bls_Pelec
    const double Ptot = Pm + Pg - bls::P0 - Pac + PE + Pv + Pel;
    dy[0] = Ptot * ainvR * (1.0 / bls::rhoL) - 1.5 * U * U * invR;
    dy[1] = U;
    dy[2] = 2.0 * bls::PI * (a2 + Z * Z) * bls::Dgl * (bls::C0 - Pg * (1.0 / bls::kH)) * (1.0 / bls::xi);
}

// The same right-hand side with its analytic Jacobian, for the Rosenbrock path of the detailed model
// (full_core.hpp: configurations whose gates turn stiff). Jm[r][c]: rows (U, Z, ng)', columns (U, Z, ng, Qm);
// fUt = d U' / d t (the acoustic pressure is the only explicit time dependence). The clamp of the
// deflection at Zmin is not differentiated (unreachable within the lookup's amplitude range).
SONIC_HD void bls_rhs_jac(const BLSParams &p, const MechDrive &d, double t, const double *y, double Qm,
                          double *dy, double (*Jm)[4], double &fUt, bool &clamped)
{
    const double U = y[0], ng = y[2];
    double Z = y[1];
    const double Zmin = bls::rel_Zmin * p.Delta;
    clamped = clamped || Z < Zmin;
    Z = Z < Zmin ? Zmin : Z;
    const double a2 = p.a * p.a;
    const double s = a2 + Z * Z, is = 1.0 / s;
    const double invR = 2.0 * Z * is, dinvR = 2.0 * (a2 - Z * Z) * is * is;
    const double sg = invR < 0.0 ? -1.0 : 1.0;
    const double ainvR = fabs(invR), dainvR = sg * dinvR;
    const double V = bls_volume(p, Z), iV = 1.0 / V;
    const double RT = bls::Rg * bls::T;
    const double Pg = ng * RT * iV, dPg_dng = RT * iV, dPg_dZ = -Pg * bls::PI * s * iV;
    // Lennard-Jones pressure: C (r^nrep - r^nattr), r = x0 / (2 Z + Delta)
    const double den = 2.0 * Z + p.Delta;
    const double lr = log(p.LJ_x0 / den);
    const double prep = exp(p.LJ_nrep * lr), patt = exp(p.LJ_nattr * lr);
    const double Pm = p.LJ_C * (prep - patt);
    const double dPm_dZ = p.LJ_C * (p.LJ_nrep * prep - p.LJ_nattr * patt) * (-2.0 / den);
    const double ph = d.w * t - d.phi;
    const double Pac = d.A * sin(ph), dPac_dt = d.A * d.w * cos(ph);
    const double cS = 12.0 * bls::delta0 * bls::muS, cL = 4.0 * bls::muL;
    const double Pv = -U * (cS * invR * invR + cL * ainvR);
    const double dPv_dU = -(cS * invR * invR + cL * ainvR);
    const double dPv_dZ = -U * (2.0 * cS * invR * dinvR + cL * dainvR);
    const double kAt = bls::kA + p.kA_tissue;
    const double strain = Z * Z / a2;
    const double PE = -kAt * strain * invR;
    const double dPE_dZ = -kAt * (2.0 * Z / a2 * invR + strain * dinvR);
    const double ke = 1.0 / (2.0 * bls::epsilon0 * bls::epsilonR);
    const double Pel = -(a2 * is) * Qm * Qm * ke;
    const double dPel_dQ = -(a2 * is) * 2.0 * Qm * ke, dPel_dZ = -Pel * 2.0 * Z * is;
    const double Ptot = Pm + Pg - bls::P0 - Pac + PE + Pv + Pel;
    const double dPtot_dZ = dPm_dZ + dPg_dZ + dPE_dZ + dPv_dZ + dPel_dZ;
    const double ir = 1.0 / bls::rhoL;
    dy[0] = Ptot * ainvR * ir - 1.5 * U * U * invR;
    dy[1] = U;
    const double kg = 2.0 * bls::PI * bls::Dgl / bls::xi;
    dy[2] = kg * s * (bls::C0 - Pg * (1.0 / bls::kH));
    Jm[0][0] = dPv_dU * ainvR * ir - 3.0 * U * invR;
    Jm[0][1] = dPtot_dZ * ainvR * ir + Ptot * dainvR * ir - 1.5 * U * U * dinvR;
    Jm[0][2] = dPg_dng * ainvR * ir;
    Jm[0][3] = dPel_dQ * ainvR * ir;
    Jm[1][0] = 1.0; Jm[1][1] = 0.0; Jm[1][2] = 0.0; Jm[1][3] = 0.0;
    Jm[2][0] = 0.0;
    Jm[2][1] = kg * (2.0 * Z * (bls::C0 - Pg * (1.0 / bls::kH)) - s * dPg_dZ * (1.0 / bls::kH));
    Jm[2][2] = -kg * s * dPg_dng * (1.0 / bls::kH);
    Jm[2][3] = 0.0;
    fUt = -dPac_dt * ainvR * ir;
}

// capacitance (bls_capacitance) and its derivative with respect to the deflection
SONIC_HD void bls_capacitance_d(const BLSParams &p, double Z, double &Cm, double &dCm)
{
    const double Zs = Z == 0.0 ? 1e-3 * p.Delta : Z;
    const double a2 = p.a * p.a;
    const double Z2 = (a2 - Zs * Zs - Zs * p.Delta) / (2.0 * Zs);
    const double den = 2.0 * Zs + p.Delta;
    const double lw = den > 0.0 ? log(den / p.Delta) : NAN;
    const double k = p.Cm0 * p.Delta / a2;
    const double dZ2 = -den / (2.0 * Zs) - Z2 / Zs;
    dCm = k * (1.0 + dZ2 * lw + Z2 * 2.0 / den);
    Cm = Z == 0.0 ? p.Cm0 : k * (Zs + Z2 * lw);
}

// ---------------------------------------------------------------------------------------------
// Dormand-Prince 5(4) with the standard 4th-order continuous extension (Hairer, Norsett, Wanner,
// "Solving ODEs I", II.5 / II.6; coefficients of DOPRI5).
// ---------------------------------------------------------------------------------------------
namespace dp5 {
constexpr double c2 = 0.2, c3 = 0.3, c4 = 0.8, c5 = 8.0 / 9.0;
constexpr double a21 = 0.2;
constexpr double a31 = 3.0 / 40.0, a32 = 9.0 / 40.0;
constexpr double a41 = 44.0 / 45.0, a42 = -56.0 / 15.0, a43 = 32.0 / 9.0;
constexpr double a51 = 19372.0 / 6561.0, a52 = -25360.0 / 2187.0, a53 = 64448.0 / 6561.0,
                 a54 = -212.0 / 729.0;
constexpr double a61 = 9017.0 / 3168.0, a62 = -355.0 / 33.0, a63 = 46732.0 / 5247.0,
                 a64 = 49.0 / 176.0, a65 = -5103.0 / 18656.0;
constexpr double a71 = 35.0 / 384.0, a73 = 500.0 / 1113.0, a74 = 125.0 / 192.0,
                 a75 = -2187.0 / 6784.0, a76 = 11.0 / 84.0;
constexpr double e1 = 71.0 / 57600.0, e3 = -71.0 / 16695.0, e4 = 71.0 / 1920.0,
                 e5 = -17253.0 / 339200.0, e6 = 22.0 / 525.0, e7 = -1.0 / 40.0;
constexpr double d1 = -12715105075.0 / 11282082432.0, d3 = 87487479700.0 / 32700410799.0,
                 d4 = -10690763975.0 / 1880347072.0, d5 = 701980252875.0 / 199316789632.0,
                 d6 = -1453857185.0 / 822651844.0, d7 = 69997945.0 / 29380423.0;
}  // namespace dp5

// One DOPRI5 step attempt. F(t, y, dy) evaluates the right-hand side. k1 = f(t, y) must be
// provided (FSAL); on return k7 = f(t + h, ynew) and r4 holds the one dense-output vector that
// needs all the stages. Continuous extension (Hairer et al., II.6):
//   y(t + s h) = y + s (d + (1-s) (b + s (d - h k7 - b + (1-s) r4))),  d = ynew - y, b = h k1 - d
template <int N, class RHS>
SONIC_HD void dopri5_step_looped(RHS &&F, double t, const double *y, const double *k1, double h,
                                 double *ynew, double *k7, double *err, double *r4, double *hlambda = nullptr);

// all stages written out: the three-equation mechanical system (small right-hand side)
template <int N, class RHS>
SONIC_HD void dopri5_step_unrolled(RHS &&F, double t, const double *y, const double *k1, double h,
                          double *ynew, double *k7, double *err, double *r4)
{
    using namespace dp5;
    double k2[N], k3[N], k4[N], k5[N], k6[N], yt[N];
#pragma unroll
    for (int i = 0; i < N; i++) yt[i] = y[i] + h * a21 * k1[i];
    F(t + c2 * h, yt, k2);
#pragma unroll
    for (int i = 0; i < N; i++) yt[i] = y[i] + h * (a31 * k1[i] + a32 * k2[i]);
    F(t + c3 * h, yt, k3);
#pragma unroll
    for (int i = 0; i < N; i++) yt[i] = y[i] + h * (a41 * k1[i] + a42 * k2[i] + a43 * k3[i]);
    F(t + c4 * h, yt, k4);
#pragma unroll
    for (int i = 0; i < N; i++)
        yt[i] = y[i] + h * (a51 * k1[i] + a52 * k2[i] + a53 * k3[i] + a54 * k4[i]);
    F(t + c5 * h, yt, k5);
#pragma unroll
    for (int i = 0; i < N; i++)
        yt[i] = y[i] + h * (a61 * k1[i] + a62 * k2[i] + a63 * k3[i] + a64 * k4[i] + a65 * k5[i]);
    F(t + h, yt, k6);
#pragma unroll
    for (int i = 0; i < N; i++)
        ynew[i] = y[i] + h * (a71 * k1[i] + a73 * k3[i] + a74 * k4[i] + a75 * k5[i] + a76 * k6[i]);
    F(t + h, ynew, k7);
#pragma unroll
    for (int i = 0; i < N; i++) {
        err[i] = h * (e1 * k1[i] + e3 * k3[i] + e4 * k4[i] + e5 * k5[i] + e6 * k6[i] + e7 * k7[i]);
        r4[i] = h * (d1 * k1[i] + d3 * k3[i] + d4 * k4[i] + d5 * k5[i] + d6 * k6[i] + d7 * k7[i]);
    }
}

// The six evaluations run in ONE loop that is not unrolled, so a kernel holds one inlined copy of its
// right-hand side (~10^3 FP64 instructions with two pow and a dozen exp for the detailed models)
// instead of seven: the seven-copy kernels needed 256 VGPRs, 260 - 480 spilled SGPRs and up to 500
// spilled VGPRs, and their results changed from build to build. The stage
// derivatives live in private memory (dynamic stage index).
// `hlambda` (optional; in: N component scales, out: hlambda[0]): h times an estimate of the dominant local
// eigenvalue, the test of Hairer's DOPRI5 code -- h |k7 - k6| / |ynew - g6| (g6 the argument of the sixth stage)
// in the scaled norm: a value above ~3.3, the stability bound of the pair on the real axis, on step after step
// means the steps are limited by stability, not accuracy.
template <int N, class RHS>
SONIC_HD void dopri5_step_looped(RHS &&F, double t, const double *y, const double *k1, double h,
                          double *ynew, double *k7, double *err, double *r4, double *hlambda)
{
    using namespace dp5;
    double k[7][N], yt[N], g6[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        k[0][i] = k1[i];
#pragma unroll
        for (int j = 1; j < 7; j++) k[j][i] = 0.0;
    }
#if defined(__clang__)
#pragma clang loop unroll(disable)
#endif
    for (int s = 0; s < 6; s++) {
        double a0, a1 = 0.0, a2 = 0.0, a3 = 0.0, a4 = 0.0, a5 = 0.0, cs;
        switch (s) {
        case 0: a0 = a21; cs = c2; break;
        case 1: a0 = a31; a1 = a32; cs = c3; break;
        case 2: a0 = a41; a1 = a42; a2 = a43; cs = c4; break;
        case 3: a0 = a51; a1 = a52; a2 = a53; a3 = a54; cs = c5; break;
        case 4: a0 = a61; a1 = a62; a2 = a63; a3 = a64; a4 = a65; cs = 1.0; break;
        default: a0 = a71; a2 = a73; a3 = a74; a4 = a75; a5 = a76; cs = 1.0; break;
        }
#pragma unroll
        for (int i = 0; i < N; i++)
            yt[i] = y[i] + h * (a0 * k[0][i] + a1 * k[1][i] + a2 * k[2][i] + a3 * k[3][i] +
                                a4 * k[4][i] + a5 * k[5][i]);
        if (s == 4) {
#pragma unroll
            for (int i = 0; i < N; i++) g6[i] = yt[i];
        }
        F(t + cs * h, yt, k[s + 1]);
    }
    if (hlambda) {
        // on entry *hlambda..hlambda[N-1] hold the scales of the components (the error norm's)
        double num = 0.0, den = 0.0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const double isc = 1.0 / hlambda[i];
            const double dk = (k[6][i] - k[5][i]) * isc, dyi = (yt[i] - g6[i]) * isc;
            num += dk * dk;
            den += dyi * dyi;
        }
        *hlambda = den > 0.0 ? h * sqrt(num / den) : 0.0;
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        ynew[i] = yt[i];           // the last stage point is the new solution (FSAL)
        k7[i] = k[6][i];
        err[i] = h * (e1 * k[0][i] + e3 * k[2][i] + e4 * k[3][i] + e5 * k[4][i] + e6 * k[5][i] + e7 * k[6][i]);
        r4[i] = h * (d1 * k[0][i] + d3 * k[2][i] + d4 * k[3][i] + d5 * k[4][i] + d6 * k[5][i] + d7 * k[6][i]);
    }
}

template <int N, class RHS>
SONIC_HD void dopri5_step(RHS &&F, double t, const double *y, const double *k1, double h,
                          double *ynew, double *k7, double *err, double *r4, double *hlambda = nullptr)
{
    if constexpr (N <= 3) dopri5_step_unrolled<N>(F, t, y, k1, h, ynew, k7, err, r4);
    else dopri5_step_looped<N>(F, t, y, k1, h, ynew, k7, err, r4, hlambda);
}

// component i of the continuous extension at t + s h
SONIC_HD double dopri5_dense(double yi, double ynewi, double k1i, double k7i, double r4i, double h,
                             double s)
{
    const double s1 = 1.0 - s;
    const double d = ynewi - yi;
    const double b = h * k1i - d;
    return yi + s * (d + s1 * (b + s * (d - h * k7i - b + s1 * r4i)));
}

// ---------------------------------------------------------------------------------------------
// Dormand-Prince 8(5,3) (Hairer's DOP853; coefficients in dop853_coeffs.hpp) for a system of N
// components held by one lane. K[i][s] = stage derivative s of component i (K[i][0] = f(t, y) on
// entry, first same as last). At the tolerances of the lookup generation it needs a third of the
// right-hand sides of the 5(4) pair for the same accuracy (RS, 500 kHz, 100 / 600 kPa: 3.0e4 / 1.3e5
// per acoustic period at rtol 1e-9 against 9.3e4 / 3.4e5 at rtol 1e-10, both 1 - 3e-10 of the
// deflection range from the converged solution).
// ---------------------------------------------------------------------------------------------
struct ScalarOps {
    typedef double V;
    SONIC_HD static V splat(double a) { return a; }
    SONIC_HD static V mul(V a, V b) { return a * b; }
    SONIC_HD static V fma_(V a, V b, V c) { return fma(a, b, c); }
};

template <int S, int N, class RHS>
SONIC_HD void dop853_stage(RHS &&F, double t, const double *y, double h, double cs, double (*K)[16])
{
    double yt[N], dy[N];
#pragma unroll
    for (int i = 0; i < N; i++) yt[i] = fma(h, dp8::stage_sum<S, ScalarOps>(K[i]), y[i]);
    F(t + cs * h, yt, dy);
#pragma unroll
    for (int i = 0; i < N; i++) K[i][S] = dy[i];
}

// one step attempt; returns the error norm of Hairer's code: err5^2 / sqrt(err5^2 + 0.01 err3^2), RMS
// over the components scaled by rtol max(|y|, |ynew|, floor)
template <int N, class RHS>
SONIC_HD double dop853_step(RHS &&F, double t, const double *y, double h, double (*K)[16], double *ynew,
                            double rtol, const double *floor_)
{
    dop853_stage<1, N>(F, t, y, h, dp8::c1, K);
    dop853_stage<2, N>(F, t, y, h, dp8::c2, K);
    dop853_stage<3, N>(F, t, y, h, dp8::c3, K);
    dop853_stage<4, N>(F, t, y, h, dp8::c4, K);
    dop853_stage<5, N>(F, t, y, h, dp8::c5, K);
    dop853_stage<6, N>(F, t, y, h, dp8::c6, K);
    dop853_stage<7, N>(F, t, y, h, dp8::c7, K);
    dop853_stage<8, N>(F, t, y, h, dp8::c8, K);
    dop853_stage<9, N>(F, t, y, h, dp8::c9, K);
    dop853_stage<10, N>(F, t, y, h, dp8::c10, K);
    dop853_stage<11, N>(F, t, y, h, dp8::c11, K);
    double dy[N];
#pragma unroll
    for (int i = 0; i < N; i++) ynew[i] = fma(h, dp8::b_sum<ScalarOps>(K[i]), y[i]);
    F(t + h, ynew, dy);
    double n5 = 0.0, n3 = 0.0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        K[i][12] = dy[i];
        const double sc = rtol * fmax(fmax(fabs(y[i]), fabs(ynew[i])), floor_[i]);
        const double r5 = qdiv(dp8::e5_sum<ScalarOps>(K[i]), sc), r3 = qdiv(dp8::e3_sum<ScalarOps>(K[i]), sc);
        n5 += r5 * r5;
        n3 += r3 * r3;
    }
    const double den = n5 + 0.01 * n3;
    if (!(den == den)) return NAN;
    return den > 0.0 ? fabs(h) * n5 / sqrt(den * N) : 0.0;
}

// the three extra stages and the coefficients Fc[i][0..6] of the 7th-order continuous extension
//   y(t + x h) = y + x (F0 + (1 - x) (F1 + x (F2 + (1 - x) (F3 + x (F4 + (1 - x) (F5 + x F6))))))
template <int N, class RHS>
SONIC_HD void dop853_dense_prepare(RHS &&F, double t, const double *y, const double *ynew, double h,
                                   double (*K)[16], double (*Fc)[7])
{
    dop853_stage<13, N>(F, t, y, h, dp8::c13, K);
    dop853_stage<14, N>(F, t, y, h, dp8::c14, K);
    dop853_stage<15, N>(F, t, y, h, dp8::c15, K);
#pragma unroll
    for (int i = 0; i < N; i++) {
        const double d = ynew[i] - y[i];
        Fc[i][0] = d;
        Fc[i][1] = h * K[i][0] - d;
        Fc[i][2] = 2.0 * d - h * (K[i][12] + K[i][0]);
        Fc[i][3] = h * dp8::d_sum<0, ScalarOps>(K[i]);
        Fc[i][4] = h * dp8::d_sum<1, ScalarOps>(K[i]);
        Fc[i][5] = h * dp8::d_sum<2, ScalarOps>(K[i]);
        Fc[i][6] = h * dp8::d_sum<3, ScalarOps>(K[i]);
    }
}
SONIC_HD double dop853_dense(const double *Fc, double yi, double x)
{
    const double x1 = 1.0 - x;
    double r = fma(x, Fc[6], Fc[5]);
    r = fma(x1, r, Fc[4]);
    r = fma(x, r, Fc[3]);
    r = fma(x1, r, Fc[2]);
    r = fma(x, r, Fc[1]);
    r = fma(x1, r, Fc[0]);
    return fma(x, r, yi);
}

// ---------------------------------------------------------------------------------------------
// True rate constants of each neuron in the reference's effRates() order
// (translators.py:287-327): x_inf / tau_x gates contribute alpha = xinf / tau,
// beta = (1 - xinf) / tau.
// ---------------------------------------------------------------------------------------------
SONIC_HD double vtrap(double x, double y) { return qdiv(x, fast_exp(qdiv(x, y)) - 1.0); }

SONIC_HD void put_inf_tau(double *out, int k, double inf, double tau)
{
    const double itau = qdiv(1.0, tau);
    out[k] = inf * itau;
    out[k + 1] = (1.0 - inf) * itau;
}

// m, h, n kinetics shared by the cortical and thalamic neurons (cortical.py:36-58)
SONIC_HD void hh_mhn_rates(double Vm, double VT, double *out)
{
    const double v = Vm - VT;
    out[0] = 0.32 * vtrap(13.0 - v, 4.0) * 1e3;
    out[1] = 0.28 * vtrap(v - 40.0, 5.0) * 1e3;
    out[2] = 0.128 * fast_exp(-(v - 17.0) * (1.0 / 18.0)) * 1e3;
    out[3] = qdiv(4.0, 1.0 + fast_exp(-(v - 40.0) * (1.0 / 5.0))) * 1e3;
    out[4] = 0.032 * vtrap(15.0 - v, 5.0) * 1e3;
    out[5] = 0.5 * fast_exp(-(v - 10.0) * (1.0 / 40.0)) * 1e3;
}

SONIC_HD void ctx_p_rates(double Vm, double TauMax, double *out, int k)
{
    const double pinf = qdiv(1.0, 1.0 + fast_exp(-(Vm + 35.0) * (1.0 / 10.0)));
    const double taup = qdiv(TauMax, 3.3 * fast_exp((Vm + 35.0) * (1.0 / 20.0)) + fast_exp(-(Vm + 35.0) * (1.0 / 20.0)));
    put_inf_tau(out, k, pinf, taup);
}

// T-type calcium gates of LTS / TC (cortical.py:254-272, thalamic.py:289-307)
SONIC_HD void lts_su_rates(double Vm, double Vx, double *out, int k)
{
    const double v = Vm + Vx;
    const double sinf = qdiv(1.0, 1.0 + fast_exp(-(v + 57.0) * (1.0 / 6.2)));
    const double xs = fast_exp(-(v + 132.0) * (1.0 / 16.7)) + fast_exp((v + 16.8) * (1.0 / 18.2));
    const double taus = 1.0 / 3.7 * (0.612 + qdiv(1.0, xs)) * 1e-3;
    const double uinf = qdiv(1.0, 1.0 + fast_exp((v + 81.0) * (1.0 / 4.0)));
    // both branches of the reference's piecewise tau_u evaluated, then selected: no divergent
    // control flow inside the right-hand side
    const double tu_lo = fast_exp((v + 467.0) * (1.0 / 66.6)), tu_hi = fast_exp(-(v + 22.0) * (1.0 / 10.5)) + 28.0;
    const double tauu = 1.0 / 3.7 * (v < -80.0 ? tu_lo : tu_hi) * 1e-3;
    put_inf_tau(out, k, sinf, taus);
    put_inf_tau(out, k + 2, uinf, tauu);
}

SONIC_HD double stn_xinf(double v, double theta, double k) { return qdiv(1.0, 1.0 + fast_exp(qdiv(v - theta, k))); }
SONIC_HD double stn_tau1(double V, double th, double sg, double t0, double t1)
{
    return t0 + qdiv(t1, 1.0 + fast_exp(-qdiv(V - th, sg)));
}
SONIC_HD double stn_tau2(double V, double th1, double th2, double s1, double s2, double t0, double t1)
{
    return t0 + qdiv(t1, fast_exp(-qdiv(V - th1, s1)) + fast_exp(-qdiv(V - th2, s2)));
}

// neuron_id as in include/pysonic_amd.h; returns the number of rates written
template <int NEURON>
struct NeuronRates;

template <>
struct NeuronRates<0> {   // RS (cortical.py:122-160)
    static constexpr int NR = 8;
    SONIC_HD static void eval(double Vm, double *out) { hh_mhn_rates(Vm, -56.2, out); ctx_p_rates(Vm, 0.608, out, 6); }
};
template <>
struct NeuronRates<1> {   // FS (cortical.py:163-201)
    static constexpr int NR = 8;
    SONIC_HD static void eval(double Vm, double *out) { hh_mhn_rates(Vm, -57.9, out); ctx_p_rates(Vm, 0.502, out, 6); }
};
template <>
struct NeuronRates<2> {   // LTS (cortical.py:204-303)
    static constexpr int NR = 12;
    SONIC_HD static void eval(double Vm, double *out)
    {
        hh_mhn_rates(Vm, -50.0, out);
        ctx_p_rates(Vm, 4.0, out, 6);
        lts_su_rates(Vm, -7.0, out, 8);
    }
};
template <>
struct NeuronRates<3> {   // RE (thalamic.py:117-179)
    static constexpr int NR = 10;
    SONIC_HD static void eval(double Vm, double *out)
    {
        hh_mhn_rates(Vm, -67.0, out);
        const double sinf = qdiv(1.0, 1.0 + fast_exp(-(Vm + 52.0) * (1.0 / 7.4)));
        const double taus = (1.0 + qdiv(0.33, fast_exp((Vm + 27.0) * (1.0 / 10.0)) + fast_exp(-(Vm + 102.0) * (1.0 / 15.0)))) * 1e-3;
        const double uinf = qdiv(1.0, 1.0 + fast_exp((Vm + 80.0) * (1.0 / 5.0)));
        const double tauu = (28.3 + qdiv(0.33, fast_exp((Vm + 48.0) * (1.0 / 4.0)) + fast_exp(-(Vm + 407.0) * (1.0 / 50.0)))) * 1e-3;
        put_inf_tau(out, 6, sinf, taus);
        put_inf_tau(out, 8, uinf, tauu);
    }
};
template <>
struct NeuronRates<4> {   // TC (thalamic.py:182-323)
    static constexpr int NR = 12;
    SONIC_HD static void eval(double Vm, double *out)
    {
        hh_mhn_rates(Vm, -52.0, out);
        lts_su_rates(Vm, 0.0, out, 6);
        const double oinf = qdiv(1.0, 1.0 + fast_exp((Vm + 75.0) * (1.0 / 5.5)));
        const double tauo = qdiv(1.0, fast_exp(-14.59 - 0.086 * Vm) + fast_exp(-1.87 + 0.0701 * Vm)) * 1e-3;
        put_inf_tau(out, 10, oinf, tauo);
    }
};
template <>
struct NeuronRates<6> {   // IB (cortical.py:307-400): RS kinetics + alpha / beta gates q, r of iCaL
    static constexpr int NR = 12;
    SONIC_HD static void eval(double Vm, double *out)
    {
        hh_mhn_rates(Vm, -56.2, out);
        ctx_p_rates(Vm, 0.608, out, 6);
        out[8] = 0.055 * vtrap(-(Vm + 27.0), 3.8) * 1e3;
        out[9] = 0.94 * fast_exp(-(Vm + 75.0) / 17.0) * 1e3;
        out[10] = 0.000457 * fast_exp(-(Vm + 13.0) / 50.0) * 1e3;
        out[11] = 0.0065 / (fast_exp(-(Vm + 15.0) / 28.0) + 1.0) * 1e3;
    }
};
template <>
struct NeuronRates<7> {   // HHseg (hh.py:44-86), q10 = 3^((36 - 6.3) / 10)
    static constexpr int NR = 6;
    SONIC_HD static void eval(double Vm, double *out)
    {
        const double q10 = 26.1246286895632;
        out[0] = q10 * 0.1 * vtrap(-(Vm + 40.0), 10.0) * 1e3;
        out[1] = q10 * 4.0 * fast_exp(-(Vm + 65.0) / 18.0) * 1e3;
        out[2] = q10 * 0.07 * fast_exp(-(Vm + 65.0) / 20.0) * 1e3;
        out[3] = q10 * 1.0 / (fast_exp(-(Vm + 35.0) / 10.0) + 1.0) * 1e3;
        out[4] = q10 * 0.01 * vtrap(-(Vm + 55.0), 10.0) * 1e3;
        out[5] = q10 * 0.125 * fast_exp(-(Vm + 65.0) / 80.0) * 1e3;
    }
};
template <>
struct NeuronRates<8> {   // SWnode (sweeney.py:41-60)
    static constexpr int NR = 4;
    SONIC_HD static void eval(double Vm, double *out)
    {
        const double am = (126.0 + 0.363 * Vm) / (1.0 + fast_exp(-(Vm + 49.0) / 5.3)) * 1e3;
        const double bh = 15.6 / (1.0 + fast_exp(-(Vm + 56.0) / 10.0)) * 1e3;
        out[0] = am;
        out[1] = am / fast_exp((Vm + 56.2) / 4.17);
        out[2] = bh / fast_exp((Vm + 74.5) / 5.0);
        out[3] = bh;
    }
};
template <>
struct NeuronRates<9> {   // MRGnode (mrg.py:60-108): q10 = 2.2^1.6, 2.9^1.6, 3^0; m / h shifted by 3 mV
    static constexpr int NR = 8;
    SONIC_HD static void eval(double Vm, double *out)
    {
        const double q_mp = 3.530825783474764, q_h = 5.493344008948558, Vs = Vm + 3.0, Vt = Vm + 80.0;
        out[0] = q_mp * 1.86 * vtrap(-(Vs + 18.4), 10.3) * 1e3;
        out[1] = q_mp * 0.086 * vtrap(Vs + 22.7, 9.16) * 1e3;
        out[2] = q_h * 0.062 * vtrap(Vs + 111.0, 11.0) * 1e3;
        out[3] = q_h * 2.3 / (1.0 + fast_exp(-(Vs + 28.8) / 13.4)) * 1e3;
        out[4] = q_mp * 0.01 * vtrap(-(Vm + 27.0), 10.2) * 1e3;
        out[5] = q_mp * 0.00025 * vtrap(Vm + 34.0, 10.0) * 1e3;
        out[6] = 0.3 / (1.0 + fast_exp(-(Vt - 27.0) / 5.0)) * 1e3;
        out[7] = 0.03 / (1.0 + fast_exp(-(Vt + 10.0) / 1.0)) * 1e3;
    }
};
template <>
struct NeuronRates<10> {   // SUseg (sundt.py:70-117): Traub sodium gates (q10 = 3^0.6, shifts -6 / +6 mV from
                           // a -65 mV rest), Borg-Graham potassium gates with x = (Vm - Vref) F / (R T) 1e-3
    static constexpr int NR = 8;
    SONIC_HD static void eval(double Vm, double *out)
    {
        const double q10 = 1.9331820449317627, k = 0.037541548196719836;   // F / (Rg T) 1e-3, T = 309.15 K
        const double vm = Vm + 65.0 - 6.0, vh = Vm + 65.0 + 6.0;
        out[0] = q10 * 0.32 * vtrap(13.1 - vm, 4.0) * 1e3;
        out[1] = q10 * 0.28 * vtrap(vm - 40.1, 5.0) * 1e3;
        out[2] = q10 * 0.128 * fast_exp((17.0 - vh) / 18.0) * 1e3;
        out[3] = q10 * 4.0 / (1.0 + fast_exp((40.0 - vh) / 5.0)) * 1e3;
        const double xn = (Vm + 32.0) * k, xl = (Vm + 61.0) * k;
        out[4] = q10 * 0.03 * fast_exp(5.0 * 0.4 * xn) * 1e3;        // alphaBG(0.03, -5, 0.4, -32)
        out[5] = q10 * 0.03 * fast_exp(-5.0 * 0.6 * xn) * 1e3;       // betaBG
        out[6] = q10 * 0.001 * fast_exp(-2.0 * 1.0 * xl) * 1e3;      // alphaBG(0.001, 2, 1, -61)
        out[7] = q10 * 0.001 * fast_exp(2.0 * 0.0 * xl) * 1e3;       // betaBG: (1 - gamma) = 0
    }
};
template <>
struct NeuronRates<11> {   // FHnode (fh.py:61-98): q10 = 3^1.6, voltages relative to the -70 mV rest
    static constexpr int NR = 8;
    SONIC_HD static void eval(double Vm, double *out)
    {
        const double q10 = 5.799546134795289, v = Vm + 70.0;
        out[0] = q10 * 0.36 * vtrap(22.0 - v, 3.0) * 1e3;
        out[1] = q10 * 0.4 * vtrap(v - 13.0, 20.0) * 1e3;
        out[2] = q10 * 0.1 * vtrap(v + 10.0, 6.0) * 1e3;
        out[3] = q10 * 4.5 / (fast_exp((45.0 - v) / 10.0) + 1.0) * 1e3;
        out[4] = q10 * 0.02 * vtrap(35.0 - v, 10.0) * 1e3;
        out[5] = q10 * 0.05 * vtrap(v - 10.0, 10.0) * 1e3;
        out[6] = q10 * 0.006 * vtrap(40.0 - v, 10.0) * 1e3;
        out[7] = q10 * 0.09 * vtrap(v + 25.0, 20.0) * 1e3;
    }
};
template <>
struct NeuronRates<12> {   // passive neuron (pas.py): no gate; the padding gate of the device model
    static constexpr int NR = 2;
    SONIC_HD static void eval(double, double *out) { out[0] = 0.0; out[1] = 0.0; }
};
template <>
struct NeuronRates<5> {   // STN (stn.py:52-136, 209-338): order a b c d1 m h n p q
    static constexpr int NR = 18;
    SONIC_HD static void eval(double V, double *out)
    {
        put_inf_tau(out, 0, stn_xinf(V, -45, -14.7), stn_tau1(V, -40, -0.5, 1e-3, 1e-3));
        put_inf_tau(out, 2, stn_xinf(V, -90, 7.5), stn_tau2(V, -60, -40, -30, 10, 0e-3, 200e-3));
        put_inf_tau(out, 4, stn_xinf(V, -30.6, -5), stn_tau2(V, -27, -50, -20, 15, 45e-3, 10e-3));
        put_inf_tau(out, 6, stn_xinf(V, -60, 7.5), stn_tau2(V, -40, -20, -15, 20, 400e-3, 500e-3));
        put_inf_tau(out, 8, stn_xinf(V, -40, -8), stn_tau1(V, -53, -0.7, 0.2e-3, 3e-3));
        put_inf_tau(out, 10, stn_xinf(V, -45.5, 6.4), stn_tau2(V, -50, -50, -15, 16, 0e-3, 24.5e-3));
        put_inf_tau(out, 12, stn_xinf(V, -41, -14), stn_tau2(V, -40, -40, -40, 50, 0e-3, 11e-3));
        put_inf_tau(out, 14, stn_xinf(V, -56, -6.7), stn_tau2(V, -27, -102, -10, 15, 5e-3, 0.33e-3));
        put_inf_tau(out, 16, stn_xinf(V, -85, 5.8), stn_tau2(V, -50, -50, -15, 16, 0e-3, 400e-3));
    }
};

// ---------------------------------------------------------------------------------------------
// computeEffVars for one (f, A, Qm) cell -- nbls.py:153-222 on top of simCycles
// (bls.py:749-789) and PeriodicSolver.solve (solvers.py:336-365):
//   * y0 = (0, Z_qs, ng0) with Z_qs the quasi-static deflection at Pac(t = dt) (bls.py:720-747)
//   * cycles of length T = 1/f, sampled at t_c + k T / 999, k = 1..999 (np.linspace with 1000
//     points, first point dropped: solvers.py:166-169,332-334)
//   * after >= 2 cycles stop when rmse(last, previous) / ptp(last) < 1e-4 for Z and ng, or after
//     11 cycles (loop counter quirk; A = 0 gives 0/0 = NaN and always runs 11)
//   * effective variables = means over the LAST 1000 ROWS (.tail(1000): the 999 samples of the
//     last cycle preceded by the last sample of the cycle before) of Vm = Qm / Cm_eff(Z) * 1e3 and
//     of every rate(Vm).
// `zs`, `ngs`: per-cell scratch of NPC-1 samples each, strided by `stride` (coalesced across
// lanes). Returns the number of cycles; status bit 1: Z clamped, 2: no sign change for Z_qs,
// 4: step budget exhausted, 8: not converged after 11 cycles (reference logs a warning).
// ---------------------------------------------------------------------------------------------
constexpr int MECH_NPC = 1000;

struct MechOpts {
    double rtol;          // relative tolerance of the DOPRI5 controller
    int max_steps;        // per-cell budget of step attempts
    int nmax_cycles;      // NCYCLES_MAX (constants.py:34) -> cap is nmax + 1 cycles in total
};

// Fourier overtones of the imposed charge (nbls.py:169-178): Qm(t) is the reference's profile of
// MECH_NPC samples over the acoustic period, irfft([Qm0, A_i fast_exp(j phi_i)], n) * n, i.e.
// Qm_k = Qm0 + 2 sum_i A_i cos(2 pi i k / n + phi_i), held constant over [k dt, (k + 1) dt)
// (bls.py:767-769: Qm[int((t % T) / dt)]). `out`: [n_fs][2 n] amplitude and phase of the first n
// Fourier coefficients of Vm over the last cycle (nbls.py:194-201).
struct MechOvertones {
    int n;
    const double *A, *phi;
    double *out;
};

SONIC_HD double mech_charge_sample(double Qm0, const MechOvertones &ov, int k)
{
    double q = Qm0;
    for (int i = 0; i < ov.n; i++)
        q += 2.0 * ov.A[i] * cos(2.0 * bls::PI * (double)((i + 1) * k) / (double)MECH_NPC + ov.phi[i]);
    return q;
}

template <int NEURON>
SONIC_HD int mech_cell(const BLSParams &p, double f, double A, double phi, double Qm0,
                       const double *fs, int n_fs, const MechOpts &o, double *zs, double *ngs,
                       long stride, double *effvars /* [n_fs][1 + NR] */, int *status_out,
                       const MechOvertones ov = MechOvertones{0, nullptr, nullptr, nullptr})
{
    constexpr int NR = NeuronRates<NEURON>::NR;
    constexpr int NS = MECH_NPC - 1;                 // samples per cycle
    int status = 0;
    const double Tper = 1.0 / f;
    const double dt = 1.0 / (MECH_NPC * f);          // drives.py:276-279
    const MechDrive d{2.0 * bls::PI * f, A, phi};
    bool clamped = false;

    // initial conditions (with overtones: from the first sample of the profile, bls.py:769)
    double Qm = mech_charge_sample(Qm0, ov, 0);
    const double Pac_dt = A * sin(d.w * dt - phi);
    const double Zqs = bls_balancedefQS(p, p.ng0, Qm, Pac_dt);
    if (!(Zqs == Zqs)) {
        for (int i = 0; i < n_fs * (1 + NR); i++) effvars[i] = NAN;
        *status_out = 2;
        return 0;
    }
    double y[3] = {0.0, Zqs, p.ng0};
    bool trial_clamped = false;     // clamp seen by the stages of the current step attempt
    auto F = [&](double t, const double *yy, double *dy) { bls_rhs(p, d, t, yy, Qm, dy, trial_clamped); };

    // absolute error floors: variables smaller than these are controlled absolutely
    const double floor_[3] = {1e-6, 1e-13, 1e-25};
    // Dormand-Prince 8(5,3): K[i][s] = stage derivative s of component i, K[i][0] = f(t, y)
    double K[3][16], ynew[3], k1[3], Fc[3][7];
    double t = 0.0, h = dt;
    F(t, y, k1);
#pragma unroll
    for (int i = 0; i < 3; i++) K[i][0] = k1[i];
    int nsteps = 0, ncycles = 0;
    double z_last_start = Zqs;     // Z at the start of the last cycle run (= row before its samples)
    bool converged = false;

    for (int cyc = 0; cyc <= o.nmax_cycles && !converged; cyc++) {
        const double t0c = t, t1c = t + Tper;
        const double step = (t1c - t0c) / (double)NS;         // np.linspace(t0, t0 + T, 1000)
        // the samples are interpolated with the continuous extension, which the error estimate does not
        // control: no step longer than two sample intervals (see COOP_HMAX_DENSE in full_coop.hpp)
        const double hmax = 2.0 * step;
        int ks = 1;                                           // next sample index (1..NS)
        double sse_z = 0.0, sse_n = 0.0, zmin = INFINITY, zmax = -INFINITY, nmin = INFINITY,
               nmax = -INFINITY;
        z_last_start = y[1];
        int kq = 0;                                           // index of the current charge sample
        if (ov.n > 0) {
            Qm = mech_charge_sample(Qm0, ov, 0);
            F(t, y, k1);
#pragma unroll
            for (int i = 0; i < 3; i++) K[i][0] = k1[i];
        }
        while (ks <= NS) {
            bool last = false, lastq = false;
            h = fmin(h, hmax);
            const double hwant = h;
            if (ov.n > 0 && kq < MECH_NPC - 1) {
                // the charge is piecewise constant: steps end on its discontinuities
                const double tb = t0c + (double)(kq + 1) * dt;
                if (t + 1.0001 * h >= tb) { h = tb - t; lastq = true; }
            }
            if (!lastq && t + 1.0001 * h >= t1c) { h = t1c - t; last = true; }
            trial_clamped = false;
            const double en = dop853_step<3>(F, t, y, h, K, ynew, o.rtol, floor_);
            nsteps++;
            // standard controller: h_new = h * min(6, max(0.2, 0.9 * en^(-1/8)))
            double fac = 0.9 * fast_exp(-0.125 * fast_log(fmax(en, 1e-12)));
            fac = fmin(6.0, fmax(0.2, fac));
            if (!(en == en)) fac = 0.2;
            if (en <= 1.0) {
                clamped = clamped || trial_clamped;
                const double tnew = last ? t1c : (lastq ? t0c + (double)(kq + 1) * dt : t + h);
                // samples inside (t, tnew]: the three extra stages of the continuous extension only then
                const double ts0 = (ks == NS) ? t1c : t0c + (double)ks * step;
                if (last || ts0 <= tnew) {
                    bool prepared = false;
                    while (ks <= NS) {
                        const double ts = (ks == NS) ? t1c : t0c + (double)ks * step;
                        if (!last && ts > tnew) break;
                        double zv, nv;
                        if (ts >= tnew) { zv = ynew[1]; nv = ynew[2]; }
                        else {
                            if (!prepared) { dop853_dense_prepare<3>(F, t, y, ynew, h, K, Fc); prepared = true; }
                            const double sg = (ts - t) / h;
                            zv = dop853_dense(Fc[1], y[1], sg);
                            nv = dop853_dense(Fc[2], y[2], sg);
                        }
                        const long idx = (long)(ks - 1) * stride;
                        if (cyc > 0) {
                            const double dz = zv - zs[idx], dn = nv - ngs[idx];
                            sse_z += dz * dz;
                            sse_n += dn * dn;
                        }
                        zs[idx] = zv;
                        ngs[idx] = nv;
                        zmin = fmin(zmin, zv); zmax = fmax(zmax, zv);
                        nmin = fmin(nmin, nv); nmax = fmax(nmax, nv);
                        ks++;
                    }
                }
#pragma unroll
                for (int i = 0; i < 3; i++) { y[i] = ynew[i]; K[i][0] = K[i][12]; }
                t = tnew;
                h = h * fac;
                if (lastq) {
                    // next charge sample: the derivative at the new point changes with it
                    kq++;
                    Qm = mech_charge_sample(Qm0, ov, kq);
                    F(t, y, k1);
#pragma unroll
                    for (int i = 0; i < 3; i++) K[i][0] = k1[i];
                    h = fmax(h, hwant);
                }
            } else {
                h = h * fmin(fac, 1.0);
            }
            if (nsteps >= o.max_steps || !(h > 1e-18)) { status |= 4; ks = NS + 1; cyc = o.nmax_cycles + 1; }
        }
        t = t1c;
        ncycles++;
        if (cyc >= 1 && !(status & 4)) {
            // isPeriodicallyStable (solvers.py:317-330): rmse / ptp < MAX_RMSE_PTP_RATIO
            const double rz = sqrt(sse_z / NS) / (zmax - zmin);
            const double rn = sqrt(sse_n / NS) / (nmax - nmin);
            converged = (rz < 1e-4) && (rn < 1e-4);
#ifdef MECH_DEBUG
            printf("cyc %d rz %.6e rn %.6e ptpz %.3e ptpn %.3e nsteps %d\n", cyc, rz, rn, zmax - zmin, nmax - nmin, nsteps);
#endif
        }
    }
    if (!converged) status |= 8;
    if (clamped) status |= 1;

    // cycle averages over the last 1000 rows
    for (int j = 0; j < n_fs; j++) {
        const double fsj = fs[j];
        double sumV = 0.0, sumR[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) sumR[r] = 0.0;
        for (int i = 0; i < 2 * ov.n; i++) ov.out[(long)j * 2 * ov.n + i] = 0.0;
        for (int ks = 0; ks <= NS; ks++) {
            const double zv = (ks == 0) ? z_last_start : zs[(long)(ks - 1) * stride];
            const double Cm = bls_capacitance(p, zv);
            // row ks of the last 1000 rows pairs with sample ks of the charge profile (nbls.py:181-188)
            const double Qk = mech_charge_sample(Qm0, ov, ks);
            const double Vm = Qk / (fsj * Cm + (1.0 - fsj) * p.Cm0) * 1e3;     // nbls.py:148-151,188
            for (int i = 0; i < ov.n; i++) {      // rfft(Vm)[i + 1] / n, accumulated (re, -im)
                const double ang = 2.0 * bls::PI * (double)((i + 1) * ks) / (double)MECH_NPC;
                ov.out[(long)j * 2 * ov.n + 2 * i] += Vm * cos(ang);
                ov.out[(long)j * 2 * ov.n + 2 * i + 1] -= Vm * sin(ang);
            }
            double rates[NR];
            NeuronRates<NEURON>::eval(Vm, rates);
            sumV += Vm;
#pragma unroll
            for (int r = 0; r < NR; r++) sumR[r] += rates[r];
        }
        for (int i = 0; i < ov.n; i++) {          // amplitude-phase form (nbls.py:197-201)
            double *c = ov.out + (long)j * 2 * ov.n + 2 * i;
            const double re = c[0] * (1.0 / MECH_NPC), im = c[1] * (1.0 / MECH_NPC);
            c[0] = sqrt(re * re + im * im);
            c[1] = atan2(im, re);
        }
        double *ev = effvars + (long)j * (1 + NR);
        ev[0] = sumV * (1.0 / MECH_NPC);
#pragma unroll
        for (int r = 0; r < NR; r++) ev[1 + r] = sumR[r] * (1.0 / MECH_NPC);
    }
    *status_out = status;
    return ncycles;
}

}  // namespace sonic
