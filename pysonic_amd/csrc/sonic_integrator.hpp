// pysonic_amd/csrc/sonic_integrator.hpp
//
// Per-configuration integrator of the SONIC effective system  y' = f(y; level tables)  used by
// the HIP kernel (one stimulus configuration per lane) -- the device-side replacement for the
// reference's  EventDrivenSolver + scipy odeint(LSODA)  loop (PySONIC/core/solvers.py:150-170,
// 445-480) applied to NeuronalBilayerSonophore.effDerivatives (PySONIC/core/nbls.py:280-315).
//
// Method: RODAS4 (Hairer & Wanner, "Solving ODEs II", sec. IV.7/IV.10): 6-stage, order 4(3),
// stiffly accurate, L-stable Rosenbrock method with 3rd-order dense output. No Newton iteration
// and no data-dependent inner loop, so the 64 lanes of a wavefront stay in lockstep; the only
// per-lane divergence is step acceptance (predicated) and the lookup-cell reload.
//   * Linear algebra: the Jacobian is "arrow + small dense core" (sonic_models.hpp), so
//     (I/(h gamma) - J) k = r  is solved in O(NG + NC^3) with everything in registers.
//   * Lookup tables: effective coefficients are piecewise linear in Q on the reference's charge
//     grid (np.interp semantics, PySONIC/core/lookups.py:309-322). Each lane caches the record
//     of the grid cell it is in (left node value + slope for every table) in registers and
//     reloads it from the level table only when Q leaves the cell.
//   * Output: rows are produced on the reference's per-segment np.linspace grid
//     (solvers.py:77-97) with the method's dense output, so step sizes are not tied to the
//     50 us output step; steps are clipped at segment (event) boundaries only.
//
// This header is compiled by hipcc for the device and by g++ for the CPU test harness
// (tests/native/) -- the harness exists for development and sanitizer runs; the product never
// falls back to it.
#pragma once
#include <math.h>
#include "sonic_models.hpp"

#ifndef SONIC_HOME_CELL
#define SONIC_HOME_CELL 1
#endif
#ifndef SONIC_OV_TARGET
#define SONIC_OV_TARGET 0.005   // aim this fraction of a cell width past the node
#endif
#ifndef SONIC_OV_MAX
#define SONIC_OV_MAX 0.03       // reject steps that end further than this outside the home cell
#endif
// The same two fractions for the lane-per-configuration kernels (LTS, RE, TC, STN ...). The sliver of a
// step that lies past the node is integrated with the home cell's lines, an error NO tolerance
// controls; the bursting / rebounding neurons amplify it (RE golden c0: 2.9e-7 C/m2 RMS over the part of
// the trace where the reference still agrees with itself, at any rtol from 1e-6 to 1e-8; 1.0e-7 with
// these values, LTS / TC 5 - 10 x closer as well) for ~25 % more steps.
#ifndef SONIC_LANE_OV_TARGET
#define SONIC_LANE_OV_TARGET 0.002
#endif
#ifndef SONIC_LANE_OV_MAX
#define SONIC_LANE_OV_MAX 0.01
#endif

namespace sonic {

// Status bits reported per configuration (mirrors of reference behaviours)
enum : int {
    ST_OK = 0,
    ST_Q_OUT_OF_RANGE = 1,   // Qm left the lookup's charge range -> NaN rows (lookups.py:322)
    ST_STEP_UNDERFLOW = 2,   // step size underflow / non-finite state
    ST_MAX_STEPS = 4,        // step budget exhausted
};

// What set the steps of a configuration (metric columns SONIC_M_NCAPPED .. SONIC_M_NCROSS): accepted steps whose
// size was the node predictor's, not the error controller's; attempts rejected because they ended too far past
// the node (not by the error estimate: those are NREJ minus these); table cells crossed.
struct StepCounts {
    int capped = 0, over = 0, cross = 0;
};

struct SolverOpts {
    double rtol;        // relative tolerance
    double atol;        // absolute tolerance (same for every component, like odeint's scalar atol)
    double h0;          // initial step at the start of every segment (s)
    double hmin;        // step underflow threshold (s). Far below any physical time scale on purpose:
                        // when the amplitude switches, gates whose effective rates reach 1e20 /s (STN
                        // `b`, `h`, `q` above ~450 kPa) relax within 1e-20 s, and like LSODA (hmin = 0)
                        // the controller walks down to that scale and back up in ~50 steps. Steps
                        // below ulp(t) leave t unchanged and still advance y; they sum to < ulp(t).
    int max_steps;      // per-configuration budget of step attempts
    int qss_gates;      // bit i: device gate i is a quasi-steady-state variable (0: none)
    double qdrive;      // constant added to dQm/dt: Idrive 1e-3 of DrivenNeuronalBilayerSonophore
                        // (nbls.py:717-721), 0 otherwise
};

// A segment shorter than this is not integrated: its rows repeat the state (odeint: "tout too close
// to t to start integration"), e.g. a progress-log event one ulp away from a stimulus event.
constexpr double SONIC_SEG_EPS = 1e-14;

// Time for the charge to travel `dist` (signed like its velocity fq) to the table node it is heading to, from the
// second-order expansion q(t + h) ~ q + fq h + (fp / 2) h^2 with fp = (J f)_Q, both known at the start of a step:
// the root of the quadratic that continues the linear estimate dist / fq,  h = 2 dist / (fq + sign(fq) sqrt(fq^2 +
// 2 fp dist)). The linear estimate alone misses by the curvature: on the upstroke of a spike the step overshoots the
// node (rejected, retried with a secant), on the way down it stops short (one more tiny step, and the 6x growth
// limit makes the next one small too) -- four attempts per table cell where this takes one (RS, 600 kPa, loose
// tolerance: 10 837 -> 5 680 attempts for 3 816 cells). Single precision: it is only a proposal. A charge that the
// quadratic never brings to the node (discriminant < 0: it turns round first) gets 4/3 of the linear estimate.
SONIC_HD float node_time_denominator(float fq, float fp, float dist, float root)
{
    (void)fp; (void)dist;
    return fq + (fq > 0.0f ? root : -root);
}
SONIC_HD float node_time_discriminant(float fq, float fp, float dist)
{
    return fmaxf(fq * fq + 2.0f * fp * dist, 0.25f * fq * fq);
}

// RODAS4 coefficients (Hairer & Wanner, RODAS code, method 1)
namespace rodas4 {
constexpr double gamma = 0.25;
constexpr double a21 = 0.1544000000000000e+01;
constexpr double a31 = 0.9466785280815826e+00;
constexpr double a32 = 0.2557011698983284e+00;
constexpr double a41 = 0.3314825187068521e+01;
constexpr double a42 = 0.2896124015972201e+01;
constexpr double a43 = 0.9986419139977817e+00;
constexpr double a51 = 0.1221224509226641e+01;
constexpr double a52 = 0.6019134481288629e+01;
constexpr double a53 = 0.1253708332932087e+02;
constexpr double a54 = -0.6878860361058950e+00;
constexpr double c21 = -0.5668800000000000e+01;
constexpr double c31 = -0.2430093356833875e+01;
constexpr double c32 = -0.2063599157091915e+00;
constexpr double c41 = -0.1073529058151375e+00;
constexpr double c42 = -0.9594562251023355e+01;
constexpr double c43 = -0.2047028614809616e+02;
constexpr double c51 = 0.7496443313967647e+01;
constexpr double c52 = -0.1024680431464352e+02;
constexpr double c53 = -0.3399990352819905e+02;
constexpr double c54 = 0.1170890893206160e+02;
constexpr double c61 = 0.8083246795921522e+01;
constexpr double c62 = -0.7981132988064893e+01;
constexpr double c63 = -0.3152159432874371e+02;
constexpr double c64 = 0.1631930543123136e+02;
constexpr double c65 = -0.6058818238834054e+01;
constexpr double d21 = 0.1012623508344586e+02;
constexpr double d22 = -0.7487995877610167e+01;
constexpr double d23 = -0.3480091861555747e+02;
constexpr double d24 = -0.7992771707568823e+01;
constexpr double d25 = 0.1025137723295662e+01;
constexpr double d31 = -0.6762803392801253e+00;
constexpr double d32 = 0.6087714651680015e+01;
constexpr double d33 = 0.1643084320892478e+02;
constexpr double d34 = 0.2476722511418386e+02;
constexpr double d35 = -0.6594389125716872e+01;
}  // namespace rodas4

// ---------------------------------------------------------------------------------------------
// Level tables: for every distinct drive amplitude ("level") of a batch the host projects the
// 2-D (A, Q) lookup at that amplitude exactly as the reference does (Lookup.project,
// lookups.py:230-271) and packs one record per charge-grid cell j = [Q_j, Q_{j+1}):
//     rec = { Q_j, Q_{j+1}, (fp_k[j], slope_k[j]) for k in 0..NT-1 },
//     slope_k[j] = (fp_k[j+1] - fp_k[j]) / (Q_{j+1} - Q_j)        (np.interp's slope)
// so that  table_k(Q) = slope_k[j] * (Q - Q_j) + fp_k[j]  is np.interp's own expression.
// ---------------------------------------------------------------------------------------------
template <int NT, bool LDS = false>
struct CellRec {
    double xlo, xhi;
    double v[NT];
    double s[NT];
};

// The same record with its 2 NT values and slopes in LDS (lane-interleaved: entry k of lane l at
// base[k * 64 + l], conflict-free) instead of registers: for the models whose per-lane state does
// not fit the register file (TC: 13 tables, STN: 19 tables -> 54 / 78 VGPRs for the home cell).
struct LdsLaneArray {
    double *p;
    SONIC_HD double &operator[](int k) const { return p[k * 64]; }
};
template <int NT>
struct CellRec<NT, true> {
    double xlo, xhi;
    LdsLaneArray v, s;
};

template <int NT>
constexpr int cell_rec_doubles() { return 2 + 2 * NT; }

struct LevelGrid {
    const double *recs;   // [n_levels][n_cells][2 + 2 NT]
    int n_cells;
    double q0;            // Q_0
    double qmax;          // Q_{n_cells}
    double inv_dq;        // 1 / nominal grid step
};

template <int NT, class C>
SONIC_HD void load_cell(const LevelGrid &G, int level, int j, C &c)
{
    const double *r = G.recs + ((size_t)level * G.n_cells + j) * cell_rec_doubles<NT>();
    c.xlo = r[0];
    c.xhi = r[1];
#pragma unroll
    for (int k = 0; k < NT; k++) {
        c.v[k] = r[2 + 2 * k];
        c.s[k] = r[3 + 2 * k];
    }
}

// Make `c` the record of the cell containing q and return its index; returns -1 if q is outside
// the charge range (np.interp(..., left=nan, right=nan)) or not finite.
template <int NT, class C>
SONIC_HD int locate_cell(const LevelGrid &G, int level, double q, C &c)
{
    if (!(q >= G.q0 && q <= G.qmax)) return -1;
    int j = (int)((q - G.q0) * G.inv_dq);
    if (j < 0) j = 0;
    if (j > G.n_cells - 1) j = G.n_cells - 1;
    load_cell<NT>(G, level, j, c);
    // the grid is np.arange-generated: the guess can be off by one cell at most
    if (q < c.xlo && j > 0) {
        j -= 1;
        load_cell<NT>(G, level, j, c);
    } else if (q >= c.xhi && j < G.n_cells - 1) {
        j += 1;
        load_cell<NT>(G, level, j, c);
    }
    return j;
}


// Quasi-steady-state gates (qss_vars of NBLS.effDerivatives, nbls.py:296-303): gate i is not
// integrated but set to x_inf = alpha_i / (alpha_i + beta_i) with the rates interpolated at the
// current charge. In the arrow structure it becomes a function of Q: its column of the core rows
// folds into the Q column (d x_inf / dQ), its own row and column vanish.
template <class M>
SONIC_HD void qss_substitute(int qss, const double *lk, double *yq)
{
#pragma unroll
    for (int i = 0; i < M::NG; i++)
        if (qss & (1 << i)) yq[M::NC + i] = lk[1 + 2 * i] / (lk[1 + 2 * i] + lk[2 + 2 * i]);
}

template <class M>
SONIC_HD void qss_fold(int qss, const double *lk, const double *dlk, double *f,
                       Jac<M::NC, M::NG> *J)
{
#pragma unroll
    for (int i = 0; i < M::NG; i++) {
        if (!(qss & (1 << i))) continue;
        f[M::NC + i] = 0.0;
        if (J) {
            const double a = lk[1 + 2 * i], b = lk[2 + 2 * i], da = dlk[1 + 2 * i], db = dlk[2 + 2 * i];
            const double r = 1.0 / (a + b);
            const double dxdq = (da * b - a * db) * r * r;
#pragma unroll
            for (int c = 0; c < M::NC; c++) {
                J->Jcc[c][0] += J->Jcg[c][i] * dxdq;
                J->Jcg[c][i] = 0.0;
            }
            J->Jgq[i] = 0.0;
            J->Dg[i] = 0.0;
        }
    }
}

// f(y) with the lookup lines of `cell`, whether or not y[0] lies inside it (home-cell stepping)
template <class M, class C>
SONIC_HD void eval_home(const typename M::Params &P, const C &cell, const double *y,
                        double *f, int qss = 0, double qdrive = 0.0)
{
    double lk[M::NT];
    const double dq = y[0] - cell.xlo;
#pragma unroll
    for (int k = 0; k < M::NT; k++) lk[k] = cell.s[k] * dq + cell.v[k];
    if (qss) {
        double yq[M::NY];
#pragma unroll
        for (int i = 0; i < M::NY; i++) yq[i] = y[i];
        qss_substitute<M>(qss, lk, yq);
        M::template eval<false>(P, lk, nullptr, yq, f, nullptr);      // slopes: Jacobian only
        qss_fold<M>(qss, lk, nullptr, f, nullptr);
        f[0] += qdrive;
        return;
    }
    M::template eval<false>(P, lk, nullptr, y, f, nullptr);
    f[0] += qdrive;
}

template <class M, class C>
SONIC_HD void eval_home_jac(const typename M::Params &P, const C &cell,
                            const double *y, double *f, Jac<M::NC, M::NG> &J, int qss = 0,
                            double qdrive = 0.0)
{
    double lk[M::NT], dlk[M::NT];
    const double dq = y[0] - cell.xlo;
#pragma unroll
    for (int k = 0; k < M::NT; k++) {
        dlk[k] = cell.s[k];
        lk[k] = dlk[k] * dq + cell.v[k];
    }
    if (qss) {
        double yq[M::NY];
#pragma unroll
        for (int i = 0; i < M::NY; i++) yq[i] = y[i];
        qss_substitute<M>(qss, lk, yq);
        M::template eval<true>(P, lk, dlk, yq, f, &J);
        qss_fold<M>(qss, lk, dlk, f, &J);
        f[0] += qdrive;
        return;
    }
    M::template eval<true>(P, lk, dlk, y, f, &J);
    f[0] += qdrive;
}

// Factorisation of  W = I/(h gamma) - J  for the arrow + core structure
template <class M>
struct WFactor {
    double invd[M::NG];          // 1 / (1/(h gamma) - Dg_i)
    double w[M::NC][M::NG];      // Jcg[c][i] * invd_i
    double core[M::NC][M::NC];   // LU of the Schur-complemented core block (no pivoting needed:
                                 // NC == 1 -> scalar reciprocal)
};

template <class M>
SONIC_HD void factor_W(const Jac<M::NC, M::NG> &J, double inv_hg, WFactor<M> &F)
{
    constexpr int NC = M::NC, NG = M::NG;
#pragma unroll
    for (int i = 0; i < NG; i++) F.invd[i] = fast_rcp(inv_hg - J.Dg[i]);
    double A[NC][NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < NG; i++) {
            F.w[c][i] = J.Jcg[c][i] * F.invd[i];
            s += F.w[c][i] * J.Jgq[i];
        }
#pragma unroll
        for (int d = 0; d < NC; d++) A[c][d] = (c == d ? inv_hg : 0.0) - J.Jcc[c][d];
        A[c][0] -= s;
    }
    // in-place Doolittle LU without pivoting; diagonal stored as reciprocal
#pragma unroll
    for (int k = 0; k < NC; k++) {
        A[k][k] = fast_rcp(A[k][k]);
#pragma unroll
        for (int r = k + 1; r < NC; r++) {
            A[r][k] *= A[k][k];
#pragma unroll
            for (int d = k + 1; d < NC; d++) A[r][d] -= A[r][k] * A[k][d];
        }
    }
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int d = 0; d < NC; d++) F.core[c][d] = A[c][d];
}

// Solve W k = r in place (r -> k). Layout [core | gates].
template <class M>
SONIC_HD void solve_W(const Jac<M::NC, M::NG> &J, const WFactor<M> &F, double *r)
{
    constexpr int NC = M::NC, NG = M::NG;
    double b[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        double s = r[c];
#pragma unroll
        for (int i = 0; i < NG; i++) s += F.w[c][i] * r[NC + i];
        b[c] = s;
    }
    // forward / backward substitution
#pragma unroll
    for (int c = 1; c < NC; c++)
#pragma unroll
        for (int d = 0; d < c; d++) b[c] -= F.core[c][d] * b[d];
#pragma unroll
    for (int c = NC - 1; c >= 0; c--) {
#pragma unroll
        for (int d = c + 1; d < NC; d++) b[c] -= F.core[c][d] * b[d];
        b[c] *= F.core[c][c];
    }
#pragma unroll
    for (int c = 0; c < NC; c++) r[c] = b[c];
#pragma unroll
    for (int i = 0; i < NG; i++) r[NC + i] = (r[NC + i] + J.Jgq[i] * b[0]) * F.invd[i];
}

// One RODAS4 step attempt from y with step h. All six stages evaluate the lookup lines of the
// home cell `cell` (see integrate_config). Outputs ynew, the scaled error norm and the stage
// increments k1..k5 (needed by the dense output, which is only evaluated when a row falls
// inside the step). f0 = f(y) and J = df/dy(y) are evaluated by the caller: they survive a
// rejected step.
template <class M, class C>
SONIC_HD void rodas4_step(const typename M::Params &P, const C &cell,
                          const double *y, const double *f0, const Jac<M::NC, M::NG> &J,
                          double inv_h, const SolverOpts &o, double *ynew, double (*k)[M::NY],
                          float &errnorm)
{
    using namespace rodas4;
    constexpr int NY = M::NY;
    WFactor<M> F;
    double k6[NY], yt[NY];
    double *k1 = k[0], *k2 = k[1], *k3 = k[2], *k4 = k[3], *k5 = k[4];

#pragma unroll
    for (int i = 0; i < NY; i++) k1[i] = f0[i];
    factor_W<M>(J, inv_h * (1.0 / gamma), F);
    solve_W<M>(J, F, k1);

#pragma unroll
    for (int i = 0; i < NY; i++) yt[i] = y[i] + a21 * k1[i];
    eval_home<M>(P, cell, yt, k2, o.qss_gates, o.qdrive);
    {
        const double g1 = c21 * inv_h;
#pragma unroll
        for (int i = 0; i < NY; i++) k2[i] += g1 * k1[i];
    }
    solve_W<M>(J, F, k2);

#pragma unroll
    for (int i = 0; i < NY; i++) yt[i] = y[i] + a31 * k1[i] + a32 * k2[i];
    eval_home<M>(P, cell, yt, k3, o.qss_gates, o.qdrive);
    {
        const double g1 = c31 * inv_h, g2 = c32 * inv_h;
#pragma unroll
        for (int i = 0; i < NY; i++) k3[i] += g1 * k1[i] + g2 * k2[i];
    }
    solve_W<M>(J, F, k3);

#pragma unroll
    for (int i = 0; i < NY; i++) yt[i] = y[i] + a41 * k1[i] + a42 * k2[i] + a43 * k3[i];
    eval_home<M>(P, cell, yt, k4, o.qss_gates, o.qdrive);
    {
        const double g1 = c41 * inv_h, g2 = c42 * inv_h, g3 = c43 * inv_h;
#pragma unroll
        for (int i = 0; i < NY; i++) k4[i] += g1 * k1[i] + g2 * k2[i] + g3 * k3[i];
    }
    solve_W<M>(J, F, k4);

#pragma unroll
    for (int i = 0; i < NY; i++)
        yt[i] = y[i] + a51 * k1[i] + a52 * k2[i] + a53 * k3[i] + a54 * k4[i];
    eval_home<M>(P, cell, yt, k5, o.qss_gates, o.qdrive);
    {
        const double g1 = c51 * inv_h, g2 = c52 * inv_h, g3 = c53 * inv_h, g4 = c54 * inv_h;
#pragma unroll
        for (int i = 0; i < NY; i++) k5[i] += g1 * k1[i] + g2 * k2[i] + g3 * k3[i] + g4 * k4[i];
    }
    solve_W<M>(J, F, k5);

#pragma unroll
    for (int i = 0; i < NY; i++) yt[i] += k5[i];
    eval_home<M>(P, cell, yt, k6, o.qss_gates, o.qdrive);
    {
        const double g1 = c61 * inv_h, g2 = c62 * inv_h, g3 = c63 * inv_h, g4 = c64 * inv_h,
                     g5 = c65 * inv_h;
#pragma unroll
        for (int i = 0; i < NY; i++)
            k6[i] += g1 * k1[i] + g2 * k2[i] + g3 * k3[i] + g4 * k4[i] + g5 * k5[i];
    }
    solve_W<M>(J, F, k6);

    // embedded error estimate = k6; scaled RMS norm in single precision (a 3-digit quantity)
    float e2 = 0.0f;
    const float rtol = (float)o.rtol, atol = (float)o.atol;
#pragma unroll
    for (int i = 0; i < NY; i++) {
        ynew[i] = yt[i] + k6[i];
        const float sc = atol + rtol * fmaxf(fabsf((float)y[i]), fabsf((float)ynew[i]));
        const float e = (float)k6[i] / sc;
        e2 += e * e;
    }
    errnorm = sqrtf(e2 * (1.0f / NY));
}

// RODAS3 (Sandu et al. 1997, "Benchmarking stiff ODE solvers for atmospheric chemistry problems
// II: Rosenbrock solvers"): 4 stages / 3 function evaluations, order 3(2), L-stable, stiffly
// accurate. In the same transformed form as above:
//   gamma = 1/2;  Y2 = y, Y3 = y + 2 k1, Y4 = y + 2 k1 + k3
//   c21 = 4, c31 = 1, c32 = -1, c41 = 1, c42 = -1, c43 = -8/3;  ynew = Y4 + k4, err = k4
template <class M, class C>
SONIC_HD void rodas3_step(const typename M::Params &P, const C &cell,
                          const double *y, const double *f0, const Jac<M::NC, M::NG> &J,
                          double inv_h, const SolverOpts &o, double *ynew, float &errnorm)
{
    constexpr int NY = M::NY;
    WFactor<M> F;
    double k1[NY], k2[NY], k3[NY], k4[NY], yt[NY];
#pragma unroll
    for (int i = 0; i < NY; i++) k1[i] = f0[i];
    factor_W<M>(J, inv_h * 2.0, F);
    solve_W<M>(J, F, k1);
    {
        const double g1 = 4.0 * inv_h;
#pragma unroll
        for (int i = 0; i < NY; i++) k2[i] = f0[i] + g1 * k1[i];
    }
    solve_W<M>(J, F, k2);
#pragma unroll
    for (int i = 0; i < NY; i++) yt[i] = y[i] + 2.0 * k1[i];
    eval_home<M>(P, cell, yt, k3, o.qss_gates, o.qdrive);
#pragma unroll
    for (int i = 0; i < NY; i++) k3[i] += inv_h * (k1[i] - k2[i]);
    solve_W<M>(J, F, k3);
#pragma unroll
    for (int i = 0; i < NY; i++) yt[i] += k3[i];
    eval_home<M>(P, cell, yt, k4, o.qss_gates, o.qdrive);
    {
        const double g3 = -(8.0 / 3.0) * inv_h;
#pragma unroll
        for (int i = 0; i < NY; i++) k4[i] += inv_h * (k1[i] - k2[i]) + g3 * k3[i];
    }
    solve_W<M>(J, F, k4);
    float e2 = 0.0f;
    const float rtol = (float)o.rtol, atol = (float)o.atol;
#pragma unroll
    for (int i = 0; i < NY; i++) {
        ynew[i] = yt[i] + k4[i];
        const float sc = atol + rtol * fmaxf(fabsf((float)y[i]), fabsf((float)ynew[i]));
        const float e = (float)k4[i] / sc;
        e2 += e * e;
    }
    errnorm = sqrtf(e2 * (1.0f / NY));
}

// ROS4 with Shampine's parameters (Kaps & Rentrop 1979; Shampine 1982, "Implementation of Rosenbrock
// methods"): 4 stages / 3 function evaluations, order 4(3), A-stable (not stiffly accurate).
// Transformed form as above:
//   gamma = 1/2;  Y2 = y + 2 k1;  Y3 = Y4 = y + 48/25 k1 + 6/25 k2
//   c21 = -8, c31 = 372/25, c32 = 12/5, c41 = -112/125, c42 = -54/125, c43 = -2/5
//   ynew = y + 19/9 k1 + 1/2 k2 + 25/108 k3 + 125/108 k4;  err = 17/54 k1 + 7/36 k2 + 125/108 k4
template <class M, class C>
SONIC_HD void ros4s_step(const typename M::Params &P, const C &cell,
                         const double *y, const double *f0, const Jac<M::NC, M::NG> &J,
                         double inv_h, const SolverOpts &o, double *ynew, float &errnorm)
{
    constexpr int NY = M::NY;
    WFactor<M> F;
    double k1[NY], k2[NY], k3[NY], k4[NY], yt[NY];
#pragma unroll
    for (int i = 0; i < NY; i++) k1[i] = f0[i];
    factor_W<M>(J, inv_h * 2.0, F);
    solve_W<M>(J, F, k1);
#pragma unroll
    for (int i = 0; i < NY; i++) yt[i] = y[i] + 2.0 * k1[i];
    eval_home<M>(P, cell, yt, k2, o.qss_gates, o.qdrive);
#pragma unroll
    for (int i = 0; i < NY; i++) k2[i] += (-8.0 * inv_h) * k1[i];
    solve_W<M>(J, F, k2);
#pragma unroll
    for (int i = 0; i < NY; i++) yt[i] = y[i] + (48.0 / 25.0) * k1[i] + (6.0 / 25.0) * k2[i];
    eval_home<M>(P, cell, yt, k3, o.qss_gates, o.qdrive);
#pragma unroll
    for (int i = 0; i < NY; i++) {
        k4[i] = k3[i] + inv_h * ((-112.0 / 125.0) * k1[i] + (-54.0 / 125.0) * k2[i]);
        k3[i] += inv_h * ((372.0 / 25.0) * k1[i] + (12.0 / 5.0) * k2[i]);
    }
    solve_W<M>(J, F, k3);
#pragma unroll
    for (int i = 0; i < NY; i++) k4[i] += (-2.0 / 5.0) * inv_h * k3[i];
    solve_W<M>(J, F, k4);
    float e2 = 0.0f;
    const float rtol = (float)o.rtol, atol = (float)o.atol;
#pragma unroll
    for (int i = 0; i < NY; i++) {
        ynew[i] = y[i] + (19.0 / 9.0) * k1[i] + 0.5 * k2[i] + (25.0 / 108.0) * k3[i] + (125.0 / 108.0) * k4[i];
        const float sc = atol + rtol * fmaxf(fabsf((float)y[i]), fabsf((float)ynew[i]));
        const float e = (float)((17.0 / 54.0) * k1[i] + (7.0 / 36.0) * k2[i] + (125.0 / 108.0) * k4[i]) / sc;
        e2 += e * e;
    }
    errnorm = sqrtf(e2 * (1.0f / NY));
}

// Dense-output vectors of a step:  y(t + s h) = y (1-s) + s (ynew + (1-s) (c3 + s c4))
template <int NY>
SONIC_HD void rodas4_dense(const double (*k)[NY], double *c3, double *c4)
{
    using namespace rodas4;
#pragma unroll
    for (int i = 0; i < NY; i++) {
        c3[i] = d21 * k[0][i] + d22 * k[1][i] + d23 * k[2][i] + d24 * k[3][i] + d25 * k[4][i];
        c4[i] = d31 * k[0][i] + d32 * k[1][i] + d33 * k[2][i] + d34 * k[3][i] + d35 * k[4][i];
    }
}

// ---------------------------------------------------------------------------------------------
// Segment schedule of one configuration (host-built, mirrors EventDrivenSolver.solve,
// solvers.py:445-480 and Appendix B of SURVEY.md): segment s integrates from t0[s] to t1[s]
// with the tables of level[s], and emits n[s] rows on np.linspace(t0, t1, n) -- the first of
// which repeats the state at t0 (solvers.py:166-170) -- all labelled stimstate = x[s].
// ---------------------------------------------------------------------------------------------
struct Schedule {
    const double *t0;
    const double *t1;
    const double *x;
    const int *n;
    const int *level;
    int nseg;
};

// np.linspace(t0, t1, n) (numpy/_core/function_base.py): step = (t1 - t0) / (n - 1);
// y = arange(n) * step + t0 (product and sum rounded separately), y[-1] = t1; if step == 0,
// y = arange(n) / (n - 1) * (t1 - t0) + t0. Contraction is disabled to round like numpy.
struct Linspace {
    double t0, t1, delta, step;
    int n;
};
SONIC_HD Linspace linspace_make(double t0, double t1, int n)
{
    Linspace g;
    g.t0 = t0; g.t1 = t1; g.n = n;
    g.delta = t1 - t0;
    g.step = g.delta / (double)(n - 1);
    return g;
}
SONIC_HD double linspace_at(const Linspace &g, int i)
{
#pragma clang fp contract(off)
    if (i == g.n - 1) return g.t1;
    if (g.step == 0.0) return ((double)i / (double)(g.n - 1)) * g.delta + g.t0;
    const double prod = (double)i * g.step;
    return prod + g.t0;
}

// Integrate one configuration. `emit(row, t, x, y, Vm)` is called once per output row, in row
// order (row 0 = initial condition with stimstate 0).
//
// The loop is a FLAT state machine -- every iteration is one step attempt, whatever segment the
// configuration is in -- so that the lanes of a wavefront (one configuration each) re-converge
// once per step and never wait for each other at segment or output-row boundaries.
//
// Lookup access ("home cell"): the right-hand side is only C0 at the nodes of the charge grid
// (piecewise-linear tables), so a step that straddles a node sees a jump in f' and is usually
// rejected by the error estimate. Instead the step proposal of the error controller
// (Hairer & Wanner IV.7) is capped by the predicted time at which Q reaches the node it is
// heading to, plus SONIC_OV_TARGET of a cell; all six stages of the step then use the lines of
// ONE cell (the home cell, in registers), and a step that ends more than SONIC_OV_MAX of a
// cell outside its home cell is rejected and retried with a secant-corrected size. The next home
// cell is one cell up or down in nearly every case, so it is loaded by index without a search.
// Returns status bits; *nsteps / *nrej are filled if non-null.
// Rosenbrock method of a model: 4 = RODAS4 (default), 5 = ROS4 with Shampine's parameters (models that
// declare `static constexpr int METHOD`), 3 = RODAS3 (experiments). SONIC_METHOD overrides for all.
template <class M, class = void>
struct ModelMethod { static constexpr int value = 4; };
template <class M>
struct ModelMethod<M, decltype((void)M::METHOD)> { static constexpr int value = M::METHOD; };

template <class M, class Emit, class C>
SONIC_HD int integrate_config(const typename M::Params &P, const LevelGrid &G,
                              const Schedule &S, const double *y0, const SolverOpts &o,
                              Emit &&emit, int *nsteps_out, int *nrej_out, C &home, StepCounts *counts = nullptr)
{
    StepCounts cnt;
    constexpr int NY = M::NY;
    constexpr int NT = M::NT;
#ifdef SONIC_METHOD
    constexpr int METHOD = SONIC_METHOD;
#else
    constexpr int METHOD = ModelMethod<M>::value;
#endif
    double y[NY];                // `home`: the home cell of y[0] (registers or LDS, see CellRec)
    int jh = -1;                 // its index
    int status = ST_OK;
#pragma unroll
    for (int i = 0; i < NY; i++) y[i] = y0[i];

    int nsteps = 0, nrej = 0;
    long row = 0;
    bool dead = false;   // once true, remaining rows are NaN (reference: NaN derivatives)

    // row 0: initial conditions, stimstate 0 (solvers.py:99-117, 404-406); Vm from the A = 0
    // level, which the host always places at level index 0
    jh = locate_cell<NT>(G, 0, y[0], home);
    if (jh < 0) { dead = true; status |= ST_Q_OUT_OF_RANGE; }
    {
        const double Vm = dead ? NAN : home.s[0] * (y[0] - home.xlo) + home.v[0];
        emit(row++, S.nseg > 0 ? S.t0[0] : 0.0, 0.0, y, Vm);
    }

    int s = 0;
    bool seg_init = true;
    bool have_f0 = false;          // f0 / J valid for the current y and level
    double x = 0.0, t = 0.0, h = o.h0;
    int level = 0, irow = 0;
    Linspace grid = linspace_make(0.0, 0.0, 2);
    double tr = 0.0;               // time of the next row to emit
    double f0[NY];
    Jac<M::NC, M::NG> J;
    double k[5][NY];

    while (s < S.nseg) {
        if (seg_init) {
            seg_init = false;
            grid = linspace_make(S.t0[s], S.t1[s], S.n[s]);
            x = S.x[s]; level = S.level[s];
            // new level -> new tables: reload the home cell
            if (!dead) {
                jh = locate_cell<NT>(G, level, y[0], home);
                if (jh < 0) { dead = true; status |= ST_Q_OUT_OF_RANGE; }
            }
            have_f0 = false;
            // first row of the segment = state at t0 under the new stimstate
            const double Vm = dead ? NAN : home.s[0] * (y[0] - home.xlo) + home.v[0];
            if (dead) {
#pragma unroll
                for (int i = 0; i < NY; i++) y[i] = NAN;
            }
            emit(row++, grid.t0, x, y, Vm);
            irow = 1;
            t = grid.t0;
            h = fmin(o.h0, grid.delta);
            if (dead || !(grid.t1 - grid.t0 > SONIC_SEG_EPS)) {
                // dead lane, or a segment of zero length -- or shorter than the smallest step, e.g. a
                // progress-log event one ulp away from a stimulus event (odeint: "tout too close
                // to t to start integration") --: rows repeat the state
                for (; irow < grid.n; irow++) emit(row++, linspace_at(grid, irow), x, y, Vm);
                s++;
                seg_init = true;
                continue;
            }
            tr = linspace_at(grid, irow);
        }

        const double cellw = home.xhi - home.xlo;
        if (!have_f0) {
            // f(y), J(y) with the home cell's lines; kept across rejected steps
            eval_home_jac<M>(P, home, y, f0, J, o.qss_gates, o.qdrive);
            have_f0 = true;
        }

        // kink-aware cap: time for Q to reach the node it is heading to, plus a sliver
        // (single precision: it is only a proposal)
        bool capped = false;
        {
            const double dq = f0[0];
            const double dist = dq > 0.0 ? (home.xhi - y[0]) + SONIC_LANE_OV_TARGET * cellw
                                         : (home.xlo - y[0]) - SONIC_LANE_OV_TARGET * cellw;
            double fp = 0.0;                       // (J f)_Q: second derivative of the charge (node_time_*)
#pragma unroll
            for (int b = 0; b < M::NC; b++) fp += J.Jcc[0][b] * f0[b];
#pragma unroll
            for (int i = 0; i < M::NG; i++) fp += J.Jcg[0][i] * f0[M::NC + i];
            const float fq = (float)dq, dd = (float)dist;
            const float root = sqrtf(node_time_discriminant(fq, (float)fp, dd));
            const float hc = 2.0f * dd / node_time_denominator(fq, (float)fp, dd, root);
            capped = hc > 0.0f && (double)hc < h;
            if (capped) h = fmax((double)hc, 1e-3 * h);
        }

        bool last = false;
        if (t + 1.0001 * h >= grid.t1) { h = grid.t1 - t; last = true; }
        const double inv_h = fast_rcp(h);
        double ynew[NY];
        float err;
        if constexpr (METHOD == 4) rodas4_step<M>(P, home, y, f0, J, inv_h, o, ynew, k, err);
        else if constexpr (METHOD == 5) ros4s_step<M>(P, home, y, f0, J, inv_h, o, ynew, err);
        else rodas3_step<M>(P, home, y, f0, J, inv_h, o, ynew, err);
        nsteps++;
        // step-size controller (Hairer & Wanner IV.7): h_new = h / fac, fac = err^(1/4) / 0.9
        // clipped to [1/6, 5] <=> rfac = 0.9 err^(-1/4) clipped to [0.2, 6]; single precision
        float rfac = (METHOD == 4 || METHOD == 5) ? 0.9f / sqrtf(sqrtf(err)) : 0.9f / cbrtf(err);
        rfac = fminf(6.0f, fmaxf(0.2f, rfac));
        if (!(err == err)) rfac = 0.2f;   // NaN -> shrink
        double hnew = h * (double)rfac;
        // all stages used the home cell's lines: only valid if the step ended (almost) inside it
        const double over = fmax(home.xlo - ynew[0], ynew[0] - home.xhi);
        bool accept = err <= 1.0f;
        if (over > SONIC_LANE_OV_MAX * cellw) {
            accept = false;
            cnt.over++;
            // secant estimate of the step that ends SONIC_LANE_OV_TARGET past the node
            const double moved = fabs(ynew[0] - y[0]);
            const double want = moved - over + SONIC_LANE_OV_TARGET * cellw;
            hnew = h * fmax(0.1, fmin(0.9, want / moved));
        }
        if (accept) {
            cnt.capped += capped ? 1 : 0;
            const double tnew = last ? grid.t1 : t + h;
            // dense output for every grid row inside (t, tnew]
            if (irow < grid.n && (last || tr <= tnew)) {
                // RODAS4: the method's dense output; otherwise cubic Hermite from (y, f0) and
                // (ynew, f(ynew)), f(ynew) with the home cell's lines (ynew is at most SONIC_LANE_OV_MAX
                // of a cell outside it)
                double c3[NY], c4[NY];
                if constexpr (METHOD == 4) rodas4_dense<NY>(k, c3, c4);
                else eval_home<M>(P, home, ynew, c3, o.qss_gates, o.qdrive);      // c3 = f(ynew)
                while (irow < grid.n && (last || tr <= tnew)) {
                    double yr[NY];
                    if (tr >= tnew) {
#pragma unroll
                        for (int i = 0; i < NY; i++) yr[i] = ynew[i];
                    } else {
                        const double sg = (tr - t) * inv_h, s1 = 1.0 - sg;
                        if constexpr (METHOD == 4) {
#pragma unroll
                            for (int i = 0; i < NY; i++)
                                yr[i] = y[i] * s1 + sg * (ynew[i] + s1 * (c3[i] + sg * c4[i]));
                        } else {
                            const double hh = h;
#pragma unroll
                            for (int i = 0; i < NY; i++) {
                                const double d = ynew[i] - y[i];
                                yr[i] = y[i] + sg * (d + s1 * ((hh * f0[i] - d) * s1 - (hh * c3[i] - d) * sg));
                            }
                        }
                    }
                    // Vm = lerp of the V table at the row's charge (nbls.py:426-428): the row lies
                    // in the home cell or, past the node, in the prefetched neighbour
                    double Vm;
                    if (yr[0] >= home.xlo && yr[0] < home.xhi) {
                        Vm = home.s[0] * (yr[0] - home.xlo) + home.v[0];
                    } else {
                        CellRec<NT> cr;
                        Vm = NAN;
                        if (locate_cell<NT>(G, level, yr[0], cr) >= 0)
                            Vm = cr.s[0] * (yr[0] - cr.xlo) + cr.v[0];
                    }
                    emit(row++, tr, x, yr, Vm);
                    irow++;
                    if (irow < grid.n) tr = linspace_at(grid, irow);
                }
            }
#pragma unroll
            for (int i = 0; i < NY; i++) y[i] = ynew[i];
            t = tnew;
            h = hnew;
            have_f0 = false;
            // new home cell: unchanged, the prefetched neighbour, or (rarely) a demand load
            if (!(y[0] >= home.xlo && y[0] < home.xhi)) {
                cnt.cross++;
                // one cell up or down in nearly every case (kink-aware steps): try that first
                const int jg = y[0] >= home.xhi ? jh + 1 : jh - 1;
                if (jg >= 0 && jg < G.n_cells) {
                    load_cell<NT>(G, level, jg, home);
                    jh = jg;
                }
                if (!(y[0] >= home.xlo && y[0] < home.xhi)) {
                    jh = locate_cell<NT>(G, level, y[0], home);
                    if (jh < 0) { dead = true; status |= ST_Q_OUT_OF_RANGE; }
                }
            }
            if (last) { s++; seg_init = true; }
        } else {
            nrej++;
            h = fmin(hnew, h);
            if (!(h >= o.hmin)) { dead = true; status |= ST_STEP_UNDERFLOW; }
        }
        if (nsteps >= o.max_steps && !dead && !seg_init) { dead = true; status |= ST_MAX_STEPS; }
        if (dead && !seg_init) {
            // fill the rest of this segment with NaN rows; later segments take the dead path above
#pragma unroll
            for (int i = 0; i < NY; i++) y[i] = NAN;
            for (; irow < grid.n; irow++) emit(row++, linspace_at(grid, irow), x, y, NAN);
            s++;
            seg_init = true;
        }
    }
    if (nsteps_out) *nsteps_out = nsteps;
    if (nrej_out) *nrej_out = nrej;
    if (counts) *counts = cnt;
    return status;
}


// ---------------------------------------------------------------------------------------------
// On-device spike detection on the Qm rows of one configuration, while they are produced.
//
// Reference: detectSpikes / find_tpeaks (PySONIC/postpro.py:175-284) = scipy.signal.find_peaks
// with height >= SPIKE_MIN_QAMP (3e-5 C/m2), prominence >= SPIKE_MIN_QPROM (20e-5 C/m2, full
// window) and distance >= SPIKE_MIN_DT (0.5 ms), applied to the signal resampled linearly at
// 1e-7 s. Local maxima of a linear interpolant sit on original samples, so peaks are detected on
// the rows themselves:
//   * local maximum: strict rise followed by a strict fall; a plateau (e.g. the duplicated rows
//     at events) counts once (scipy _local_maxima_1d);
//   * prominence = height - max(lowest value between the peak and the previous higher sample,
//     lowest value between the peak and the next higher sample), computed online with a stack of
//     unfinished peaks (each carries the running minimum since it occurred; minima are merged on
//     pop), exactly scipy's peak_prominences with wlen=None;
//   * the distance rule only matters if two prominent peaks are closer than 0.5 ms: such a pair
//     raises SPK_CLOSE_PEAKS in the returned flags and the host falls back to the reference's
//     full procedure for that configuration.
// Candidates (height >= mph) are appended to `cand` (5 doubles each: t, height, prominence,
// left minimum, running minimum) in order of occurrence; `stack` holds indices of unfinished ones.
// ---------------------------------------------------------------------------------------------
enum : int { SPK_OVERFLOW = 1, SPK_CLOSE_PEAKS = 2 };

struct SpikeSummary {
    double nspikes, t_first, t_last, sum_inv_isi;
    int flags;
};

struct SpikeTracker {
    // SPIKE_MIN_QAMP, SPIKE_MIN_QPROM, SPIKE_MIN_DT (constants.py:49-51)
    static constexpr double mph = 3e-5, mpp = 20e-5, mpt = 5e-4;
    double *cand;       // [cap][5]
    int *stack;         // [cap]
    int cap, ncand, depth, flags;
    // signal history for local-maximum detection (registers; memory is touched only when a
    // candidate peak is found)
    double v_prev, t_rise;   // last distinct value; time of the first sample of the current plateau
    double run_min;          // minimum of the signal since the innermost unfinished peak (or start)
    int trend;               // +1: last distinct move was a rise, -1: fall, 0: none yet

    SONIC_HD void init(double *cand_, int *stack_, int cap_)
    {
        cand = cand_; stack = stack_; cap = cap_; ncand = 0; depth = 0; flags = 0;
        v_prev = NAN; t_rise = 0.0; trend = 0; run_min = INFINITY;
    }

    SONIC_HD void on_peak(double tp, double hp)
    {
        if (ncand >= cap) { flags |= SPK_OVERFLOW; return; }
        // finish every unfinished peak lower than this one: its right window ends here.
        // `merged` = minimum of the signal since the peak being examined
        double merged = run_min;
        while (depth > 0) {
            double *top = cand + (long)stack[depth - 1] * 5;
            merged = fmin(top[4], merged);
            if (!(top[1] < hp)) { top[4] = merged; break; }
            top[2] = top[1] - fmax(top[3], merged);
            depth--;
        }
        double *c = cand + (long)ncand * 5;
        c[0] = tp; c[1] = hp; c[2] = NAN; c[3] = merged; c[4] = INFINITY;
        stack[depth++] = ncand++;
        run_min = INFINITY;
    }

    SONIC_HD void feed(double t, double v)
    {
        // written with selects: the only branch is the (rare) candidate peak. A NaN sample (rows
        // of a dead configuration) compares false everywhere and leaves the history untouched.
        run_min = fmin(run_min, v);
        const bool up = v > v_prev, down = v < v_prev, first = !(v_prev == v_prev) && v == v;
        if (down && trend == 1 && v_prev >= mph) on_peak(t_rise, v_prev);
        t_rise = (up || first) ? t : t_rise;          // a plateau keeps the time of its first sample
        trend = up ? 1 : (down ? -1 : trend);
        v_prev = v == v ? v : v_prev;
    }

    SONIC_HD SpikeSummary finish()
    {
        // unfinished peaks: right window runs to the end of the signal
        double merged = run_min;
        while (depth > 0) {
            double *top = cand + (long)stack[depth - 1] * 5;
            merged = fmin(top[4], merged);
            top[2] = top[1] - fmax(top[3], merged);
            depth--;
        }
        SpikeSummary s{0.0, NAN, NAN, 0.0, flags};
        double tprev = NAN;
        for (int i = 0; i < ncand; i++) {
            const double *c = cand + (long)i * 5;
            if (!(c[2] >= mpp)) continue;
            if (s.nspikes > 0.0) {
                const double isi = c[0] - tprev;
                if (isi < mpt) s.flags |= SPK_CLOSE_PEAKS;
                s.sum_inv_isi += 1.0 / isi;
            } else {
                s.t_first = c[0];
            }
            tprev = c[0];
            s.t_last = c[0];
            s.nspikes += 1.0;
        }
        return s;
    }
};

}  // namespace sonic
