// pysonic_amd/csrc/full_row.hpp
//
// ROW-COOPERATIVE integration of the detailed NICE model (method='full') of the neurons the octet kernel of
// full_coop.hpp does not cover -- LTS, IB, RE, TC, STN: one configuration per ROW of 16 adjacent lanes (four per
// wavefront) instead of one per lane (full_core.hpp).
//
// Reference: NeuronalBilayerSonophore.__simFull / fullDerivatives (PySONIC/core/nbls.py:265-278, 331-354) =
// BilayerSonophore.derivatives (bls.py:681-718) coupled to PointNeuron.derivatives (pneuron.py:485-505) with the true
// rate functions of the neuron (neurons/cortical.py:254-303, thalamic.py:117-366, stn.py:52-136, 209-338), integrated
// on a dense grid of 1000 points per acoustic period and resampled to 10 ns (solvers.py:184-191, 213-221).
//
// Why: one lane per configuration makes every step the latency of ~6000 dependent FP64 instructions issued for a
// single lane, with the stage vectors in private memory (1 - 3 KB of scratch per lane): 15 - 18 us per step. Here
//   * EVERY STATE IS ONE LANE of the row: the gates where sonic_group.hpp puts them (its lane roles, current
//     ownership and Ca2+ machinery are reused as they are), the other membrane states (Cai, P0, O, C) and the
//     mechanical ones (U, Z, ng) and Qm on lanes the gates leave free. A stage vector is one double per lane, the
//     sixteen of the 8(5,3) pair 32 registers, and a stage combination is one FMA per coefficient whatever the
//     number of states;
//   * the right-hand side broadcasts what every lane needs (U, Z, ng, Qm, the Ca2+ states: row_newbcast DPP moves),
//     evaluates the mechanical system and Vm = Qm / Cm(Z) once -- replicated arithmetic: the same instructions for
//     all lanes, one logarithm shared by the capacitance and the Lennard-Jones powers --, then the TWO RATE CONSTANTS
//     OF EVERY GATE ON ITS OWN LANE from one generic form (three exponentials and three reciprocals for all gates at
//     once, per-lane coefficients: RowRate), the currents with group_rhs, and scatters the derivatives back with
//     0 / 1 lane masks.
// The integrator is the Dormand-Prince 8(5,3) pair with its continuous extension, as in full_coop.hpp (whose
// stage sums, dense output and pressure rotation it shares), at rtol 1e-7.
//
// Configurations whose gates turn ultra-stiff (STN above ~450 kPa) are not for an explicit pair: the kernel gives
// them up as soon as its steps collapse (status bit FULL_ST_STIFF) and the host runs those on the lane kernel, which
// hands them to RODAS4 (full_core.hpp).
//
// Written once over the Ops backends of sonic_group.hpp: GroupOpsDev (DPP) on the device, GroupOpsHost (16-element
// arrays) in the CPU harness.
#pragma once
#include "full_coop.hpp"
#include "sonic_group.hpp"

namespace sonic {

// ---- the two rate constants of a gate from one generic form ------------------------------------------------
//   u_j = (Vm - v_j) k_j,  e_j = exp(u_j)  (arguments capped at +-700)
//   R1 = (n0 + n1 u1 + n2 e1) / (d0 + d1 e1)
//   X2 = (m0 + m1 u2 + m2 e2) / (g0 + g1 e2 + g2 e3)
//   alpha / beta gate (it = 0):   dx/dt = a - r x with a = R1 (alpha), r = R1 + X2 (alpha + beta)
//   x_inf / tau gate  (it = 1):   tau = t0 + X2 + s2 e2 + s3 e3 -- below Vm = vth: t0b + X2 + s2b e2 + s3b e3 --,
//                                 r = 1 / tau, a = R1 r (R1 = x_inf)
// which covers vtrap (n1 u / (e - 1)), c exp(u), c / (1 + e) and every tau of the neurons above. Lanes without a
// voltage-gated state carry zeros with unit denominators: a = r = 0.
enum : int { RR_V1, RR_K1, RR_N0, RR_N1, RR_N2, RR_D0, RR_D1, RR_V2, RR_K2, RR_M0, RR_M1, RR_M2, RR_V3, RR_K3,
              RR_G0, RR_G1, RR_G2, RR_T0, RR_S2, RR_S3, RR_VTH, RR_T0B, RR_S2B, RR_S3B, RR_IT,
              // the lane's component: 0 / 1 masks (U, Z, ng, Qm, core state c >= 1), error floor and weight
              RR_MU, RR_MZ, RR_MNG, RR_MQ, RR_MC1, RR_MC2, RR_MC3, RR_MC4, RR_FLOOR, RR_ERRW, RR_COUNT };

struct RowLaneSpec {
    double v[RR_COUNT];
    int col;       // output column of the lane's component (t stim Z ng Qm states... Vm), -1: none
    int extra;     // 1: the lane also stores t and stim (columns 0, 1); 2: also Vm (last column)
};

inline RowLaneSpec row_lane_none()
{
    RowLaneSpec s;
    for (int i = 0; i < RR_COUNT; i++) s.v[i] = 0.0;
    s.v[RR_D0] = 1.0; s.v[RR_G0] = 1.0; s.v[RR_T0] = 1.0; s.v[RR_T0B] = 1.0; s.v[RR_VTH] = -1e300;
    s.v[RR_FLOOR] = 1.0;      // (lanes without a state: weight 0 in the error norm, any finite scale)
    s.col = -1; s.extra = 0;
    return s;
}

// builders of the two halves of a gate: `which` = 1 (R1: v1 k1 n d) or 2 (X2: v2 k2 m g)
inline void rr_put(RowLaneSpec &s, int which, double v, double k, double c0, double c1, double c2, double q0, double q1)
{
    if (which == 1) {
        s.v[RR_V1] = v; s.v[RR_K1] = k; s.v[RR_N0] = c0; s.v[RR_N1] = c1; s.v[RR_N2] = c2; s.v[RR_D0] = q0; s.v[RR_D1] = q1;
    } else {
        s.v[RR_V2] = v; s.v[RR_K2] = k; s.v[RR_M0] = c0; s.v[RR_M1] = c1; s.v[RR_M2] = c2; s.v[RR_G0] = q0; s.v[RR_G1] = q1;
        s.v[RR_G2] = 0.0;
    }
}
// c 1e3 vtrap(sgn (Vm - vx), y), vtrap(x, y) = x / (exp(x / y) - 1) = y u / (e - 1), u = x / y
inline void rr_vtrap(RowLaneSpec &s, int which, double c, double sgn, double vx, double y)
{
    rr_put(s, which, vx, sgn / y, 0.0, c * 1e3 * y, 0.0, -1.0, 1.0);
}
// c 1e3 exp((Vm - vx) k)
inline void rr_exp(RowLaneSpec &s, int which, double c, double vx, double k) { rr_put(s, which, vx, k, 0.0, 0.0, c * 1e3, 1.0, 0.0); }
// c 1e3 / (1 + exp((Vm - vx) k))
inline void rr_sig(RowLaneSpec &s, int which, double c, double vx, double k) { rr_put(s, which, vx, k, c * 1e3, 0.0, 0.0, 1.0, 1.0); }
// x_inf = 1 / (1 + exp((Vm - vx) k)) ; tau = t0 + t1 / (g0 + g1 exp((Vm - v2) k2) + g2 exp((Vm - v3) k3))
inline void rr_inf_tau(RowLaneSpec &s, double vx, double k, double t0, double t1, double g0, double g1, double v2,
                       double k2, double g2, double v3, double k3)
{
    rr_put(s, 1, vx, k, 1.0, 0.0, 0.0, 1.0, 1.0);
    s.v[RR_V2] = v2; s.v[RR_K2] = k2; s.v[RR_M0] = t1; s.v[RR_M1] = 0.0; s.v[RR_M2] = 0.0;
    s.v[RR_G0] = g0; s.v[RR_G1] = g1; s.v[RR_G2] = g2; s.v[RR_V3] = v3; s.v[RR_K3] = k3;
    s.v[RR_T0] = t0; s.v[RR_T0B] = t0; s.v[RR_IT] = 1.0;
}

// gate `g` (table order of the model: sonic_models.hpp, mech_core.hpp NeuronRates) of neuron `id` -- the same
// functions as NeuronRates<id>::eval, as data
inline bool row_gate_rate(int id, int g, RowLaneSpec &s)
{
    auto mhn = [&](double VT, int k) {      // cortical.py:36-58
        switch (k) {
        case 0: rr_vtrap(s, 1, 0.32, -1.0, VT + 13.0, 4.0); rr_vtrap(s, 2, 0.28, 1.0, VT + 40.0, 5.0); break;
        case 1: rr_exp(s, 1, 0.128, VT + 17.0, -1.0 / 18.0); rr_sig(s, 2, 4.0, VT + 40.0, -1.0 / 5.0); break;
        default: rr_vtrap(s, 1, 0.032, -1.0, VT + 15.0, 5.0); rr_exp(s, 2, 0.5, VT + 10.0, -1.0 / 40.0); break;
        }
    };
    auto ctx_p = [&](double TauMax) {       // cortical.py:60-66
        rr_inf_tau(s, -35.0, -1.0 / 10.0, 0.0, TauMax, 0.0, 3.3, -35.0, 1.0 / 20.0, 1.0, -35.0, -1.0 / 20.0);
    };
    auto lts_su = [&](double Vx, int k) {   // cortical.py:254-272, thalamic.py:289-307 (v = Vm + Vx)
        if (k == 0)
            rr_inf_tau(s, -(57.0 + Vx), -1.0 / 6.2, 0.612e-3 / 3.7, 1e-3 / 3.7, 0.0, 1.0, -(132.0 + Vx), -1.0 / 16.7, 1.0,
                       -(16.8 + Vx), 1.0 / 18.2);
        else {
            // tau_u = (v < -80 ? exp((v + 467) / 66.6) : exp(-(v + 22) / 10.5) + 28) / 3.7 ms
            rr_inf_tau(s, -(81.0 + Vx), 1.0 / 4.0, 28e-3 / 3.7, 0.0, 1.0, 0.0, -(22.0 + Vx), -1.0 / 10.5, 0.0,
                       -(467.0 + Vx), 1.0 / 66.6);
            s.v[RR_S2] = 1e-3 / 3.7;
            s.v[RR_VTH] = -80.0 - Vx; s.v[RR_T0B] = 0.0; s.v[RR_S2B] = 0.0; s.v[RR_S3B] = 1e-3 / 3.7;
        }
    };
    auto stn1 = [&](double th, double k, double tth, double sg, double t0, double t1) {      // stn.py: x_inf, tau with one exponential
        rr_inf_tau(s, th, 1.0 / k, t0, t1, 1.0, 1.0, tth, -1.0 / sg, 0.0, 0.0, 0.0);
    };
    auto stn2 = [&](double th, double k, double th1, double th2, double s1, double s2, double t0, double t1) {
        rr_inf_tau(s, th, 1.0 / k, t0, t1, 0.0, 1.0, th1, -1.0 / s1, 1.0, th2, -1.0 / s2);
    };
    switch (id) {
    case 2:   // LTS: m h n p s u
        if (g < 3) mhn(-50.0, g); else if (g == 3) ctx_p(4.0); else lts_su(-7.0, g - 4);
        return g < 6;
    case 6:   // IB: m h n p q r (cortical.py:307-400)
        if (g < 3) mhn(-56.2, g); else if (g == 3) ctx_p(0.608);
        else if (g == 4) { rr_vtrap(s, 1, 0.055, -1.0, -27.0, 3.8); rr_exp(s, 2, 0.94, -75.0, -1.0 / 17.0); }
        else { rr_exp(s, 1, 0.000457, -13.0, -1.0 / 50.0); rr_sig(s, 2, 0.0065, -15.0, -1.0 / 28.0); }
        return g < 6;
    case 3:   // RE: m h n s u (thalamic.py:117-179)
        if (g < 3) mhn(-67.0, g);
        else if (g == 3) rr_inf_tau(s, -52.0, -1.0 / 7.4, 1e-3, 0.33e-3, 0.0, 1.0, -27.0, 1.0 / 10.0, 1.0, -102.0, -1.0 / 15.0);
        else rr_inf_tau(s, -80.0, 1.0 / 5.0, 28.3e-3, 0.33e-3, 0.0, 1.0, -48.0, 1.0 / 4.0, 1.0, -407.0, -1.0 / 50.0);
        return g < 5;
    case 4:   // TC: m h n s u, then the O gate of iH (thalamic.py:182-323)
        if (g < 3) mhn(-52.0, g); else if (g < 5) lts_su(0.0, g - 3);
        else rr_inf_tau(s, -75.0, 1.0 / 5.5, 0.0, 1e-3, 0.0, 1.0, -14.59 / 0.086, -0.086, 1.0, 1.87 / 0.0701, 0.0701);
        return g < 6;
    case 5:   // STN: a b c d1 m h n p q (stn.py:52-136, 209-338)
        switch (g) {
        case 0: stn1(-45.0, -14.7, -40.0, -0.5, 1e-3, 1e-3); break;
        case 1: stn2(-90.0, 7.5, -60.0, -40.0, -30.0, 10.0, 0e-3, 200e-3); break;
        case 2: stn2(-30.6, -5.0, -27.0, -50.0, -20.0, 15.0, 45e-3, 10e-3); break;
        case 3: stn2(-60.0, 7.5, -40.0, -20.0, -15.0, 20.0, 400e-3, 500e-3); break;
        case 4: stn1(-40.0, -8.0, -53.0, -0.7, 0.2e-3, 3e-3); break;
        case 5: stn2(-45.5, 6.4, -50.0, -50.0, -15.0, 16.0, 0e-3, 24.5e-3); break;
        case 6: stn2(-41.0, -14.0, -40.0, -40.0, -40.0, 50.0, 0e-3, 11e-3); break;
        case 7: stn2(-56.0, -6.7, -27.0, -102.0, -10.0, 15.0, 5e-3, 0.33e-3); break;
        case 8: stn2(-85.0, 5.8, -50.0, -50.0, -15.0, 16.0, 0e-3, 400e-3); break;
        default: return false;
        }
        return true;
    // ---- the data-driven neurons (sonic_models.hpp: GatedModel; rate functions: mech_core.hpp NeuronRates<id>) ----
    case 7: {  // HHseg: m h n (hh.py:44-86), q10 = 3^((36 - 6.3) / 10)
        const double q = 26.1246286895632;
        switch (g) {
        case 0: rr_vtrap(s, 1, q * 0.1, -1.0, -40.0, 10.0); rr_exp(s, 2, q * 4.0, -65.0, -1.0 / 18.0); break;
        case 1: rr_exp(s, 1, q * 0.07, -65.0, -1.0 / 20.0); rr_sig(s, 2, q * 1.0, -35.0, -1.0 / 10.0); break;
        case 2: rr_vtrap(s, 1, q * 0.01, -1.0, -55.0, 10.0); rr_exp(s, 2, q * 0.125, -65.0, -1.0 / 80.0); break;
        default: return false;
        }
        return true;
    }
    case 9: {  // MRGnode: m h p s (mrg.py:60-108): q10 = 2.2^1.6, 2.9^1.6, 3^0; m / h shifted by 3 mV
        const double qm = 3.530825783474764, qh = 5.493344008948558;
        switch (g) {
        case 0: rr_vtrap(s, 1, qm * 1.86, -1.0, -21.4, 10.3); rr_vtrap(s, 2, qm * 0.086, 1.0, -25.7, 9.16); break;
        case 1: rr_vtrap(s, 1, qh * 0.062, 1.0, -114.0, 11.0); rr_sig(s, 2, qh * 2.3, -31.8, -1.0 / 13.4); break;
        case 2: rr_vtrap(s, 1, qm * 0.01, -1.0, -27.0, 10.2); rr_vtrap(s, 2, qm * 0.00025, 1.0, -34.0, 10.0); break;
        case 3: rr_sig(s, 1, 0.3, -53.0, -1.0 / 5.0); rr_sig(s, 2, 0.03, -90.0, -1.0); break;
        default: return false;
        }
        return true;
    }
    case 10: {  // SUseg: m h n l (sundt.py:70-117): Traub sodium gates, Borg-Graham potassium gates
        const double q = 1.9331820449317627, k = 0.037541548196719836;   // 3^0.6; F / (Rg T) 1e-3 at 309.15 K
        switch (g) {
        case 0: rr_vtrap(s, 1, q * 0.32, -1.0, -45.9, 4.0); rr_vtrap(s, 2, q * 0.28, 1.0, -18.9, 5.0); break;
        case 1: rr_exp(s, 1, q * 0.128, -54.0, -1.0 / 18.0); rr_sig(s, 2, q * 4.0, -31.0, -1.0 / 5.0); break;
        case 2: rr_exp(s, 1, q * 0.03, -32.0, 2.0 * k); rr_exp(s, 2, q * 0.03, -32.0, -3.0 * k); break;
        case 3: rr_exp(s, 1, q * 0.001, -61.0, -2.0 * k); rr_exp(s, 2, q * 0.001, -61.0, 0.0); break;
        default: return false;
        }
        return true;
    }
    case 11: {  // FHnode: m h n p (fh.py:61-98): q10 = 3^1.6, voltages relative to the -70 mV rest
        const double q = 5.799546134795289;
        switch (g) {
        case 0: rr_vtrap(s, 1, q * 0.36, -1.0, -48.0, 3.0); rr_vtrap(s, 2, q * 0.4, 1.0, -57.0, 20.0); break;
        case 1: rr_vtrap(s, 1, q * 0.1, 1.0, -80.0, 6.0); rr_sig(s, 2, q * 4.5, -25.0, -1.0 / 10.0); break;
        case 2: rr_vtrap(s, 1, q * 0.02, -1.0, -35.0, 10.0); rr_vtrap(s, 2, q * 0.05, 1.0, -60.0, 10.0); break;
        case 3: rr_vtrap(s, 1, q * 0.006, -1.0, -30.0, 10.0); rr_vtrap(s, 2, q * 0.09, 1.0, -95.0, 20.0); break;
        default: return false;
        }
        return true;
    }
    case 8: {  // SWnode: m h (sweeney.py:41-60)
        if (g == 0) {
            // alpha_m = (126 + 0.363 Vm) / (1 + e1), e1 = exp(-(Vm + 49) / 5.3): the numerator in u1 = -(Vm + 49) / 5.3
            rr_put(s, 1, -49.0, -1.0 / 5.3, (126.0 - 0.363 * 49.0) * 1e3, -0.363 * 5.3 * 1e3, 0.0, 1.0, 1.0);
            // beta_m = alpha_m / e2, e2 = exp((Vm + 56.2) / 4.17): (126 + 0.363 Vm) / (e2 + e2 e1), e2 e1 = one exponential
            rr_put(s, 2, -56.2, 1.0 / 4.17, (126.0 - 0.363 * 56.2) * 1e3, 0.363 * 4.17 * 1e3, 0.0, 0.0, 1.0);
            const double k3 = 1.0 / 4.17 - 1.0 / 5.3;
            s.v[RR_G2] = 1.0; s.v[RR_K3] = k3; s.v[RR_V3] = -(56.2 / 4.17 - 49.0 / 5.3) / k3;
        } else if (g == 1) {
            // beta_h = 15.6 / (1 + eb), eb = exp(-(Vm + 56) / 10); alpha_h = beta_h ea, ea = exp(-(Vm + 74.5) / 5):
            // h_inf = 1 / (1 + exp((Vm + 74.5) / 5)), tau_h = (1 + eb) / (15.6e3 (1 + ea))
            rr_inf_tau(s, -74.5, 1.0 / 5.0, 0.0, 1.0 / 15.6e3, 1.0, 0.0, -56.0, -1.0 / 10.0, 1.0, -74.5, -1.0 / 5.0);
            s.v[RR_M2] = 1.0 / 15.6e3;
        } else
            return false;
        return true;
    }
    }
    return false;
}

template <class O>
struct RowConsts {
    typename O::V r[RR_COUNT];
    typename O::I col, extra;
};

// (a, r) of every lane's gate at the potential Vm (the generic form above); DERIV: also d a / d Vm, d r / d Vm
// (analytic; the caps of the exponentials' arguments are not differentiated: rates beyond exp(700) are beyond use)
// `hitch` (may be null): two more exponentials the caller needs -- exp(hitch[0]), exp(hitch[1]), returned in place --
// evaluated on the lane of U, which carries no gate, as that lane's e1 / e2: one vector exponential serves the gates
// and the two Lennard-Jones powers of the mechanical system (their arguments stay far inside the +-700 cap).
template <class O, bool DERIV, int LHITCH = 12>
SONIC_HD void row_rates(const RowConsts<O> &R, double Vm, typename O::V &a, typename O::V &r, typename O::V &da,
                        typename O::V &dr, double *hitch = nullptr)
{
    typedef typename O::V V;
    const V Vv = O::splat(Vm), cap = O::splat(700.0), ncap = O::splat(-700.0);
    V u1 = O::mul(O::sub(Vv, R.r[RR_V1]), R.r[RR_K1]), u2 = O::mul(O::sub(Vv, R.r[RR_V2]), R.r[RR_K2]);
    const V u3 = O::mul(O::sub(Vv, R.r[RR_V3]), R.r[RR_K3]);
    if (hitch) {
        u1 = O::lt_pick(O::splat(0.5), R.r[RR_MU], O::splat(hitch[0]), u1);
        u2 = O::lt_pick(O::splat(0.5), R.r[RR_MU], O::splat(hitch[1]), u2);
    }
    const V e1 = O::exp_(O::max_(O::min_(u1, cap), ncap)), e2 = O::exp_(O::max_(O::min_(u2, cap), ncap)),
            e3 = O::exp_(O::max_(O::min_(u3, cap), ncap));
    if (hitch) {
        hitch[0] = O::template bcast<LHITCH>(e1);
        hitch[1] = O::template bcast<LHITCH>(e2);
    }
    const V iD1 = O::rcp(O::fma_(R.r[RR_D1], e1, R.r[RR_D0]));
    const V iD2 = O::rcp(O::fma_(R.r[RR_G2], e3, O::fma_(R.r[RR_G1], e2, R.r[RR_G0])));
    const V R1 = O::mul(O::fma_(R.r[RR_N2], e1, O::fma_(R.r[RR_N1], u1, R.r[RR_N0])), iD1);
    const V X2 = O::mul(O::fma_(R.r[RR_M2], e2, O::fma_(R.r[RR_M1], u2, R.r[RR_M0])), iD2);
    const V t0 = O::lt_pick(Vv, R.r[RR_VTH], R.r[RR_T0B], R.r[RR_T0]), s2 = O::lt_pick(Vv, R.r[RR_VTH], R.r[RR_S2B], R.r[RR_S2]),
            s3 = O::lt_pick(Vv, R.r[RR_VTH], R.r[RR_S3B], R.r[RR_S3]);
    const V tau = O::fma_(s3, e3, O::fma_(s2, e2, O::add(t0, X2)));
    const V rit = O::rcp(tau);
    // it = 1: (a, r) = (R1 / tau, 1 / tau); it = 0: (R1, R1 + X2)
    const V it = R.r[RR_IT];
    const V ra = O::add(R1, X2);
    r = O::fma_(it, O::sub(rit, ra), ra);
    a = O::fma_(it, O::sub(O::mul(R1, rit), R1), R1);
    if constexpr (DERIV) {
        const V k1e1 = O::mul(R.r[RR_K1], e1), k2e2 = O::mul(R.r[RR_K2], e2), k3e3 = O::mul(R.r[RR_K3], e3);
        // d R1 = (n1 k1 + n2 k1 e1 - R1 d1 k1 e1) / D1 ; d X2 = (m1 k2 + m2 k2 e2 - X2 (g1 k2 e2 + g2 k3 e3)) / D2
        const V dR1 = O::mul(O::fma_(O::sub(R.r[RR_N2], O::mul(R1, R.r[RR_D1])), k1e1, O::mul(R.r[RR_N1], R.r[RR_K1])), iD1);
        const V dX2 = O::mul(O::sub(O::fma_(R.r[RR_M2], k2e2, O::mul(R.r[RR_M1], R.r[RR_K2])),
                                    O::mul(X2, O::fma_(R.r[RR_G2], k3e3, O::mul(R.r[RR_G1], k2e2)))), iD2);
        const V dtau = O::fma_(s3, k3e3, O::fma_(s2, k2e2, dX2));
        const V drit = O::sub(O::splat(0.0), O::mul(O::mul(rit, rit), dtau));           // d (1 / tau)
        const V dra = O::add(dR1, dX2);
        dr = O::fma_(it, O::sub(drit, dra), dra);
        const V dait = O::fma_(dR1, rit, O::mul(R1, drit));                             // d (R1 / tau)
        da = O::fma_(it, O::sub(dait, dR1), dR1);
    }
}

// Where the states that are not gates live, per model (the gates sit where GroupModel<M>::lanes puts them):
//   LU, LZ, LNG, LQ  lanes of U, Z, ng, Qm;  core_lane(c), c >= 1: lane of core state c (GroupModel's z[c]);
//   LX  lane that evaluates the rate constants the core needs (TC: the O gate of iH), -1: none.
//   DEVICE_STIFF  whether the device library builds the Rosenbrock kernel (full_row_config MODE 2) for the model.
template <class M>
struct RowModel {
    static constexpr int LU = 12, LZ = 13, LNG = 14, LQ = 15, LX = -1;
    static constexpr bool DEVICE_STIFF = true;
    SONIC_HD static constexpr int core_lane(int) { return LQ; }
};
template <>
struct RowModel<ThalamoCortical> {      // gates on lanes 0 1 2 4 5; Cai P0 O C on 8 .. 11; the O rates on lane 6
    static constexpr int LU = 12, LZ = 13, LNG = 14, LQ = 15, LX = 6;
    static constexpr bool DEVICE_STIFF = true;
    SONIC_HD static constexpr int core_lane(int c) { return 7 + c; }
};
template <>
struct RowModel<OtsukaSTN> {            // gates on lanes 0 1 2 4 .. 11; Cai on the free lane of the first quad
    static constexpr int LU = 12, LZ = 13, LNG = 14, LQ = 15, LX = -1;
    static constexpr bool DEVICE_STIFF = true;
    SONIC_HD static constexpr int core_lane(int) { return 3; }
};

// lane descriptions of neuron `id` (model M) for the row kernel, from the group kernel's LaneSpec
template <class M>
bool row_lane_specs(int id, const LaneSpec *gl, RowLaneSpec *rl)
{
    typedef GroupModel<M> GM;
    typedef RowModel<M> RM;
    constexpr int NY = M::NY;
    for (int i = 0; i < GRP; i++) {
        rl[i] = row_lane_none();
        if (gl[i].tab >= 0) {
            if (!row_gate_rate(id, gl[i].tab, rl[i])) return false;
        }
        if (gl[i].colx >= 0) {            // a gate (voltage- or calcium-gated): sonic column + 2 (Z, ng come first)
            rl[i].col = gl[i].colx + 2;
            rl[i].v[RR_ERRW] = 1.0;
            rl[i].v[RR_FLOOR] = FULL_FLOOR_Y;
        }
    }
    if (RM::LX >= 0) {
        if (gl[RM::LX].tab >= 0 || gl[RM::LX].colx >= 0) return false;
        if (!row_gate_rate(id, M::NG, rl[RM::LX])) return false;     // the first rate pair after the gates'
    }
    auto state = [&](int lane, int mask, int col, double floor_) {
        if (gl[lane].colx >= 0 || gl[lane].tab >= 0 || rl[lane].col >= 0) return false;
        rl[lane].v[mask] = 1.0;
        rl[lane].col = col;
        rl[lane].v[RR_ERRW] = 1.0;
        rl[lane].v[RR_FLOOR] = floor_;
        return true;
    };
    bool ok = state(RM::LU, RR_MU, -1, FULL_FLOOR_U) && state(RM::LZ, RR_MZ, 2, FULL_FLOOR_Z) &&
              state(RM::LNG, RR_MNG, 3, 1e-25) && state(RM::LQ, RR_MQ, 4, FULL_FLOOR_Y);
    for (int c = 1; c < GM::NC && ok; c++) ok = state(RM::core_lane(c), RR_MC1 + c - 1, GM::core_col(c) + 2, FULL_FLOOR_Y);
    rl[RM::LU].extra = 1;
    rl[RM::LQ].extra = 2;
    (void)NY;
    return ok;
}

// ---- one evaluation of the right-hand side --------------------------------------------------------------------
// y: one component per lane. pac = acoustic pressure at the time of the evaluation (replicated).
// `live_rate` (may be null): receives the largest rate constant among the gates that are LIVE at this state, for the
// stiffness test of the integrator. A gate with rate r limits an explicit step to ~6 / r only if its equation can
// carry a perturbation: the sodium inactivation gate of the cortical / thalamic neurons reaches r = 1e12 /s in every
// hyperpolarised half period, but sits at h = 1.0 exactly with beta_h = 1e-35 -- its derivative is 0 to the last bit
// and nothing grows (every neuron here integrates explicitly through that). "Live": |a - r x| above the rounding
// level 1e-12 r max(|x|, 1e-6). The gates of STN (tau down to 1e-23 s) lag their moving x_inf by tau dx_inf/dt and are.
// The membrane part of the right-hand side at the potential Vm (replicated): the derivatives fz of the core
// z = (Qm, Ca2+ states ...) and fg of the gates x (one per lane) -- PointNeuron.derivatives (pneuron.py:485-505) with
// the true rate functions. Also the whole right-hand side of the sparse phase of the hybrid scheme, where Vm = Qm / Cm
// at the capacitance of the replayed deflection (hybrid_row.hpp).
template <class O, class M>
SONIC_HD void row_membrane(const typename M::Params &P, const GroupConsts<O> &C, const RowConsts<O> &R, double qdrive,
                           double Vm, const double *z, typename O::V y, double *fz, typename O::V &fg,
                           double *live_rate, double *hitch = nullptr)
{
    typedef typename O::V V;
    typedef GroupModel<M> GM;
    typedef RowModel<M> RM;
    const double Qm = z[0];
    // ---- rate constants, one gate per lane (RowRate form) ----
    V a, r;
    row_rates<O, false, RowModel<M>::LU>(R, Vm, a, r, a, r, hitch);

    // ---- membrane: the group kernel's right-hand side on a "cell" that holds the rates at Vm ----
    GroupCell<O, GM::NX> H;
    H.av = a; H.as = O::splat(0.0);
    H.bv = O::sub(r, a); H.bs = O::splat(0.0);
    H.xlo = Qm; H.xhi = Qm; H.vv = Vm; H.vs = 0.0;
    if constexpr (GM::NX > 0) {
        static_assert(GM::NX == 2 && RM::LX >= 0, "core rate constants: one (alpha, beta) pair on lane LX");
        H.xv[0] = O::template bcast<RM::LX < 0 ? 0 : RM::LX>(a);
        H.xv[1] = O::template bcast<RM::LX < 0 ? 0 : RM::LX>(r) - H.xv[0];
        H.xs[0] = 0.0; H.xs[1] = 0.0;
    }
    GroupRhs<O> G;
    group_rhs<O, GM>(P, H, C, z, y, G);
    const double sQ = O::allsum(G.cur);
    double sC = 0.0;
    if constexpr (GM::HAS_CAI) sC = O::allsum(O::mul(C.kap, G.cur));
    GM::template core<false>(P, H, G.Vm, z, sQ, sC, qdrive, fz, 0.0, 0.0, nullptr);

    if (live_rate) {
        const V lvl = O::mul(O::mul(O::splat(1e-12), G.r), O::max_(O::abs_(y), O::splat(1e-6)));
        double lr_ = O::allmax(O::lt_pick(lvl, O::abs_(G.fg), G.r, O::splat(0.0)));
        if constexpr (GM::NX > 0) {
            // the O / C pair of TC's iH: dC/dt = beta_o O - alpha_o C
            const double fo = H.xv[1] * z[3], fc = H.xv[0] * z[4];
            lr_ = fmax(lr_, fabs(fo - fc) > 1e-12 * (fo + fc) ? H.xv[0] + H.xv[1] : 0.0);
        }
        *live_rate = lr_;
    }

    fg = G.fg;
}

// The right-hand side proper, at U, Z (unclamped), ng, the core z = (Qm, Ca2+ states ...) -- all replicated -- and the
// gates x (one per lane; the other lanes' values do not matter): dU / dt, dng / dt, the core derivatives fz and the
// gate derivatives fg (0 on the lanes without a gate). pac = acoustic pressure at the time of the evaluation.
template <class O, class M>
SONIC_HD void row_eval(const BLSParams &p, const typename M::Params &P, const GroupConsts<O> &C, const RowConsts<O> &R,
                       double fs, double qdrive, double U, double Zraw, double ng, const double *z, typename O::V y,
                       double pac, bool &clamped, double &dU, double &dng, double *fz, typename O::V &fg,
                       double *live_rate)
{
    typedef typename O::V V;
    typedef GroupModel<M> GM;
    typedef RowModel<M> RM;
    constexpr int NC = GM::NC;
    const double Qm = z[0];
    // ---- mechanical system (bls.py:681-718) and capacitance (bls.py:334-345), replicated; one logarithm ----
    const double Zmin = bls::rel_Zmin * p.Delta;
    clamped = clamped || Zraw < Zmin;
    const double Z = Zraw < Zmin ? Zmin : Zraw;
    const double a2 = p.a * p.a;
    const double is = fast_rcp(a2 + Z * Z);
    const double invR = 2.0 * Z * is, ainvR = fabs(invR);
    const double vol = bls::PI * a2 * p.Delta + Z * (bls::PI * a2 + (bls::PI / 3.0) * Z * Z);      // bls.py:311-319
    const double Pg = ng * (bls::Rg * bls::T) * fast_rcp(vol);
    const double den = 2.0 * Z + p.Delta;                       // > 0: Z >= -0.49 Delta
    const double lw = fast_log(den * (1.0 / p.Delta));
    const double lr = fast_log(p.LJ_x0 / p.Delta) - lw;         // (the first term folds to a constant per sonophore)
    // capacitance at the UNclamped deflection, as full_rhs (Z = 0: Cm0)
    double Cm;
    {
        const double Zs = Zraw == 0.0 ? p.Delta : Zraw;
        const double Z2 = (a2 - Zs * Zs - Zs * p.Delta) * fast_rcp(2.0 * Zs);
        const double w = (2.0 * Zs + p.Delta) * (1.0 / p.Delta);
        const double lws = Zraw == Z && Zraw != 0.0 ? lw : (w > 0.0 ? fast_log(w) : NAN);
        Cm = (p.Cm0 * p.Delta * (1.0 / a2)) * (Zs + Z2 * lws);
        Cm = Zraw == 0.0 ? p.Cm0 : Cm;
    }
    const double Ceff = fs * Cm + (1.0 - fs) * p.Cm0;
    const double Vm = Qm * fast_rcp(Ceff) * 1e3;

    // the membrane part; the two Lennard-Jones powers exp(nrep lr), exp(nattr lr) ride on the free lane of its vector
    // exponentials (row_rates: hitch) instead of costing two replicated ones
    double lj[2] = {p.LJ_nrep * lr, p.LJ_nattr * lr};
    row_membrane<O, M>(P, C, R, qdrive, Vm, z, y, fz, fg, live_rate, lj);

    const double Pm = p.LJ_C * (lj[0] - lj[1]);
    const double Pv = -12.0 * U * bls::delta0 * bls::muS * invR * invR - 4.0 * U * bls::muL * ainvR;
    const double PE = -(bls::kA + p.kA_tissue) * (Z * Z * (1.0 / a2)) * invR;
    const double Pel = -(a2 * is) * Qm * Qm * (1.0 / (2.0 * bls::epsilon0 * bls::epsilonR));
    const double Ptot = Pm + Pg - bls::P0 - pac + PE + Pv + Pel;
    dU = Ptot * ainvR * (1.0 / bls::rhoL) - 1.5 * U * U * invR;
    dng = 2.0 * bls::PI * (a2 + Z * Z) * bls::Dgl * (bls::C0 - Pg * (1.0 / bls::kH)) * (1.0 / bls::xi);
}

// the lanes that carry a gate: those with a state that is none of U, Z, ng, the core
template <class O>
SONIC_HD typename O::V row_gate_mask(const RowConsts<O> &R)
{
    typename O::V g = O::sub(R.r[RR_ERRW], O::add(O::add(R.r[RR_MU], R.r[RR_MZ]), O::add(R.r[RR_MNG], R.r[RR_MQ])));
    return O::sub(g, O::add(O::add(R.r[RR_MC1], R.r[RR_MC2]), O::add(R.r[RR_MC3], R.r[RR_MC4])));
}

template <class O, class M>
SONIC_HD typename O::V row_rhs(const BLSParams &p, const typename M::Params &P, const GroupConsts<O> &C,
                               const RowConsts<O> &R, double fs, double qdrive, typename O::V y, double pac,
                               bool &clamped, double *live_rate = nullptr)
{
    typedef typename O::V V;
    typedef GroupModel<M> GM;
    typedef RowModel<M> RM;
    constexpr int NC = GM::NC;
    const double U = O::template bcast<RM::LU>(y), Zraw = O::template bcast<RM::LZ>(y),
                 ng = O::template bcast<RM::LNG>(y), Qm = O::template bcast<RM::LQ>(y);
    double z[NC];
    z[0] = Qm;
    if constexpr (NC > 1) z[1] = O::template bcast<RM::core_lane(1)>(y);
    if constexpr (NC > 2) z[2] = O::template bcast<RM::core_lane(2)>(y);
    if constexpr (NC > 3) z[3] = O::template bcast<RM::core_lane(3)>(y);
    if constexpr (NC > 4) z[4] = O::template bcast<RM::core_lane(4)>(y);

    double dU, dng, fz[NC];
    V fg;
    row_eval<O, M>(p, P, C, R, fs, qdrive, U, Zraw, ng, z, y, pac, clamped, dU, dng, fz, fg, live_rate);

    // ---- the derivative of every lane's component ----
    V dy = O::mul(fg, row_gate_mask<O>(R));            // gates: a - r x (the lane of TC's O rates carries no state)
    dy = O::fma_(R.r[RR_MU], O::splat(dU), dy);
    dy = O::fma_(R.r[RR_MZ], O::splat(U), dy);
    dy = O::fma_(R.r[RR_MNG], O::splat(dng), dy);
    dy = O::fma_(R.r[RR_MQ], O::splat(fz[0]), dy);
    if constexpr (NC > 1) dy = O::fma_(R.r[RR_MC1], O::splat(fz[1]), dy);
    if constexpr (NC > 2) dy = O::fma_(R.r[RR_MC2], O::splat(fz[2]), dy);
    if constexpr (NC > 3) dy = O::fma_(R.r[RR_MC3], O::splat(fz[3]), dy);
    if constexpr (NC > 4) dy = O::fma_(R.r[RR_MC4], O::splat(fz[4]), dy);
    return dy;
}

// ---- the stiff path: RODAS4 on the whole system, on the row -------------------------------------------------
// What full_core.hpp does one configuration per lane (see "The stiff path" there), with the structure of the group
// kernel: the gates are a diagonal that every lane eliminates for itself, bordered by the "extended core"
// E = (U, Z, ng, Qm, Ca2+ states ...) and by the Vm column: every membrane equation sees Z and Qm through
// Vm = Qm / Cm(Z) alone. Everything stays ON THE LANES: the state, the stage increments and the right-hand sides are
// row vectors (one component per lane, as for the explicit pair), and the (3 + NC)^2 Schur complement of the extended
// core is held one ROW PER LANE -- column b is the row vector A[b], whose lane of equation i holds W_ib -- and
// factorised across the lanes: the pivot row goes round by row_newbcast, every lane updates its own row. (The first
// version of this path kept E, its six stage increments and the E x E matrix replicated on every lane: 8 x 8 + 6 x 8
// doubles for TC, a kernel of 256 + 256 registers plus scratch that did not integrate on the device.)
// The Jacobian is analytic throughout: mechanical block from bls_rhs_jac, d (a, r) / d Vm of the generic rate form
// (row_rates), and the current / core derivatives of sonic_group.hpp evaluated on a "cell" whose slopes are those
// d / d Vm (what the effective model differentiates with respect to Q, the detailed one with respect to Vm).
// MECH = false: the membrane system alone (the sparse phase of the hybrid scheme), E = the core, Vm = Qm dVdQ.
template <int I, int N, class F>
SONIC_HD void row_static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        row_static_for<I + 1, N>(f);
    }
}

// equation I of the extended core: its lane in the row, and its 0 / 1 mask among the row constants
template <class M, bool MECH, int I>
struct RowEq {
    typedef RowModel<M> RM;
    static constexpr int lane = MECH ? (I == 0 ? RM::LU : I == 1 ? RM::LZ : I == 2 ? RM::LNG : I == 3 ? RM::LQ
                                                                                       : RM::core_lane(I < 4 ? 1 : I - 3))
                                     : (I == 0 ? RM::LQ : RM::core_lane(I < 1 ? 1 : I));
    static constexpr int mask = (MECH ? RR_MU : RR_MQ) + I;
    static_assert(RR_MZ == RR_MU + 1 && RR_MNG == RR_MU + 2 && RR_MQ == RR_MU + 3 && RR_MC1 == RR_MU + 4 &&
                  RR_MC4 == RR_MU + 7, "masks in the order of the equations");
};

template <class O, class M, bool MECH = true>
struct RowJac {
    typedef typename O::V V;
    static constexpr int NC = GroupModel<M>::NC, M0 = MECH ? 3 : 0, E = M0 + GroupModel<M>::NC;
    V A[E];                  // column b of d f_E / d y_E, row i on the lane of equation i, 0 on the other lanes; then
                             // (row_factor) the LU of the Schur complement of W: L below the diagonal, U on and above
    V dinv;                  // 1 / U_ii on the lane of equation i
    V eidx;                  // i on the lane of equation i, -1 on the other lanes
    V gmask;                 // 1 on the lanes that carry a gate
    V jq, rr, JgV, JgC;      // d (sum of currents) / d gate; rate (-d f_g / d x_g); d f_g / d Vm; d f_g / d Cai (gate lanes)
    V invd, wq;              // 1 / (1 / (h gamma) + r) on the gate lanes, 1 elsewhere; jq invd on the gate lanes, 0 elsewhere
    double dVdZ, dVdQ, fUt;
    double live;             // MECH: the largest rate constant among the gates, for the way back to the explicit pair
};

template <class O, class M, bool MECH>
SONIC_HD void row_jac_lanes(const RowConsts<O> &R, RowJac<O, M, MECH> &J)
{
    typedef typename O::V V;
    constexpr int E = RowJac<O, M, MECH>::E;
    J.gmask = row_gate_mask<O>(R);
    V idx = O::splat(-1.0);
    row_static_for<0, E>([&](auto ic) SONIC_COOP_INLINE {
        constexpr int i = decltype(ic)::value;
        idx = O::fma_(R.r[RowEq<M, MECH, i>::mask], O::splat(1.0 + i), idx);
    });
    J.eidx = idx;
}

// the core states z[c] (replicated) from their lanes
template <class O, class M>
SONIC_HD void row_gather_core(typename O::V y, double *z)
{
    typedef RowModel<M> RM;
    constexpr int NC = GroupModel<M>::NC;
    z[0] = O::template bcast<RM::LQ>(y);
    if constexpr (NC > 1) z[1] = O::template bcast<RM::core_lane(1)>(y);
    if constexpr (NC > 2) z[2] = O::template bcast<RM::core_lane(2)>(y);
    if constexpr (NC > 3) z[3] = O::template bcast<RM::core_lane(3)>(y);
    if constexpr (NC > 4) z[4] = O::template bcast<RM::core_lane(4)>(y);
}

// core derivatives fz[c] (replicated) onto their lanes, added to dy
template <class O, class M>
SONIC_HD typename O::V row_scatter_core(const RowConsts<O> &R, const double *fz, typename O::V dy)
{
    constexpr int NC = GroupModel<M>::NC;
    dy = O::fma_(R.r[RR_MQ], O::splat(fz[0]), dy);
    if constexpr (NC > 1) dy = O::fma_(R.r[RR_MC1], O::splat(fz[1]), dy);
    if constexpr (NC > 2) dy = O::fma_(R.r[RR_MC2], O::splat(fz[2]), dy);
    if constexpr (NC > 3) dy = O::fma_(R.r[RR_MC3], O::splat(fz[3]), dy);
    if constexpr (NC > 4) dy = O::fma_(R.r[RR_MC4], O::splat(fz[4]), dy);
    return dy;
}

// the membrane part with its Jacobian at the potential Vm: core derivatives f0z, d f0z / d (Vm, z_1 ...) in Jzz
// (column 0 = d / d Vm), gate derivatives fg, and the gate parts of J (jq, rr, JgV, JgC)
template <class O, class M, bool MECH>
SONIC_HD void row_membrane_jac(const typename M::Params &P, const GroupConsts<O> &C, const RowConsts<O> &R, double qdrive,
                               double Vm, const double *z, typename O::V xg, double *f0z,
                               double (*Jzz)[GroupModel<M>::NC], typename O::V &fg, RowJac<O, M, MECH> &J)
{
    typedef typename O::V V;
    typedef GroupModel<M> GM;
    typedef RowModel<M> RM;
    V a, r, da, dr;
    row_rates<O, true>(R, Vm, a, r, da, dr);
    GroupCell<O, GM::NX> H;
    H.av = a; H.as = da;
    H.bv = O::sub(r, a); H.bs = O::sub(dr, da);
    H.xlo = z[0]; H.xhi = z[0]; H.vv = Vm; H.vs = 1.0;          // "slopes" = d / d Vm, evaluated at distance 0
    if constexpr (GM::NX > 0) {
        H.xv[0] = O::template bcast<RM::LX < 0 ? 0 : RM::LX>(a);
        H.xv[1] = O::template bcast<RM::LX < 0 ? 0 : RM::LX>(r) - H.xv[0];
        H.xs[0] = O::template bcast<RM::LX < 0 ? 0 : RM::LX>(da);
        H.xs[1] = O::template bcast<RM::LX < 0 ? 0 : RM::LX>(dr) - H.xs[0];
    }
    GroupRhs<O> G;
    group_rhs<O, GM>(P, H, C, z, xg, G);
    // the Jacobian parts of the group kernel's step (integrate_stream_group), d / d Q read as d / d Vm
    const V other = GM::HAS_X2 ? O::mul(G.f1, G.f2) : G.f1;
    const V cond = O::mul(G.gpw, other);
    const double sQ = O::allsum(G.cur);
    double sCond;
    if constexpr (GM::HAS_GHK) sCond = O::allsum(O::mul(cond, G.ddrive));
    else sCond = O::allsum(cond);
    double sC = 0.0, sKCond = 0.0;
    if constexpr (GM::HAS_CAI) {
        sC = O::allsum(O::mul(C.kap, G.cur));
        sKCond = O::allsum(O::mul(C.kap, cond));
    }
    GM::template core<true>(P, H, G.Vm, z, sQ, sC, qdrive, f0z, sCond, sKCond, Jzz);
    const V dpw = O::fma_(xg, O::fma_(xg, O::fma_(xg, C.d4, C.d3), C.d2), C.c1);
    const V own = O::mul(O::mul(O::mul(C.G, dpw), other), G.drive);
    const V gd = O::mul(G.gpw, G.drive);
    V jq = O::fma_(O::swap1(GM::HAS_X2 ? O::mul(gd, G.f2) : gd), C.r1, own);
    if constexpr (GM::HAS_X2) jq = O::fma_(O::swap2(O::mul(gd, G.f1)), C.r2, jq);
    J.jq = O::mul(jq, J.gmask);
    J.rr = G.r;
    J.JgV = O::mul(O::sub(H.as, O::mul(O::add(H.as, H.bs), xg)), J.gmask);
    J.JgC = O::splat(0.0);
    if constexpr (GM::HAS_CAIGATE)
        J.JgC = O::mul(O::mul(O::mul(O::mul(G.xinf, O::sub(G.xinf, O::splat(1.0))), C.ikx), C.itau), J.gmask);
    fg = O::mul(G.fg, J.gmask);
    if constexpr (MECH) {
        // the largest rate constant of ANY gate, live or not (row_membrane's test of liveness keeps the explicit pair
        // from giving up too early; the way back must be safe: a gate of STN that sits on its x_inf with tau = 1e-23 s
        // is not live, and an explicit stage would still throw it off)
        double lr_ = O::allmax(O::mul(G.r, J.gmask));
        if constexpr (GM::NX > 0) lr_ = fmax(lr_, H.xv[0] + H.xv[1]);
        J.live = lr_;
    }
}

// f(t, y) (one component per lane) and its Jacobian
template <class O, class M>
SONIC_HD typename O::V row_rhs_jac(const BLSParams &p, const typename M::Params &P, const GroupConsts<O> &C,
                                   const RowConsts<O> &R, double fs, double qdrive, const MechDrive &d, double t,
                                   typename O::V y, RowJac<O, M, true> &J, bool &clamped)
{
    typedef typename O::V V;
    typedef GroupModel<M> GM;
    typedef RowModel<M> RM;
    constexpr int NC = GM::NC;
    row_jac_lanes<O, M, true>(R, J);
    double ym[3], z[NC];
    ym[0] = O::template bcast<RM::LU>(y); ym[1] = O::template bcast<RM::LZ>(y); ym[2] = O::template bcast<RM::LNG>(y);
    row_gather_core<O, M>(y, z);
    double Jm[3][4], dym[3];
    bls_rhs_jac(p, d, t, ym, z[0], dym, Jm, J.fUt, clamped);
    double Cm, dCm;
    bls_capacitance_d(p, ym[1], Cm, dCm);
    const double Ceff = fs * Cm + (1.0 - fs) * p.Cm0;
    const double Vm = z[0] / Ceff * 1e3;
    J.dVdQ = 1e3 / Ceff;
    J.dVdZ = -Vm / Ceff * fs * dCm;
    double f0z[NC], Jzz[NC][NC];
    V fg;
    row_membrane_jac<O, M, true>(P, C, R, qdrive, Vm, z, y, f0z, Jzz, fg, J);
    // ---- the right-hand side on its lanes ----
    V f0 = fg;
    f0 = O::fma_(R.r[RR_MU], O::splat(dym[0]), f0);
    f0 = O::fma_(R.r[RR_MZ], O::splat(dym[1]), f0);
    f0 = O::fma_(R.r[RR_MNG], O::splat(dym[2]), f0);
    f0 = row_scatter_core<O, M>(R, f0z, f0);
    // ---- d f_E / d y_E, one row per lane: rows U, Z, ng from the mechanical block (columns U, Z, ng, Qm), the core
    //      rows through Vm (columns Z, Qm) and the core states ----
    double jv[NC], jq_[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) { jv[c] = Jzz[c][0] * J.dVdZ; jq_[c] = Jzz[c][0] * J.dVdQ; }
    auto mech_col = [&](int b) SONIC_COOP_INLINE {
        V a_ = O::mul(R.r[RR_MU], O::splat(Jm[0][b]));
        a_ = O::fma_(R.r[RR_MZ], O::splat(Jm[1][b]), a_);
        return O::fma_(R.r[RR_MNG], O::splat(Jm[2][b]), a_);
    };
    J.A[0] = mech_col(0);
    J.A[1] = row_scatter_core<O, M>(R, jv, mech_col(1));
    J.A[2] = mech_col(2);
    J.A[3] = row_scatter_core<O, M>(R, jq_, mech_col(3));
    row_static_for<1, NC>([&](auto ec) SONIC_COOP_INLINE {
        constexpr int e = decltype(ec)::value;
        double col[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) col[c] = Jzz[c][e];
        J.A[3 + e] = row_scatter_core<O, M>(R, col, O::splat(0.0));
    });
    return f0;
}

// W = I c0 - J, c0 = 1 / (h gamma): gates eliminated lane-wise, Schur complement of the extended core factorised
// across the lanes (no pivoting, as group_lu: the diagonal c0 dominates)
template <class O, class M, bool MECH>
SONIC_HD void row_factor(const GroupConsts<O> &C, const RowConsts<O> &R, RowJac<O, M, MECH> &J, double c0)
{
    typedef typename O::V V;
    typedef GroupModel<M> GM;
    constexpr int E = RowJac<O, M, MECH>::E, Q = RowJac<O, M, MECH>::M0;       // Q: row / column of Qm
    const V one = O::splat(1.0);
    const V invd = O::rcp(O::add(O::splat(c0), J.rr));
    J.invd = O::lt_pick(O::splat(0.5), J.gmask, invd, one);       // (a select: 1 + gmask (invd - 1) would round invd ~ h to 1e-16)
    J.wq = O::mul(J.jq, invd);
    row_static_for<0, E>([&](auto bc) SONIC_COOP_INLINE {
        constexpr int b = decltype(bc)::value;
        J.A[b] = O::sub(O::mul(R.r[RowEq<M, MECH, b>::mask], O::splat(c0)), J.A[b]);
    });
    const V wv = O::mul(J.wq, J.JgV);
    V sq = O::mul(R.r[RR_MQ], O::splat(O::allsum(wv)));                // the Schur terms of the rows of Qm and Cai
    if constexpr (GM::HAS_CAI) sq = O::fma_(R.r[RR_MC1], O::splat(O::allsum(O::mul(C.kap, wv))), sq);
    if constexpr (MECH) J.A[1] = O::sub(J.A[1], O::mul(sq, O::splat(J.dVdZ)));
    J.A[Q] = O::sub(J.A[Q], O::mul(sq, O::splat(J.dVdQ)));
    if constexpr (GM::HAS_CAIGATE) {
        const V wc = O::mul(J.wq, J.JgC);
        V sc = O::mul(R.r[RR_MQ], O::splat(O::allsum(wc)));
        sc = O::fma_(R.r[RR_MC1], O::splat(O::allsum(O::mul(C.kap, wc))), sc);
        J.A[Q + 1] = O::sub(J.A[Q + 1], sc);
    }
    J.dinv = one;
    row_static_for<0, E>([&](auto kc) SONIC_COOP_INLINE {
        constexpr int k = decltype(kc)::value;
        constexpr int lk = RowEq<M, MECH, k>::lane;
        const double pinv = fast_rcp1(O::template bcast<lk>(J.A[k]));
        J.dinv = O::lt_pick(O::splat(0.5), R.r[RowEq<M, MECH, k>::mask], O::splat(pinv), J.dinv);
        // l_ik on the lanes of the equations below k (0 elsewhere); the lanes up to k keep their u_ik
        const V lik = O::lt_pick(O::splat((double)k), J.eidx, O::mul(J.A[k], O::splat(pinv)), O::splat(0.0));
        J.A[k] = O::lt_pick(O::splat((double)k), J.eidx, lik, J.A[k]);
        row_static_for<k + 1, E>([&](auto jc) SONIC_COOP_INLINE {
            constexpr int j = decltype(jc)::value;
            J.A[j] = O::sub(J.A[j], O::mul(lik, O::splat(O::template bcast<lk>(J.A[j]))));
        });
    });
}

// W k = r in place (r: one component per lane); kt = the stage's increment of the time variable
template <class O, class M, bool MECH>
SONIC_HD void row_solve(const GroupConsts<O> &C, const RowConsts<O> &R, const RowJac<O, M, MECH> &J, typename O::V &r,
                        double kt)
{
    typedef typename O::V V;
    typedef GroupModel<M> GM;
    typedef RowModel<M> RM;
    constexpr int E = RowJac<O, M, MECH>::E, Q = RowJac<O, M, MECH>::M0;
    if constexpr (MECH) r = O::fma_(R.r[RR_MU], O::splat(J.fUt * kt), r);
    const V wr = O::mul(J.wq, r);
    r = O::fma_(R.r[RR_MQ], O::splat(O::allsum(wr)), r);
    if constexpr (GM::HAS_CAI) r = O::fma_(R.r[RR_MC1], O::splat(O::allsum(O::mul(C.kap, wr))), r);
    // forward (unit lower triangle), then backward
    row_static_for<0, E - 1>([&](auto kc) SONIC_COOP_INLINE {
        constexpr int k = decltype(kc)::value;
        const double xk = O::template bcast<RowEq<M, MECH, k>::lane>(r);
        r = O::lt_pick(O::splat((double)k), J.eidx, O::sub(r, O::mul(J.A[k], O::splat(xk))), r);
    });
    row_static_for<0, E>([&](auto ic) SONIC_COOP_INLINE {
        constexpr int k = E - 1 - decltype(ic)::value;
        r = O::lt_pick(O::splat(0.5), R.r[RowEq<M, MECH, k>::mask], O::mul(r, J.dinv), r);
        if constexpr (k > 0) {
            const double xk = O::template bcast<RowEq<M, MECH, k>::lane>(r);
            // (the lanes without an equation hold 0 in A: unchanged)
            r = O::lt_pick(J.eidx, O::splat((double)k), O::sub(r, O::mul(J.A[k], O::splat(xk))), r);
        }
    });
    // the gates, each on its own lane
    double kv = J.dVdQ * O::template bcast<RM::LQ>(r);
    if constexpr (MECH) kv += J.dVdZ * O::template bcast<RM::LZ>(r);
    V num = O::fma_(J.JgV, O::splat(kv), r);
    if constexpr (GM::HAS_CAIGATE) num = O::fma_(J.JgC, O::splat(O::template bcast<RM::core_lane(1)>(r)), num);
    r = O::mul(num, J.invd);
    (void)Q;
}

// One RODAS4 step attempt from (t, y) with f0 = f(t, y) and J (not yet factorised for this h; consumed).
// F(ts, ys) = the right-hand side on its lanes. On return k[0 .. 4] are the increments the dense output needs,
// k[5] the error estimate, ynew the new state.
template <class O, class M, bool MECH, class RHS>
SONIC_HD void row_rodas4_attempt(RHS &&F, const GroupConsts<O> &C, const RowConsts<O> &R, RowJac<O, M, MECH> &J,
                                 double t, typename O::V y, typename O::V f0, double h, typename O::V &ynew,
                                 typename O::V *k)
{
    using namespace rodas4;
    typedef typename O::V V;
    const double inv_h = 1.0 / h;
    row_factor<O, M, MECH>(C, R, J, inv_h * (1.0 / gamma));
    double kt[6];
    k[0] = f0;
    kt[0] = h * gamma;
    row_solve<O, M, MECH>(C, R, J, k[0], kt[0]);
    V yt = y;
#define ROW_RODAS_STAGE(S, A_EXPR, A_EXPR_T, C_EXPR, C_EXPR_T)                                                    \
    {                                                                                                             \
        yt = O::add(y, A_EXPR);                                                                                   \
        V rs = F(t + (A_EXPR_T), yt);                                                                             \
        kt[S] = h * gamma * (1.0 + inv_h * (C_EXPR_T));                                                           \
        rs = O::fma_(O::splat(inv_h), C_EXPR, rs);                                                                \
        row_solve<O, M, MECH>(C, R, J, rs, kt[S]);                                                                \
        k[S] = rs;                                                                                                \
    }
#define L1(c1) (c1) * kt[0]
#define L2(c1, c2) ((c1) * kt[0] + (c2) * kt[1])
#define L3(c1, c2, c3) ((c1) * kt[0] + (c2) * kt[1] + (c3) * kt[2])
#define L4(c1, c2, c3, c4) ((c1) * kt[0] + (c2) * kt[1] + (c3) * kt[2] + (c4) * kt[3])
#define L5(c1, c2, c3, c4, c5) ((c1) * kt[0] + (c2) * kt[1] + (c3) * kt[2] + (c4) * kt[3] + (c5) * kt[4])
#define G1(c1) O::mul(O::splat(c1), k[0])
#define G2(c1, c2) O::fma_(O::splat(c2), k[1], G1(c1))
#define G3(c1, c2, c3) O::fma_(O::splat(c3), k[2], G2(c1, c2))
#define G4(c1, c2, c3, c4) O::fma_(O::splat(c4), k[3], G3(c1, c2, c3))
#define G5(c1, c2, c3, c4, c5) O::fma_(O::splat(c5), k[4], G4(c1, c2, c3, c4))
    ROW_RODAS_STAGE(1, G1(a21), L1(a21), G1(c21), L1(c21))
    ROW_RODAS_STAGE(2, G2(a31, a32), L2(a31, a32), G2(c31, c32), L2(c31, c32))
    ROW_RODAS_STAGE(3, G3(a41, a42, a43), L3(a41, a42, a43), G3(c41, c42, c43), L3(c41, c42, c43))
    ROW_RODAS_STAGE(4, G4(a51, a52, a53, a54), L4(a51, a52, a53, a54), G4(c51, c52, c53, c54), L4(c51, c52, c53, c54))
    // Y6 = Y5 + k5 (stiffly accurate)
    ROW_RODAS_STAGE(5, G5(a51, a52, a53, a54, 1.0), L5(a51, a52, a53, a54, 1.0), G5(c61, c62, c63, c64, c65),
                    L5(c61, c62, c63, c64, c65))
#undef ROW_RODAS_STAGE
#undef L1
#undef L2
#undef L3
#undef L4
#undef L5
#undef G1
#undef G2
#undef G3
#undef G4
#undef G5
    ynew = O::add(yt, k[5]);              // yt = Y6 after the last stage
}

// error norm of a RODAS4 attempt over the NSTATE components (weights and floors of the lanes: RR_ERRW, RR_FLOOR) and
// the step-size factor of the order-4 controller (Hairer & Wanner IV.7)
template <class O>
SONIC_HD double row_rodas4_error(const RowConsts<O> &R, typename O::V y, typename O::V ynew, typename O::V err,
                                 double rtol, int nstate, double &fac)
{
    typedef typename O::V V;
    const V sc = O::mul(O::splat(rtol), O::max_(O::max_(O::abs_(y), O::abs_(ynew)), R.r[RR_FLOOR]));
    const V e = O::mul(O::mul(err, O::rcp(sc)), R.r[RR_ERRW]);
    const double en = sqrt(O::allsum(O::mul(e, e)) * (1.0 / nstate));
    fac = 0.9 * O::fast_pow(fmax(en, 1e-10), -0.25);
    fac = fmin(6.0, fmax(0.2, fac));
    if (!(en == en)) fac = 0.2;
    return en;
}

// The membrane system (core z, gates) at a frozen capacitance, Vm = Qm kV, over an interval of length `span`: the
// sparse phase of the hybrid scheme (solvers.py:590-633; the reference uses scipy's explicit dop853 -- stiff at high
// amplitudes, see hybrid_coop.hpp: coop_membrane_rodas4). RODAS4 with the exact Jacobian, gates eliminated lane-wise.
// y: one component per lane (the mechanical lanes are left alone). `hs` carries the step size. Returns false if
// the step budget runs out.
template <class O, class M>
SONIC_HD bool row_membrane_rodas4(const typename M::Params &P, const GroupConsts<O> &C, const RowConsts<O> &R,
                                  double qdrive, double kV, double rtol, double span, typename O::V &y, double &hs,
                                  int &nsteps, int max_steps)
{
    typedef typename O::V V;
    typedef GroupModel<M> GM;
    constexpr int NC = GM::NC, NSTATE = M::NY + 1;
    const V mech3 = O::add(O::add(R.r[RR_MU], R.r[RR_MZ]), R.r[RR_MNG]);           // lanes this integrator leaves alone
    const V memb = O::sub(R.r[RR_ERRW], mech3);
    // membrane right-hand side on its lanes (0 on the lanes of U, Z, ng)
    auto F = [&](double, V ys) SONIC_COOP_INLINE {
        double z[NC], fz[NC];
        row_gather_core<O, M>(ys, z);
        V fg;
        row_membrane<O, M>(P, C, R, qdrive, z[0] * kV, z, ys, fz, fg, nullptr);
        return row_scatter_core<O, M>(R, fz, O::mul(fg, row_gate_mask<O>(R)));
    };
    V ycur = y;
    double tcur = 0.0;
    hs = fmin(hs, span);
    while (tcur < span) {
        bool last = false;
        double h = hs;
        if (tcur + 1.0001 * h >= span) { h = span - tcur; last = true; }
        RowJac<O, M, false> J;
        row_jac_lanes<O, M, false>(R, J);
        double z[NC], f0z[NC], Jzz[NC][NC];
        row_gather_core<O, M>(ycur, z);
        V fg, ynew, k[6];
        J.dVdQ = kV; J.dVdZ = 0.0; J.fUt = 0.0;
        row_membrane_jac<O, M, false>(P, C, R, qdrive, z[0] * kV, z, ycur, f0z, Jzz, fg, J);
        const V f0 = row_scatter_core<O, M>(R, f0z, fg);
        row_static_for<0, NC>([&](auto bc) SONIC_COOP_INLINE {
            constexpr int b = decltype(bc)::value;
            double col[NC];
#pragma unroll
            for (int a = 0; a < NC; a++) col[a] = b == 0 ? Jzz[a][0] * kV : Jzz[a][b];
            J.A[b] = row_scatter_core<O, M>(R, col, O::splat(0.0));
        });
        row_rodas4_attempt<O, M, false>(F, C, R, J, 0.0, ycur, f0, h, ynew, k);
        nsteps++;
        double fac;
        const double en = row_rodas4_error<O>(R, ycur, ynew, O::mul(k[5], memb), rtol, NSTATE, fac);
        if (en <= 1.0) {
            ycur = ynew;
            tcur = last ? span : tcur + h;
            hs = h * fac;
        } else {
            hs = h * fmin(fac, 1.0);
        }
        if (nsteps >= max_steps || !(hs > 1e-18)) return false;
    }
    // U, Z, ng as they were
    y = O::fma_(mech3, y, O::mul(memb, ycur));
    return true;
}

#ifndef ROW_ERR_GUARD
#define ROW_ERR_GUARD 1.0
#endif

// ---- Dormand-Prince 8(5,3) on row vectors (stage sums, dense output: dop853_coeffs.hpp, full_coop.hpp) --------
// error norm of Hairer's DOP853 over the NS state components (errw: 1 on the lanes that carry one)
template <class O, class RHS>
SONIC_HD double row_dp8_attempt(RHS &&rhs, typename O::V y, typename O::V *K, double h, typename O::V floor_,
                                typename O::V errw, double rtol, int ns, typename O::V &ynew)
{
    typedef typename O::V V;
    const V hv = O::splat(h);
#define DP8_STAGE(SI) K[SI] = rhs(std::integral_constant<int, SI>{}, O::fma_(hv, dp8::stage_sum<SI, O>(K), y))
    DP8_STAGE(1);
    DP8_STAGE(2);
    DP8_STAGE(3);
    DP8_STAGE(4);
    DP8_STAGE(5);
    DP8_STAGE(6);
    DP8_STAGE(7);
    DP8_STAGE(8);
    DP8_STAGE(9);
    DP8_STAGE(10);
    DP8_STAGE(11);
#undef DP8_STAGE
    ynew = O::fma_(hv, dp8::b_sum<O>(K), y);
    K[12] = rhs(std::integral_constant<int, 12>{}, ynew);
    const V sc = O::mul(O::splat(rtol), O::max_(O::max_(O::abs_(y), O::abs_(ynew)), floor_));
    const V isc = O::mul(errw, O::rcp(sc));                         // sc >= rtol floor > 0; 0 on lanes without a state
    const V r5 = O::mul(dp8::e5_sum<O>(K), isc), r3 = O::mul(dp8::e3_sum<O>(K), isc);
    const V q5 = O::mul(r5, r5), q3 = O::mul(r3, r3);
    const double n5 = O::allsum(q5), n3 = O::allsum(q3);
    const double den = n5 + 0.01 * n3;
    double en = den > 0.0 ? fabs(h) * n5 / sqrt(den * ns) : (den == den ? 0.0 : NAN);
    // No single state more than ROW_ERR_GUARD scales off: the RMS over a dozen states lets one of them miss its own
    // tolerance threefold, and that is what a gate with a DISCONTINUOUS rate function does at every crossing (tau_u of
    // LTS / TC jumps by 20 % at -80 mV, which Vm = Qm / Cm(Z) crosses twice per acoustic period): steps accepted
    // across the jump left u 5 bars off the reference at any tolerance; with the guard the controller closes in on it.
    const V dc = O::fma_(O::splat(0.01), q3, q5);
    const double emax = O::allmax(O::mul(q5, O::rsqrt_pos(dc)));         // per state: r5^2 / sqrt(r5^2 + 0.01 r3^2)
    en = fmax(en, fabs(h) * emax * (1.0 / ROW_ERR_GUARD));
    if (!(n5 == n5) || !(n3 == n3)) en = NAN;
    return en;
}

// stage times of the pair as fractions of the step, by lane: lane SI = c_SI (lane 12: the end of the step)
SONIC_HD double row_stage_fraction(int lane)
{
    switch (lane) {
    case 1: return dp8::c1; case 2: return dp8::c2; case 3: return dp8::c3; case 4: return dp8::c4;
    case 5: return dp8::c5; case 6: return dp8::c6; case 7: return dp8::c7; case 8: return dp8::c8;
    case 9: return dp8::c9; case 10: return dp8::c10; case 11: return 1.0; case 12: return 1.0;
    case 13: return dp8::c13; case 14: return dp8::c14; case 15: return dp8::c15;
    default: return 0.0;
    }
}

// Integrate y from t0 to t1 under the drive amplitude As, calling dense(td, yd) at the points 1 .. ns - 1 of
// np.linspace(t0, t1, ns) in order (coop_integrate_segment of full_coop.hpp on rows). Returns 0, status bit 4 if
// the step budget ran out, FULL_ST_STIFF if the steps collapsed.
// t_stop / i_stop: on entry where to start (time, next dense point: t0 and 1 for a whole segment), on return with
// FULL_ST_STIFF where the pair gave up.
template <class O, class M, class Dense>
SONIC_HD int row_integrate_segment(const BLSParams &p, const typename M::Params &P, const GroupConsts<O> &C,
                                   const RowConsts<O> &R, double fs, double qdrive, double w, double phi, double rtol,
                                   double As, double t0, double t1, int ns, double dt, typename O::V &y,
                                   typename O::V *K, double &h, int &nsteps, int max_steps, bool &clamped, int &iasti,
                                   int &nonsti, double &t_stop, int &i_stop, Dense &&dense)
{
    typedef typename O::V V;
    constexpr int NSTATE = 3 + M::NY;
    const V cS = O::lane_values(row_stage_fraction);
    const Linspace grid = linspace_make(t0, t1, ns);
    bool trial_clamped = false;
    double t = t_stop;
    int i_d = i_stop;
    if (i_d >= ns) return 0;
    double td = linspace_at(grid, i_d);
    // phase of the drive carried from step to step by rotation, re-seeded every 32 steps (see full_coop.hpp)
    double S0 = sin(w * t - phi), C0 = cos(w * t - phi);
    int nseed = 0;
    K[0] = row_rhs<O, M>(p, P, C, R, fs, qdrive, y, As * S0, trial_clamped);      // the amplitude changed: no FSAL
    h = fmin(h, t1 - t);
    while (i_d < ns) {
        bool last = false;
        if (t + 1.0001 * h >= t1) { h = t1 - t; last = true; }
        trial_clamped = false;
        const double wh = w * h;
        const bool small = wh < 0.25;
        // Pac at the stage times, one stage per lane
        V pS;
        if (small) {
            V sd, cd;
            oct_sincos_small<O>(O::mul(cS, O::splat(wh)), sd, cd);
            pS = O::mul(O::splat(As), O::fma_(O::splat(S0), cd, O::mul(O::splat(C0), sd)));
        } else {
            pS = O::mul(O::splat(As), O::sin_(O::sub(O::mul(O::splat(w), O::fma_(cS, O::splat(h), O::splat(t))), O::splat(phi))));
        }
        V ynew;
        double live_rate = 0.0;
        auto rhs = [&](auto si, V yt) SONIC_COOP_INLINE {
            constexpr int SI = decltype(si)::value;
            return row_rhs<O, M>(p, P, C, R, fs, qdrive, yt, O::template bcast<SI>(pS), trial_clamped,
                                 SI == 12 ? &live_rate : nullptr);
        };
        const double en = row_dp8_attempt<O>(rhs, y, K, h, R.r[RR_FLOOR], R.r[RR_ERRW], rtol, NSTATE, ynew);
        const double tnew_ = last ? t1 : t + h;
        if (en <= 1.0 && i_d < ns && (last || td <= tnew_)) coop_dp8_dense_stages<O>(rhs, y, K, h);
        nsteps++;
        double fac = 0.9 * O::fast_pow(fmax(en, 1e-12), -0.125);
        fac = fmin(6.0, fmax(0.2, fac));
        if (!(en == en)) fac = 0.2;
        if (en <= 1.0) {
            clamped = clamped || trial_clamped;
            const double tnew = last ? t1 : t + h;
            if (i_d < ns && (last || td <= tnew)) {
                CoopDense8<O> ext;
                ext.prepare(y, ynew, K, h);
                while (i_d < ns && (last || td <= tnew)) {
                    V yd = ynew;
                    if (td < tnew) yd = ext.at((td - t) / h);
                    dense(td, yd);
                    i_d++;
                    if (i_d < ns) td = linspace_at(grid, i_d);
                }
            }
            // stiffness test (the bookkeeping of DOP853's, Hairer, Norsett, Wanner II.5): steps within a factor of two
            // of the pair's stability limit (h x the largest live rate = 6.1 on the negative real axis) fifteen times
            // without six steps in between that are not -- the steps are then limited by stability, not accuracy
            if (h * live_rate > 3.0) { nonsti = 0; iasti++; }
            else if (iasti > 0 && ++nonsti >= 6) iasti = 0;
            y = ynew;
            K[0] = K[12];
            if (small && ++nseed < 32) {
                const double d = w * (tnew - t), z2 = d * d;
                // sin d, cos d for |d| <= 0.25 (oct_sincos_small, scalar)
                double ps = -1.0 / 39916800.0;
                ps = ps * z2 + 1.0 / 362880.0; ps = ps * z2 - 1.0 / 5040.0; ps = ps * z2 + 1.0 / 120.0; ps = ps * z2 - 1.0 / 6.0;
                const double s_ = ps * z2 * d + d;
                double pc = 1.0 / 479001600.0;
                pc = pc * z2 - 1.0 / 3628800.0; pc = pc * z2 + 1.0 / 40320.0; pc = pc * z2 - 1.0 / 720.0; pc = pc * z2 + 1.0 / 24.0;
                pc = pc * z2 - 0.5;
                const double c_ = pc * z2 + 1.0;
                const double S1 = S0 * c_ + C0 * s_;
                C0 = C0 * c_ - S0 * s_;
                S0 = S1;
            } else {
                S0 = sin(w * tnew - phi); C0 = cos(w * tnew - phi);
                nseed = 0;
            }
            t = tnew;
            h = fmin(h * fac, COOP_HMAX_DENSE * dt);
        } else {
            h *= fmin(fac, 1.0);
        }
        if (iasti >= 15 || !(h > 1e-7 * dt)) { t_stop = t; i_stop = i_d; return FULL_ST_STIFF; }
        if (nsteps >= max_steps) return 4;
    }
    return 0;
}

// The rest of a segment -- from time t_from, next dense point i_from -- on RODAS4 (the stiff path above). y: one
// component per lane, as for the explicit pair. Returns 0 or status bit 4 (step budget, step-size underflow).
// may_leave: the integrator hands the segment back (ROW_ST_LEFT; t_from / i_from then say where) once the largest
// rate constant among the gates has stayed below 1 / (spacing of the dense grid) -- half the step the explicit
// pair takes on the mechanical system -- for ten accepted steps: the gates of TC / RE / STN are stiff in the
// hyperpolarised part of the acoustic period only, and there RODAS4 takes ~9 x the steps of the 8(5,3) pair.
constexpr int ROW_ST_LEFT = 128;      // (internal: never stored as a status)
template <class O, class M, class Dense>
SONIC_HD int row_rodas_segment(const BLSParams &p, const typename M::Params &P, const GroupConsts<O> &C,
                               const RowConsts<O> &R, double fs, double qdrive, double w, double phi, double rtol,
                               double As, double &t_from, int &i_from, double t0, double t1, int ns, typename O::V &y,
                               double &h, int &nsteps, int max_steps, bool &clamped, bool may_leave, Dense &&dense)
{
    typedef typename O::V V;
    constexpr int NSTATE = 3 + M::NY;
    const Linspace grid = linspace_make(t0, t1, ns);
    const MechDrive d{w, As, phi};
    double t = t_from;
    int i_d = i_from;
    double td = i_d < ns ? linspace_at(grid, i_d) : t1;
    h = fmin(h, t1 - t);
    int calm = 0;
    while (i_d < ns) {
        bool last = false;
        if (t + 1.0001 * h >= t1) { h = t1 - t; last = true; }
        bool trial_clamped = false;
        V ynew, k[6];
        RowJac<O, M, true> J;
        const V f0 = row_rhs_jac<O, M>(p, P, C, R, fs, qdrive, d, t, y, J, trial_clamped);
        auto F = [&](double ts, V ys) SONIC_COOP_INLINE {
            return row_rhs<O, M>(p, P, C, R, fs, qdrive, ys, As * sin(w * ts - phi), trial_clamped);
        };
        row_rodas4_attempt<O, M, true>(F, C, R, J, t, y, f0, h, ynew, k);
        nsteps++;
        double fac;
        const double en = row_rodas4_error<O>(R, y, ynew, k[5], rtol, NSTATE, fac);
        if (en <= 1.0) {
            clamped = clamped || trial_clamped;
            const double tnew = last ? t1 : t + h;
            if (i_d < ns && (last || td <= tnew)) {
                // RODAS4's third-order dense output (sonic_integrator.hpp: rodas4_dense)
                using namespace rodas4;
                const V c3 = O::fma_(O::splat(d25), k[4], O::fma_(O::splat(d24), k[3], O::fma_(O::splat(d23), k[2],
                             O::fma_(O::splat(d22), k[1], O::mul(O::splat(d21), k[0])))));
                const V c4 = O::fma_(O::splat(d35), k[4], O::fma_(O::splat(d34), k[3], O::fma_(O::splat(d33), k[2],
                             O::fma_(O::splat(d32), k[1], O::mul(O::splat(d31), k[0])))));
                while (i_d < ns && (last || td <= tnew)) {
                    V yd = ynew;
                    if (td < tnew) {
                        const double sg = (td - t) / h, s1 = 1.0 - sg;
                        const V mid = O::fma_(O::splat(s1), O::fma_(O::splat(sg), c4, c3), ynew);
                        yd = O::fma_(O::splat(sg), mid, O::mul(y, O::splat(s1)));
                    }
                    dense(td, yd);
                    i_d++;
                    if (i_d < ns) td = linspace_at(grid, i_d);
                }
            }
            y = ynew;
            t = tnew;
            h *= fac;
            calm = J.live * grid.step < 1.0 ? calm + 1 : 0;
            if (may_leave && calm >= 10 && i_d < ns) { t_from = t; i_from = i_d; return ROW_ST_LEFT; }
        } else {
            h *= fmin(fac, 1.0);
        }
        if (nsteps >= max_steps || !(h > 1e-18)) return 4;
    }
    return 0;
}

// A segment on the explicit pair, on RODAS4 while its steps are limited by stability (stiff_mode 1; 2: RODAS4
// throughout; 0: the explicit pair alone, which gives a stiff configuration up with FULL_ST_STIFF). `stiff`: which of
// the two the configuration is on, carried from segment to segment.
template <class O, class M, class Dense>
SONIC_HD int row_switching_segment(const BLSParams &p, const typename M::Params &P, const GroupConsts<O> &C,
                                   const RowConsts<O> &R, double fs, double qdrive, double w, double phi, double rtol,
                                   double rtol_stiff, double As, double t0, double t1, int ns, double dt,
                                   typename O::V &y, typename O::V *K, double &h, int &nsteps, int max_steps,
                                   bool &clamped, int &iasti, int &nonsti, bool &stiff, int stiff_mode, Dense &&dense)
{
    double t_from = t0;
    int i_from = 1;
    for (;;) {
        if (!stiff) {
            const int bad = row_integrate_segment<O, M>(p, P, C, R, fs, qdrive, w, phi, rtol, As, t0, t1, ns, dt, y, K, h,
                                                        nsteps, max_steps, clamped, iasti, nonsti, t_from, i_from, dense);
            if (bad != FULL_ST_STIFF || stiff_mode == 0) return bad;
            stiff = true;
            h = fmax(h, 1e-15);
        }
        const int bad = row_rodas_segment<O, M>(p, P, C, R, fs, qdrive, w, phi, rtol_stiff, As, t_from, i_from, t0, t1, ns,
                                                y, h, nsteps, max_steps, clamped, stiff_mode == 1, dense);
        if (bad != ROW_ST_LEFT) return bad;
        stiff = false;
        iasti = 0; nonsti = 0;
        h = fmin(h, dt);
    }
}

// One configuration on the sixteen lanes of a row; flow and resampling as full_coop_config (full_coop.hpp).
// `store`: false for a shadow copy of a configuration (same arithmetic, no stores).
// MODE: which integrators this instance contains -- 0: the explicit pair alone (a configuration that turns stiff is
// given up with FULL_ST_STIFF), 1: both, alternating (row_switching_segment; opts.stiff_mode 2: RODAS4 from the
// start), 2: RODAS4 alone. The device library builds 0 and 1 as separate kernels and restarts the few stiff
// configurations on the second: holding both integrators costs ~400 scalar spills, which the explicit pair would pay
// for in every step of every configuration; the restart costs the microsecond the explicit pair had integrated.
template <class O, class M, int MODE>
SONIC_HD void full_row_config(const FullDev &D, const BLSParams &p, const typename M::Params &P,
                              const LaneSpec *glanes, const RowLaneSpec *rlanes, long long c, bool store)
{
    typedef typename O::V V;
    typedef GroupModel<M> GM;
    typedef RowModel<M> RM;
    constexpr int NCOL = M::NY + 5;              // t stim Z ng Qm states... Vm
    const double f = D.f[c], fs = D.fs[c];
    const double w = 2.0 * bls::PI * f;
    const double dt = 1.0 / (MECH_NPC * f);
    const int max_steps = full_step_budget(D.opts, f, D.tstop[c]);
    int status = 0;
    bool clamped = false;

    GroupConsts<O> C;
    O::load_consts(glanes, C);
    RowConsts<O> R;
    O::load_row_consts(rlanes, R);

    // initial conditions (nbls.py:321-329, bls.py:720-747), as full_config
    const double Pac_dt = D.A[c] * sin(w * dt - D.phi);
    const double Zqs = bls_balancedefQS(p, p.ng0, D.y0[0], Pac_dt);
    if (!(Zqs == Zqs)) status |= 2;
    V y = O::init_gates(D.y0, C.colx);                         // gates (0 elsewhere)
    y = O::fma_(R.r[RR_MZ], O::splat(Zqs), y);
    y = O::fma_(R.r[RR_MNG], O::splat(p.ng0), y);
    y = O::fma_(R.r[RR_MQ], O::splat(D.y0[0]), y);
    if constexpr (GM::NC > 1) y = O::fma_(R.r[RR_MC1], O::splat(D.y0[GM::core_col(1) - 2]), y);
    if constexpr (GM::NC > 2) y = O::fma_(R.r[RR_MC2], O::splat(D.y0[GM::core_col(2) - 2]), y);
    if constexpr (GM::NC > 3) y = O::fma_(R.r[RR_MC3], O::splat(D.y0[GM::core_col(3) - 2]), y);
    if constexpr (GM::NC > 4) y = O::fma_(R.r[RR_MC4], O::splat(D.y0[GM::core_col(4) - 2]), y);

    const long long s0 = D.seg_off[c];
    const int nseg = (int)(D.seg_off[c + 1] - s0);
    const long long M_rows = D.row_off[c + 1] - D.row_off[c];
    double *rows = D.traces + D.row_off[c] * NCOL;
    const Linspace out = linspace_make(0.0, D.tstop[c], (int)M_rows);
    long long j = 0;
    double tau = linspace_at(out, 0);
    double tp = 0.0;
    V yp = y;
    int nsteps = 0, iasti = 0, nonsti = 0;

    auto consume = [&](double ti, V yi, double xs) {
        while (j < M_rows && tau <= ti) {
            V r = yi;
            if (ti > tp) {
                const V wgt = O::splat((tau - tp) / (ti - tp));
                r = O::fma_(O::sub(yi, yp), wgt, yp);                           // np.interp
            }
            // Vm from the RESAMPLED Qm and Z (nbls.py:317-319, 349-351)
            const double Zr = O::template bcast<RM::LZ>(r), Qr = O::template bcast<RM::LQ>(r);
            const double Vm = Qr / (fs * bls_capacitance(p, Zr) + (1.0 - fs) * p.Cm0) * 1e3;
            if (store) O::store_full_row(rows + j * NCOL, R, NCOL, tau, (j == 0) ? 0.0 : xs, r, Vm);
            j++;
            if (j < M_rows) tau = linspace_at(out, (int)j);
        }
        tp = ti;
        yp = yi;
    };

    V K[16];                                  // stage derivatives; K[0] = f(t, y) (first same as last)
    double h = 0.25 * dt;
    // explicit pair until its steps turn out to be limited by stability, then RODAS4 for the rest of the
    // configuration (stiff_mode 0: explicit pair only -- the configuration is given up with FULL_ST_STIFF)
    bool stiff = MODE == 2 || (MODE == 1 && D.opts.stiff_mode == 2);
    for (int s = 0; s < nseg && !(status & (6 | FULL_ST_STIFF)); s++) {
        const double t0 = D.seg_t0[s0 + s], t1 = D.seg_t1[s0 + s], xs = D.seg_x[s0 + s];
        const int ns = D.seg_n[s0 + s];
        const double As = D.A[c] * xs;                    // eventfunc: drive.xvar * x (nbls.py:337)
        consume(t0, y, xs);                               // first dense row of the segment (duplicate)
        if (!(t1 > t0)) { consume(t1, y, xs); continue; }
        auto dense = [&](double td, V yd) SONIC_COOP_INLINE { consume(td, yd, xs); };
        int bad = 0;
        if constexpr (MODE == 0) {
            double t_from = t0;
            int i_from = 1;
            bad = row_integrate_segment<O, M>(p, P, C, R, fs, D.opts.qdrive, w, D.phi, D.opts.rtol, As, t0, t1, ns, dt, y, K, h,
                                              nsteps, max_steps, clamped, iasti, nonsti, t_from, i_from, dense);
        } else if constexpr (MODE == 1) {
            bad = row_switching_segment<O, M>(p, P, C, R, fs, D.opts.qdrive, w, D.phi, D.opts.rtol, D.opts.rtol_stiff, As, t0,
                                              t1, ns, dt, y, K, h, nsteps, max_steps, clamped, iasti, nonsti, stiff,
                                              D.opts.stiff_mode, dense);
        } else {
            double t_from = t0;
            int i_from = 1;
            bad = row_rodas_segment<O, M>(p, P, C, R, fs, D.opts.qdrive, w, D.phi, D.opts.rtol_stiff, As, t_from, i_from, t0, t1,
                                          ns, y, h, nsteps, max_steps, clamped, false, dense);
        }
        if (bad) { status |= bad; break; }
    }
    // rows not produced (failed configuration): NaN
    for (; j < M_rows; j++)
        if (store) O::fill_full_row_nan(rows + j * NCOL, R, NCOL, linspace_at(out, (int)j));
    if (clamped) status |= 1;
    if (store && O::leader()) {
        D.status[c] = status;
        D.nsteps[c] = nsteps;
    }
}

}  // namespace sonic
