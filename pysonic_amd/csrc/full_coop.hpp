// pysonic_amd/csrc/full_coop.hpp
//
// OCTET-COOPERATIVE integration of the detailed NICE model (method='full') of the cortical RS / FS
// neurons: one configuration per group of 8 adjacent lanes instead of one per lane.
//
// Reference: NeuronalBilayerSonophore.__simFull / fullDerivatives (PySONIC/core/nbls.py:265-278,
// 331-354) = BilayerSonophore.derivatives (bls.py:681-718) coupled to PointNeuron.derivatives
// (pneuron.py:485-505) with the true rate functions (neurons/cortical.py:36-66), integrated on a
// dense grid of 1000 points per acoustic period and resampled to 10 ns (solvers.py:184-191,213-221).
// Same equations, integrator (DOPRI5 with dense output) and on-the-fly resampling as
// full_core.hpp; what changes is WHO computes what.
//
// Why: a configuration is a strictly sequential chain of ~3e6 steps per simulated millisecond, and a
// batch is a few hundred configurations (BASELINE config 5: 256). One lane per configuration
// leaves the GPU empty and makes every step the latency of ~4700 dependent FP64 instructions issued
// for a single lane (44 us per step measured in round 1 -- a host core runs the same code in 1 us).
// The right-hand side has eight state components and about a dozen expensive sub-expressions (six
// divisions, two logarithms, ten exponentials) that are independent given (Z, Qm, Vm):
//
//   lane   owns      phase A (division, log, exp)                      phase B (one rate each)
//   0      U         1/R = 2 Z / (a^2 + Z^2)  -> elastic + viscous      beta_m
//   1      Z         w = (2 Z + D) / D, log w -> capacitance            beta_h
//   2      ng        Pg = ng Rg T / V(Z)      -> gas pressure, flux     beta_n
//   3      Qm        Z2 = (a^2 - Z^2 - Z D) / (2 Z) -> Cm -> Vm         beta_p
//   4      m         a^2 / (a^2 + Z^2)        -> electrical pressure    alpha_m
//   5      h         r = x0 / (2 Z + D), exp(nrep log r)   -> LJ rep.   alpha_h
//   6      n         r,                  exp(nattr log r)  -> LJ attr.  alpha_n
//   7      p         -P0 - Pac(t)                                       alpha_p
//
// Every lane executes the SAME instruction stream (one division, one log, one exp, one rational
// function of an exponential ...) on its own operands, selected by per-lane constants; the octet
// exchanges values with DPP moves (register to register): broadcasts of Z, Qm and Vm, two all-reduce
// sums (net pressure, net current) and a shift by four lanes (beta_x to the lane of gate x). A
// right-hand side is ~270 wavefront instructions instead of ~750, the stage vectors are one double per
// lane (no private memory), and the acoustic pressure at the six stage times is evaluated once per
// step, one stage per lane.
//
// Written once over an `Ops` backend like sonic_quad.hpp: on the device an octet vector is one double
// per lane (OctOpsDev, DPP); the CPU test harness uses 8-element arrays (OctOpsHost, development).
#pragma once
#include "full_core.hpp"

namespace sonic {

constexpr int OCT = 8;

// ---- CPU emulation backend: V = 8 values, one per lane of the octet ---------------------------
struct OctOpsHost {
    struct V {
        double v[OCT];
    };
    struct VF {
        float v[OCT];
    };
    static V splat(double a) { V r; for (int i = 0; i < OCT; i++) r.v[i] = a; return r; }
    static V roles(double a0, double a1, double a2, double a3, double a4, double a5, double a6, double a7)
    {
        return V{{a0, a1, a2, a3, a4, a5, a6, a7}};
    }
#define OCT_UN(name, expr) static V name(V a) { V r; for (int i = 0; i < OCT; i++) { const double x = a.v[i]; r.v[i] = (expr); } return r; }
#define OCT_BIN(name, expr) static V name(V a, V b) { V r; for (int i = 0; i < OCT; i++) { const double x = a.v[i], y = b.v[i]; r.v[i] = (expr); } return r; }
    OCT_BIN(add, x + y)
    OCT_BIN(sub, x - y)
    OCT_BIN(mul, x * y)
    OCT_BIN(div, x / y)
    OCT_BIN(max_, x > y ? x : y)
    OCT_UN(neg, -x)
    OCT_UN(abs_, fabs(x))
    OCT_UN(exp_, exp(x))
    OCT_UN(log_, log(x))
    OCT_UN(sin_, sin(x))
#undef OCT_UN
#undef OCT_BIN
    static V fma_(V a, V b, V c) { V r; for (int i = 0; i < OCT; i++) r.v[i] = a.v[i] * b.v[i] + c.v[i]; return r; }
    // r[i] = c[i] != 0 ? a[i] : b[i]   (c: per-lane 0 / 1 constant)
    static V pick(V c, V a, V b) { V r; for (int i = 0; i < OCT; i++) r.v[i] = c.v[i] != 0.0 ? a.v[i] : b.v[i]; return r; }
    static V lt_pick(V x, V y, V a, V b) { V r; for (int i = 0; i < OCT; i++) r.v[i] = x.v[i] < y.v[i] ? a.v[i] : b.v[i]; return r; }
    static V eq0_pick(V x, V a, V b) { V r; for (int i = 0; i < OCT; i++) r.v[i] = x.v[i] == 0.0 ? a.v[i] : b.v[i]; return r; }
    template <int SRC>
    static V bcast(V a) { return splat(a.v[SRC]); }
    // lanes 4..7 receive the value of lanes 0..3 (lanes 0..3: unspecified)
    static V shr4(V a) { V r = a; for (int i = 4; i < OCT; i++) r.v[i] = a.v[i - 4]; return r; }
    static V swap1(V a) { V r; for (int i = 0; i < OCT; i++) r.v[i] = a.v[i ^ 1]; return r; }
    static V allsum(V a)
    {
        V b, c, d;
        for (int i = 0; i < OCT; i++) b.v[i] = a.v[i] + a.v[i ^ 1];
        for (int i = 0; i < OCT; i++) c.v[i] = b.v[i] + b.v[i ^ 2];
        for (int i = 0; i < OCT; i++) d.v[i] = c.v[i] + c.v[7 - i];
        return d;
    }
    static double first(V a) { return a.v[0]; }          // a value known to be replicated
    static bool any_lt(V x, V y) { for (int i = 0; i < OCT; i++) if (x.v[i] < y.v[i]) return true; return false; }
    static int lane() { return -1; }
    // store a.v[i] to base[col[i]] for the lanes with col >= 0
    static void scatter(double *base, const int (&col)[OCT], V a)
    {
        for (int i = 0; i < OCT; i++)
            if (col[i] >= 0) base[col[i]] = a.v[i];
    }
    static void store_lane(double *addr, V a, int lane_) { *addr = a.v[lane_]; }
    static double fast_pow_m02(double en) { return exp(-0.2 * log(en)); }
};

#if defined(__HIPCC__)
// ---- device backend: V = one double per lane; lane & 7 = component index ----------------------
struct OctOpsDev {
    typedef double V;
    template <int CTRL>
    static __device__ __forceinline__ double dpp(double x)
    {
        // all eight lanes of an octet are active whenever the octet is: no `old` value is needed
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
        return __hiloint2double(hi, lo);
    }
    static __device__ __forceinline__ int lane() { return threadIdx.x & 7; }
    static __device__ __forceinline__ V splat(double a) { return a; }
    static __device__ __forceinline__ V roles(double a0, double a1, double a2, double a3, double a4,
                                              double a5, double a6, double a7)
    {
        const int l = lane();
        const double lo = (l & 2) ? ((l & 1) ? a3 : a2) : ((l & 1) ? a1 : a0);
        const double hi = (l & 2) ? ((l & 1) ? a7 : a6) : ((l & 1) ? a5 : a4);
        return (l & 4) ? hi : lo;
    }
    static __device__ __forceinline__ V add(V a, V b) { return a + b; }
    static __device__ __forceinline__ V sub(V a, V b) { return a - b; }
    static __device__ __forceinline__ V mul(V a, V b) { return a * b; }
    static __device__ __forceinline__ V div(V a, V b) { return qdiv(a, b); }
    static __device__ __forceinline__ V max_(V a, V b) { return fmax(a, b); }
    static __device__ __forceinline__ V neg(V a) { return -a; }
    static __device__ __forceinline__ V abs_(V a) { return fabs(a); }
    static __device__ __forceinline__ V exp_(V a) { return exp(a); }
    static __device__ __forceinline__ V log_(V a) { return log(a); }
    static __device__ __forceinline__ V sin_(V a) { return sin(a); }
    static __device__ __forceinline__ V fma_(V a, V b, V c) { return fma(a, b, c); }
    static __device__ __forceinline__ V pick(V c, V a, V b) { return c != 0.0 ? a : b; }
    static __device__ __forceinline__ V lt_pick(V x, V y, V a, V b) { return x < y ? a : b; }
    static __device__ __forceinline__ V eq0_pick(V x, V a, V b) { return x == 0.0 ? a : b; }
    // broadcast of lane SRC of the octet: quad_perm broadcast inside the source's quad, then the
    // other quad fetches it with a shift by four lanes inside the row of 16
    template <int SRC>
    static __device__ __forceinline__ V bcast(V a)
    {
        constexpr int q = SRC & 3;
        const double inq = dpp<q | (q << 2) | (q << 4) | (q << 6)>(a);       // quad_perm [q,q,q,q]
        const int l = lane();
        if (SRC < 4) {
            const double up = dpp<0x114>(inq);                                 // row_shr:4 -> lanes 4..7 <- 0..3
            return (l & 4) ? up : inq;
        } else {
            const double dn = dpp<0x104>(inq);                                 // row_shl:4 -> lanes 0..3 <- 4..7
            return (l & 4) ? inq : dn;
        }
    }
    static __device__ __forceinline__ V shr4(V a) { return dpp<0x114>(a); }
    static __device__ __forceinline__ V swap1(V a) { return dpp<0xB1>(a); }   // [1,0,3,2]
    // all lanes must add the same two rounded numbers at every stage so that the replicated result
    // is bit-identical across the octet (see QuadOpsDev::allsum): pin the operand first
    static __device__ __forceinline__ V allsum(V a)
    {
        asm volatile("" : "+v"(a));
        a += dpp<0xB1>(a);      // + neighbour              [1,0,3,2]
        asm volatile("" : "+v"(a));
        a += dpp<0x4E>(a);      // + other pair             [2,3,0,1]
        asm volatile("" : "+v"(a));
        a += dpp<0x141>(a);     // + other quad             row_half_mirror: lane i <- lane 7 - i
        return a;
    }
    static __device__ __forceinline__ double first(V a) { return a; }
    static __device__ __forceinline__ void scatter(double *base, const int (&)[OCT], V a, int col)
    {
        if (col >= 0) base[col] = a;
    }
    static __device__ __forceinline__ double fast_pow_m02(double en)
    {
        // en^(-1/5) for the step-size controller: single-precision hardware log2 / exp2
        return (double)__builtin_amdgcn_exp2f(-0.2f * __builtin_amdgcn_logf((float)en));
    }
};
#endif

// Per-lane constants of the octet for one sonophore + cortical neuron.
template <class O>
struct CoopConsts {
    typedef typename O::V V;
    // phase A: N = n0 + n1 Zc + n2 Zs + n3 Zs^2 + n4 y_own ; D = d0 + d1 Zc + d2 Zs + d3 Zc^2 + d4 Vol
    V n0, n1, n2, n3, n4, d0, d1, d2, d3, d4;
    V uselog, cexp, tE;        // log argument mask, exponent of exp(cexp * log), coefficient of the LJ term
    V is0, is1, is2, is3, is4, isgate;      // lane predicates as 0 / 1
    // phase B: u = (Vm - vc) vs ; num = a0 + a1 u + e (a2 + a4 e^2) + a3 e^2 ; den = b0 + b1 e + b2 e^2 + b3 e^3
    V vc, vs, a0, a1, a2, a3, a4, b0, b1, b2, b3, K;
    // currents (lanes 4..7), as in sonic_quad.hpp: term = G pw(x) other (Vm - E)
    V G, E, c0, c1, c3, c4, nc3;
    // DOPRI5 error floors and the sine argument offsets of the six stage times
    V floor_, cstage;
};

// rate functions of cortical.py:36-66 (RS: VT = -56.2 mV, TauMax = 0.608 s; FS: -57.9, 0.502) in the
// generic form K num(u, e) / den(e), e = exp(u), u = (Vm - vc) vs:
//   vtrap(x, y) = x / (exp(x / y) - 1) = y u / (e - 1) with u = x / y
//   p: pinf = 1 / (1 + exp(-(Vm + 35) / 10)), taup = TauMax / (3.3 exp((Vm + 35) / 20) + exp(-(Vm + 35) / 20));
//      with e = exp((Vm + 35) / 20):  alpha_p = pinf / taup = e (3.3 e^2 + 1) / ((e^2 + 1) TauMax),
//      beta_p = (1 - pinf) / taup = (3.3 e^2 + 1) / (e (e^2 + 1) TauMax)
template <class O>
SONIC_HD CoopConsts<O> coop_consts(const BLSParams &p, const CorticalParams &P, int neuron, double qdrive)
{
    CoopConsts<O> C;
    const double a2 = p.a * p.a;
    const double VT = neuron == 0 ? -56.2 : -57.9, TauMax = neuron == 0 ? 0.608 : 0.502;
    //                 l0        l1        l2              l3        l4     l5       l6       l7
    C.n0 = O::roles(0.0,      p.Delta,  0.0,            a2,       a2,    p.LJ_x0, p.LJ_x0, 1.0);
    C.n1 = O::roles(2.0,      0.0,      0.0,            0.0,      0.0,   0.0,     0.0,     0.0);
    C.n2 = O::roles(0.0,      2.0,      0.0,            -p.Delta, 0.0,   0.0,     0.0,     0.0);
    C.n3 = O::roles(0.0,      0.0,      0.0,            -1.0,     0.0,   0.0,     0.0,     0.0);
    C.n4 = O::roles(0.0,      0.0,      bls::Rg * bls::T, 0.0,    0.0,   0.0,     0.0,     0.0);
    C.d0 = O::roles(a2,       p.Delta,  0.0,            0.0,      a2,    p.Delta, p.Delta, 1.0);
    C.d1 = O::roles(0.0,      0.0,      0.0,            0.0,      0.0,   2.0,     2.0,     0.0);
    C.d2 = O::roles(0.0,      0.0,      0.0,            2.0,      0.0,   0.0,     0.0,     0.0);
    C.d3 = O::roles(1.0,      0.0,      0.0,            0.0,      1.0,   0.0,     0.0,     0.0);
    C.d4 = O::roles(0.0,      0.0,      1.0,            0.0,      0.0,   0.0,     0.0,     0.0);
    C.uselog = O::roles(0, 1, 0, 0, 0, 1, 1, 0);
    C.cexp = O::roles(0, 0, 0, 0, 0, p.LJ_nrep, p.LJ_nattr, 0);
    C.tE = O::roles(0, 0, 0, 0, 0, p.LJ_C, -p.LJ_C, 0);
    C.is0 = O::roles(1, 0, 0, 0, 0, 0, 0, 0);
    C.is1 = O::roles(0, 1, 0, 0, 0, 0, 0, 0);
    C.is2 = O::roles(0, 0, 1, 0, 0, 0, 0, 0);
    C.is3 = O::roles(0, 0, 0, 1, 0, 0, 0, 0);
    C.is4 = O::roles(0, 0, 0, 0, 1, 0, 0, 0);
    C.isgate = O::roles(0, 0, 0, 0, 1, 1, 1, 1);
    // rates: lanes 0..3 = beta_m beta_h beta_n beta_p ; lanes 4..7 = alpha_m alpha_h alpha_n alpha_p
    //   beta_m  = 0.28e3 vtrap(v - 40, 5)          u = (Vm - (VT + 40)) / 5,    K = 0.28e3 * 5
    //   beta_h  = 4e3 / (1 + exp(-(v - 40) / 5))   u = -(Vm - (VT + 40)) / 5,   K = 4e3
    //   beta_n  = 0.5e3 exp(-(v - 10) / 40)        u = -(Vm - (VT + 10)) / 40,  K = 0.5e3
    //   alpha_m = 0.32e3 vtrap(13 - v, 4)          u = -(Vm - (VT + 13)) / 4,   K = 0.32e3 * 4
    //   alpha_h = 0.128e3 exp(-(v - 17) / 18)      u = -(Vm - (VT + 17)) / 18,  K = 0.128e3
    //   alpha_n = 0.032e3 vtrap(15 - v, 5)         u = -(Vm - (VT + 15)) / 5,   K = 0.032e3 * 5
    C.vc = O::roles(VT + 40.0, VT + 40.0, VT + 10.0, -35.0, VT + 13.0, VT + 17.0, VT + 15.0, -35.0);
    C.vs = O::roles(1.0 / 5.0, -1.0 / 5.0, -1.0 / 40.0, 1.0 / 20.0, -1.0 / 4.0, -1.0 / 18.0, -1.0 / 5.0, 1.0 / 20.0);
    //               bm   bh   bn   bp   am   ah   an   ap
    C.a0 = O::roles(0.0, 1.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0);
    C.a1 = O::roles(1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 1.0, 0.0);
    C.a2 = O::roles(0.0, 0.0, 1.0, 0.0, 0.0, 1.0, 0.0, 1.0);
    C.a3 = O::roles(0.0, 0.0, 0.0, 3.3, 0.0, 0.0, 0.0, 0.0);
    C.a4 = O::roles(0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 3.3);
    C.b0 = O::roles(-1.0, 1.0, 1.0, 0.0, -1.0, 1.0, -1.0, 1.0);
    C.b1 = O::roles(1.0, 1.0, 0.0, 1.0, 1.0, 0.0, 1.0, 0.0);
    C.b2 = O::roles(0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0);
    C.b3 = O::roles(0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0);
    C.K = O::roles(0.28e3 * 5.0, 4e3, 0.5e3, 1.0 / TauMax, 0.32e3 * 4.0, 0.128e3, 0.032e3 * 5.0, 1.0 / TauMax);
    // currents (cortical.py:92-119): m lane iNa = gNa m^3 h (V - ENa), h lane iLeak, n lane iKd = gKd n^4
    // (V - EK), p lane iM = gM p (V - EK); the injected current is folded into the leak reversal potential
    const double ELeak = qdrive != 0.0 ? P.ELeak + qdrive / (1e-3 * P.gLeak) : P.ELeak;
    C.G = O::roles(0, 0, 0, 0, -1e-3 * P.gNabar, -1e-3 * P.gLeak, -1e-3 * P.gKdbar, -1e-3 * P.gMbar);
    C.E = O::roles(0, 0, 0, 0, P.ENa, ELeak, P.EK, P.EK);
    C.c0 = O::roles(0, 0, 0, 0, 0, 1, 0, 0);
    C.c1 = O::roles(0, 0, 0, 0, 0, 0, 0, 1);
    C.c3 = O::roles(0, 0, 0, 0, 1, 0, 0, 0);
    C.c4 = O::roles(0, 0, 0, 0, 0, 0, 1, 0);
    C.nc3 = O::roles(1, 1, 1, 1, 0, 1, 1, 1);
    C.floor_ = O::roles(FULL_FLOOR_U, 1e-13, 1e-25, FULL_FLOOR_Y, FULL_FLOOR_Y, FULL_FLOOR_Y, FULL_FLOOR_Y, FULL_FLOOR_Y);
    C.cstage = O::roles(0.0, dp5::c2, dp5::c3, dp5::c4, dp5::c5, 1.0, 1.0, 1.0);
    return C;
}

template <class O>
struct CoopScalars {
    double a2, inv_a2, inv_3D, volk, Zmin, Delta, Cm0, kC, fs, kE, kel, inv_rho, kng, qdrive;
};

// dy/dt of the octet's eight components (y: one component per lane). `pac` = acoustic pressure at
// the time of this evaluation (replicated). Returns true if the deflection had to be clamped.
template <class O>
SONIC_HD typename O::V coop_rhs(const CoopConsts<O> &C, const CoopScalars<O> &S, typename O::V y,
                                typename O::V pac, bool &clamped)
{
    typedef typename O::V V;
    const V Zb = O::template bcast<1>(y), Qb = O::template bcast<3>(y);
    const V Zmin = O::splat(S.Zmin);
    if (O::any_lt(Zb, Zmin)) clamped = true;
    const V Zc = O::max_(Zb, Zmin);                                   // bls.py:694-696
    const V Zs = O::eq0_pick(Zb, O::splat(S.Delta), Zb);              // capacitance: harmless Z where Z = 0
    const V Zc2 = O::mul(Zc, Zc), Zs2 = O::mul(Zs, Zs);
    // V(Z) = pi a^2 D (1 + (Z / (3 D)) (3 + Z^2 / a^2))        (bls.py:311-319)
    const V vol = O::mul(O::splat(S.volk), O::fma_(O::mul(Zc, O::splat(S.inv_3D)),
                                                   O::fma_(Zc2, O::splat(S.inv_a2), O::splat(3.0)), O::splat(1.0)));
    // phase A: one division per lane
    V N = O::fma_(C.n1, Zc, C.n0);
    N = O::fma_(C.n2, Zs, N);
    N = O::fma_(C.n3, Zs2, N);
    N = O::fma_(C.n4, y, N);
    V D = O::fma_(C.d1, Zc, C.d0);
    D = O::fma_(C.d2, Zs, D);
    D = O::fma_(C.d3, Zc2, D);
    D = O::fma_(C.d4, vol, D);
    const V q = O::div(N, D);
    // one logarithm (lanes 1, 5, 6), one exponential (lanes 5, 6: the Lennard-Jones powers, bls.py:29-41,472-480)
    const V L = O::log_(O::pick(C.uselog, q, O::splat(1.0)));
    const V Ex = O::exp_(O::mul(C.cexp, L));
    // capacitance and potential on lane 3 (bls.py:334-345, nbls.py:148-151): Cm = Cm0 D / a^2 (Z + Z2 log w)
    const V Lw = O::template bcast<1>(L);
    V Cm = O::mul(O::splat(S.kC), O::fma_(q, Lw, Zs));
    Cm = O::eq0_pick(Zb, O::splat(S.Cm0), Cm);
    const V Cme = O::fma_(O::splat(S.fs), Cm, O::splat((1.0 - S.fs) * S.Cm0));
    const V Vm = O::template bcast<3>(O::mul(O::div(Qb, Cme), O::splat(1e3)));
    // pressure terms, one per lane (bls.py:596-655, 482-491), summed over the octet
    //   lane 0: PE + Pv = -(kA + kA_tissue) (Z / a)^2 / R - 12 U delta0 muS / R^2 - 4 U muL / |R|
    //   lane 1: -P0 - Pac     lane 2: Pg     lane 4: Pelec = -a^2 / (a^2 + Z^2) Qm^2 / (2 eps0 epsR)
    //   lanes 5, 6: +- C r^n
    const V Ub = O::template bcast<0>(y);
    const V t0 = O::sub(O::mul(q, O::fma_(O::splat(-S.kE), Zc2, O::mul(O::splat(-12.0 * bls::delta0 * bls::muS), O::mul(Ub, q)))),
                        O::mul(O::splat(4.0 * bls::muL), O::mul(Ub, O::abs_(q))));
    const V t1 = O::sub(O::splat(-bls::P0), pac);
    const V t4 = O::mul(O::mul(q, O::splat(-S.kel)), O::mul(Qb, Qb));
    V T = O::mul(C.tE, Ex);
    T = O::pick(C.is0, t0, T);
    T = O::pick(C.is1, t1, T);
    T = O::pick(C.is2, q, T);
    T = O::pick(C.is4, t4, T);
    const V Ptot = O::allsum(T);
    // phase B: one rate constant per lane
    const V u = O::mul(O::sub(Vm, C.vc), C.vs);
    const V e = O::exp_(u);
    const V e2 = O::mul(e, e);
    V num = O::fma_(C.a1, u, C.a0);
    num = O::fma_(C.a3, e2, num);
    num = O::fma_(e, O::fma_(C.a4, e2, C.a2), num);
    V den = O::fma_(C.b1, e, C.b0);
    den = O::fma_(O::fma_(C.b3, e, C.b2), e2, den);
    const V rate = O::mul(C.K, O::div(num, den));
    // phase C: gates (lanes 4..7: alpha is the lane's own rate, beta comes from four lanes below)
    const V beta = O::shr4(rate);
    const V fgate = O::sub(rate, O::mul(O::add(rate, beta), y));        // alpha - (alpha + beta) x
    // currents (pneuron.py:288-296): lanes 4..7, zero elsewhere (G = 0)
    const V xo = O::swap1(y);                                          // the m lane needs h
    const V x2 = O::mul(y, y);
    const V pw = O::fma_(x2, O::fma_(C.c4, x2, O::mul(C.c3, y)), O::fma_(C.c1, y, C.c0));
    const V other = O::fma_(xo, C.c3, C.nc3);
    const V iterm = O::mul(O::mul(C.G, pw), O::mul(other, O::sub(Vm, C.E)));
    const V dQ = O::add(O::allsum(iterm), O::splat(S.qdrive));          // -1e-3 iNet (+ injected current)
    // derivatives by lane
    //   dU = Ptot / (rho |R|) - 3 U^2 / (2 R)      dZ = U       dng = 2 pi (a^2 + Z^2) Dgl (C0 - Pg / kH) / xi
    const V dU = O::sub(O::mul(O::mul(Ptot, O::abs_(q)), O::splat(S.inv_rho)), O::mul(O::mul(O::splat(1.5), O::mul(y, y)), q));
    const V dng = O::mul(O::mul(O::splat(S.kng), O::add(O::splat(S.a2), Zc2)), O::fma_(q, O::splat(-1.0 / bls::kH), O::splat(bls::C0)));
    V dy = fgate;
    dy = O::pick(C.is0, dU, dy);
    dy = O::pick(C.is1, Ub, dy);
    dy = O::pick(C.is2, dng, dy);
    dy = O::pick(C.is3, dQ, dy);
    return dy;
}

}  // namespace sonic
