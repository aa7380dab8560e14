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
//   lane   owns      phase A (one division)                             exponential of              phase B (one rate each)
//   0      U         1/R = 2 Z / (a^2 + Z^2)  -> elastic + viscous      beta_m (and beta_h)         beta_m
//   1      Z         w = (2 Z + D) / D, log w -> capacitance, log r     r^nrep  -> LJ repulsion     beta_h  (e of lane 0)
//   2      ng        Pg = ng Rg T / V(Z)      -> gas pressure, flux     beta_n                      beta_n
//   3      Qm        Z2 = (a^2 - Z^2 - Z D) / (2 Z) -> Cm -> Vm         beta_p (and alpha_p)        beta_p
//   4      m         a^2 / (a^2 + Z^2)        -> electrical pressure    alpha_m                     alpha_m
//   5      h                                                            alpha_h                     alpha_h
//   6      n                                                            alpha_n                     alpha_n
//   7      p                                                            r^nattr -> LJ attraction    alpha_p (e of lane 3)
//
// (log r = log(x0 / D) - log w, r = x0 / (2 Z + D): ONE logarithm and ONE exponential per right-hand side, each
// on all lanes at once; -P0 - Pac(t) is replicated and added to the sum of the pressure terms.)
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
#include <type_traits>

#include "full_core.hpp"
#include "dop853_coeffs.hpp"

// right-hand-side lambdas are expanded in place (their stage index is a compile-time constant)
#define SONIC_COOP_INLINE __attribute__((always_inline))

namespace sonic {

constexpr int OCT = 8;

// Largest step of the cooperative integrators, in units of the dense grid spacing (1 / (1000 f)): the error
// estimate controls the step, not the continuous extension between its ends, and the rows are interpolated
// on the dense grid -- an 8(5,3) step several dense points long leaves 2e-6 of the deflection range in a
// 40 kPa pulsed run, one at most two points long 2e-7 (the configurations that need many steps take far
// shorter ones anyway).
#ifndef COOP_HMAX_DENSE
#define COOP_HMAX_DENSE 2.0
#endif

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
    OCT_BIN(div_finite, x / y)
    OCT_BIN(max_, x > y ? x : y)
    OCT_BIN(min_, x < y ? x : y)
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
    // r = a on lane L, b elsewhere
    template <int L>
    static V on_lane(V a, V b) { V r = b; r.v[L] = a.v[L]; return r; }
    // r = a on the lanes of the bit mask M, b elsewhere
    template <int M>
    static V on_lanes(V a, V b) { V r = b; for (int i = 0; i < OCT; i++) if ((M >> i) & 1) r.v[i] = a.v[i]; return r; }
    template <int SRC>
    static V bcast(V a) { return splat(a.v[SRC]); }
    // lanes 0..3 receive lane SRC (< 4), lanes 4..7 lane 4 + SRC
    template <int SRC>
    static V quad_bcast(V a) { V r; for (int i = 0; i < OCT; i++) r.v[i] = a.v[(i & 4) + SRC]; return r; }
    // lanes 0..3 receive the value of lanes 4..7 (lanes 4..7: unspecified)
    static V shl4(V a) { V r = a; for (int i = 0; i < 4; i++) r.v[i] = a.v[i + 4]; return r; }
    // lane 1 takes the value of lane 0, lane 7 that of lane 3 (rate constants that share an exponential)
    static V exp_share(V a) { V r = a; r.v[1] = a.v[0]; r.v[7] = a.v[3]; return r; }
    // lanes 4..7 receive the value of lanes 0..3 (lanes 0..3: unspecified)
    static V shr4(V a) { V r = a; for (int i = 4; i < OCT; i++) r.v[i] = a.v[i - 4]; return r; }
    static V swap1(V a) { V r; for (int i = 0; i < OCT; i++) r.v[i] = a.v[i ^ 1]; return r; }
    // lane i receives the value of lane i - 1 (lane 0: unspecified)
    static V shr1(V a) { V r = a; for (int i = 1; i < OCT; i++) r.v[i] = a.v[i - 1]; return r; }
    // two sums in one butterfly: lanes 0..3 of `st` = sum of t over the octet, lanes 0..3 of `su` = sum of u over
    // lanes 4..7 (the other lanes: unspecified)
    static void sum_pair(V t, V u, V &st, V &su)
    {
        V m;
        for (int i = 0; i < 4; i++) { m.v[i] = t.v[i] + t.v[7 - i]; m.v[4 + i] = u.v[4 + i]; }
        V b, c;
        for (int i = 0; i < OCT; i++) b.v[i] = m.v[i] + m.v[i ^ 1];
        for (int i = 0; i < OCT; i++) c.v[i] = b.v[i] + b.v[i ^ 2];
        st = c;
        for (int i = 0; i < 4; i++) su.v[i] = su.v[4 + i] = c.v[4 + i];
    }
    static V allsum(V a)
    {
        V b, c, d;
        for (int i = 0; i < OCT; i++) b.v[i] = a.v[i] + a.v[i ^ 1];
        for (int i = 0; i < OCT; i++) c.v[i] = b.v[i] + b.v[i ^ 2];
        for (int i = 0; i < OCT; i++) d.v[i] = c.v[i] + c.v[7 - i];
        return d;
    }
    static V allmax(V a)
    {
        double m = -INFINITY;
        for (int i = 0; i < OCT; i++) m = fmax(m, a.v[i] == a.v[i] ? a.v[i] : INFINITY);   // NaN counts as +inf
        return splat(m);
    }
    static double first(V a) { return a.v[0]; }          // a value known to be replicated
    static bool any_lt(V x, V y) { for (int i = 0; i < OCT; i++) if (x.v[i] < y.v[i]) return true; return false; }
    // output row [t, stim, Z, ng, Qm, m, h, n, p, Vm]: lane l > 0 stores its component at column l + 1,
    // lane 0 (U is not an output) stores t and stim, lane 3 also stores Vm
    static void store_row(double *o, double t, double stim, V r, double Vm)
    {
        o[0] = t; o[1] = stim;
        for (int i = 1; i < OCT; i++) o[i + 1] = r.v[i];
        o[9] = Vm;
    }
    static void fill_row_nan(double *o, double t)
    {
        o[0] = t;
        for (int i = 1; i < 10; i++) o[i] = NAN;
    }
    static bool leader() { return true; }
    static double fast_pow(double en, double e) { return exp(e * log(en)); }
    // lanes 0..3 <-> four arrays `stride` doubles apart (hybrid_coop.hpp: U, Z, ng and t of a dense row)
    static void store4(double *base, long stride, long idx, V v) { for (int i = 0; i < 4; i++) base[i * stride + idx] = v.v[i]; }
    static V load4(const double *base, long stride, long idx)
    {
        V r = splat(0.0);
        for (int i = 0; i < 4; i++) r.v[i] = base[i * stride + idx];
        return r;
    }
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
    static __device__ __forceinline__ V div_finite(V a, V b) { return a * fast_rcp(b); }   // b finite, not 0
    static __device__ __forceinline__ V max_(V a, V b) { return fmax(a, b); }
    static __device__ __forceinline__ V min_(V a, V b) { return fmin(a, b); }
    static __device__ __forceinline__ V neg(V a) { return -a; }
    static __device__ __forceinline__ V abs_(V a) { return fabs(a); }
    static __device__ __forceinline__ V exp_(V x) { return fast_exp(x); }     // fast_math.hpp
    static __device__ __forceinline__ V log_(V x) { return fast_log(x); }
    static __device__ __forceinline__ V sin_(V a) { return sin(a); }
    static __device__ __forceinline__ V fma_(V a, V b, V c) { return fma(a, b, c); }
    static __device__ __forceinline__ V pick(V c, V a, V b) { return c != 0.0 ? a : b; }
    static __device__ __forceinline__ V lt_pick(V x, V y, V a, V b) { return x < y ? a : b; }
    static __device__ __forceinline__ V eq0_pick(V x, V a, V b) { return x == 0.0 ? a : b; }
    template <int L>
    static __device__ __forceinline__ V on_lane(V a, V b) { return lane() == L ? a : b; }
    template <int M>
    static __device__ __forceinline__ V on_lanes(V a, V b) { return ((M >> lane()) & 1) ? a : b; }
    // broadcast of lane SRC of the octet: quad_perm broadcast inside the source's quad, then the
    // other quad fetches it with a shift by four lanes inside the row of 16
    // DPP move that writes only the banks (groups of four lanes of a row of 16) of BANKS; the others keep `old`
    template <int CTRL, int BANKS>
    static __device__ __forceinline__ double dpp_banks(double old, double x)
    {
        const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(x), CTRL, 0xf, BANKS, false);
        const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(x), CTRL, 0xf, BANKS, false);
        return __hiloint2double(hi, lo);
    }
    template <int SRC>
    static __device__ __forceinline__ V bcast(V a)
    {
        constexpr int q = SRC & 3;
        const double inq = dpp<q | (q << 2) | (q << 4) | (q << 6)>(a);       // quad_perm [q,q,q,q]
        // the other quad of the octet takes the value with a shift by four lanes inside the row of 16, written
        // to its banks only (no select): row_shr:4 into banks 1, 3 (lanes 4..7 <- 0..3), row_shl:4 into banks 0, 2
        if (SRC < 4) return dpp_banks<0x114, 0xA>(inq, inq);
        else return dpp_banks<0x104, 0x5>(inq, inq);
    }
    static __device__ __forceinline__ V exp_share(V a)
    {
        const double b = dpp_banks<0xE0, 0x5>(a, a);       // quad_perm [0,0,2,3] in the first quad: lane 1 <- lane 0
        const double c = dpp_banks<0x114, 0xA>(b, b);      // second quad <- first quad ...
        return lane() == 7 ? c : b;                          // ... kept on lane 7 only (<- lane 3)
    }
    template <int SRC>
    static __device__ __forceinline__ V quad_bcast(V a) { return dpp<SRC | (SRC << 2) | (SRC << 4) | (SRC << 6)>(a); }
    static __device__ __forceinline__ V shl4(V a) { return dpp<0x104>(a); }
    static __device__ __forceinline__ V shr4(V a) { return dpp<0x114>(a); }
    static __device__ __forceinline__ V swap1(V a) { return dpp<0xB1>(a); }   // [1,0,3,2]
    static __device__ __forceinline__ V shr1(V a) { return dpp<0x111>(a); }    // row_shr:1
    static __device__ __forceinline__ void sum_pair(V t, V u, V &st, V &su)
    {
        t += dpp<0x141>(t);                                 // lane i + lane 7 - i: both quads hold the four pair sums
        V m = dpp_banks<0xE4, 0xA>(t, u);                   // second quad := u (identity quad_perm into banks 1, 3)
        m += dpp<0xB1>(m);
        m += dpp<0x4E>(m);                                  // first quad: sum of t, second quad: sum of u
        st = m;
        su = dpp<0x104>(m);                                 // row_shl:4: lanes 0..3 <- 4..7
    }
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
    static __device__ __forceinline__ V allmax(V a)
    {
        a = a == a ? a : INFINITY;                  // NaN counts as +inf
        a = fmax(a, dpp<0xB1>(a));
        a = fmax(a, dpp<0x4E>(a));
        a = fmax(a, dpp<0x141>(a));
        return a;
    }
    static __device__ __forceinline__ double first(V a) { return a; }
    static __device__ __forceinline__ bool any_lt(V x, V y) { return x < y; }     // operands are replicated
    static __device__ __forceinline__ void store_row(double *o, double t, double stim, V r, double Vm)
    {
        const int l = lane();
        if (l == 0) { o[0] = t; o[1] = stim; }
        else o[l + 1] = r;
        if (l == 3) o[9] = Vm;
    }
    static __device__ __forceinline__ void fill_row_nan(double *o, double t)
    {
        const int l = lane();
        if (l == 0) { o[0] = t; o[1] = NAN; }
        else o[l + 1] = NAN;
        if (l == 3) o[9] = NAN;
    }
    static __device__ __forceinline__ bool leader() { return lane() == 0; }
    static __device__ __forceinline__ void store4(double *base, long stride, long idx, V v)
    {
        const int l = lane();
        if (l < 4) base[l * stride + idx] = v;
    }
    static __device__ __forceinline__ V load4(const double *base, long stride, long idx)
    {
        const int l = lane();
        return l < 4 ? base[l * stride + idx] : 0.0;
    }
    static __device__ __forceinline__ double fast_pow(double en, float e)
    {
        // en^e for the step-size controller: single-precision hardware log2 / exp2
        return (double)__builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf((float)en));
    }
};
#endif

// Per-lane constants of the octet for one sonophore + cortical neuron.
template <class O>
struct CoopConsts {
    typedef typename O::V V;
    // phase A: N = n0 + n1 Zc + n2 Zs + n3 Zs^2 + n4 y_own ; D = d0 + d1 Zc + d2 Zs + d3 Zc^2 + d5 Zc^3
    // (lane 2: the volume V(Z) = pi a^2 Delta + pi a^2 Z + pi / 3 Z^3, bls.py:311-319)
    V n0, n1, n2, n3, n4, d0, d1, d2, d3, d5;
    V cexp, tE;                // Lennard-Jones terms: exponent of r^n = exp(n log r) and coefficient, on the lanes
                               // whose rate constant borrows its exponential (1 and 7, see coop_rhs)
    // phase B: u = (Vm - vc) vs ; num = a0 + a1 u + e (a2 + a4 e^2) + a3 e^2 ; den = b0 + b1 e + b2 e^2 + b3 e^3
    // vsx = vs, zero on lanes 1 and 7 (the shared exponential of coop_rhs)
    V vsx, uk, ncexp;          // u = Vm vsx + uk - cexp log w, uk = cexp log(x0 / Delta) - vc vsx
    V vc, vs, a0, a1, a2, a3, a4, b0, b1, b2, b3, K;
    // currents (lanes 4..7), as in sonic_quad.hpp: term = G pw(x) other (Vm - E)
    V G, E, c0, c1, c3, c4, nc3;
    // DOPRI5 error floors and the sine argument offsets of the six stage times
    V floor_, cstage;
    // pressure terms q (pa Zc^2 + pb U q + pc Qm^2) + pd U |q| + pq q (lane 0: elastic + viscous, lane 2: gas,
    // lane 4: electrical; zero elsewhere)
    // all in units of rhoL (the sum enters dU / dt = Ptot / (rhoL |R|) - ...); tE too
    V pa, pb, pc, pd, pq;
    // blend of the derivatives of lanes 0..3: dy = k0 dU + k1 U + k2 dng + k3 dQ (one of them 1 per lane)
    V k0, k1, k2, k3;
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
    // (lane 1: w = (Delta + 2 Z) / Delta with the clamped Z, which is never 0 there; lanes 5..7: no division)
    C.n0 = O::roles(0.0,      p.Delta,  0.0,            a2,       a2,    1.0,     1.0,     1.0);
    C.n1 = O::roles(2.0,      2.0,      0.0,            0.0,      0.0,   0.0,     0.0,     0.0);
    C.n2 = O::roles(0.0,      0.0,      0.0,            -p.Delta, 0.0,   0.0,     0.0,     0.0);
    C.n3 = O::roles(0.0,      0.0,      0.0,            -1.0,     0.0,   0.0,     0.0,     0.0);
    C.n4 = O::roles(0.0,      0.0,      bls::Rg * bls::T, 0.0,    0.0,   0.0,     0.0,     0.0);
    C.d0 = O::roles(a2,       p.Delta,  bls::PI * a2 * p.Delta, 0.0, a2, 1.0,     1.0,     1.0);
    C.d1 = O::roles(0.0,      0.0,      bls::PI * a2,   0.0,      0.0,   0.0,     0.0,     0.0);
    C.d2 = O::roles(0.0,      0.0,      0.0,            2.0,      0.0,   0.0,     0.0,     0.0);
    C.d3 = O::roles(1.0,      0.0,      0.0,            0.0,      1.0,   0.0,     0.0,     0.0);
    C.d5 = O::roles(0.0,      0.0,      bls::PI / 3.0,  0.0,      0.0,   0.0,     0.0,     0.0);
    C.cexp = O::roles(0, p.LJ_nrep, 0, 0, 0, 0, 0, p.LJ_nattr);
    C.tE = O::roles(0, p.LJ_C, 0, 0, 0, 0, 0, -p.LJ_C);
    // rates: lanes 0..3 = beta_m beta_h beta_n beta_p ; lanes 4..7 = alpha_m alpha_h alpha_n alpha_p
    //   beta_m  = 0.28e3 vtrap(v - 40, 5)          u = (Vm - (VT + 40)) / 5,    K = 0.28e3 * 5
    //   beta_h  = 4e3 / (1 + exp(-(v - 40) / 5))   u = (Vm - (VT + 40)) / 5,    K = 4e3: e / (1 + e), beta_m's exponential
    //   beta_n  = 0.5e3 exp(-(v - 10) / 40)        u = -(Vm - (VT + 10)) / 40,  K = 0.5e3
    //   alpha_m = 0.32e3 vtrap(13 - v, 4)          u = -(Vm - (VT + 13)) / 4,   K = 0.32e3 * 4
    //   alpha_h = 0.128e3 exp(-(v - 17) / 18)      u = -(Vm - (VT + 17)) / 18,  K = 0.128e3
    //   alpha_n = 0.032e3 vtrap(15 - v, 5)         u = -(Vm - (VT + 15)) / 5,   K = 0.032e3 * 5
    C.vc = O::roles(VT + 40.0, VT + 40.0, VT + 10.0, -35.0, VT + 13.0, VT + 17.0, VT + 15.0, -35.0);
    C.vs = O::roles(1.0 / 5.0, 1.0 / 5.0, -1.0 / 40.0, 1.0 / 20.0, -1.0 / 4.0, -1.0 / 18.0, -1.0 / 5.0, 1.0 / 20.0);
    C.vsx = O::roles(1.0 / 5.0, 0.0, -1.0 / 40.0, 1.0 / 20.0, -1.0 / 4.0, -1.0 / 18.0, -1.0 / 5.0, 0.0);
    //               bm   bh   bn   bp   am   ah   an   ap
    C.a0 = O::roles(0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0);
    C.a1 = O::roles(1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 1.0, 0.0);
    C.a2 = O::roles(0.0, 1.0, 1.0, 0.0, 0.0, 1.0, 0.0, 1.0);
    C.a3 = O::roles(0.0, 0.0, 0.0, 3.3, 0.0, 0.0, 0.0, 0.0);
    C.a4 = O::roles(0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 3.3);
    C.b0 = O::roles(-1.0, 1.0, 1.0, 0.0, -1.0, 1.0, -1.0, 1.0);
    C.b1 = O::roles(1.0, 1.0, 0.0, 1.0, 1.0, 0.0, 1.0, 0.0);
    C.b2 = O::roles(0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0);
    C.b3 = O::roles(0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0);
    C.K = O::roles(0.28e3 * 5.0, 4e3, 0.5e3, 1.0 / TauMax, 0.32e3 * 4.0, 0.128e3, 0.032e3 * 5.0, 1.0 / TauMax);
    // currents (cortical.py:92-119): m lane iNa = gNa m^3 h (V - ENa), h lane iLeak, n lane iKd = gKd n^4
    // (V - EK), p lane iM = gM p (V - EK)
    (void)qdrive;
    const double ELeak = P.ELeak;
    C.G = O::roles(0, 0, 0, 0, -1e-3 * P.gNabar, -1e-3 * P.gLeak, -1e-3 * P.gKdbar, -1e-3 * P.gMbar);
    C.E = O::roles(0, 0, 0, 0, P.ENa, ELeak, P.EK, P.EK);
    C.c0 = O::roles(0, 0, 0, 0, 0, 1, 0, 0);
    C.c1 = O::roles(0, 0, 0, 0, 0, 0, 0, 1);
    C.c3 = O::roles(0, 0, 0, 0, 1, 0, 0, 0);
    C.c4 = O::roles(0, 0, 0, 0, 0, 0, 1, 0);
    C.nc3 = O::roles(1, 1, 1, 1, 0, 1, 1, 1);
    C.floor_ = O::roles(FULL_FLOOR_U, FULL_FLOOR_Z, 1e-25, FULL_FLOOR_Y, FULL_FLOOR_Y, FULL_FLOOR_Y, FULL_FLOOR_Y, FULL_FLOOR_Y);
    C.cstage = O::roles(0.0, dp5::c2, dp5::c3, dp5::c4, dp5::c5, 1.0, 1.0, 1.0);
    const double kE = (bls::kA + p.kA_tissue) / a2, kel = 1.0 / (2.0 * bls::epsilon0 * bls::epsilonR);
    const double ir = 1.0 / bls::rhoL;
    C.pa = O::roles(-kE * ir, 0, 0, 0, 0, 0, 0, 0);
    C.pb = O::roles(-12.0 * bls::delta0 * bls::muS * ir, 0, 0, 0, 0, 0, 0, 0);
    C.pc = O::roles(0, 0, 0, 0, -kel * ir, 0, 0, 0);
    C.pd = O::roles(-4.0 * bls::muL * ir, 0, 0, 0, 0, 0, 0, 0);
    C.pq = O::roles(0, 0, ir, 0, 0, 0, 0, 0);
    C.tE = O::mul(C.tE, O::splat(ir));
    C.k0 = O::roles(1, 0, 0, 0, 0, 0, 0, 0);
    C.k1 = O::roles(0, 1, 0, 0, 0, 0, 0, 0);
    C.k2 = O::roles(0, 0, 1, 0, 0, 0, 0, 0);
    C.k3 = O::roles(0, 0, 0, 1, 0, 0, 0, 0);
    const double lr0 = log(p.LJ_x0 / p.Delta);     // log r = lr0 - log w, r = x0 / (Delta + 2 Z), w = (Delta + 2 Z) / Delta
    C.ncexp = O::neg(C.cexp);
    C.uk = O::sub(O::mul(C.cexp, O::splat(lr0)), O::mul(C.vc, C.vsx));
    return C;
}

template <class O>
struct CoopScalars {
    double a2, inv_a2, inv_3D, volk, Zmin, Delta, Cm0, kC, fs, kE, kel, inv_rho, kng, qdrive, p0r, kng_a2;
};

template <class O>
SONIC_HD CoopScalars<O> coop_scalars(const BLSParams &p, double fs, double qdrive)
{
    CoopScalars<O> S;
    S.a2 = p.a * p.a;
    S.inv_a2 = 1.0 / S.a2;
    S.inv_3D = 1.0 / (3.0 * p.Delta);
    S.volk = bls::PI * S.a2 * p.Delta;
    S.Zmin = bls::rel_Zmin * p.Delta;
    S.Delta = p.Delta;
    S.Cm0 = p.Cm0;
    S.kC = p.Cm0 * p.Delta / S.a2;
    S.fs = fs;
    S.kE = (bls::kA + p.kA_tissue) / S.a2;
    S.kel = 1.0 / (2.0 * bls::epsilon0 * bls::epsilonR);
    S.inv_rho = 1.0 / bls::rhoL;
    S.kng = 2.0 * bls::PI * bls::Dgl / bls::xi;
    S.qdrive = qdrive;
    S.p0r = -bls::P0 * S.inv_rho;
    S.kng_a2 = S.kng * S.a2;
    return S;
}

// the eight rate constants at the (replicated) potential Vm, one per lane: 0..3 = beta_m beta_h beta_n beta_p,
// 4..7 = alpha_m alpha_h alpha_n alpha_p (generic form K num(u, e) / den(e), see coop_consts)
template <class O>
SONIC_HD typename O::V coop_rate(const CoopConsts<O> &C, typename O::V Vm)
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
    return O::mul(C.K, O::div(num, den));
}

// the same from u = (Vm - vc) vs and the exponentials E of coop_rhs, in which lanes 1 and 7 hold the
// Lennard-Jones powers: beta_h takes beta_m's exponential (lane 0), alpha_p that of beta_p (lane 3)
template <class O>
SONIC_HD typename O::V coop_rate_shared(const CoopConsts<O> &C, typename O::V u, typename O::V E)
{
    typedef typename O::V V;
    const V e = O::exp_share(E);
    const V e2 = O::mul(e, e);
    V num = O::fma_(C.a1, u, C.a0);
    num = O::fma_(C.a3, e2, num);
    num = O::fma_(e, O::fma_(C.a4, e2, C.a2), num);
    V den = O::fma_(C.b1, e, C.b0);
    den = O::fma_(O::fma_(C.b3, e, C.b2), e2, den);
    return O::mul(C.K, O::div_finite(num, den));       // (the exponentials are capped: den is finite)
}

// Membrane part of the right-hand side at the (replicated) potential Vm: the gate derivatives on lanes
// 4..7 and the charge derivative (replicated). Also the whole right-hand side of the sparse phase of the
// hybrid scheme, where the capacitance is frozen (hybrid_coop.hpp).
template <class O>
SONIC_HD void coop_membrane(const CoopConsts<O> &C, const CoopScalars<O> &S, typename O::V y,
                            typename O::V Vm, typename O::V u, typename O::V E, typename O::V &fgate,
                            typename O::V &iterm)
{
    typedef typename O::V V;
    // phase B: one rate constant per lane
    const V rate = coop_rate_shared<O>(C, u, E);
    // phase C: gates (lanes 4..7: alpha is the lane's own rate, beta comes from four lanes below)
    const V beta = O::shr4(rate);
    fgate = O::sub(rate, O::mul(O::add(rate, beta), y));               // alpha - (alpha + beta) x
    // currents (pneuron.py:288-296): lanes 4..7, zero elsewhere (G = 0)
    const V xo = O::swap1(y);                                          // the m lane needs h
    const V x2 = O::mul(y, y);
    const V pw = O::fma_(x2, O::fma_(C.c4, x2, O::mul(C.c3, y)), O::fma_(C.c1, y, C.c0));
    const V other = O::fma_(xo, C.c3, C.nc3);
    iterm = O::mul(O::mul(C.G, pw), O::mul(other, O::sub(Vm, C.E)));   // summed by the caller: -1e-3 iNet
    (void)S;
}

// dy/dt of the octet's eight components (y: one component per lane). `pterm` = (-P0 - Pac(t)) / rhoL at the
// time of this evaluation, on lane 0 at least (coop_pterm). Sets `clamped` if the deflection had to be clamped.
// MEMBRANE = false: the mechanical system alone (U, Z, ng) at the imposed charge of lane 3, as
// BilayerSonophore.derivatives with a constant Qm (lookup generation, mech_coop.hpp): the potential, the
// rate constants and the currents are skipped, lanes 3..7 get a zero derivative.
template <class O, bool MEMBRANE = true>
SONIC_HD typename O::V coop_rhs(const CoopConsts<O> &C, const CoopScalars<O> &S, typename O::V y,
                                typename O::V pterm, bool &clamped)
{
    typedef typename O::V V;
    const V Zb = O::template bcast<1>(y);
    const V Zmin = O::splat(S.Zmin);
    if (O::any_lt(Zb, Zmin)) clamped = true;
    const V Zc = O::max_(Zb, Zmin);                                   // bls.py:694-696
    // lane 3 divides by 2 Z: a deflection of exactly 0 is nudged (the sum leaves any other Z as it is); the
    // capacitance of that case is Cm0, selected below
    const V Zs = O::add(Zb, O::splat(1e-30));
    const V Zc2 = O::mul(Zc, Zc), Zs2 = O::mul(Zs, Zs);
    // phase A: one division per lane
    V N = O::fma_(C.n1, Zc, C.n0);
    N = O::fma_(C.n2, Zs, N);
    N = O::fma_(C.n3, Zs2, N);
    N = O::fma_(C.n4, y, N);
    V D = O::fma_(C.d1, Zc, C.d0);
    D = O::fma_(C.d2, Zs, D);
    D = O::fma_(C.d3, Zc2, D);
    D = O::fma_(C.d5, O::mul(Zc2, Zc), D);
    const V q = O::div_finite(N, D);                  // D: a^2 + Z^2, Delta, V(Z), 2 Z (Z != 0), Delta + 2 Z > 0, 1
    // one logarithm: log w on lane 1 (the other lanes' results are not used)
    const V Lw = O::template bcast<1>(O::log_(q));
    // capacitance and potential on lane 3 (bls.py:334-345, nbls.py:148-151): Cm = Cm0 D / a^2 (Z + Z2 log w)
    V Vm = O::splat(0.0);
    if (MEMBRANE) {
        // (in mF/m2, so that Qm / Cm is in mV)
        V Cm = O::mul(O::splat(1e-3 * S.kC), O::fma_(q, Lw, Zs));
        Cm = O::eq0_pick(Zb, O::splat(1e-3 * S.Cm0), Cm);
        const V Cme = O::fma_(O::splat(S.fs), Cm, O::splat((1.0 - S.fs) * 1e-3 * S.Cm0));
        Vm = O::template bcast<3>(O::div_finite(y, Cme));                      // lane 3: y = Qm
    }
    // one exponential per lane: the six distinct ones of the rate constants (beta_m and beta_h share one, so do
    // alpha_p and beta_p), and on the two lanes they leave free (1, 7) the Lennard-Jones powers r^n = exp(n log r),
    // log r = log(x0 / Delta) - log w (bls.py:29-41,472-480). Arguments capped: 0 x inf must not reach the sums.
    V u = O::fma_(C.ncexp, Lw, C.uk);
    if (MEMBRANE) u = O::fma_(Vm, C.vsx, u);
    const V Ex = O::exp_(O::min_(u, O::splat(700.0)));
    // pressure terms, one per lane (bls.py:596-655, 482-491), summed over the octet
    //   lane 0: PE + Pv = -(kA + kA_tissue) (Z / a)^2 / R - 12 U delta0 muS / R^2 - 4 U muL / |R|
    //   (replicated: -P0 - Pac)   lane 2: Pg     lane 4: Pelec = -a^2 / (a^2 + Z^2) Qm^2 / (2 eps0 epsR)
    //   lanes 1, 7: +- C r^n (the Lennard-Jones powers of the shared exponential)
    // as linear forms with per-lane coefficients (zero where a lane has no such term: every factor is finite),
    // the replicated -P0 - Pac added after the sum. The velocity terms sit on lane 0, where y = U; the charge of
    // the electrical term comes to lane 4 from its neighbour
    const V Q4 = O::shr1(y);
    const V inner = O::fma_(C.pc, O::mul(Q4, Q4), O::fma_(C.pb, O::mul(y, q), O::fma_(C.pa, Zc2, C.pq)));
    V T = O::fma_(C.tE, Ex, O::mul(q, inner));
    T = O::fma_(C.pd, O::mul(y, O::abs_(q)), T);
    // net pressure (for lane 0) and net current (for lane 3): one butterfly for both sums
    V fgate = O::splat(0.0), Psum, dQ = O::splat(0.0);
    if (MEMBRANE) {
        V iterm;
        coop_membrane<O>(C, S, y, Vm, u, Ex, fgate, iterm);
        O::sum_pair(T, iterm, Psum, dQ);
        dQ = O::add(dQ, O::splat(S.qdrive));                           // -1e-3 iNet (+ injected current)
    } else {
        Psum = O::allsum(T);
    }
    const V Ptot = O::add(Psum, pterm);                                // / rhoL
    // derivatives by lane
    //   dU = Ptot / (rho |R|) - 3 U^2 / (2 R)      dZ = U       dng = 2 pi (a^2 + Z^2) Dgl (C0 - Pg / kH) / xi
    const V dU = O::fma_(O::mul(O::splat(-1.5), O::mul(y, y)), q, O::mul(Ptot, O::abs_(q)));
    const V dng = O::mul(O::fma_(O::splat(S.kng), Zc2, O::splat(S.kng_a2)), O::fma_(q, O::splat(-1.0 / bls::kH), O::splat(bls::C0)));
    // lanes 0..3 blend their four candidates with 0 / 1 weights (all finite), lanes 4..7 are the gates
    const V lo = O::fma_(C.k3, dQ, O::fma_(C.k2, dng, O::fma_(C.k1, O::swap1(y), O::mul(C.k0, dU))));     // swap1: lane 1 <- U
    return O::template on_lanes<0x0F>(lo, fgate);
}

// One configuration, integrated by the eight lanes of an octet. Same flow as full_config
// (full_core.hpp): segments between events on the dense grid np.linspace(t0, t1, n), DOPRI5 steps with
// the standard controller, dense points consumed on the fly by the linear resampling onto the 10 ns
// sin(d), cos(d) for |d| <= 0.25: Taylor series of degree 11 / 12 (truncation < 1e-19)
template <class O>
SONIC_HD void oct_sincos_small(typename O::V d, typename O::V &sd, typename O::V &cd)
{
    typedef typename O::V V;
    const V z = O::mul(d, d);
    V ps = O::splat(-1.0 / 39916800.0);
    ps = O::fma_(ps, z, O::splat(1.0 / 362880.0));
    ps = O::fma_(ps, z, O::splat(-1.0 / 5040.0));
    ps = O::fma_(ps, z, O::splat(1.0 / 120.0));
    ps = O::fma_(ps, z, O::splat(-1.0 / 6.0));
    sd = O::fma_(O::mul(ps, z), d, d);
    V pc = O::splat(1.0 / 479001600.0);
    pc = O::fma_(pc, z, O::splat(-1.0 / 3628800.0));
    pc = O::fma_(pc, z, O::splat(1.0 / 40320.0));
    pc = O::fma_(pc, z, O::splat(-1.0 / 720.0));
    pc = O::fma_(pc, z, O::splat(1.0 / 24.0));
    pc = O::fma_(pc, z, O::splat(-0.5));
    cd = O::fma_(pc, z, O::splat(1.0));
}

// ---- Dormand-Prince 8(5,3) on octet vectors (shared by full_coop_config and hybrid_coop.hpp) -----------
// The right-hand side is a callable rhs(std::integral_constant<int, SI>, y_stage): the stage index is a
// compile-time constant so that it can select the lane of a per-stage quantity (the acoustic pressure).

// pressure at the time of stage SI from the two octet vectors of stage pressures: pA = stages 1..8,
// pB = stages 9, 10, the end of the step (stages 11 and 12), the dense-output stages 13..15
// pAh = shl4(pA): lane 0 is the only consumer (coop_rhs), a broadcast inside the quads reaches it
template <class O, int SI>
SONIC_HD typename O::V coop_stage_pac(typename O::V pA, typename O::V pAh, typename O::V pB)
{
    if constexpr (SI <= 4) return O::template quad_bcast<SI - 1>(pA);
    else if constexpr (SI <= 8) return O::template quad_bcast<SI - 5>(pAh);
    else if constexpr (SI <= 10) return O::template quad_bcast<SI - 9>(pB);
    else if constexpr (SI <= 12) return O::template quad_bcast<2>(pB);
    else if constexpr (SI == 13) return O::template quad_bcast<3>(pB);
    else return O::template quad_bcast<SI - 14>(O::shl4(pB));
}

// One step attempt of size h from y, K[0] = f(t, y): fills K[1..12] and ynew, returns the error norm of
// Hairer's DOP853, err5^2 / sqrt(err5^2 + 0.01 err3^2), RMS over the eight components (NaN if not finite)
template <class O, class RHS>
SONIC_HD double coop_dp8_attempt(RHS &&rhs, typename O::V y, typename O::V *K, double h, typename O::V floor_,
                                 double rtol, typename O::V &ynew)
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
    const V isc = O::div_finite(O::splat(1.0), sc);                   // sc >= rtol floor > 0
    const V r5 = O::mul(dp8::e5_sum<O>(K), isc), r3 = O::mul(dp8::e3_sum<O>(K), isc);
    const double n5 = O::first(O::allsum(O::mul(r5, r5))), n3 = O::first(O::allsum(O::mul(r3, r3)));
    const double den = n5 + 0.01 * n3;
    double en = den > 0.0 ? fabs(h) * n5 / sqrt(den * OCT) : (den == den ? 0.0 : NAN);
    if (!(n5 == n5) || !(n3 == n3)) en = NAN;
    return en;
}

// the three extra stages of the 7th-order continuous extension (K[13..15]), for an accepted step
template <class O, class RHS>
SONIC_HD void coop_dp8_dense_stages(RHS &&rhs, typename O::V y, typename O::V *K, double h)
{
    const typename O::V hv = O::splat(h);
#define DP8_STAGE(SI) K[SI] = rhs(std::integral_constant<int, SI>{}, O::fma_(hv, dp8::stage_sum<SI, O>(K), y))
    DP8_STAGE(13);
    DP8_STAGE(14);
    DP8_STAGE(15);
#undef DP8_STAGE
}

// 7th-order continuous extension of DOP853 (Hairer et al., II.6): with x = (t - t_n) / h,
//   y(x) = y + x (F0 + (1 - x) (F1 + x (F2 + (1 - x) (F3 + x (F4 + (1 - x) (F5 + x F6))))))
template <class O>
struct CoopDense8 {
    typedef typename O::V V;
    V y, F0, F1, F2, F3, F4, F5, F6;
    SONIC_HD void prepare(V y_, V ynew, const V *K, double h)
    {
        const V hv = O::splat(h);
        y = y_;
        F0 = O::sub(ynew, y_);
        F1 = O::sub(O::mul(hv, K[0]), F0);
        F2 = O::sub(O::mul(O::splat(2.0), F0), O::mul(hv, O::add(K[12], K[0])));
        F3 = O::mul(hv, dp8::d_sum<0, O>(K));
        F4 = O::mul(hv, dp8::d_sum<1, O>(K));
        F5 = O::mul(hv, dp8::d_sum<2, O>(K));
        F6 = O::mul(hv, dp8::d_sum<3, O>(K));
    }
    SONIC_HD V at(double xd) const
    {
        const V x = O::splat(xd), x1 = O::splat(1.0 - xd);
        V r = O::fma_(x, F6, F5);
        r = O::fma_(x1, r, F4);
        r = O::fma_(x, r, F3);
        r = O::fma_(x1, r, F2);
        r = O::fma_(x, r, F1);
        r = O::fma_(x1, r, F0);
        return O::fma_(x, r, y);
    }
};

// Integrate y from t0 to t1 under the drive amplitude As (K[0] = f(t0, y) is computed here: the amplitude
// may have changed), calling dense(td, yd) at the points 1 .. ns - 1 of np.linspace(t0, t1, ns) in order.
// `h` carries the step size from one call to the next. Returns 0, or status bit 4 if the step budget ran
// out or the step size underflowed.
template <class O, int METHOD, class Dense, bool MEMBRANE = true>
SONIC_HD int coop_integrate_segment(const CoopConsts<O> &C, const CoopScalars<O> &S, double w, double phi,
                                    double rtol, double As, double t0, double t1, int ns, double dt,
                                    typename O::V &y, typename O::V *K, double &h, int &nsteps, int max_steps,
                                    bool &clamped, Dense &&dense)
{
    typedef typename O::V V;
    // stage times of the step, one per lane, for the acoustic pressure (drives.py:303-304): evaluated once
    // per step attempt; A: stages 1..8 (METHOD 5: stages 1..5 = c2 c3 c4 c5 1), B (METHOD 8): stages 9, 10,
    // the end of the step, and the three dense-output stages
    const V cA = METHOD == 5 ? O::roles(dp5::c2, dp5::c3, dp5::c4, dp5::c5, 1.0, 1.0, 1.0, 1.0)
                             : O::roles(dp8::c1, dp8::c2, dp8::c3, dp8::c4, dp8::c5, dp8::c6, dp8::c7, dp8::c8);
    const V cB = O::roles(dp8::c9, dp8::c10, 1.0, dp8::c13, dp8::c14, dp8::c15, 1.0, 1.0);
    const Linspace grid = linspace_make(t0, t1, ns);
    bool trial_clamped = false;
    double t = t0;
    int i_d = 1;
    double td = linspace_at(grid, i_d);
    // the drive amplitude changed: no FSAL reuse
    // phase of the drive at t, carried from step to step by rotation and re-seeded every 32 steps:
    // sin(w (t + c h) - phi) = S0 cos(c w h) + C0 sin(c w h), c w h small -> two short Taylor series
    // per lane instead of a library sine with its argument reduction (~200 instructions per call)
    double S0 = sin(w * t - phi), C0 = cos(w * t - phi);
    int nseed = 0;
    K[0] = coop_rhs<O, MEMBRANE>(C, S, y, O::splat(S.p0r - As * S.inv_rho * S0), trial_clamped);
    h = fmin(h, t1 - t0);
    while (i_d < ns) {
        bool last = false;
        if (t + 1.0001 * h >= t1) { h = t1 - t; last = true; }
        trial_clamped = false;
        const V hv = O::splat(h);
        const double wh = w * h;
        const bool small = wh < 0.25;
        // (-P0 - Pac) / rhoL at the stage times c h, one stage per lane
        const V nAr = O::splat(-As * S.inv_rho), p0r = O::splat(S.p0r);
        auto pressure = [&](V cst) {
            if (!small) return O::fma_(nAr, O::sin_(O::sub(O::mul(O::splat(w), O::fma_(cst, hv, O::splat(t))), O::splat(phi))), p0r);
            V sd, cd;
            oct_sincos_small<O>(O::mul(cst, O::splat(wh)), sd, cd);
            return O::fma_(nAr, O::fma_(O::splat(S0), cd, O::mul(O::splat(C0), sd)), p0r);
        };
        const V pA = pressure(cA);
        V ynew, err;
        double en;
        if constexpr (METHOD == 5) {
            using namespace dp5;
            V yt = O::fma_(O::mul(hv, O::splat(a21)), K[0], y);
            K[1] = coop_rhs<O, MEMBRANE>(C, S, yt, O::template bcast<0>(pA), trial_clamped);
            yt = O::fma_(hv, O::fma_(O::splat(a32), K[1], O::mul(O::splat(a31), K[0])), y);
            K[2] = coop_rhs<O, MEMBRANE>(C, S, yt, O::template bcast<1>(pA), trial_clamped);
            yt = O::fma_(hv, O::fma_(O::splat(a43), K[2], O::fma_(O::splat(a42), K[1], O::mul(O::splat(a41), K[0]))), y);
            K[3] = coop_rhs<O, MEMBRANE>(C, S, yt, O::template bcast<2>(pA), trial_clamped);
            yt = O::fma_(hv, O::fma_(O::splat(a54), K[3], O::fma_(O::splat(a53), K[2],
                                     O::fma_(O::splat(a52), K[1], O::mul(O::splat(a51), K[0])))), y);
            K[4] = coop_rhs<O, MEMBRANE>(C, S, yt, O::template bcast<3>(pA), trial_clamped);
            yt = O::fma_(hv, O::fma_(O::splat(a65), K[4], O::fma_(O::splat(a64), K[3], O::fma_(O::splat(a63), K[2],
                                     O::fma_(O::splat(a62), K[1], O::mul(O::splat(a61), K[0]))))), y);
            K[5] = coop_rhs<O, MEMBRANE>(C, S, yt, O::template bcast<4>(pA), trial_clamped);
            ynew = O::fma_(hv, O::fma_(O::splat(a76), K[5], O::fma_(O::splat(a75), K[4], O::fma_(O::splat(a74), K[3],
                                       O::fma_(O::splat(a73), K[2], O::mul(O::splat(a71), K[0]))))), y);
            K[6] = coop_rhs<O, MEMBRANE>(C, S, ynew, O::template bcast<4>(pA), trial_clamped);
            err = O::mul(hv, O::fma_(O::splat(e7), K[6], O::fma_(O::splat(e6), K[5], O::fma_(O::splat(e5), K[4],
                                     O::fma_(O::splat(e4), K[3], O::fma_(O::splat(e3), K[2], O::mul(O::splat(e1), K[0])))))));
            const V sc = O::mul(O::splat(rtol), O::max_(O::max_(O::abs_(y), O::abs_(ynew)), C.floor_));
            const V er = O::div(err, sc);
            en = sqrt(O::first(O::allsum(O::mul(er, er))) * (1.0 / OCT));
        } else {
            const V pB = pressure(cB), pAh = O::shl4(pA);
            auto rhs = [&](auto si, V yt) SONIC_COOP_INLINE {
                return coop_rhs<O, MEMBRANE>(C, S, yt, coop_stage_pac<O, decltype(si)::value>(pA, pAh, pB), trial_clamped);
            };
            en = coop_dp8_attempt<O>(rhs, y, K, h, C.floor_, rtol, ynew);
            // dense-output stages only if a dense point falls inside this (accepted) step
            const double tnew_ = last ? t1 : t + h;
            if (en <= 1.0 && i_d < ns && (last || td <= tnew_)) coop_dp8_dense_stages<O>(rhs, y, K, h);
        }
        nsteps++;
        // h_new = h * min(facmax, max(0.2, 0.9 * en^(-1 / (q + 1)))), q = 4 (5(4) pair) or 7 (8(5,3))
        double fac = 0.9 * (METHOD == 5 ? O::fast_pow(fmax(en, 1e-10), -0.2) : O::fast_pow(fmax(en, 1e-12), -0.125));
        fac = fmin(METHOD == 5 ? 5.0 : 6.0, fmax(0.2, fac));
        if (!(en == en)) fac = 0.2;
        if (en <= 1.0) {
            clamped = clamped || trial_clamped;
            const double tnew = last ? t1 : t + h;
            if (i_d < ns && (last || td <= tnew)) {
                const V dlt = O::sub(ynew, y);
                if constexpr (METHOD == 5) {
                    using namespace dp5;
                    // continuous extension (Hairer et al., II.6), see dopri5_dense
                    const V r4 = O::mul(hv, O::fma_(O::splat(d7), K[6], O::fma_(O::splat(d6), K[5], O::fma_(O::splat(d5), K[4],
                                            O::fma_(O::splat(d4), K[3], O::fma_(O::splat(d3), K[2], O::mul(O::splat(d1), K[0])))))));
                    const V bb = O::sub(O::mul(hv, K[0]), dlt);
                    const V cc = O::sub(O::sub(dlt, O::mul(hv, K[6])), bb);
                    while (i_d < ns && (last || td <= tnew)) {
                        V yd = ynew;
                        if (td < tnew) {
                            const V sg = O::splat((td - t) / h), s1 = O::splat(1.0 - (td - t) / h);
                            yd = O::fma_(sg, O::fma_(s1, O::fma_(sg, O::fma_(s1, r4, cc), bb), dlt), y);
                        }
                        dense(td, yd);
                        i_d++;
                        if (i_d < ns) td = linspace_at(grid, i_d);
                    }
                } else {
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
            }
            y = ynew;
            K[0] = K[METHOD == 5 ? 6 : 12];
            if (small && ++nseed < 32) {
                V sd, cd;
                oct_sincos_small<O>(O::splat(w * (tnew - t)), sd, cd);
                const double s_ = O::first(sd), c_ = O::first(cd);
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
        if (nsteps >= max_steps || !(h > 1e-18)) return 4;
    }
    return 0;
}

// One configuration, integrated by the eight lanes of an octet. Same flow as full_config
// (full_core.hpp): segments between events on the dense grid np.linspace(t0, t1, n), explicit Runge-Kutta
// steps with the standard controller, dense points consumed on the fly by the linear resampling onto
// the 10 ns grid. `store`: false for a shadow copy of a configuration (same arithmetic, no stores).
//
// METHOD 5: Dormand-Prince 5(4), 6 right-hand sides per step (as full_core.hpp).
// METHOD 8: Dormand-Prince 8(5,3), 12 per step + 3 for the 7th-order dense output of the steps that
//           contain a dense point. At equal accuracy it needs half the right-hand sides of the 5(4) pair
//           on this system (RS, 600 kPa: 3.2e4 per simulated microsecond at rtol 1e-7 against 6.4e4 at
//           rtol 1e-8, both 1e-7 of the deflection range from the converged solution).
template <class O, int METHOD>
SONIC_HD void full_coop_config(const FullDev &D, const BLSParams &p, const CorticalParams &P, int neuron,
                               long long c, bool store)
{
    typedef typename O::V V;
    constexpr int NCOL = 10;                     // t stim Z ng Qm m h n p Vm
    const double f = D.f[c], fs = D.fs[c];
    const double w = 2.0 * bls::PI * f;
    const double dt = 1.0 / (MECH_NPC * f);
    const int max_steps = full_step_budget(D.opts, f, D.tstop[c]);
    int status = 0;
    bool clamped = false;

    const CoopConsts<O> C = coop_consts<O>(p, P, neuron, D.opts.qdrive);
    const CoopScalars<O> S = coop_scalars<O>(p, fs, D.opts.qdrive);

    // initial conditions (nbls.py:321-329, bls.py:720-747), as full_config
    const double Pac_dt = D.A[c] * sin(w * dt - D.phi);
    const double Zqs = bls_balancedefQS(p, p.ng0, D.y0[0], Pac_dt);
    if (!(Zqs == Zqs)) status |= 2;
    V y = O::roles(0.0, Zqs, p.ng0, D.y0[0], D.y0[1], D.y0[2], D.y0[3], D.y0[4]);

    const long long s0 = D.seg_off[c];
    const int nseg = (int)(D.seg_off[c + 1] - s0);
    const long long M_rows = D.row_off[c + 1] - D.row_off[c];
    double *rows = D.traces + D.row_off[c] * NCOL;
    const Linspace out = linspace_make(0.0, D.tstop[c], (int)M_rows);
    long long j = 0;
    double tau = linspace_at(out, 0);
    double tp = 0.0;
    V yp = y;
    int nsteps = 0;

    auto consume = [&](double ti, V yi, double xs) {
        while (j < M_rows && tau <= ti) {
            V r = yi;
            if (ti > tp) {
                const V wgt = O::splat((tau - tp) / (ti - tp));
                r = O::fma_(O::sub(yi, yp), wgt, yp);                           // np.interp
            }
            // Vm from the RESAMPLED Qm and Z (nbls.py:317-319, 349-351)
            const double Zr = O::first(O::template bcast<1>(r)), Qr = O::first(O::template bcast<3>(r));
            const double Vm = Qr / (fs * bls_capacitance(p, Zr) + (1.0 - fs) * p.Cm0) * 1e3;
            if (store) O::store_row(rows + j * NCOL, tau, (j == 0) ? 0.0 : xs, r, Vm);
            j++;
            if (j < M_rows) tau = linspace_at(out, (int)j);
        }
        tp = ti;
        yp = yi;
    };

    constexpr int NK = METHOD == 5 ? 7 : 16;
    V K[NK];                                  // stage derivatives; K[0] = f(t, y) (first same as last)
    double h = 0.25 * dt;
    for (int s = 0; s < nseg && !(status & 6); s++) {
        const double t0 = D.seg_t0[s0 + s], t1 = D.seg_t1[s0 + s], xs = D.seg_x[s0 + s];
        const int ns = D.seg_n[s0 + s];
        const double As = D.A[c] * xs;                    // eventfunc: drive.xvar * x (nbls.py:337)
        consume(t0, y, xs);                               // first dense row of the segment (duplicate)
        if (!(t1 > t0)) { consume(t1, y, xs); continue; }
        const int bad = coop_integrate_segment<O, METHOD>(C, S, w, D.phi, D.opts.rtol, As, t0, t1, ns, dt, y, K, h,
                                                          nsteps, max_steps, clamped,
                                                          [&](double td, V yd) SONIC_COOP_INLINE { consume(td, yd, xs); });
        if (bad) { status |= bad; break; }
    }
    // rows not produced (failed configuration): NaN
    for (; j < M_rows; j++)
        if (store) O::fill_row_nan(rows + j * NCOL, linspace_at(out, (int)j));
    if (clamped) status |= 1;
    if (store && O::leader()) {
        D.status[c] = status;
        D.nsteps[c] = nsteps;
    }
}

}  // namespace sonic
