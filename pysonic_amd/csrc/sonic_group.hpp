// pysonic_amd/csrc/sonic_group.hpp
//
// GROUP-COOPERATIVE integrator for the neurons with more gates than the cortical RS / FS pair and, for
// two of them, a calcium core: LTS (and IB), RE, TC, STN. One stimulus configuration per row of 16
// adjacent lanes (4 per wavefront) instead of one per lane; sonic_quad.hpp generalised.
//
// Why: a configuration is a strictly sequential chain of ~10^4 Rosenbrock steps, and in the
// lane-per-configuration kernel a step of these models costs 3 - 14 us (every lane walks through all
// its gates, table lines and Jacobian columns one after the other), so a sweep of 10^4 configurations
// takes as long as its slowest member needs alone, with most of the chip idle. Here
//   * every GATE lives on its own lane: x' = a - r x with (a, r) either from the lane's two lookup
//     lines (voltage-gated states) or from a closed form of the calcium concentration (STN: d2, r);
//   * the CORE z = (Q, [Cai, ...]) is replicated on the 16 lanes: each lane evaluates the same scalar
//     arithmetic on the same operands;
//   * each ionic current is owned by the lane of its first gate: i = G pw(x) f1 f2 (Vm - E), pw a
//     monomial of the owner's gate, f1 / f2 the gates of the lanes next to it (DPP quad_perm xor 1 /
//     xor 2) -- m^3 h, n^4, p, s^2 u, a^2 b, p^2 q, c^2 d1 d2, r^2 --, E constant or the Nernst
//     potential of the current Cai; per-lane constants select all of that (no branches);
//   * dQ/dt and dCai/dt need one all-reduce over the row each (4 DPP steps);
//   * W = I/(h gamma) - J is a diagonal (gates) bordered by the core rows and by TWO columns (Q and Cai):
//     the gates are eliminated lane-wise, the Schur complement of the core (NC x NC, NC <= 5) is
//     factorised redundantly on every lane.
// The arithmetic is the Rosenbrock / home-cell scheme of sonic_integrator.hpp (see there for the
// references into PySONIC); sums over gates are formed by the butterfly, so results agree with the
// lane-per-configuration kernel to rounding, not bitwise.
//
// Written once over an `Ops` backend like sonic_quad.hpp: on the device a group vector is one double
// per lane (GroupOpsDev, DPP); the CPU test harness uses 16-element arrays (GroupOpsHost, development).
#pragma once
#include "sonic_quad.hpp"

// lambdas of the step are expanded in place: their stage index must stay a compile-time constant
#define SONIC_LAMBDA_INLINE __attribute__((always_inline))

namespace sonic {

constexpr int GRP = 16;

// What a lane is, built on the host by GroupModel<M>::lanes and uploaded once per model.
//   gate:     tab   index of the gate in the table order of the model (alpha = table 1 + 2 tab), -1: no lines
//             itau, theta, ikx   calcium-dependent gate: x_inf = 1 / (1 + exp((Cai - theta) ikx)), rate itau
//   current:  G (with the factor -1e-3 of dQ/dt = -1e-3 iNet), E0, eCa (1: E = E0 + ECa(Cai))
//             c1..c4   pw(x) = c1 x + c2 x^2 + c3 x^3 + c4 x^4
//             k1, k2   multiply by the gate of lane ^ 1 / lane ^ 2
//             r1, r2   this lane's gate is such a factor of the current owned by lane ^ 1 / lane ^ 2
//             kap      weight of the lane's current (and of its Jacobian entries) in dCai/dt
//   output:   colx  column of the gate in an output row (and 2 + its index in the initial conditions), -1: none
//             cols / ssel   a second column this lane stores, and which replicated scalar goes there
//             errw  1 if the lane carries a state of the model (it counts in the error norm)
struct LaneSpec {
    double G, E0, eCa, c1, c2, c3, c4, k1, k2, r1, r2, kap, itau, theta, ikx, errw;
    double ghk, Cin, Cout;      // current driven by the Goldman-Hodgkin-Katz force of (Cin, Cout) instead of Vm - E0
    int tab, colx, cols, ssel;
};
enum : int { GS_T = 0, GS_X = 1, GS_VM = 2, GS_Z0 = 3 };   // ssel: GS_Z0 + c = core variable c

SONIC_HD LaneSpec lane_none()
{
    return LaneSpec{0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0., 0., -1, -1, -1, -1};
}

template <class O>
struct GroupConsts {
    typename O::V G, E0, eCa, c1, c2, c3, c4, d2, d3, d4, k1, nk1, k2, nk2, r1, r2, kap, itau, theta, ikx, errw;
    typename O::V ghk, Cin, Cout;
    typename O::I tab, colx, cols, ssel;
};

// ---- CPU emulation backend: V = 16 values, one per lane of the group ---------------------------
struct GroupOpsHost {
    struct V {
        double v[GRP];
    };
    struct I {
        int v[GRP];
    };
    static V splat(double a) { V r; for (int i = 0; i < GRP; i++) r.v[i] = a; return r; }
    static double pin(double a) { return a; }
#define GRP_BIN(name, expr) static V name(V a, V b) { V r; for (int i = 0; i < GRP; i++) { const double x = a.v[i], y = b.v[i]; r.v[i] = (expr); } return r; }
    GRP_BIN(add, x + y)
    GRP_BIN(sub, x - y)
    GRP_BIN(mul, x * y)
#undef GRP_BIN
    static V fma_(V a, V b, V c) { V r; for (int i = 0; i < GRP; i++) r.v[i] = a.v[i] * b.v[i] + c.v[i]; return r; }
    static V rcp(V a) { V r; for (int i = 0; i < GRP; i++) r.v[i] = 1.0 / a.v[i]; return r; }
    static V exp_(V a) { V r; for (int i = 0; i < GRP; i++) r.v[i] = exp(a.v[i]); return r; }
    static V swap1(V a) { V r; for (int i = 0; i < GRP; i++) r.v[i] = a.v[i ^ 1]; return r; }
    static V swap2(V a) { V r; for (int i = 0; i < GRP; i++) r.v[i] = a.v[i ^ 2]; return r; }
    static double allsum(V a)
    {
        V b, c, d;
        for (int i = 0; i < GRP; i++) b.v[i] = a.v[i] + a.v[i ^ 1];
        for (int i = 0; i < GRP; i++) c.v[i] = b.v[i] + b.v[i ^ 2];
        for (int i = 0; i < GRP; i++) d.v[i] = c.v[i] + c.v[(i & 8) | (7 - (i & 7))];
        return d.v[0] + d.v[15];
    }
    // sum over the lanes of w (e / (atol + rtol max(|a|, |b|)))^2 in single precision
    static float errsum(V e, V a, V b, V w, float atol, float rtol)
    {
        float acc = 0.0f;
        for (int i = 0; i < GRP; i++) {
            const float sc = atol + rtol * fmaxf(fabsf((float)a.v[i]), fabsf((float)b.v[i]));
            const float r = (float)e.v[i] / sc;
            acc += (float)w.v[i] * r * r;
        }
        return acc;
    }
    static V select(bool c, V a, V b) { return c ? a : b; }
    static float rcpf(float a) { return 1.0f / a; }
    static float sqrtf_(float a) { return sqrtf(a); }
    static float rsqf(float a) { return 1.0f / sqrtf(a); }
    static bool leader() { return true; }
    static bool wave_any(bool c) { return c; }
    // ---- used by the row-cooperative detailed model (full_row.hpp) ----
    template <int L>
    static double bcast(V a) { return a.v[L]; }                        // lane L of the row, replicated
#define GRP_UN(name, expr) static V name(V a) { V r; for (int i = 0; i < GRP; i++) { const double x = a.v[i]; r.v[i] = (expr); } return r; }
    GRP_UN(abs_, fabs(x))
    GRP_UN(sin_, sin(x))
#undef GRP_UN
    static V max_(V a, V b) { V r; for (int i = 0; i < GRP; i++) r.v[i] = a.v[i] > b.v[i] ? a.v[i] : b.v[i]; return r; }
    static V min_(V a, V b) { V r; for (int i = 0; i < GRP; i++) r.v[i] = a.v[i] < b.v[i] ? a.v[i] : b.v[i]; return r; }
    static V lt_pick(V x, V t, V a, V b) { V r; for (int i = 0; i < GRP; i++) r.v[i] = x.v[i] < t.v[i] ? a.v[i] : b.v[i]; return r; }
    template <class F>
    static V lane_values(F f) { V r; for (int i = 0; i < GRP; i++) r.v[i] = f(i); return r; }
    static double fast_pow(double en, double e) { return exp(e * log(en)); }
    // 1 / sqrt(a), 0 where a = 0
    static V rsqrt_pos(V a) { V r; for (int i = 0; i < GRP; i++) r.v[i] = a.v[i] > 1e-290 ? 1.0 / sqrt(a.v[i]) : 0.0; return r; }
    static double allmax(V a) { double m = a.v[0]; for (int i = 1; i < GRP; i++) m = a.v[i] > m ? a.v[i] : m; return m; }   // no NaN among the operands
    template <class RL, class RC>
    static void load_row_consts(const RL *s, RC &R)
    {
        for (int i = 0; i < GRP; i++) {
            for (int k = 0; k < (int)(sizeof(R.r) / sizeof(R.r[0])); k++) R.r[k].v[i] = s[i].v[k];
            R.col.v[i] = s[i].col; R.extra.v[i] = s[i].extra;
        }
    }
    template <class RC>
    static void store_full_row(double *o, const RC &R, int ncol, double t, double stim, V r, double Vm)
    {
        for (int i = 0; i < GRP; i++) {
            if (R.col.v[i] >= 0) o[R.col.v[i]] = r.v[i];
            if (R.extra.v[i] == 1) { o[0] = t; o[1] = stim; }
            if (R.extra.v[i] == 2) o[ncol - 1] = Vm;
        }
    }
    template <class RC>
    static void fill_full_row_nan(double *o, const RC &, int ncol, double t)
    {
        o[0] = t;
        for (int i = 1; i < ncol; i++) o[i] = NAN;
    }
    // lanes 12 .. 15 <-> four arrays `stride` doubles apart (hybrid_row.hpp: U, Z, ng and t of a dense row)
    static void store_mech4(double *base, long stride, long idx, V v) { for (int i = 0; i < 4; i++) base[i * stride + idx] = v.v[12 + i]; }
    static V load_mech4(const double *base, long stride, long idx)
    {
        V r = splat(0.0);
        for (int i = 0; i < 4; i++) r.v[12 + i] = base[i * stride + idx];
        return r;
    }
    static V neg(V a) { V r; for (int i = 0; i < GRP; i++) r.v[i] = -a.v[i]; return r; }
    static void load_consts(const LaneSpec *s, GroupConsts<GroupOpsHost> &C)
    {
        for (int i = 0; i < GRP; i++) {
            C.G.v[i] = s[i].G; C.E0.v[i] = s[i].E0; C.eCa.v[i] = s[i].eCa;
            C.c1.v[i] = s[i].c1; C.c2.v[i] = s[i].c2; C.c3.v[i] = s[i].c3; C.c4.v[i] = s[i].c4;
            C.d2.v[i] = 2.0 * s[i].c2; C.d3.v[i] = 3.0 * s[i].c3; C.d4.v[i] = 4.0 * s[i].c4;
            C.k1.v[i] = s[i].k1; C.nk1.v[i] = 1.0 - s[i].k1; C.k2.v[i] = s[i].k2; C.nk2.v[i] = 1.0 - s[i].k2;
            C.r1.v[i] = s[i].r1; C.r2.v[i] = s[i].r2; C.kap.v[i] = s[i].kap;
            C.itau.v[i] = s[i].itau; C.theta.v[i] = s[i].theta; C.ikx.v[i] = s[i].ikx; C.errw.v[i] = s[i].errw;
            C.ghk.v[i] = s[i].ghk; C.Cin.v[i] = s[i].Cin; C.Cout.v[i] = s[i].Cout;
            C.tab.v[i] = s[i].tab; C.colx.v[i] = s[i].colx; C.cols.v[i] = s[i].cols; C.ssel.v[i] = s[i].ssel;
        }
    }
    // lookup lines of the lane's gate in the record of a cell: value and slope of alpha, beta
    static void load_lines(const double *rec, const I &tab, V &av, V &as, V &bv, V &bs)
    {
        for (int i = 0; i < GRP; i++) {
            const int g = tab.v[i];
            av.v[i] = g < 0 ? 0.0 : rec[4 + 4 * g]; as.v[i] = g < 0 ? 0.0 : rec[5 + 4 * g];
            bv.v[i] = g < 0 ? 0.0 : rec[6 + 4 * g]; bs.v[i] = g < 0 ? 0.0 : rec[7 + 4 * g];
        }
    }
    // the gate of every lane from the initial conditions (reference column order, Qm first)
    static V init_gates(const double *y0ref, const I &colx)
    {
        V r;
        for (int i = 0; i < GRP; i++) r.v[i] = colx.v[i] < 0 ? 0.0 : y0ref[colx.v[i] - 2];
        return r;
    }
    template <int NC>
    static void store_row(double *r, const GroupConsts<GroupOpsHost> &C, double t, double x, double Vm,
                          const double *z, V g)
    {
        for (int i = 0; i < GRP; i++) {
            if (C.colx.v[i] >= 0) r[C.colx.v[i]] = g.v[i];
            const int s = C.ssel.v[i];
            if (C.cols.v[i] >= 0) r[C.cols.v[i]] = s == GS_T ? t : (s == GS_X ? x : (s == GS_VM ? Vm : z[s - GS_Z0]));
        }
    }
};

#if defined(__HIPCC__)
// ---- device backend: V = one double per lane; lane & 15 = position in the group ---------------
struct GroupOpsDev {
    typedef double V;
    typedef int I;
    template <int CTRL>
    static __device__ __forceinline__ double dpp(double x)
    {
        // all sixteen lanes of a group are active whenever the group is: no `old` value is needed
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
        return __hiloint2double(hi, lo);
    }
    template <int CTRL>
    static __device__ __forceinline__ float dppf(float x)
    {
        return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xf, 0xf, true));
    }
    static __device__ __forceinline__ int lane() { return threadIdx.x & 15; }
    static __device__ __forceinline__ V splat(double a) { return a; }
    // a constant kept in a vector register pair for the whole kernel (see QuadOpsDev::pin)
    static __device__ __forceinline__ double pin(double a)
    {
        asm volatile("" : "+v"(a));
        return a;
    }
    static __device__ __forceinline__ V add(V a, V b) { return a + b; }
    static __device__ __forceinline__ V sub(V a, V b) { return a - b; }
    static __device__ __forceinline__ V mul(V a, V b) { return a * b; }
    static __device__ __forceinline__ V fma_(V a, V b, V c) { return fma(a, b, c); }
    static __device__ __forceinline__ V rcp(V a) { return fast_rcp1(a); }
    static __device__ __forceinline__ V exp_(V a) { return fast_exp(a); }
    static __device__ __forceinline__ V swap1(V a) { return dpp<0xB1>(a); }          // quad_perm [1,0,3,2]
    static __device__ __forceinline__ V swap2(V a) { return dpp<0x4E>(a); }          // quad_perm [2,3,0,1]
    // Butterfly over the row of 16: quad_perm xor 1, xor 2, then row_half_mirror (quads 0 <-> 1, 2 <-> 3)
    // and row_mirror (halves). Every lane adds the same two rounded numbers at every stage, so the
    // replicated result is bit-identical across the group -- provided the operand is materialised first
    // (see QuadOpsDev::allsum for what happens otherwise).
    static __device__ __forceinline__ double allsum(V a)
    {
        asm volatile("" : "+v"(a));
        a += dpp<0xB1>(a);
        a += dpp<0x4E>(a);
        a += dpp<0x141>(a);
        a += dpp<0x140>(a);
        return a;
    }
    static __device__ __forceinline__ float errsum(V e, V a, V b, V w, float atol, float rtol)
    {
        const float sc = atol + rtol * fmaxf(fabsf((float)a), fabsf((float)b));
        const float r = (float)e * __builtin_amdgcn_rcpf(sc);
        float f = (float)w * r * r;
        asm volatile("" : "+v"(f));
        f += dppf<0xB1>(f);
        f += dppf<0x4E>(f);
        f += dppf<0x141>(f);
        f += dppf<0x140>(f);
        return f;
    }
    static __device__ __forceinline__ V select(bool c, V a, V b) { return c ? a : b; }
    static __device__ __forceinline__ float rcpf(float a) { return __builtin_amdgcn_rcpf(a); }
    static __device__ __forceinline__ float sqrtf_(float a) { return __builtin_amdgcn_sqrtf(a); }
    static __device__ __forceinline__ float rsqf(float a) { return __builtin_amdgcn_rsqf(a); }
    static __device__ __forceinline__ bool leader() { return lane() == 0; }
    static __device__ __forceinline__ bool wave_any(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0; }
    static constexpr int WIDTH = 16;     // lanes per configuration
    static __device__ __forceinline__ int from_leader(int v) { return __shfl(v, threadIdx.x & ~15); }   // the row's lane 0
    // ---- used by the row-cooperative detailed model (full_row.hpp) ----
    // lane L of the row on every lane of the row: row_newbcast
    template <int L>
    static __device__ __forceinline__ double bcast(V a) { return dpp<0x150 + L>(a); }
    static __device__ __forceinline__ V abs_(V a) { return fabs(a); }
    static __device__ __forceinline__ V sin_(V a) { return sin(a); }
    static __device__ __forceinline__ V max_(V a, V b) { return fmax(a, b); }
    static __device__ __forceinline__ V min_(V a, V b) { return fmin(a, b); }
    static __device__ __forceinline__ V lt_pick(V x, V t, V a, V b) { return x < t ? a : b; }
    template <class F>
    static __device__ __forceinline__ V lane_values(F f) { return f(lane()); }
    static __device__ __forceinline__ double fast_pow(double en, float e)
    {
        return (double)__builtin_amdgcn_exp2f(e * __builtin_amdgcn_logf((float)en));      // step-size controller
    }
    // (error norm: the hardware estimate will do -- in DOUBLE: squared errors of a state at rest lie far below the
    // single-precision range, where the estimate of 1 / sqrt(0) = inf rejected every step)
    static __device__ __forceinline__ V rsqrt_pos(V a) { return a > 1e-290 ? __builtin_amdgcn_rsq(a) : 0.0; }
    static __device__ __forceinline__ double allmax(V a)
    {
        a = fmax(a, dpp<0xB1>(a));
        a = fmax(a, dpp<0x4E>(a));
        a = fmax(a, dpp<0x141>(a));
        a = fmax(a, dpp<0x140>(a));
        return a;
    }
    template <class RL, class RC>
    static __device__ __forceinline__ void load_row_consts(const RL *s, RC &R)
    {
        const RL *me = s + lane();
#pragma unroll
        for (int k = 0; k < (int)(sizeof(R.r) / sizeof(R.r[0])); k++) R.r[k] = me->v[k];
        R.col = me->col; R.extra = me->extra;
    }
    template <class RC>
    static __device__ __forceinline__ void store_full_row(double *o, const RC &R, int ncol, double t, double stim, V r,
                                                          double Vm)
    {
        if (R.col >= 0) o[R.col] = r;
        if (R.extra == 1) { o[0] = t; o[1] = stim; }
        if (R.extra == 2) o[ncol - 1] = Vm;
    }
    static __device__ __forceinline__ void store_mech4(double *base, long stride, long idx, V v)
    {
        const int l = lane();
        if (l >= 12) base[(l - 12) * stride + idx] = v;
    }
    static __device__ __forceinline__ V load_mech4(const double *base, long stride, long idx)
    {
        const int l = lane();
        return l >= 12 ? base[(l - 12) * stride + idx] : 0.0;
    }
    static __device__ __forceinline__ V neg(V a) { return -a; }
    template <class RC>
    static __device__ __forceinline__ void fill_full_row_nan(double *o, const RC &R, int ncol, double t)
    {
        if (R.col >= 0) o[R.col] = NAN;
        if (R.extra == 1) { o[0] = t; o[1] = NAN; }
        if (R.extra == 2) o[ncol - 1] = NAN;
    }
    static __device__ __forceinline__ void load_consts(const LaneSpec *specs, GroupConsts<GroupOpsDev> &C)
    {
        const LaneSpec s = specs[lane()];
        C.G = s.G; C.E0 = s.E0; C.eCa = s.eCa;
        C.c1 = s.c1; C.c2 = s.c2; C.c3 = s.c3; C.c4 = s.c4;
        C.d2 = 2.0 * s.c2; C.d3 = 3.0 * s.c3; C.d4 = 4.0 * s.c4;
        C.k1 = s.k1; C.nk1 = 1.0 - s.k1; C.k2 = s.k2; C.nk2 = 1.0 - s.k2;
        C.r1 = s.r1; C.r2 = s.r2; C.kap = s.kap;
        C.itau = s.itau; C.theta = s.theta; C.ikx = s.ikx; C.errw = s.errw;
        C.ghk = s.ghk; C.Cin = s.Cin; C.Cout = s.Cout;
        C.tab = s.tab; C.colx = s.colx; C.cols = s.cols; C.ssel = s.ssel;
    }
    static __device__ __forceinline__ void load_lines(const double *rec, const I &tab, V &av, V &as, V &bv,
                                                      V &bs)
    {
        // lanes without lines read those of gate 0 and discard them (a load is cheaper than a branch)
        const double2 *p = (const double2 *)(rec + 4 + 4 * (tab < 0 ? 0 : tab));
        const double2 a = p[0], b = p[1];
        const bool on = tab >= 0;
        av = on ? a.x : 0.0; as = on ? a.y : 0.0; bv = on ? b.x : 0.0; bs = on ? b.y : 0.0;
    }
    static __device__ __forceinline__ V init_gates(const double *y0ref, const I &colx)
    {
        return colx < 0 ? 0.0 : y0ref[colx - 2];
    }
    template <int NC>
    static __device__ __forceinline__ void store_row(double *r, const GroupConsts<GroupOpsDev> &C, double t,
                                                     double x, double Vm, const double *z, V g)
    {
        if (C.colx >= 0) r[C.colx] = g;
        if (C.cols >= 0) {
            double v = C.ssel == GS_T ? t : (C.ssel == GS_X ? x : Vm);
#pragma unroll
            for (int c = 0; c < NC; c++) v = C.ssel == GS_Z0 + c ? z[c] : v;
            r[C.cols] = v;
        }
    }
};
#endif

// ---------------------------------------------------------------------------------------------
// Per-model description: lanes, core right-hand side and core Jacobian (replicated arithmetic).
//   NC      core variables, z[0] = Q, z[1] = Cai if HAS_CAI
//   NX      extra lookup lines the core needs (replicated), xtab(i) their table index
//   core<JAC>(P, H, Vm, z, sQ, sC, qdrive, fz, sCond, sKCond, Jzz):
//           fz = core derivatives given sQ = sum of the lane currents (already x -1e-3) and sC = sum of
//           kap x lane currents; JAC: Jzz = d fz / d z given sCond = sum of G pw f1 f2, sKCond = sum of
//           kap x that. Entries of Jzz that the gates add to (Schur complement) are handled by the caller.
//   core_col(c)  output column of core variable c
// ---------------------------------------------------------------------------------------------
template <class M>
struct GroupModel;

SONIC_HD LaneSpec lane_gate(int tab, int colx)
{
    LaneSpec s = lane_none();
    s.tab = tab; s.colx = colx; s.errw = 1.0;
    return s;
}
// owner of a current G x^e (V - E) [x the gate of lane ^ 1 if k1] [x the gate of lane ^ 2 if k2]
SONIC_HD LaneSpec lane_owner(LaneSpec s, double g, double E, int expo, bool k1, bool k2 = false)
{
    s.G = -1e-3 * g; s.E0 = E;
    s.c1 = expo == 1; s.c2 = expo == 2; s.c3 = expo == 3; s.c4 = expo == 4;
    s.k1 = k1; s.k2 = k2;
    return s;
}
SONIC_HD void lane_scalars(LaneSpec *s, int n, const int *cols, const int *ssel)
{
    for (int i = 0; i < n; i++) { s[i].cols = cols[i]; s[i].ssel = ssel[i]; }
}

template <>
struct GroupModel<CorticalLTS> {
    typedef CorticalLTS M;
    typedef LTSParams Params;
    static constexpr int NC = 1, NX = 0, NT = M::NT, NY = M::NY, NCOL = NY + 3;
    static constexpr bool HAS_CAI = false, HAS_CAIGATE = false, HAS_ECA = false, HAS_X2 = false, HAS_GHK = false;
    static constexpr bool PIN_COEFFS = true;   // the stage coefficients in vector registers, where there is room for them
    SONIC_HD static int xtab(int) { return 0; }
    SONIC_HD static int core_col(int) { return 2; }
    // reference columns: t stim Qm m h n p s u Vm
    static bool lanes(const Params &P, LaneSpec *s)
    {
        for (int i = 0; i < GRP; i++) s[i] = lane_none();
        s[0] = lane_owner(lane_gate(0, 3), P.gNabar, P.ENa, 3, true);     // m: iNa = gNa m^3 h (V - ENa)
        s[1] = lane_gate(1, 4); s[1].r1 = 1.0;                            // h
        s[2] = lane_owner(lane_gate(2, 5), P.gKdbar, P.EK, 4, false);     // n: iKd = gKd n^4 (V - EK)
        s[3] = lane_owner(lane_gate(3, 6), P.gMbar, P.EK, 1, false);      // p: iM = gM p (V - EK)
        s[4] = lane_owner(lane_gate(4, 7), P.gCaTbar, P.ECa, 2, true);    // s: iCaT = gCaT s^2 u (V - ECa)
        s[5] = lane_gate(5, 8); s[5].r1 = 1.0;                            // u
        const int cols[4] = {0, 1, 2, 9}, ssel[4] = {GS_T, GS_X, GS_Z0, GS_VM};
        lane_scalars(s, 4, cols, ssel);
        return true;
    }
    template <bool JAC, class Cell>
    SONIC_HD static void core(const Params &P, const Cell &H, double Vm, const double *, double sQ, double,
                              double qdrive, double *fz, double sCond, double, double (*Jzz)[NC])
    {
        const double GL = -1e-3 * P.gLeak;
        fz[0] = sQ + GL * (Vm - P.ELeak) + qdrive;
        if (JAC) Jzz[0][0] = (sCond + GL) * H.vs;
    }
    SONIC_HD static double eca(const Params &, double) { return 0.0; }
};

template <>
struct GroupModel<ThalamicRE> {
    typedef ThalamicRE M;
    typedef REParams Params;
    static constexpr int NC = 1, NX = 0, NT = M::NT, NY = M::NY, NCOL = NY + 3;
    static constexpr bool HAS_CAI = false, HAS_CAIGATE = false, HAS_ECA = false, HAS_X2 = false, HAS_GHK = false;
    static constexpr bool PIN_COEFFS = true;   // the stage coefficients in vector registers, where there is room for them
    SONIC_HD static int xtab(int) { return 0; }
    SONIC_HD static int core_col(int) { return 2; }
    // reference columns: t stim Qm m h n s u Vm
    static bool lanes(const Params &P, LaneSpec *s)
    {
        for (int i = 0; i < GRP; i++) s[i] = lane_none();
        s[0] = lane_owner(lane_gate(0, 3), P.gNabar, P.ENa, 3, true);
        s[1] = lane_gate(1, 4); s[1].r1 = 1.0;
        s[2] = lane_owner(lane_gate(2, 5), P.gKdbar, P.EK, 4, false);
        s[4] = lane_owner(lane_gate(3, 6), P.gCaTbar, P.ECa, 2, true);
        s[5] = lane_gate(4, 7); s[5].r1 = 1.0;
        const int cols[4] = {0, 1, 2, 8}, ssel[4] = {GS_T, GS_X, GS_Z0, GS_VM};
        lane_scalars(s, 4, cols, ssel);
        return true;
    }
    template <bool JAC, class Cell>
    SONIC_HD static void core(const Params &P, const Cell &H, double Vm, const double *, double sQ, double,
                              double qdrive, double *fz, double sCond, double, double (*Jzz)[NC])
    {
        const double GL = -1e-3 * P.gLeak;
        fz[0] = sQ + GL * (Vm - P.ELeak) + qdrive;
        if (JAC) Jzz[0][0] = (sCond + GL) * H.vs;
    }
    SONIC_HD static double eca(const Params &, double) { return 0.0; }
};

// TC: core [Q, Cai, P0, O, C]; extra lines alphao, betao (tables 11, 12); thalamic.py:182-366
template <>
struct GroupModel<ThalamoCortical> {
    typedef ThalamoCortical M;
    typedef TCParams Params;
    static constexpr int NC = 5, NX = 2, NT = M::NT, NY = M::NY, NCOL = NY + 3;
    static constexpr bool HAS_CAI = true, HAS_CAIGATE = false, HAS_ECA = false, HAS_X2 = false, HAS_GHK = false;
    static constexpr bool PIN_COEFFS = false;   // the stage coefficients in vector registers, where there is room for them
    SONIC_HD static int xtab(int i) { return 11 + i; }
    // reference columns: t stim Qm m h n s u Cai P0 O C Vm
    SONIC_HD static int core_col(int c) { return c == 0 ? 2 : 7 + c; }
    static bool lanes(const Params &P, LaneSpec *s)
    {
        for (int i = 0; i < GRP; i++) s[i] = lane_none();
        const double kap = 1e3 * P.c2m;                 // dCai/dt = ... - c2m iCaT, lane currents are -1e-3 i
        s[0] = lane_owner(lane_gate(0, 3), P.gNabar, P.ENa, 3, true);
        s[1] = lane_gate(1, 4); s[1].r1 = 1.0;
        s[2] = lane_owner(lane_gate(2, 5), P.gKdbar, P.EK, 4, false);
        s[4] = lane_owner(lane_gate(3, 6), P.gCaTbar, P.ECa, 2, true); s[4].kap = kap;
        s[5] = lane_gate(4, 7); s[5].r1 = 1.0; s[5].kap = kap;
        const int cols[8] = {0, 1, 2, 8, 9, 10, 11, 12};
        const int ssel[8] = {GS_T, GS_X, GS_Z0, GS_Z0 + 1, GS_Z0 + 2, GS_Z0 + 3, GS_Z0 + 4, GS_VM};
        lane_scalars(s, 8, cols, ssel);
        return true;
    }
    template <bool JAC, class Cell>
    SONIC_HD static void core(const Params &P, const Cell &H, double Vm, const double *z, double sQ, double sC,
                              double qdrive, double *fz, double sCond, double sKCond, double (*Jzz)[NC])
    {
        const double dq = z[0] - H.xlo;
        const double ao = H.xs[0] * dq + H.xv[0], bo = H.xs[1] * dq + H.xv[1];
        const double Cai = z[1], P0 = z[2], O = z[3], C = z[4];
        const double dH = Vm - P.EH;
        const double gH = P.gHbar * (O + 2.0 * (1.0 - O - C));
        const double gpas = P.gKLeak + P.gLeak + gH;
        fz[0] = sQ - 1e-3 * (P.gKLeak * (Vm - P.EK) + P.gLeak * (Vm - P.ELeak) + gH * dH) + qdrive;
        const double inv_tr = 1.0 / P.taur_Cai;
        const double Cai2 = Cai * Cai, Cai4 = Cai2 * Cai2;                  // nCa = 4
        fz[1] = (P.Cai_min - Cai) * inv_tr + sC;
        fz[2] = P.k2 * (1.0 - P0) - P.k1 * P0 * Cai4;
        fz[3] = ao * C - bo * O - P.k3 * O * (1.0 - P0) + P.k4 * (1.0 - O - C);
        fz[4] = bo * O - ao * C;
        if (JAC) {
#pragma unroll
            for (int a = 0; a < NC; a++)
#pragma unroll
                for (int b = 0; b < NC; b++) Jzz[a][b] = 0.0;
            Jzz[0][0] = (sCond - 1e-3 * gpas) * H.vs;
            Jzz[0][3] = 1e-3 * P.gHbar * dH;            // d gH / dO = -gHbar
            Jzz[0][4] = 2e-3 * P.gHbar * dH;            // d gH / dC = -2 gHbar
            Jzz[1][0] = sKCond * H.vs;
            Jzz[1][1] = -inv_tr;
            Jzz[2][1] = -4.0 * P.k1 * P0 * Cai2 * Cai;
            Jzz[2][2] = -P.k2 - P.k1 * Cai4;
            Jzz[3][0] = H.xs[0] * C - H.xs[1] * O;
            Jzz[3][2] = P.k3 * O;
            Jzz[3][3] = -bo - P.k3 * (1.0 - P0) - P.k4;
            Jzz[3][4] = ao - P.k4;
            Jzz[4][0] = H.xs[1] * O - H.xs[0] * C;
            Jzz[4][3] = bo;
            Jzz[4][4] = -ao;
        }
    }
    SONIC_HD static double eca(const Params &, double) { return 0.0; }
};

// STN: core [Q, Cai]; d2 and r are gates driven by Cai; ECa is the Nernst potential of Cai; stn.py:157-430
template <>
struct GroupModel<OtsukaSTN> {
    typedef OtsukaSTN M;
    typedef STNParams Params;
    static constexpr int NC = 2, NX = 0, NT = M::NT, NY = M::NY, NCOL = NY + 3;
    static constexpr bool HAS_CAI = true, HAS_CAIGATE = true, HAS_ECA = true, HAS_X2 = true, HAS_GHK = false;
    static constexpr bool PIN_COEFFS = false;   // the stage coefficients in vector registers, where there is room for them
    SONIC_HD static int xtab(int) { return 0; }
    // reference columns: t stim Qm m h n a b p q c d1 d2 r Cai Vm; table order a b c d1 m h n p q
    SONIC_HD static int core_col(int c) { return c == 0 ? 2 : 14; }
    static bool lanes(const Params &P, LaneSpec *s)
    {
        for (int i = 0; i < GRP; i++) s[i] = lane_none();
        const double kap = 1e3 * P.c2m;
        s[0] = lane_owner(lane_gate(4, 3), P.gNabar, P.ENa, 3, true);          // m: iNa = gNa m^3 h (V - ENa)
        s[1] = lane_gate(5, 4); s[1].r1 = 1.0;                                 // h
        s[2] = lane_owner(lane_gate(6, 5), P.gKdbar, P.EK, 4, false);          // n: iKd = gKd n^4 (V - EK)
        s[4] = lane_owner(lane_gate(0, 6), P.gAbar, P.EK, 2, true);            // a: iA = gA a^2 b (V - EK)
        s[5] = lane_gate(1, 7); s[5].r1 = 1.0;                                 // b
        s[6] = lane_owner(lane_gate(7, 8), P.gCaTbar, 0.0, 2, true);           // p: iCaT = gCaT p^2 q (V - ECa)
        s[6].eCa = 1.0; s[6].kap = kap;
        s[7] = lane_gate(8, 9); s[7].r1 = 1.0; s[7].kap = kap;                 // q
        s[8] = lane_owner(lane_gate(2, 10), P.gCaLbar, 0.0, 2, true, true);    // c: iCaL = gCaL c^2 d1 d2 (V - ECa)
        s[8].eCa = 1.0; s[8].kap = kap;
        s[9] = lane_gate(3, 11); s[9].r1 = 1.0; s[9].kap = kap;                // d1
        s[10] = lane_gate(-1, 12); s[10].r2 = 1.0; s[10].kap = kap;            // d2 (Cai)
        s[10].itau = 1.0 / P.tau_d2; s[10].theta = P.thetax_d2; s[10].ikx = 1.0 / P.kx_d2;
        s[11] = lane_owner(lane_gate(-1, 13), P.gKCabar, P.EK, 2, false);      // r (Cai): iKCa = gKCa r^2 (V - EK)
        s[11].itau = 1.0 / P.tau_r; s[11].theta = P.thetax_r; s[11].ikx = 1.0 / P.kx_r;
        const int cols[5] = {0, 1, 2, 14, 15}, ssel[5] = {GS_T, GS_X, GS_Z0, GS_Z0 + 1, GS_VM};
        lane_scalars(s, 5, cols, ssel);
        return true;
    }
    // nernst(Z_Ca, Cai, Cao, T) (pneuron.py:339-349)
    SONIC_HD static double eca(const Params &P, double Cai) { return P.nernst_mV * fast_log(qdiv(P.Cao, Cai)); }
    template <bool JAC, class Cell>
    SONIC_HD static void core(const Params &P, const Cell &H, double Vm, const double *z, double sQ, double sC,
                              double qdrive, double *fz, double sCond, double sKCond, double (*Jzz)[NC])
    {
        const double GL = -1e-3 * P.gLeak;
        const double inv_tr = 1.0 / P.taur_Cai;
        fz[0] = sQ + GL * (Vm - P.ELeak) + qdrive;
        fz[1] = sC - z[1] * inv_tr;
        if (JAC) {
            const double dE = qdiv(P.nernst_mV, z[1]);        // -dECa/dCai
            Jzz[0][0] = (sCond + GL) * H.vs;
            Jzz[0][1] = sKCond * (1.0 / (1e3 * P.c2m)) * dE;    // the currents that see ECa are those that feed Cai
            Jzz[1][0] = sKCond * H.vs;
            Jzz[1][1] = sKCond * dE - inv_tr;
        }
    }
};

// Data-driven gated neurons (HHseg, SWnode, MRGnode, SUseg, FHnode, passive; sonic_models.hpp: GatedModel): up to
// four currents g_c prod_k x_k^e_ck (Vm - E_c) -- or the Goldman-Hodgkin-Katz force of (Cin_c, Cout_c) -- and a
// leak. Current c takes the quad of lanes 4c .. 4c + 3: its gate of highest exponent owns it, the other one or
// two gates (exponent 1) sit on the next lanes. Layouts this scheme cannot express -- a gate shared by two
// currents, a second gate with an exponent other than 1, more than three gates in a current, a current without
// gates -- make lanes() return false and the batch runs on the lane-per-configuration kernel.
template <int NGATES>
struct GroupModel<GatedModel<NGATES>> {
    typedef GatedModel<NGATES> M;
    typedef GatedParams<NGATES> Params;
    static constexpr int NC = 1, NX = 0, NT = M::NT, NY = M::NY, NCOL = NY + 3;
    static constexpr bool HAS_CAI = false, HAS_CAIGATE = false, HAS_ECA = false, HAS_X2 = true, HAS_GHK = true;
    static constexpr bool PIN_COEFFS = true;   // the stage coefficients in vector registers, where there is room for them
    SONIC_HD static int xtab(int) { return 0; }
    SONIC_HD static int core_col(int) { return 2; }
    static bool lanes(const Params &P, LaneSpec *s)
    {
        for (int i = 0; i < GRP; i++) s[i] = lane_none();
        bool used[NGATES];
        for (int k = 0; k < NGATES; k++) used[k] = false;
        for (int c = 0; c < GATED_MAX_CURRENTS; c++) {
            int gates[3], ng = 0, total = 0;
            for (int k = 0; k < NGATES; k++) {
                const int e = (int)P.expo[c][k];
                if (e <= 0) continue;
                if (used[k] || ng == 3) return false;
                gates[ng++] = k;
                total++;
            }
            if (P.g[c] == 0.0 && total == 0) continue;
            if (total == 0) return false;                       // a gate-free current: not a leak we know of
            // owner = the gate of highest exponent
            int io = 0;
            for (int i = 1; i < ng; i++)
                if (P.expo[c][gates[i]] > P.expo[c][gates[io]]) io = i;
            const int ko = gates[io];
            for (int i = 0; i < ng; i++)
                if (i != io && (int)P.expo[c][gates[i]] != 1) return false;
            const int base = 4 * c;
            s[base] = lane_owner(lane_gate(ko, 3 + ko), P.g[c], P.E[c], (int)P.expo[c][ko], ng >= 2, ng >= 3);
            if ((int)P.expo[c][ko] > 4) return false;
            s[base].ghk = P.ghk[c] != 0.0 ? 1.0 : 0.0;
            s[base].Cin = P.Cin[c];
            s[base].Cout = P.Cout[c];
            used[ko] = true;
            int slot = 1;
            for (int i = 0; i < ng; i++) {
                if (i == io) continue;
                const int k = gates[i];
                s[base + slot] = lane_gate(k, 3 + k);
                if (slot == 1) s[base + slot].r1 = 1.0; else s[base + slot].r2 = 1.0;
                used[k] = true;
                slot++;
            }
        }
        // gates without a current (the padding gate of the passive neuron): a lane of their own, after the quads in use
        int free_lane = 0;
        for (int k = 0; k < NGATES; k++) {
            if (used[k]) continue;
            while (free_lane < GRP && (s[free_lane].tab >= 0)) free_lane++;
            if (free_lane >= GRP) return false;
            s[free_lane] = lane_gate(k, 3 + k);
        }
        // t, stimstate, Qm, Vm on the first four lanes that are free of a second column (all are)
        const int cols[4] = {0, 1, 2, 3 + NGATES}, ssel[4] = {GS_T, GS_X, GS_Z0, GS_VM};
        lane_scalars(s, 4, cols, ssel);
        return true;
    }
    SONIC_HD static bool any_ghk(const Params &P)
    {
        bool any = false;
#pragma unroll
        for (int c = 0; c < GATED_MAX_CURRENTS; c++) any = any || P.ghk[c] != 0.0;
        return any;
    }
    template <bool JAC, class Cell>
    SONIC_HD static void core(const Params &P, const Cell &H, double Vm, const double *, double sQ, double,
                              double qdrive, double *fz, double sCond, double, double (*Jzz)[NC])
    {
        const double GL = -1e-3 * P.gLeak;
        fz[0] = sQ + GL * (Vm - P.ELeak) + qdrive;
        if (JAC) Jzz[0][0] = (sCond + GL) * H.vs;
    }
    SONIC_HD static double eca(const Params &, double) { return 0.0; }
};

// ---------------------------------------------------------------------------------------------
template <class O, int NX>
struct GroupCell {
    typename O::V av, as, bv, bs;    // lines of the lane's gate in this cell
    double xlo, xhi, vv, vs;         // cell bounds and V line (replicated)
    double xv[NX > 0 ? NX : 1], xs[NX > 0 ? NX : 1];     // lines the core reads (replicated)
};

// Level records in HBM / L2, in the layout of the lane kernels (sonic_integrator.hpp):
// [Q_j, Q_j+1, (value, slope) of table 0 = V, (value, slope) of tables 1 .. NT-1]
template <class O, class GM>
struct GroupTab {
    typedef const double *Ref;
    static constexpr int REC = 2 + 2 * GM::NT;
    const double *recs;
    int level_stride;        // doubles per level = n_cells * REC
    SONIC_HD Ref level(int id) const { return recs + (size_t)id * level_stride; }
    SONIC_HD void load(Ref lvl, int j, const GroupConsts<O> &C, GroupCell<O, GM::NX> &S) const
    {
        const double *r = lvl + (size_t)j * REC;
        S.xlo = r[0]; S.xhi = r[1]; S.vv = r[2]; S.vs = r[3];
#pragma unroll
        for (int i = 0; i < GM::NX; i++) { S.xv[i] = r[2 + 2 * GM::xtab(i)]; S.xs[i] = r[3 + 2 * GM::xtab(i)]; }
        O::load_lines(r, C.tab, S.av, S.as, S.bv, S.bs);
    }
    SONIC_HD void vline(Ref lvl, int j, double &xlo, double &xhi, double &vv, double &vs) const
    {
        const double *r = lvl + (size_t)j * REC;
        xlo = r[0]; xhi = r[1]; vv = r[2]; vs = r[3];
    }
};

// V table (np.interp semantics) at charge q, any cell: only for output rows whose charge lies
// outside the home cell of the step that produced them
template <class Tab>
SONIC_HD double group_vm_at(const QuadGrid &G, const Tab &T, typename Tab::Ref lvl, double q)
{
    if (!(q >= G.q0 && q <= G.qmax)) return NAN;
    int j = (int)((q - G.q0) * G.inv_dq);
    j = j < 0 ? 0 : (j > G.n_cells - 1 ? G.n_cells - 1 : j);
    for (;;) {
        double xlo, xhi, vv, vs;
        T.vline(lvl, j, xlo, xhi, vv, vs);
        if (q < xlo && j > 0) j--;
        else if (q >= xhi && j < G.n_cells - 1) j++;
        else return vs * (q - xlo) + vv;
    }
}

// One evaluation of the lane parts at (z, x) with the lines of cell H
template <class O>
struct GroupRhs {
    typename O::V fg, r, gpw, f1, f2, drive, ddrive, cur, xinf;
    double Vm;
};

template <class O, class GM, class Cell>
SONIC_HD void group_rhs(const typename GM::Params &P, const Cell &H, const GroupConsts<O> &C,
                        const double *z, const typename O::V &x, GroupRhs<O> &R)
{
    typedef typename O::V V;
    const double dq = z[0] - H.xlo;
    const V dqv = O::splat(dq);
    V a = O::fma_(H.as, dqv, H.av);
    const V b = O::fma_(H.bs, dqv, H.bv);
    V r = O::add(a, b);
    if constexpr (GM::HAS_CAIGATE) {
        // x_inf(Cai) = 1 / (1 + exp((Cai - theta) / kx)): a += x_inf / tau, r += 1 / tau (zero on the other lanes)
        const V u = O::mul(O::sub(O::splat(z[1]), C.theta), C.ikx);
        R.xinf = O::rcp(O::add(O::splat(1.0), O::exp_(u)));
        a = O::fma_(R.xinf, C.itau, a);
        r = O::add(r, C.itau);
    }
    R.r = r;
    R.fg = O::sub(a, O::mul(r, x));
    const V pw = O::mul(x, O::fma_(x, O::fma_(x, O::fma_(x, C.c4, C.c3), C.c2), C.c1));
    R.f1 = O::fma_(O::swap1(x), C.k1, C.nk1);
    R.Vm = H.vs * dq + H.vv;
    V E = C.E0;
    if constexpr (GM::HAS_ECA) E = O::fma_(C.eCa, O::splat(GM::eca(P, z[1])), C.E0);
    R.drive = O::sub(O::splat(R.Vm), E);
    if constexpr (GM::HAS_GHK) {
        // Goldman-Hodgkin-Katz force of a monovalent ion (pneuron.py:361-375, ghk_drive of sonic_models.hpp) on the
        // lanes that ask for it: the exponentials are those of the replicated potential, the concentrations the lane's
        R.ddrive = O::splat(1.0);
        if (GM::any_ghk(P)) {
            const double xv = GHK_X_PER_MV * R.Vm;
            const double ep = exp(xv) - 1.0, em = exp(-xv) - 1.0;
            const double fp = xv / ep, fm = -xv / em;
            const double dfp = (1.0 - fp * (ep + 1.0)) / ep, dfm = -(1.0 - fm * (em + 1.0)) / em;
            const V dg = O::mul(O::splat(GHK_FARADAY * 1e6), O::sub(O::mul(C.Cin, O::splat(fm)), O::mul(C.Cout, O::splat(fp))));
            const V ddg = O::mul(O::splat(GHK_FARADAY * 1e6 * GHK_X_PER_MV),
                                 O::sub(O::mul(C.Cin, O::splat(dfm)), O::mul(C.Cout, O::splat(dfp))));
            // drive = ghk ? dg : Vm - E ; ddrive = ghk ? ddg : 1   (ghk is 0 or 1)
            R.drive = O::fma_(C.ghk, O::sub(dg, R.drive), R.drive);
            R.ddrive = O::fma_(C.ghk, O::sub(ddg, O::splat(1.0)), O::splat(1.0));
        }
    }
    R.gpw = O::mul(C.G, pw);
    V cond = O::mul(R.gpw, R.f1);
    if constexpr (GM::HAS_X2) {
        R.f2 = O::fma_(O::swap2(x), C.k2, C.nk2);
        cond = O::mul(cond, R.f2);
    }
    R.cur = O::mul(cond, R.drive);
}

// in-place Doolittle LU without pivoting of the replicated core block; diagonal stored as reciprocal
template <int NC>
SONIC_HD void group_lu(double (*A)[NC])
{
#pragma unroll
    for (int k = 0; k < NC; k++) {
        A[k][k] = fast_rcp1(A[k][k]);
#pragma unroll
        for (int r = k + 1; r < NC; r++) {
            A[r][k] *= A[k][k];
#pragma unroll
            for (int d = k + 1; d < NC; d++) A[r][d] -= A[r][k] * A[k][d];
        }
    }
}
template <int NC>
SONIC_HD void group_lu_solve(const double (*A)[NC], double *b)
{
#pragma unroll
    for (int c = 1; c < NC; c++)
#pragma unroll
        for (int d = 0; d < c; d++) b[c] -= A[c][d] * b[d];
#pragma unroll
    for (int c = NC - 1; c >= 0; c--) {
#pragma unroll
        for (int d = c + 1; d < NC; d++) b[c] -= A[c][d] * b[d];
        b[c] *= A[c][c];
    }
}

// Integrate a STREAM of configurations with the group layout. emit(row, t, x, z, gates V, Vm).
// Loop structure as integrate_stream_quad (sonic_quad.hpp): `src` hands the row its configurations one after the
// other (next / done), one place loads lookup lines, every iteration is one step attempt, and the switch to the
// next configuration sits inside that flat loop -- a row whose configuration ends takes the next of the batch's
// work queue while the other rows of its wavefront go on stepping.
// STREAM = false: one configuration, the hand-in after the loop (see integrate_stream_quad).
template <bool STREAM, class O, class GM, class Tab, class Emit, class Source>
SONIC_HD void integrate_stream_group(const typename GM::Params &P, const GroupConsts<O> &C, const QuadGrid &G,
                                     const Tab &T, const double *y0ref, const SolverOpts &o, Emit &&emit, Source &src)
{
    int ncap = 0, nover = 0, ncross = 0;
    using namespace rodas4;
    typedef typename O::V V;
    constexpr int NC = GM::NC;
    // the 25 stage coefficients as locals that shadow rodas4's: held in vector registers where the model leaves
    // room (a 64-bit literal costs two scalar moves per use, and in a wavefront alone on its SIMD a scalar move
    // takes the issue slot of an FMA), plain constants otherwise
#define GROUP_COEFF(name) const double name = GM::PIN_COEFFS ? O::pin(rodas4::name) : rodas4::name
    GROUP_COEFF(a21); GROUP_COEFF(a31); GROUP_COEFF(a32); GROUP_COEFF(a41); GROUP_COEFF(a42); GROUP_COEFF(a43);
    GROUP_COEFF(a51); GROUP_COEFF(a52); GROUP_COEFF(a53); GROUP_COEFF(a54);
    GROUP_COEFF(c21); GROUP_COEFF(c31); GROUP_COEFF(c32); GROUP_COEFF(c41); GROUP_COEFF(c42); GROUP_COEFF(c43);
    GROUP_COEFF(c51); GROUP_COEFF(c52); GROUP_COEFF(c53); GROUP_COEFF(c54);
    GROUP_COEFF(c61); GROUP_COEFF(c62); GROUP_COEFF(c63); GROUP_COEFF(c64); GROUP_COEFF(c65);
#undef GROUP_COEFF
    GroupCell<O, GM::NX> H;              // home cell
    double z[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) z[c] = y0ref[GM::core_col(c) - 2];
    V xg = O::init_gates(y0ref, C.colx);
    int status = ST_OK, nsteps = 0, nrej = 0;
    long row = 0;
    bool dead = false;
    Schedule S{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    bool have = src.next(S);

    typename Tab::Ref lvl = T.level(0);  // records of the current level (row 0: level 0)
    int jh = (int)((z[0] - G.q0) * G.inv_dq);   // cell index (hint until need_cell has run)
    bool need_cell = true, seg_init = true, row0 = true;

    int s = 0, irow = 0;
    double x = 0.0, t = 0.0, h = o.h0, tr = 0.0;
    Linspace grid = linspace_make(0.0, 0.0, 2);
    const float rtol = (float)o.rtol, atol = (float)o.atol;

    auto kill = [&]() SONIC_LAMBDA_INLINE {
#pragma unroll
        for (int c = 0; c < NC; c++) z[c] = NAN;
        xg = O::splat(NAN);
    };

    // STREAM: the switch to the next configuration sits OUTSIDE the loop of step attempts, which every lane of the
    // wavefront leaves together -- a uniform branch -- as soon as the configuration of one of its quads (rows) has
    // ended: the quad (row) concerned takes its next configuration, the others pass, and all go on stepping. With
    // the switch inside the loop (`if (s >= nseg) { ...; continue; }`) the compiler pays for the two dozen values the
    // switch resets with register copies in EVERY iteration (+10 % vector, +27 % scalar instructions per step on
    // the 4096-cell map, profiles/r03i_stream_ab.txt).
    if (!have) return;
    for (;;) {
        if constexpr (STREAM) {
            while (have && s >= S.nseg) {
                // this configuration has ended: hand in its counters, take the next one and start over
                StepCounts cnt_;
                cnt_.capped = ncap; cnt_.over = nover; cnt_.cross = ncross;
                src.done(status, nsteps, nrej, cnt_);
                have = src.next(S);
#pragma unroll
                for (int c = 0; c < NC; c++) z[c] = y0ref[GM::core_col(c) - 2];
                xg = O::init_gates(y0ref, C.colx);
                status = ST_OK; nsteps = 0; nrej = 0; ncap = 0; nover = 0; ncross = 0;
                row = 0; dead = false;
                lvl = T.level(0);
                jh = (int)((z[0] - G.q0) * G.inv_dq);
                need_cell = true; seg_init = true; row0 = true;
                s = 0; irow = 0;
                x = 0.0; t = 0.0; h = o.h0; tr = 0.0;
            }
            if (!O::wave_any(have)) break;
        }
        if (STREAM ? have : s < S.nseg) do {
        if (need_cell | seg_init) {       // (one test for both)
            if (need_cell) {
                need_cell = false;
                if (!dead) {
                    const double q = z[0];
                    if (!(q >= G.q0 && q <= G.qmax)) { dead = true; status |= ST_Q_OUT_OF_RANGE; }
                    else {
                        int j = jh < 0 ? 0 : (jh > G.n_cells - 1 ? G.n_cells - 1 : jh);
                        for (;;) {
                            T.load(lvl, j, C, H);
                            if (q < H.xlo && j > 0) j--;
                            else if (q >= H.xhi && j < G.n_cells - 1) j++;
                            else break;
                        }
                        jh = j;
                    }
                }
            }
            if (seg_init) {
                if (row0) {
                    // row 0: initial conditions, stimstate 0, Vm from the A = 0 tables (level 0)
                    row0 = false;
                    emit(row++, S.t0[0], 0.0, z, xg, dead ? NAN : H.vs * (z[0] - H.xlo) + H.vv);
                    if (S.level[0] != 0) {
                        lvl = T.level(S.level[0]);
                        need_cell = true;
                        continue;
                    }
                }
                seg_init = false;
                grid = linspace_make(S.t0[s], S.t1[s], S.n[s]);
                x = S.x[s];
                const double Vm = dead ? NAN : H.vs * (z[0] - H.xlo) + H.vv;
                if (dead) kill();
                emit(row++, grid.t0, x, z, xg, Vm);
                irow = 1;
                t = grid.t0;
                h = fmin(o.h0, grid.delta);
                if (dead || !(grid.t1 - grid.t0 > SONIC_SEG_EPS)) {
                    for (; irow < grid.n; irow++) emit(row++, linspace_at(grid, irow), x, z, xg, Vm);
                    s++;
                    seg_init = true;
                    if (s < S.nseg) {
                        lvl = T.level(S.level[s]);
                        need_cell = true;
                    }
                    continue;
                }
                tr = quad_linspace_at(grid, irow);
            }
        }

        const double cellw = H.xhi - H.xlo;
        bool capped;
        // ---- f(y) and the Jacobian with the home cell's lines (re-evaluated after a rejected step too) ----
        double f0z[NC], Jzz[NC][NC];
        V f0g, rr, jq, JgQ, JgC = O::splat(0.0);
        {
            GroupRhs<O> R;
            group_rhs<O, GM>(P, H, C, z, xg, R);
            f0g = R.fg;
            rr = R.r;
            const V other = GM::HAS_X2 ? O::mul(R.f1, R.f2) : R.f1;
            const V cond = O::mul(R.gpw, other);
            const double sQ = O::allsum(R.cur);
            double sCond;
            if constexpr (GM::HAS_GHK) sCond = O::allsum(O::mul(cond, R.ddrive));     // d (sum of currents) / d Vm
            else sCond = O::allsum(cond);
            double sC = 0.0, sKCond = 0.0;
            if constexpr (GM::HAS_CAI) {
                sC = O::allsum(O::mul(C.kap, R.cur));
                sKCond = O::allsum(O::mul(C.kap, cond));
            }
            GM::template core<true>(P, H, R.Vm, z, sQ, sC, o.qdrive, f0z, sCond, sKCond, Jzz);
            // d (sum of the currents) / d x_g: the owner through pw', its neighbours through f1 / f2
            const V dpw = O::fma_(xg, O::fma_(xg, O::fma_(xg, C.d4, C.d3), C.d2), C.c1);
            const V own = O::mul(O::mul(O::mul(C.G, dpw), other), R.drive);
            const V gd = O::mul(R.gpw, R.drive);
            jq = O::fma_(O::swap1(GM::HAS_X2 ? O::mul(gd, R.f2) : gd), C.r1, own);
            if constexpr (GM::HAS_X2) jq = O::fma_(O::swap2(O::mul(gd, R.f1)), C.r2, jq);
            JgQ = O::sub(H.as, O::mul(O::add(H.as, H.bs), xg));
            if constexpr (GM::HAS_CAIGATE)     // d x_inf / d Cai / tau = -x_inf (1 - x_inf) / kx / tau
                JgC = O::mul(O::mul(O::mul(R.xinf, O::sub(R.xinf, O::splat(1.0))), C.ikx), C.itau);
        }
        {
            // kink-aware cap: time for Q to reach the node it is heading to, plus a sliver
            const double dist = f0z[0] > 0.0 ? (H.xhi - z[0]) + SONIC_LANE_OV_TARGET * cellw
                                             : (H.xlo - z[0]) - SONIC_LANE_OV_TARGET * cellw;
            // second-order prediction of the time to the node (node_time_*, sonic_integrator.hpp)
            double fp = O::allsum(O::mul(jq, f0g));
#pragma unroll
            for (int b = 0; b < NC; b++) fp += Jzz[0][b] * f0z[b];
            const float fq = (float)f0z[0], dd = (float)dist;
            const float root = O::sqrtf_(node_time_discriminant(fq, (float)fp, dd));
            const float hc = 2.0f * dd * O::rcpf(node_time_denominator(fq, (float)fp, dd, root));
            capped = hc > 0.0f && (double)hc < h;
            h = capped ? fmax((double)hc, 1e-3 * h) : h;
        }
        const bool last = t + 1.0001 * h >= grid.t1;
        h = last ? grid.t1 - t : h;
        const double inv_h = fast_rcp1(h);

        // ---- W = I/(h gamma) - J: gates eliminated lane-wise, Schur complement of the core replicated ----
        const double c0 = inv_h * (1.0 / gamma);
        const V invd = O::rcp(O::add(O::splat(c0), rr));
        const V wq = O::mul(jq, invd);
        double A[NC][NC];
#pragma unroll
        for (int a = 0; a < NC; a++)
#pragma unroll
            for (int b = 0; b < NC; b++) A[a][b] = (a == b ? c0 : 0.0) - Jzz[a][b];
        {
            const V wj = O::mul(wq, JgQ);
            A[0][0] -= O::allsum(wj);
            if constexpr (GM::HAS_CAI) A[1][0] -= O::allsum(O::mul(C.kap, wj));
            if constexpr (GM::HAS_CAIGATE) {
                const V wc = O::mul(wq, JgC);
                A[0][1] -= O::allsum(wc);
                A[1][1] -= O::allsum(O::mul(C.kap, wc));
            }
        }
        group_lu<NC>(A);

        // k = W^-1 (rz | rg): one butterfly per core row that the gates feed
        double kz[6][NC], zt[NC];
        V kg[6], xt;
        auto solve = [&](int i, const double *fz_, const V &tsum, const V &rg) SONIC_LAMBDA_INLINE {
            double b[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) b[c] = fz_[c];
            b[0] += O::allsum(tsum);
            if constexpr (GM::HAS_CAI) b[1] += O::allsum(O::mul(C.kap, tsum));
            group_lu_solve<NC>(A, b);
#pragma unroll
            for (int c = 0; c < NC; c++) kz[i][c] = b[c];
            V num = O::fma_(JgQ, O::splat(b[0]), rg);
            if constexpr (GM::HAS_CAIGATE) num = O::fma_(JgC, O::splat(b[1]), num);
            kg[i] = O::mul(num, invd);
        };
        // stage i >= 1 at (zt, xt) with the increments cz (core) / cg (gates) of the earlier stages
        auto stage = [&](int i, const double *cz, const V &cg) SONIC_LAMBDA_INLINE {
            GroupRhs<O> R;
            group_rhs<O, GM>(P, H, C, zt, xt, R);
            const V rg = O::add(R.fg, cg);
            double fz_[NC];
            // the sums of the currents go through the same butterfly as the eliminated gates
            GM::template core<false>(P, H, R.Vm, zt, 0.0, 0.0, o.qdrive, fz_, 0.0, 0.0, nullptr);
#pragma unroll
            for (int c = 0; c < NC; c++) fz_[c] += cz[c];
            solve(i, fz_, O::fma_(wq, rg, R.cur), rg);
        };
        {
            double fz_[NC];
            // f0z already holds the sums of the currents: only the eliminated gates go through the butterfly
#pragma unroll
            for (int c = 0; c < NC; c++) fz_[c] = f0z[c];
            solve(0, fz_, O::mul(wq, f0g), f0g);
        }
        double cz[NC];
        {
            const double g1 = c21 * inv_h;
#pragma unroll
            for (int c = 0; c < NC; c++) { zt[c] = z[c] + a21 * kz[0][c]; cz[c] = g1 * kz[0][c]; }
            xt = O::fma_(O::splat(a21), kg[0], xg);
            stage(1, cz, O::mul(O::splat(g1), kg[0]));
        }
        {
            const double g1 = c31 * inv_h, g2 = c32 * inv_h;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                zt[c] = z[c] + a31 * kz[0][c] + a32 * kz[1][c];
                cz[c] = g1 * kz[0][c] + g2 * kz[1][c];
            }
            xt = O::fma_(O::splat(a32), kg[1], O::fma_(O::splat(a31), kg[0], xg));
            stage(2, cz, O::fma_(O::splat(g2), kg[1], O::mul(O::splat(g1), kg[0])));
        }
        {
            const double g1 = c41 * inv_h, g2 = c42 * inv_h, g3 = c43 * inv_h;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                zt[c] = z[c] + a41 * kz[0][c] + a42 * kz[1][c] + a43 * kz[2][c];
                cz[c] = g1 * kz[0][c] + g2 * kz[1][c] + g3 * kz[2][c];
            }
            xt = O::fma_(O::splat(a43), kg[2], O::fma_(O::splat(a42), kg[1], O::fma_(O::splat(a41), kg[0], xg)));
            stage(3, cz, O::fma_(O::splat(g3), kg[2], O::fma_(O::splat(g2), kg[1], O::mul(O::splat(g1), kg[0]))));
        }
        {
            const double g1 = c51 * inv_h, g2 = c52 * inv_h, g3 = c53 * inv_h, g4 = c54 * inv_h;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                zt[c] = z[c] + a51 * kz[0][c] + a52 * kz[1][c] + a53 * kz[2][c] + a54 * kz[3][c];
                cz[c] = g1 * kz[0][c] + g2 * kz[1][c] + g3 * kz[2][c] + g4 * kz[3][c];
            }
            xt = O::fma_(O::splat(a54), kg[3], O::fma_(O::splat(a53), kg[2],
                 O::fma_(O::splat(a52), kg[1], O::fma_(O::splat(a51), kg[0], xg))));
            stage(4, cz, O::fma_(O::splat(g4), kg[3], O::fma_(O::splat(g3), kg[2],
                         O::fma_(O::splat(g2), kg[1], O::mul(O::splat(g1), kg[0])))));
        }
        {
            const double g1 = c61 * inv_h, g2 = c62 * inv_h, g3 = c63 * inv_h, g4 = c64 * inv_h, g5 = c65 * inv_h;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                zt[c] += kz[4][c];
                cz[c] = g1 * kz[0][c] + g2 * kz[1][c] + g3 * kz[2][c] + g4 * kz[3][c] + g5 * kz[4][c];
            }
            xt = O::add(xt, kg[4]);
            stage(5, cz, O::fma_(O::splat(g5), kg[4], O::fma_(O::splat(g4), kg[3], O::fma_(O::splat(g3), kg[2],
                         O::fma_(O::splat(g2), kg[1], O::mul(O::splat(g1), kg[0]))))));
        }
        nsteps++;

        double znew[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) znew[c] = zt[c] + kz[5][c];
        const V xnew = O::add(xt, kg[5]);
        // embedded error estimate = k6; scaled RMS norm over all states, single precision
        float err;
        {
            float e2 = O::errsum(kg[5], xg, xnew, C.errw, atol, rtol);
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const float sc = atol + rtol * fmaxf(fabsf((float)z[c]), fabsf((float)znew[c]));
                const float e = (float)kz[5][c] * O::rcpf(sc);
                e2 += e * e;
            }
            err = O::sqrtf_(e2 * (1.0f / GM::NY));
        }
        // step-size controller (Hairer & Wanner IV.7): rfac = 0.9 err^(-1/4) clipped to [0.2, 6]
        float rfac = 0.9f * O::rsqf(O::sqrtf_(err));
        rfac = fminf(6.0f, fmaxf(0.2f, rfac));
        rfac = err == err ? rfac : 0.2f;
        // all stages used the home cell's lines: a step that ends too far outside it is retried
        // with a secant-corrected size that ends SONIC_LANE_OV_TARGET past the node
        const double qnew = znew[0];
        const double over = fmax(H.xlo - qnew, qnew - H.xhi);
        const bool overshoot = over > SONIC_LANE_OV_MAX * cellw;
        const float moved = fabsf((float)(qnew - z[0]));
        const float want = moved - (float)over + (float)(SONIC_LANE_OV_TARGET * cellw);
        const float sfac = fmaxf(0.1f, fminf(0.9f, want * O::rcpf(moved)));
        const double hnew = h * (double)(overshoot ? sfac : rfac);
        const bool accept = err <= 1.0f && !overshoot;
        ncap += (accept && capped) ? 1 : 0;
        nover += overshoot ? 1 : 0;
        const double tnew = last ? grid.t1 : t + h;
        if (accept & (irow < grid.n) & (last | (tr <= tnew))) {       // (one combined test: no short-circuit branches)
            // dense output for every grid row inside (t, tnew]
            double c3z[NC], c4z[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) {
                c3z[c] = d21 * kz[0][c] + d22 * kz[1][c] + d23 * kz[2][c] + d24 * kz[3][c] + d25 * kz[4][c];
                c4z[c] = d31 * kz[0][c] + d32 * kz[1][c] + d33 * kz[2][c] + d34 * kz[3][c] + d35 * kz[4][c];
            }
            const V c3g = O::fma_(O::splat(d25), kg[4], O::fma_(O::splat(d24), kg[3],
                          O::fma_(O::splat(d23), kg[2], O::fma_(O::splat(d22), kg[1],
                          O::mul(O::splat(d21), kg[0])))));
            const V c4g = O::fma_(O::splat(d35), kg[4], O::fma_(O::splat(d34), kg[3],
                          O::fma_(O::splat(d33), kg[2], O::fma_(O::splat(d32), kg[1],
                          O::mul(O::splat(d31), kg[0])))));
            do {
                const bool end = tr >= tnew;
                const double sg = end ? 1.0 : (tr - t) * inv_h, s1 = 1.0 - sg;
                // y s1 + sg (ynew + s1 (c3 + sg c4)); the row at tnew is ynew itself
                double zr[NC];
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const double zi = z[c] * s1 + sg * (znew[c] + s1 * (c3z[c] + sg * c4z[c]));
                    zr[c] = end ? znew[c] : zi;
                }
                const V mid = O::fma_(O::splat(s1), O::fma_(O::splat(sg), c4g, c3g), xnew);
                const V xi = O::fma_(O::splat(sg), mid, O::mul(xg, O::splat(s1)));
                const V xr = O::select(end, xnew, xi);
                // Vm = lerp of the V table at the row's charge (nbls.py:426-428)
                double Vm = H.vs * (zr[0] - H.xlo) + H.vv;
                if (!(zr[0] >= H.xlo && zr[0] < H.xhi)) Vm = group_vm_at(G, T, lvl, zr[0]);
                emit(row++, tr, x, zr, xr, Vm);
                irow++;
                tr = quad_linspace_at(grid, irow);
            } while (irow < grid.n && (last || tr <= tnew));
        }
        // state update (selects: accepted and rejected steps share the path)
        nrej += accept ? 0 : 1;
#pragma unroll
        for (int c = 0; c < NC; c++) z[c] = accept ? znew[c] : z[c];
        xg = O::select(accept, xnew, xg);
        t = accept ? tnew : t;
        h = accept ? hnew : fmin(hnew, h);
        // kink-aware steps end just past a node: the new home cell is the neighbour
        const bool cross = accept && !(z[0] >= H.xlo && z[0] < H.xhi);
        ncross += cross ? 1 : 0;
        jh += cross ? (z[0] >= H.xhi ? 1 : -1) : 0;
        need_cell = need_cell || cross;
        // the rare endings of a step behind ONE test (see integrate_stream_quad)
        if ((accept & last) | !(h >= o.hmin) | (nsteps >= o.max_steps) | dead) {
            if (accept && last) {
                s++;
                seg_init = true;
                if (s < S.nseg) {
                    lvl = T.level(S.level[s]);
                    need_cell = true;
                }
            }
            if (!(h >= o.hmin)) { dead = true; status |= ST_STEP_UNDERFLOW; }
            if (nsteps >= o.max_steps && !dead && !seg_init) { dead = true; status |= ST_MAX_STEPS; }
            if (dead && !seg_init) {
                // fill the rest of this segment with NaN rows; later segments take the dead path
                kill();
                for (; irow < grid.n; irow++) emit(row++, linspace_at(grid, irow), x, z, xg, NAN);
                s++;
                seg_init = true;
                if (s < S.nseg) lvl = T.level(S.level[s]);
            }
        }
        } while (STREAM ? !O::wave_any(s >= S.nseg) : s < S.nseg);
        if constexpr (!STREAM) break;
    }
    if constexpr (!STREAM) {
        StepCounts cnt_;
        cnt_.capped = ncap; cnt_.over = nover; cnt_.cross = ncross;
        src.done(status, nsteps, nrej, cnt_);
    }
}

// One configuration (the CPU harness, tests): a source of one (QuadSingleSource, sonic_quad.hpp)
template <class O, class GM, class Tab, class Emit>
SONIC_HD int integrate_config_group(const typename GM::Params &P, const GroupConsts<O> &C, const QuadGrid &G,
                                    const Tab &T, const Schedule &S, const double *y0ref, const SolverOpts &o,
                                    Emit &&emit, int *nsteps_out, int *nrej_out, StepCounts *counts = nullptr)
{
    QuadSingleSource src{S};
    integrate_stream_group<false, O, GM>(P, C, G, T, y0ref, o, emit, src);
    if (nsteps_out) *nsteps_out = src.nsteps;
    if (nrej_out) *nrej_out = src.nrej;
    if (counts) *counts = src.counts;
    return src.status;
}

}  // namespace sonic
