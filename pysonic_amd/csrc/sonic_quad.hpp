// pysonic_amd/csrc/sonic_quad.hpp
//
// QUAD-COOPERATIVE integrator for the cortical RS / FS neurons: one stimulus configuration per
// quad of 4 adjacent lanes (16 configurations per wavefront) instead of one per lane.
//
// Why: a single configuration is a strictly sequential chain of ~10^4 Rosenbrock steps, so a
// batch of a few thousand configurations (BASELINE config 2: 4096) cannot fill 1024 SIMDs with
// one lane per configuration and its run time is the LATENCY of the slowest configuration. The
// four gates m, h, n, p are independent given the charge, so the four lanes of a quad each own
// one gate (state, stage increments, lookup lines, Jacobian column) and the charge equation is
// replicated; the quad exchanges data with DPP quad_perm moves only (register-to-register, no
// LDS): one swap (h -> the m lane, m^3 -> the h lane) and one butterfly all-reduce per
// right-hand side / solve. The per-lane instruction stream of a step is ~3x shorter than in the
// lane-per-configuration kernel, and so is the critical path. Registers drop from ~320 to ~100
// per lane, so 4 wavefronts fit per SIMD.
//
// The arithmetic is the same Rosenbrock / home-cell scheme as sonic_integrator.hpp (see there for the
// references into PySONIC); sums over gates are formed by the butterfly, so results agree with
// the lane-per-configuration kernel to rounding (not bitwise).
//
// The code is written once over an `Ops` backend: on the device a "quad vector" is one double per
// lane and Ops uses DPP; the CPU test harness uses 4-element arrays (tests/native, development).
#pragma once
#include "sonic_integrator.hpp"

#ifndef SONIC_QUAD_METHOD
#define SONIC_QUAD_METHOD 5     // 4: RODAS4 (6 stages); 5: ROS4 with Shampine's parameters (4 stages, 3 evaluations)
#endif

namespace sonic {

#ifdef SONIC_QUAD_STATS
// development (CPU harness only): what limits the steps of a configuration
struct QuadStats {
    long steps, capped, capped_acc, errlim_acc, last_acc, rej_err, rej_over, cross;
    long err_hist_capped[8];      // accepted node-capped steps by err: <1e-4, <1e-3, <1e-2, <0.1, <0.3, <0.6, <1, >=1
    long err_hist_free[8];
    double sum_h_capped, sum_h_free;
    int last_dom; long dom_free[5], dom_cap[5];
    double *log;                  // [cap][5]: t, h, err, capped, accept
    long nlog, caplog;
};
inline QuadStats &quad_stats() { static QuadStats s{}; return s; }
#endif

// Record layout of the quad kernel, per level and charge cell (20 doubles):
//   [0] Q_j  [1] Q_{j+1}  [2] V value  [3] V slope  then for gate g = 0..3 (m h n p):
//   [4 + 4g] alpha value, alpha slope, beta value, beta slope
constexpr int QUAD_REC = 20;

struct QuadGrid {
    const double *recs;   // [n_levels][n_cells][QUAD_REC]
    int n_cells;
    double q0, qmax, inv_dq;
};

// ---- CPU emulation backend: V = 4 values, one per gate --------------------------------------
struct QuadOpsHost {
    struct V {
        double v[4];
    };
    static V splat(double a) { return V{{a, a, a, a}}; }
    static double pin(double a) { return a; }
    static V roles(double a0, double a1, double a2, double a3) { return V{{a0, a1, a2, a3}}; }
    static V add(V a, V b) { V r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] + b.v[i]; return r; }
    static V sub(V a, V b) { V r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] - b.v[i]; return r; }
    static V mul(V a, V b) { V r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] * b.v[i]; return r; }
    static V fma_(V a, V b, V c) { V r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] * b.v[i] + c.v[i]; return r; }
    static V rcp(V a) { V r; for (int i = 0; i < 4; i++) r.v[i] = 1.0 / a.v[i]; return r; }
    static V swap1(V a) { return V{{a.v[1], a.v[0], a.v[3], a.v[2]}}; }
    static double allsum(V a) { return (a.v[0] + a.v[1]) + (a.v[2] + a.v[3]); }
    static float allsumf(V a) { return (float)((a.v[0] + a.v[1]) + (a.v[2] + a.v[3])); }
    static double pick(V a, int g) { return a.v[g]; }
    // per-role selection: result[g] = (g == 0 ? a0 : g == 1 ? a1 : g == 2 ? a2 : a3)
    static V byrole(V a0, V a1, V a2, V a3) { return V{{a0.v[0], a1.v[1], a2.v[2], a3.v[3]}}; }
    // lookup lines of the home cell: value and slope of alpha_g, beta_g for every gate
    static void load_gate_lines(const double *rec, V &av, V &as, V &bv, V &bs)
    {
        for (int g = 0; g < 4; g++) {
            av.v[g] = rec[4 + 4 * g]; as.v[g] = rec[5 + 4 * g];
            bv.v[g] = rec[6 + 4 * g]; bs.v[g] = rec[7 + 4 * g];
        }
    }
    static void store_row(double *r, double t, double x, double q, V g, double Vm)
    {
        r[0] = t; r[1] = x; r[2] = q;
        for (int i = 0; i < 4; i++) r[3 + i] = g.v[i];
        r[7] = Vm;
    }
    // sum over the gates of (e_g / (atol + rtol max(|a_g|, |b_g|)))^2 in single precision
    static float errsum(V e, V a, V b, float atol, float rtol)
    {
        float acc[4];
        for (int i = 0; i < 4; i++) {
            const float sc = atol + rtol * fmaxf(fabsf((float)a.v[i]), fabsf((float)b.v[i]));
            const float r = (float)e.v[i] / sc;
            acc[i] = r * r;
        }
        return (acc[0] + acc[1]) + (acc[2] + acc[3]);
    }
    static V select(bool c, V a, V b) { return c ? a : b; }
    static float rcpf(float a) { return 1.0f / a; }
    static float sqrtf_(float a) { return sqrtf(a); }
    static float rsqf(float a) { return 1.0f / sqrtf(a); }
    static bool leader() { return true; }
    static bool wave_any(bool c) { return c; }       // the harness integrates one configuration at a time
};

#if defined(__HIPCC__)
// ---- device backend: V = one double per lane; lane & 3 = gate index -------------------------
struct QuadOpsDev {
    typedef double V;
    // a constant held in a vector register pair for the whole kernel: a 64-bit literal is otherwise rebuilt with
    // two scalar moves at every use, and in a wavefront that runs alone on its SIMD a scalar move takes the
    // same issue slot as an FMA
    static __device__ __forceinline__ double pin(double a)
    {
#ifndef SONIC_QUAD_NOPIN
        asm volatile("" : "+v"(a));
#endif
        return a;
    }
    template <int CTRL>
    static __device__ __forceinline__ double dpp(double x)
    {
        // every lane of a quad is active whenever its quad is: no `old` value is needed
        const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
        return __hiloint2double(hi, lo);
    }
    static __device__ __forceinline__ int role() { return threadIdx.x & 3; }
    static __device__ __forceinline__ V splat(double a) { return a; }
    static __device__ __forceinline__ V roles(double a0, double a1, double a2, double a3)
    {
        const int g = role();
        return g == 0 ? a0 : (g == 1 ? a1 : (g == 2 ? a2 : a3));
    }
    static __device__ __forceinline__ V add(V a, V b) { return a + b; }
    static __device__ __forceinline__ V sub(V a, V b) { return a - b; }
    static __device__ __forceinline__ V mul(V a, V b) { return a * b; }
    static __device__ __forceinline__ V fma_(V a, V b, V c) { return fma(a, b, c); }
    static __device__ __forceinline__ V rcp(V a) { return fast_rcp1(a); }
    static __device__ __forceinline__ V swap1(V a) { return dpp<0xB1>(a); }          // [1,0,3,2]
    // The butterfly gives every lane of the quad the SAME bits only if each lane adds the same
    // two rounded numbers: `a` must therefore be materialised first. Without the barrier the
    // compiler contracts a product feeding `a` into the first add (fma(c_i, d_i, a_j) on lane i
    // vs fma(c_j, d_j, a_i) on lane j), the replicated results differ in the last bit, the
    // lanes of a quad eventually take different branches and DPP reads disabled lanes.
    static __device__ __forceinline__ double allsum(V a)
    {
        asm volatile("" : "+v"(a));
        a += dpp<0xB1>(a);      // + neighbour          [1,0,3,2]
        a += dpp<0x4E>(a);      // + other pair         [2,3,0,1]
        return a;
    }
    static __device__ __forceinline__ float allsumf(V a)
    {
        float f = (float)a;
        asm volatile("" : "+v"(f));
        f += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(f), 0xB1, 0xf, 0xf, true));
        f += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(f), 0x4E, 0xf, 0xf, true));
        return f;
    }
    static __device__ __forceinline__ double pick(V a, int g)
    {
        switch (g) {
        case 0: return dpp<0x00>(a);
        case 1: return dpp<0x55>(a);
        case 2: return dpp<0xAA>(a);
        default: return dpp<0xFF>(a);
        }
    }
    static __device__ __forceinline__ V byrole(V a0, V a1, V a2, V a3)
    {
        const int g = role();
        return g == 0 ? a0 : (g == 1 ? a1 : (g == 2 ? a2 : a3));
    }
    static __device__ __forceinline__ void load_gate_lines(const double *rec, V &av, V &as, V &bv,
                                                           V &bs)
    {
        const double2 *p = (const double2 *)(rec + 4 + 4 * role());
        const double2 a = p[0], b = p[1];
        av = a.x; as = a.y; bv = b.x; bs = b.y;
    }
    // row = [t, stim, Qm, m, h, n, p, Vm]: lane g stores doubles 2g, 2g+1 -> one coalesced 64-B row
    static __device__ __forceinline__ void store_row(double *r, double t, double x, double q, V g,
                                                     double Vm)
    {
        const double prev = dpp<0x93>(g);     // lane i gets gate i-1   [3,0,1,2]
        const int role_ = role();
        double2 w;
        w.x = role_ == 0 ? t : (role_ == 1 ? q : (role_ == 2 ? prev : g));   // t | Qm | h | p
        w.y = role_ == 0 ? x : (role_ == 3 ? Vm : (role_ == 1 ? prev : g));   // stim | m | n | Vm
        ((double2 *)r)[role_] = w;
    }
    static __device__ __forceinline__ float errsum(V e, V a, V b, float atol, float rtol)
    {
        const float sc = atol + rtol * fmaxf(fabsf((float)a), fabsf((float)b));
        const float r = (float)e * __builtin_amdgcn_rcpf(sc);
        float f = r * r;
        asm volatile("" : "+v"(f));     // see allsum
        f += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(f), 0xB1, 0xf, 0xf, true));
        f += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(f), 0x4E, 0xf, 0xf, true));
        return f;
    }
    static __device__ __forceinline__ V select(bool c, V a, V b) { return c ? a : b; }
    // single-precision helpers of the step-size controller: hardware approximations (1 ulp)
    static __device__ __forceinline__ float rcpf(float a) { return __builtin_amdgcn_rcpf(a); }
    static __device__ __forceinline__ float sqrtf_(float a) { return __builtin_amdgcn_sqrtf(a); }
    static __device__ __forceinline__ float rsqf(float a) { return __builtin_amdgcn_rsqf(a); }
    static __device__ __forceinline__ bool leader() { return role() == 0; }
    static constexpr int WIDTH = 4;      // lanes per configuration
    // the leader's value on every lane of the quad
    static __device__ __forceinline__ int from_leader(int v) { return __builtin_amdgcn_mov_dpp(v, 0x00, 0xf, 0xf, true); }   // quad_perm [0,0,0,0]
    // true on every active lane of the wavefront if `c` holds on any of them (a scalar condition: branching on it
    // does not touch the exec mask)
    static __device__ __forceinline__ bool wave_any(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0; }
};
#endif

// Per-lane constants of the current term each lane owns -- cortical.py:92-119, pneuron.py:288-296
//   m lane: iNa  = gNa m^3 h (V - ENa)        h lane: iLeak = gLeak (V - ELeak)
//   n lane: iKd  = gKd n^4 (V - EK)           p lane: iM    = gM p (V - EK)
// written without per-lane branches or selects: the gate factor is the polynomial
//   pw(x) = c0 + c1 x + c3 x^3 + c4 x^4   with (c0, c1, c3, c4) = m:(0,0,1,0) h:(1,0,0,0) n:(0,0,0,1) p:(0,1,0,0)
// and the m lane multiplies by the h it fetches from its neighbour: other = xo * c3 + (1 - c3).
// G carries the factor -1e-3 of dQ/dt = -1e-3 iNet (nbls.py:307).
template <class O>
struct QuadConsts {
    typename O::V G, E, c0, c1, c3, c4, nc3, d2, d3;   // d2 = 3 c3, d3 = 4 c4 (derivative of pw)
};

// `qdrive` (the injected current of DrivenNeuronalBilayerSonophore, as a term of dQ/dt) is folded into
// the leak: -1e-3 gLeak (V - ELeak) + qdrive = -1e-3 gLeak (V - (ELeak + qdrive / (1e-3 gLeak))), so it
// costs no instruction in the loop (the host keeps gLeak = 0 with a current on the lane kernel).
template <class O>
SONIC_HD QuadConsts<O> quad_consts(const CorticalParams &P, double qdrive = 0.0)
{
    QuadConsts<O> C;
    C.G = O::roles(-1e-3 * P.gNabar, -1e-3 * P.gLeak, -1e-3 * P.gKdbar, -1e-3 * P.gMbar);
    const double ELeak = qdrive != 0.0 ? P.ELeak + qdrive / (1e-3 * P.gLeak) : P.ELeak;
    C.E = O::roles(P.ENa, ELeak, P.EK, P.EK);
    C.c0 = O::roles(0.0, 1.0, 0.0, 0.0);
    C.c1 = O::roles(0.0, 0.0, 0.0, 1.0);
    C.c3 = O::roles(1.0, 0.0, 0.0, 0.0);
    C.c4 = O::roles(0.0, 0.0, 1.0, 0.0);
    C.nc3 = O::roles(0.0, 1.0, 1.0, 1.0);
    C.d2 = O::roles(3.0, 0.0, 0.0, 0.0);
    C.d3 = O::roles(0.0, 0.0, 4.0, 0.0);
    return C;
}

template <class O>
struct QuadCell {
    typename O::V av, as, bv, bs;    // lines of alpha_g, beta_g in this cell (per lane)
    double xlo, xhi, vv, vs;         // cell bounds and V line (replicated)
};

// linspace_at without the `step == 0` special case of np.linspace: that case only differs for a
// subnormal step (a zero-length segment gives t0 either way)
SONIC_HD double quad_linspace_at(const Linspace &g, int i)
{
#pragma clang fp contract(off)
    const double prod = (double)i * g.step;
    const double v = prod + g.t0;
    return i >= g.n - 1 ? g.t1 : v;
}

// Where the level records live. TabGlobal reads them from HBM/L2 (any batch); TabLds (device only,
// sonic_lib.hip) from the copy of the wavefront's two levels in LDS. `Ref` designates one level.
template <class O>
struct TabGlobal {
    typedef const double *Ref;
    const double *recs;
    int level_stride;        // doubles per level = n_cells * QUAD_REC
    SONIC_HD Ref level(int id) const { return recs + (size_t)id * level_stride; }
    SONIC_HD void load(Ref lvl, int j, QuadCell<O> &S) const
    {
        const double *r = lvl + j * QUAD_REC;
        S.xlo = r[0]; S.xhi = r[1]; S.vv = r[2]; S.vs = r[3];
        O::load_gate_lines(r, S.av, S.as, S.bv, S.bs);
    }
    SONIC_HD void vline(Ref lvl, int j, double &xlo, double &xhi, double &vv, double &vs) const
    {
        const double *r = lvl + j * QUAD_REC;
        xlo = r[0]; xhi = r[1]; vv = r[2]; vs = r[3];
    }
};

// V table (np.interp semantics) at charge q, any cell: only for output rows whose charge lies
// outside the home cell of the step that produced them
template <class Tab>
SONIC_HD double quad_vm_at(const QuadGrid &G, const Tab &T, typename Tab::Ref lvl, double q)
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

// Gate part of the right-hand side at (q, x) with the lines of cell S, and the current term of
// each lane: returns fg (per lane), cond = G pw other (per lane), drive = V - E (per lane).
template <class O>
SONIC_HD void quad_rhs(const QuadCell<O> &S, const QuadConsts<O> &C, double q,
                       const typename O::V &x, typename O::V &fg, typename O::V &r,
                       typename O::V &gpw, typename O::V &other, typename O::V &drive)
{
    typedef typename O::V V;
    const double dq = q - S.xlo;
    const V dqv = O::splat(dq);
    const V a = O::fma_(S.as, dqv, S.av), b = O::fma_(S.bs, dqv, S.bv);
    r = O::add(a, b);
    fg = O::sub(a, O::mul(r, x));
    const V x2 = O::mul(x, x), x3 = O::mul(x2, x), x4 = O::mul(x2, x2);
    const V pw = O::fma_(x4, C.c4, O::fma_(x3, C.c3, O::fma_(x, C.c1, C.c0)));
    other = O::fma_(O::swap1(x), C.c3, C.nc3);      // m lane: h ; other lanes: 1
    drive = O::sub(O::splat(S.vs * dq + S.vv), C.E);
    gpw = O::mul(C.G, pw);
}

// Integrate a STREAM of configurations with the quad layout. emit(row, t, x, q, gates V, Vm).
//
// `src` hands the quad its configurations one after the other:
//     bool src.next(Schedule &S)                      the next one (false: none left); also called to get the first
//     void src.done(status, nsteps, nrej, counts)     the one just integrated has ended
// In the kernel a quad whose configuration ends takes the next of the batch's work queue at once, while the other
// quads of its wavefront go on stepping: the wavefront stays full instead of waiting, masked, for its slowest
// member (the reference's pool hands a free worker its next item the same way, batches.py:33-43, 108-128).
//
// Loop structure: ONE place loads lookup lines (`need_cell`, top of the loop) and every iteration
// is one step attempt. The quads of a wavefront diverge (one emits rows, another crosses a node,
// a third starts a segment), and a wavefront issues the union of the paths its quads take, so the
// loop body is kept small rather than fast on any single path.
//
// STREAM = false: the source holds ONE configuration. The loop is then the plain `while (s < nseg)` with the
// hand-in after it: keeping the switch inside the loop costs the 4096-cell map 11 % (11.9 against 10.7 ms per launch,
// profiles/r03i_stream_ab.txt) although it never runs there -- done() and the state it needs stay live across the
// step block.
template <bool STREAM, class O, class Tab, class Emit, class Source>
SONIC_HD void integrate_stream_quad(const CorticalParams &P, const QuadGrid &G, const Tab &T, const double *y0,
                                    const SolverOpts &o, Emit &&emit, Source &src)
{
    int ncap = 0, nover = 0, ncross = 0;
    using namespace rodas4;
    typedef typename O::V V;
    const QuadConsts<O> C = quad_consts<O>(P, o.qdrive);
    QuadCell<O> H;                       // home cell
    Schedule S{nullptr, nullptr, nullptr, nullptr, nullptr, 0};
    bool have = src.next(S);
    double q = y0[0];
    V xg = O::roles(y0[1], y0[2], y0[3], y0[4]);
    int status = ST_OK, nsteps = 0, nrej = 0;
    long row = 0;
    bool dead = false;

    typename Tab::Ref lvl = T.level(0);  // records of the current level (row 0: level 0)
    int jh = (int)((q - G.q0) * G.inv_dq);   // cell index (hint until need_cell has run)
    // STREAM (launches with more wavefronts than SIMDs): the cell the charge is heading for, requested a step ahead.
    // A step that the node predictor caps ends just past the node and the next one needs the neighbour's record at
    // once -- an L2 round trip that grows with the number of wavefronts in flight (550 clocks with one per compute
    // unit, 2 500 with four). Requested as soon as the direction is known, the record arrives while the stages of
    // the step run: the 65 536-configuration sweep 20.9 -> 20.0 ms. Not in the plain build: one wavefront per SIMD
    // (the 4096-cell map) gains nothing from it (10.25 / 10.32 ms) -- its SIMD issues vector instructions in 56 % of
    // the cycles and scalar ones in 12 %, and waits for memory in 2 % (profiles/r03t_bench_sq_counters.json).
    QuadCell<O> N = QuadCell<O>();
    int jn = -1;                         // cell index of N at level `lvl`, -1: none
    bool need_cell = true, seg_init = true, row0 = true;

    int s = 0, irow = 0;
    double x = 0.0, t = 0.0, h = o.h0, tr = 0.0;
    Linspace grid = linspace_make(0.0, 0.0, 2);
    double f0Q = 0.0, Jqq = 0.0;
    V f0g = O::splat(0.0), Jqg = f0g, Jgq = f0g, Dg = f0g;
    const float rtol = (float)o.rtol, atol = (float)o.atol;

#if SONIC_QUAD_METHOD != 4
    // coefficients of ROS4 (Shampine) used in every step (O::pin)
    const double A31 = O::pin(48.0 / 25.0), A32 = O::pin(6.0 / 25.0), C21 = O::pin(-8.0), C31 = O::pin(372.0 / 25.0),
                 C32 = O::pin(12.0 / 5.0), C41 = O::pin(-112.0 / 125.0), C42 = O::pin(-54.0 / 125.0),
                 C43 = O::pin(-2.0 / 5.0), B1 = O::pin(19.0 / 9.0), B3 = O::pin(25.0 / 108.0),
                 B4 = O::pin(125.0 / 108.0), E1 = O::pin(17.0 / 54.0), E2 = O::pin(7.0 / 36.0);
#endif

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
                q = y0[0];
                xg = O::roles(y0[1], y0[2], y0[3], y0[4]);
                status = ST_OK; nsteps = 0; nrej = 0; ncap = 0; nover = 0; ncross = 0;
                row = 0; dead = false;
                lvl = T.level(0);
                jh = (int)((q - G.q0) * G.inv_dq);
                jn = -1;
                need_cell = true; seg_init = true; row0 = true;
                s = 0; irow = 0;
                x = 0.0; t = 0.0; h = o.h0; tr = 0.0;
            }
            if (!O::wave_any(have)) break;
        }
        if (STREAM ? have : s < S.nseg) do {
        // (both rare-ish starts of an iteration behind one test, see the end of the loop body)
        if (need_cell | seg_init) {
            if (need_cell) {
                need_cell = false;
                if (!dead) {
                    if (!(q >= G.q0 && q <= G.qmax)) { dead = true; status |= ST_Q_OUT_OF_RANGE; }
                    else {
                        int j = jh < 0 ? 0 : (jh > G.n_cells - 1 ? G.n_cells - 1 : jh);
                        for (;;) {
                            if (STREAM && j == jn) H = N;        // (fetched during the last step)
                            else T.load(lvl, j, H);
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
                    emit(row++, S.t0[0], 0.0, q, xg, dead ? NAN : H.vs * (q - H.xlo) + H.vv);
                    if (S.level[0] != 0) {
                        lvl = T.level(S.level[0]);
                        jn = -1;
                        need_cell = true;
                        continue;
                    }
                }
                seg_init = false;
                grid = linspace_make(S.t0[s], S.t1[s], S.n[s]);
                x = S.x[s];
                const double Vm = dead ? NAN : H.vs * (q - H.xlo) + H.vv;
                if (dead) { q = NAN; xg = O::splat(NAN); }
                emit(row++, grid.t0, x, q, xg, Vm);
                irow = 1;
                t = grid.t0;
                h = fmin(o.h0, grid.delta);
                if (dead || !(grid.t1 - grid.t0 > SONIC_SEG_EPS)) {
                    for (; irow < grid.n; irow++) emit(row++, linspace_at(grid, irow), x, q, xg, Vm);
                    s++;
                    seg_init = true;
                    if (s < S.nseg) {
                        lvl = T.level(S.level[s]);
                        jn = -1;
                        need_cell = true;
                    }
                    continue;
                }
                tr = quad_linspace_at(grid, irow);
            }
        }

        const double cellw = H.xhi - H.xlo;
        bool capped;
        {
            // f(y) and the Jacobian with the home cell's lines. Re-evaluated after a rejected
            // step too (8 % of the steps): cheaper than a divergent branch around it.
            V r, gpw, other, drive;
            quad_rhs<O>(H, C, q, xg, f0g, r, gpw, other, drive);
            const V cond = O::mul(gpw, other);
            f0Q = O::allsum(O::mul(cond, drive));
            Jqq = H.vs * O::allsum(cond);
            // d fQ / d x_g: own gate through pw'; the h lane's entry is the m lane's G m^3 (V - ENa)
            const V x2 = O::mul(xg, xg), x3 = O::mul(x2, xg);
            const V dpw = O::fma_(x3, C.d3, O::fma_(x2, C.d2, C.c1));
            const V own = O::mul(O::mul(O::mul(C.G, dpw), other), drive);
            Jqg = O::fma_(O::swap1(O::mul(gpw, drive)), C.c0, own);
            Jgq = O::sub(H.as, O::mul(O::add(H.as, H.bs), xg));
            Dg = O::sub(O::splat(0.0), r);
        }
        {
            // kink-aware cap: time for Q to reach the node it is heading to, plus a sliver
            const double dist = f0Q > 0.0 ? (H.xhi - q) + SONIC_OV_TARGET * cellw
                                          : (H.xlo - q) - SONIC_OV_TARGET * cellw;
            // second-order prediction of the time to the node (node_time_*, sonic_integrator.hpp)
            const float fq = (float)f0Q, dd = (float)dist;
            const float fp = (float)(Jqq * f0Q + O::allsum(O::mul(Jqg, f0g)));
            const float root = O::sqrtf_(node_time_discriminant(fq, fp, dd));
            const float hc = 2.0f * dd * O::rcpf(node_time_denominator(fq, fp, dd, root));
            capped = hc > 0.0f && (double)hc < h;
            h = capped ? fmax((double)hc, 1e-3 * h) : h;
            // the neighbour this step is heading into (see N)
            if constexpr (STREAM) {
                const int jt = jh + (f0Q > 0.0 ? 1 : -1);
                if (capped & (jt != jn) & (jt >= 0) & (jt < G.n_cells)) {
                    T.load(lvl, jt, N);
                    jn = jt;
                }
            }
        }
        const bool last = t + 1.0001 * h >= grid.t1;
        h = last ? grid.t1 - t : h;
        const double inv_h = fast_rcp1(h);   // enters W and the c_ij / h terms alike: 1e-14 relative is round-off

#if SONIC_QUAD_METHOD == 4
        // ---- W = I/(h gamma) - J (arrow matrix): per lane invd, w; replicated pivot ----
        const double c0 = inv_h * (1.0 / gamma);
        const V invd = O::rcp(O::sub(O::splat(c0), Dg));
        const V w = O::mul(Jqg, invd);
        const double piv = fast_rcp1(c0 - Jqq - O::allsum(O::mul(w, Jgq)));

        // ---- six stages; k = W^-1 (f(Y_s) + sum_j c_sj/h k_j) ----
        // stage 1 from f0; stages 2..6: ONE butterfly gives sum_g (current term + w_g r_g)
        double k1Q, k2Q, k3Q, k4Q, k5Q, k6Q, qt;
        V k1g, k2g, k3g, k4g, k5g, k6g, xt;
        {
            const double b = (f0Q + O::allsum(O::mul(w, f0g))) * piv;
            k1Q = b;
            k1g = O::mul(O::fma_(Jgq, O::splat(b), f0g), invd);
        }
#define QUAD_STAGE(KQ, KG, CQ, CG)                                                        \
        {                                                                                 \
            V fg_, r_, gpw_, other_, drive_;                                              \
            quad_rhs<O>(H, C, qt, xt, fg_, r_, gpw_, other_, drive_);                     \
            const V rg_ = O::add(fg_, CG);                                                \
            const V term_ = O::mul(O::mul(gpw_, other_), drive_);                         \
            const double b_ = (O::allsum(O::fma_(w, rg_, term_)) + (CQ)) * piv;           \
            KQ = b_;                                                                      \
            KG = O::mul(O::fma_(Jgq, O::splat(b_), rg_), invd);                           \
        }
        qt = q + a21 * k1Q;
        xt = O::fma_(O::splat(a21), k1g, xg);
        {
            const double g1 = c21 * inv_h;
            QUAD_STAGE(k2Q, k2g, g1 * k1Q, O::mul(O::splat(g1), k1g));
        }
        qt = q + a31 * k1Q + a32 * k2Q;
        xt = O::fma_(O::splat(a32), k2g, O::fma_(O::splat(a31), k1g, xg));
        {
            const double g1 = c31 * inv_h, g2 = c32 * inv_h;
            QUAD_STAGE(k3Q, k3g, g1 * k1Q + g2 * k2Q,
                       O::fma_(O::splat(g2), k2g, O::mul(O::splat(g1), k1g)));
        }
        qt = q + a41 * k1Q + a42 * k2Q + a43 * k3Q;
        xt = O::fma_(O::splat(a43), k3g,
                     O::fma_(O::splat(a42), k2g, O::fma_(O::splat(a41), k1g, xg)));
        {
            const double g1 = c41 * inv_h, g2 = c42 * inv_h, g3 = c43 * inv_h;
            QUAD_STAGE(k4Q, k4g, g1 * k1Q + g2 * k2Q + g3 * k3Q,
                       O::fma_(O::splat(g3), k3g,
                               O::fma_(O::splat(g2), k2g, O::mul(O::splat(g1), k1g))));
        }
        qt = q + a51 * k1Q + a52 * k2Q + a53 * k3Q + a54 * k4Q;
        xt = O::fma_(O::splat(a54), k4g,
                     O::fma_(O::splat(a53), k3g,
                             O::fma_(O::splat(a52), k2g, O::fma_(O::splat(a51), k1g, xg))));
        {
            const double g1 = c51 * inv_h, g2 = c52 * inv_h, g3 = c53 * inv_h, g4 = c54 * inv_h;
            QUAD_STAGE(k5Q, k5g, g1 * k1Q + g2 * k2Q + g3 * k3Q + g4 * k4Q,
                       O::fma_(O::splat(g4), k4g,
                               O::fma_(O::splat(g3), k3g,
                                       O::fma_(O::splat(g2), k2g, O::mul(O::splat(g1), k1g)))));
        }
        qt += k5Q;
        xt = O::add(xt, k5g);
        {
            const double g1 = c61 * inv_h, g2 = c62 * inv_h, g3 = c63 * inv_h, g4 = c64 * inv_h,
                         g5 = c65 * inv_h;
            QUAD_STAGE(k6Q, k6g, g1 * k1Q + g2 * k2Q + g3 * k3Q + g4 * k4Q + g5 * k5Q,
                       O::fma_(O::splat(g5), k5g,
                               O::fma_(O::splat(g4), k4g,
                                       O::fma_(O::splat(g3), k3g,
                                               O::fma_(O::splat(g2), k2g,
                                                       O::mul(O::splat(g1), k1g))))));
        }
#undef QUAD_STAGE
        nsteps++;

        const double qnew = qt + k6Q;
        const V xnew = O::add(xt, k6g);
        // embedded error estimate = k6; scaled RMS norm over (Q, m, h, n, p), single precision
        float err;
        {
            const float scQ = atol + rtol * fmaxf(fabsf((float)q), fabsf((float)qnew));
            const float eQ = (float)k6Q * O::rcpf(scQ);
            err = O::sqrtf_((eQ * eQ + O::errsum(k6g, xg, xnew, atol, rtol)) * 0.2f);
        }
#else
        // ---- W = I/(h gamma) - J (arrow matrix), gamma = 1/2: per lane invd, w; replicated pivot ----
        const double c0 = inv_h * 2.0;
        const V invd = O::rcp(O::sub(O::splat(c0), Dg));
        const V w = O::mul(Jqg, invd);
        const double piv = fast_rcp1(c0 - Jqq - O::allsum(O::mul(w, Jgq)));

        // ---- four stages, three evaluations (ros4s_step, sonic_integrator.hpp); stage 1 from f0,
        //      stages 2..4: ONE butterfly gives sum_g (current term + w_g r_g) ----
        double k1Q, k2Q, k3Q, k4Q, qt;
        V k1g, k2g, k3g, k4g, xt;
        {
            const double b = (f0Q + O::allsum(O::mul(w, f0g))) * piv;
            k1Q = b;
            k1g = O::mul(O::fma_(Jgq, O::splat(b), f0g), invd);
        }
        V fg_, term_;
#define QUAD_EVAL()                                                                       \
        {                                                                                 \
            V r_, gpw_, other_, drive_;                                                   \
            quad_rhs<O>(H, C, qt, xt, fg_, r_, gpw_, other_, drive_);                     \
            term_ = O::mul(O::mul(gpw_, other_), drive_);                                 \
        }
#define QUAD_SOLVE(KQ, KG, CQ, CG)                                                        \
        {                                                                                 \
            const V rg_ = O::add(fg_, CG);                                                \
            const double b_ = (O::allsum(O::fma_(w, rg_, term_)) + (CQ)) * piv;           \
            KQ = b_;                                                                      \
            KG = O::mul(O::fma_(Jgq, O::splat(b_), rg_), invd);                           \
        }
        qt = q + 2.0 * k1Q;
        xt = O::fma_(O::splat(2.0), k1g, xg);
        QUAD_EVAL();
        {
            const double g1 = C21 * inv_h;
            QUAD_SOLVE(k2Q, k2g, g1 * k1Q, O::mul(O::splat(g1), k1g));
        }
        qt = q + A31 * k1Q + A32 * k2Q;
        xt = O::fma_(O::splat(A32), k2g, O::fma_(O::splat(A31), k1g, xg));
        QUAD_EVAL();
        {
            const double g1 = C31 * inv_h, g2 = C32 * inv_h;
            QUAD_SOLVE(k3Q, k3g, g1 * k1Q + g2 * k2Q,
                       O::fma_(O::splat(g2), k2g, O::mul(O::splat(g1), k1g)));
        }
        {
            const double g1 = C41 * inv_h, g2 = C42 * inv_h, g3 = C43 * inv_h;
            QUAD_SOLVE(k4Q, k4g, g1 * k1Q + g2 * k2Q + g3 * k3Q,
                       O::fma_(O::splat(g3), k3g,
                               O::fma_(O::splat(g2), k2g, O::mul(O::splat(g1), k1g))));
        }
#undef QUAD_SOLVE
        nsteps++;

        const double qnew = q + B1 * k1Q + 0.5 * k2Q + B3 * k3Q + B4 * k4Q;
        const V xnew = O::fma_(O::splat(B4), k4g, O::fma_(O::splat(B3), k3g,
                       O::fma_(O::splat(0.5), k2g, O::fma_(O::splat(B1), k1g, xg))));
        // embedded error estimate; scaled RMS norm over (Q, m, h, n, p), single precision
        float err;
        {
            const double eQd = E1 * k1Q + E2 * k2Q + B4 * k4Q;
            const V eg = O::fma_(O::splat(B4), k4g, O::fma_(O::splat(E2), k2g, O::mul(O::splat(E1), k1g)));
            const float scQ = atol + rtol * fmaxf(fabsf((float)q), fabsf((float)qnew));
            const float eQ = (float)eQd * O::rcpf(scQ);
            err = O::sqrtf_((eQ * eQ + O::errsum(eg, xg, xnew, atol, rtol)) * 0.2f);
#ifdef SONIC_QUAD_STATS
            {
                float comp[5]; comp[0] = eQ * eQ;
                for (int gi = 0; gi < 4; gi++) {
                    const float a_ = (float)O::pick(xg, gi), b_ = (float)O::pick(xnew, gi), e_ = (float)O::pick(eg, gi);
                    const float sc_ = atol + rtol * fmaxf(fabsf(a_), fabsf(b_)); comp[1 + gi] = (e_ / sc_) * (e_ / sc_);
                }
                int im = 0; for (int k = 1; k < 5; k++) if (comp[k] > comp[im]) im = k;
                quad_stats().last_dom = im;
            }
#endif
        }
#endif
        // step-size controller (Hairer & Wanner IV.7): rfac = 0.9 err^(-1/4) clipped to [0.2, 6]
        float rfac = 0.9f * O::rsqf(O::sqrtf_(err));
        rfac = fminf(6.0f, fmaxf(0.2f, rfac));
        rfac = err == err ? rfac : 0.2f;
        // all stages used the home cell's lines: a step that ends too far outside it is retried
        // with a secant-corrected size that ends SONIC_OV_TARGET past the node
        const double over = fmax(H.xlo - qnew, qnew - H.xhi);
        const bool overshoot = over > SONIC_OV_MAX * cellw;
        const float moved = fabsf((float)(qnew - q));
        const float want = moved - (float)over + (float)(SONIC_OV_TARGET * cellw);
        const float sfac = fmaxf(0.1f, fminf(0.9f, want * O::rcpf(moved)));
        const double hnew = h * (double)(overshoot ? sfac : rfac);
        const bool accept = err <= 1.0f && !overshoot;
        const double tnew = last ? grid.t1 : t + h;
#if SONIC_QUAD_METHOD == 4
        if (accept & (irow < grid.n) & (last | (tr <= tnew))) {       // (one combined test: no short-circuit branches)
            // dense output for every grid row inside (t, tnew]
            const double c3Q = d21 * k1Q + d22 * k2Q + d23 * k3Q + d24 * k4Q + d25 * k5Q;
            const double c4Q = d31 * k1Q + d32 * k2Q + d33 * k3Q + d34 * k4Q + d35 * k5Q;
            const V c3g = O::fma_(O::splat(d25), k5g, O::fma_(O::splat(d24), k4g,
                          O::fma_(O::splat(d23), k3g, O::fma_(O::splat(d22), k2g,
                          O::mul(O::splat(d21), k1g)))));
            const V c4g = O::fma_(O::splat(d35), k5g, O::fma_(O::splat(d34), k4g,
                          O::fma_(O::splat(d33), k3g, O::fma_(O::splat(d32), k2g,
                          O::mul(O::splat(d31), k1g)))));
            do {
                const double sg = tr >= tnew ? 1.0 : (tr - t) * inv_h, s1 = 1.0 - sg;
                // y s1 + sg (ynew + s1 (c3 + sg c4)); sg = 1 gives ynew exactly
                const double qi = q * s1 + sg * (qnew + s1 * (c3Q + sg * c4Q));
                const double qr = sg == 1.0 ? qnew : qi;
                const V mid = O::fma_(O::splat(s1), O::fma_(O::splat(sg), c4g, c3g), xnew);
                const V xr = O::fma_(O::splat(sg), mid, O::mul(xg, O::splat(s1)));
                // Vm = lerp of the V table at the row's charge (nbls.py:426-428)
                double Vm = H.vs * (qr - H.xlo) + H.vv;
                if (!(qr >= H.xlo && qr < H.xhi)) Vm = quad_vm_at(G, T, lvl, qr);
                emit(row++, tr, x, qr, xr, Vm);
                irow++;
                tr = quad_linspace_at(grid, irow);
            } while (irow < grid.n && (last || tr <= tnew));
        }
#else
        if (accept & (irow < grid.n) & (last | (tr <= tnew))) {       // (one combined test: no short-circuit branches)
            // rows inside (t, tnew]: cubic Hermite on (y, f0), (ynew, f(ynew)); f(ynew) with the home
            // cell's lines (ynew lies at most SONIC_OV_MAX of a cell outside it)
            qt = qnew;
            xt = xnew;
            QUAD_EVAL();
            const double f1Q = O::allsum(term_);
            const double dQ = qnew - q, aQ = h * f0Q - dQ, bQ = h * f1Q - dQ;
            const V dg = O::sub(xnew, xg);
            const V ag = O::fma_(O::splat(h), f0g, O::sub(O::splat(0.0), dg));
            const V bg = O::fma_(O::splat(h), fg_, O::sub(O::splat(0.0), dg));
            do {
                const bool end = tr >= tnew;
                const double sg = end ? 1.0 : (tr - t) * inv_h, s1 = 1.0 - sg;
                // y + sg (d + s1 (a s1 - b sg)); the row at tnew is ynew itself
                const double qi = q + sg * (dQ + s1 * (aQ * s1 - bQ * sg));
                const double qr = end ? qnew : qi;
                const V cub = O::sub(O::mul(ag, O::splat(s1)), O::mul(bg, O::splat(sg)));
                const V xi = O::fma_(O::splat(sg), O::fma_(O::splat(s1), cub, dg), xg);
                const V xr = O::select(end, xnew, xi);
                // Vm = lerp of the V table at the row's charge (nbls.py:426-428)
                double Vm = H.vs * (qr - H.xlo) + H.vv;
                if (!(qr >= H.xlo && qr < H.xhi)) Vm = quad_vm_at(G, T, lvl, qr);
                emit(row++, tr, x, qr, xr, Vm);
                irow++;
                tr = quad_linspace_at(grid, irow);
            } while (irow < grid.n && (last || tr <= tnew));
        }
#undef QUAD_EVAL
#endif
#ifdef SONIC_QUAD_STATS
        {
            QuadStats &Q = quad_stats();
            if (Q.log && Q.nlog < Q.caplog) {
                double *r = Q.log + 6 * Q.nlog++;
                r[0] = t; r[1] = h; r[2] = err; r[3] = capped; r[4] = accept; r[5] = q;
            }
            Q.steps++;
            if (capped) Q.capped++;
            const int b = err < 1e-4f ? 0 : err < 1e-3f ? 1 : err < 1e-2f ? 2 : err < 0.1f ? 3 : err < 0.3f ? 4 : err < 0.6f ? 5 : err < 1.f ? 6 : 7;
            if (accept) { (capped ? Q.dom_cap : Q.dom_free)[Q.last_dom]++; }
            if (accept) {
                if (capped) { Q.capped_acc++; Q.err_hist_capped[b]++; Q.sum_h_capped += h; }
                else if (last) Q.last_acc++;
                else { Q.errlim_acc++; Q.err_hist_free[b]++; Q.sum_h_free += h; }
                if (!(qnew >= H.xlo && qnew < H.xhi)) Q.cross++;
            } else if (overshoot) Q.rej_over++;
            else Q.rej_err++;
        }
#endif
        // state update (selects: accepted and rejected steps share the path)
        nrej += accept ? 0 : 1;
        ncap += (accept && capped) ? 1 : 0;
        nover += overshoot ? 1 : 0;
        q = accept ? qnew : q;
        xg = O::select(accept, xnew, xg);
        t = accept ? tnew : t;
        h = accept ? hnew : fmin(hnew, h);
        // kink-aware steps end just past a node: the new home cell is the neighbour
        const bool cross = accept && !(q >= H.xlo && q < H.xhi);
        jh += cross ? (q >= H.xhi ? 1 : -1) : 0;
        ncross += cross ? 1 : 0;
        need_cell = need_cell || cross;
        // the rare endings of a step -- the segment is over, the step size underflowed, the step budget is spent --
        // behind ONE test: a wavefront alone on its SIMD pays an issue slot for every exec-mask instruction of a
        // branch it does not take (DESIGN.md 5.0 iii)
        if ((accept & last) | !(h >= o.hmin) | (nsteps >= o.max_steps) | dead) {
            if (accept && last) {
                s++;
                seg_init = true;
                if (s < S.nseg) {
                    lvl = T.level(S.level[s]);
                    jn = -1;
                    need_cell = true;
                }
            }
            if (!(h >= o.hmin)) { dead = true; status |= ST_STEP_UNDERFLOW; }
            if (nsteps >= o.max_steps && !dead && !seg_init) { dead = true; status |= ST_MAX_STEPS; }
            if (dead && !seg_init) {
                // fill the rest of this segment with NaN rows; later segments take the dead path
                q = NAN; xg = O::splat(NAN);
                for (; irow < grid.n; irow++) emit(row++, linspace_at(grid, irow), x, q, xg, NAN);
                s++;
                seg_init = true;
                if (s < S.nseg) { lvl = T.level(S.level[s]); jn = -1; }
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

// One configuration (the CPU harness, tests): a source of one
struct QuadSingleSource {
    const Schedule &S0;
    bool taken = false;
    int status = 0, nsteps = 0, nrej = 0;
    StepCounts counts;
    SONIC_HD bool next(Schedule &S)
    {
        if (taken) return false;
        taken = true;
        S = S0;
        return true;
    }
    SONIC_HD void done(int st, int ns, int nr, const StepCounts &c) { status = st; nsteps = ns; nrej = nr; counts = c; }
};

template <class O, class Tab, class Emit>
SONIC_HD int integrate_config_quad(const CorticalParams &P, const QuadGrid &G, const Tab &T,
                                   const Schedule &S, const double *y0, const SolverOpts &o,
                                   Emit &&emit, int *nsteps_out, int *nrej_out, StepCounts *counts = nullptr)
{
    QuadSingleSource src{S};
    integrate_stream_quad<false, O>(P, G, T, y0, o, emit, src);
    if (nsteps_out) *nsteps_out = src.nsteps;
    if (nrej_out) *nrej_out = src.nrej;
    if (counts) *counts = src.counts;
    return src.status;
}

}  // namespace sonic
