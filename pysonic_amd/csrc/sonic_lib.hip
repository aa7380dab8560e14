// pysonic_amd/csrc/sonic_lib.hip -- libpysonic_amd.so: HIP kernels (gfx950) + the C ABI of
// include/pysonic_amd.h. Written for MI355X only (wave64, FP64 VALU path; no MFMA: the path has
// no dense contraction).
//
// Data layout in HBM (all float64 unless noted):
//   level records  [n_levels][n_cells][2 + 2 NT]   projected lookups, one record per charge cell
//   segments       SoA: t0[], t1[], x[], n[] (i32), level[] (i32), CSR-indexed by seg_off[cfg]
//   traces         [total_rows][NS + 4] row-major: t, stimstate, Qm, states (reference order), Vm
//                  -> rows of one configuration are contiguous = the reference's DataFrame block
//   metrics        [n_cfg][SONIC_NMETRICS];  status [n_cfg] (i32)
//
// Kernel mapping: one stimulus configuration per lane, 64-lane workgroups (one wavefront each) so
// that a finished wavefront frees its SIMD slot immediately; state vector, Rosenbrock stages and
// the cached lookup cell live in VGPRs; model parameters are kernel arguments (SGPRs).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <climits>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/pysonic_amd.h"
#include "lib_common.hpp"
#include "sonic_integrator.hpp"
#include <chrono>
#include "sonic_quad.hpp"
#include "sonic_group.hpp"

using namespace sonic;

// ------------------------------------------------------------------------------------------
// kernel
// ------------------------------------------------------------------------------------------
struct BatchDev {
    const double *recs;
    int n_cells;
    double q0, qmax, inv_dq;
    const double *seg_t0, *seg_t1, *seg_x;
    const int *seg_n, *seg_level;
    const long long *seg_off;   // [n_cfg + 1]
    const long long *row_off;   // [n_cfg + 1]
    const double *y0;           // [NY] reference column order (Qm, states...)
    double *traces;             // may be null (metrics only)
    double *spk_cand;           // [n_cfg][SPK_CAP][5] spike-candidate scratch
    int *spk_stack;             // [n_cfg][SPK_CAP]
    double *metrics;
    int *status;
    long long n_cfg;
    int qpw;                    // quad kernel: configurations (quads) per wavefront, 1..16
    int diag;                   // what goes to the RESERVED metric: 0 placement id, 1 shader MHz
    // quad kernel, LDS-resident tables: slots grouped by amplitude level and padded with -1
    const int *lds_order;       // [n_slots] slot -> configuration or -1: 16 (quad kernel, qpw) or 64 (lane
                                // kernels) slots per wavefront, in order of descending estimated cost
    const int *wave_level;      // [n_slots / qpw] the non-zero level of the wavefront's configurations
    const LaneSpec *lanes;      // group kernel: what each of the 16 lanes of a group is (sonic_group.hpp)
    long long n_slots;
    // work queue (quad / group kernels): the configurations that no wavefront holds at the start, in order of
    // descending estimated cost; a quad (row) of a FULL wavefront that ends its configuration takes the next one
    const int *queue;           // [n_queue]
    int *queue_head;            // next entry of `queue` to hand out (zeroed before every launch)
    int n_queue;
    SolverOpts opts;
};

constexpr int SPK_CAP = 512;   // candidate peaks (height >= 3e-5 C/m2) per configuration

// The batch description as the RARE paths of the work-queue kernels read it (a configuration ends, the next one is
// taken): straight from the kernel-argument segment, behind a barrier the compiler cannot move loads across. Read
// through the by-value parameter instead, the two dozen pointers of BatchDev stay in scalar registers across the
// step loop -- which has none to spare: 46 scalar spills (v_writelane / v_readlane in every step) against 6.
// BatchDev is the first kernel parameter of these kernels: offset 0 of the segment.
typedef __attribute__((address_space(4))) const BatchDev *BatchArgs;
__device__ __forceinline__ BatchArgs batch_args()
{
    BatchArgs p = (BatchArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}


// The configurations a quad (quad kernel) or a row of 16 lanes (group kernel) integrates, one after the other: the
// one the host placed in its slot, then -- in a wavefront all of whose slots hold a configuration -- whatever the
// batch's work queue still holds (integrate_stream_quad / integrate_stream_group call next / done).
template <class O, int NCOL, bool QUEUE>
struct ConfigSource {
    long long cfg;
    bool shadow, refill;
    long long clk0, wall0;
    bool first = true;
    double *rows = nullptr;
    double qmin, qmax, qlast;
    long long nrows;
    long long t_begin = 0;
    SpikeTracker spk;

    template <class Args>
    __device__ __forceinline__ void begin(const Args &A, Schedule &S)
    {
        if (A.diag == 4) t_begin = wall_clock64();
        const long long s0 = A.seg_off[cfg];
        S = Schedule{A.seg_t0 + s0, A.seg_t1 + s0, A.seg_x + s0, A.seg_n + s0, A.seg_level + s0,
                     (int)(A.seg_off[cfg + 1] - s0)};
        rows = A.traces ? A.traces + A.row_off[cfg] * NCOL : nullptr;
        qmin = INFINITY; qmax = -INFINITY; qlast = NAN;
        nrows = 0;
        spk.init(A.spk_cand + cfg * (long long)SPK_CAP * 5, A.spk_stack + cfg * (long long)SPK_CAP, SPK_CAP);
    }
    // the first configuration: from the kernel's own copy of the arguments (the entry of the kernel)
    __device__ __forceinline__ bool start(const BatchDev &B, Schedule &S)
    {
        first = false;
        begin(B, S);
        return true;
    }
    __device__ __forceinline__ bool next(const BatchDev &B, Schedule &S)
    {
        if (first) return start(B, S);
        if constexpr (!QUEUE) return false;
        else {
            if (!refill) return false;
            const auto &A = *batch_args();
            int idx = 0;
            if (O::leader()) idx = atomicAdd(A.queue_head, 1);
            idx = O::from_leader(idx);
            if (idx >= A.n_queue) return false;
            cfg = A.queue[idx];
            begin(A, S);
            return true;
        }
    }
    template <class Args>
    __device__ __forceinline__ void finish(const Args &A, int st, int nsteps, int nrej, const StepCounts &cnt)
    {
        if (shadow) return;
        const SpikeSummary ss = spk.finish();
        if (!O::leader()) return;
        double *m = A.metrics + cfg * SONIC_NMETRICS;
        m[SONIC_M_NSTEPS] = (double)nsteps;
        m[SONIC_M_NREJ] = (double)nrej;
        m[SONIC_M_NROWS] = (double)nrows;
        m[SONIC_M_QMIN] = qmin;
        m[SONIC_M_QMAX] = qmax;
        m[SONIC_M_QLAST] = qlast;
        m[SONIC_M_NSPIKES] = ss.nspikes;
        m[SONIC_M_TFIRST] = ss.t_first;
        m[SONIC_M_TLAST] = ss.t_last;
        m[SONIC_M_SUMINVISI] = ss.sum_inv_isi;
        m[SONIC_M_SPKFLAGS] = (double)ss.flags;
        // diagnostics: where the wavefront ran -- HW_ID (wave, SIMD, CU, SE ids) + XCC_ID << 32
        if (A.diag == 1)   // average shader clock over the life of the wavefront (wall clock = 100 MHz)
            m[SONIC_M_RESERVED] = 100.0 * (double)(clock64() - clk0) / (double)(wall_clock64() - wall0);
        else
            m[SONIC_M_RESERVED] = (double)(((unsigned long long)(__builtin_amdgcn_s_getreg(63508) & 0xf) << 32) |
                                           (unsigned)__builtin_amdgcn_s_getreg(63492));
        m[SONIC_M_NCAPPED] = (double)cnt.capped;
        m[SONIC_M_NREJ_NODE] = (double)cnt.over;
        m[SONIC_M_NCROSS] = (double)cnt.cross;
        m[SONIC_M_SPARE] = 0.0;
        if (A.diag == 4) {      // development: when the configuration ran (100 MHz wall clock) and where
            m[SONIC_M_SPARE] = (double)t_begin;
            m[SONIC_M_RESERVED] = (double)wall_clock64();
            m[SONIC_M_NREJ_NODE] = (double)(blockIdx.x * (64 / O::WIDTH) + threadIdx.x / O::WIDTH);
        }
        A.status[cfg] = st;
    }
};

// what the integrators see: next(S) / done(...) / y0() (the initial conditions, reference column order)
template <class O, int NCOL, bool QUEUE>
struct KernelSource : ConfigSource<O, NCOL, QUEUE> {
    const BatchDev &B;
    __device__ __forceinline__ KernelSource(const BatchDev &B_, long long cfg_, bool shadow_, bool refill_, long long clk0_,
                                            long long wall0_)
        : ConfigSource<O, NCOL, QUEUE>{cfg_, shadow_, refill_, clk0_, wall0_}, B(B_) {}
    __device__ __forceinline__ bool next(Schedule &S) { return ConfigSource<O, NCOL, QUEUE>::next(B, S); }
    __device__ __forceinline__ void done(int st, int nsteps, int nrej, const StepCounts &cnt)
    {
        if constexpr (QUEUE) this->finish(*batch_args(), st, nsteps, nrej, cnt);
        else this->finish(B, st, nsteps, nrej, cnt);
    }
};

template <class M>
__global__ void __launch_bounds__(64)
sonic_integrate_kernel(const BatchDev B, const typename M::Params P)
{
    // one configuration per lane; the host decides how many of the 64 slots of each wavefront
    // carry one (pack_wavefronts). The lanes of an empty slot run a shadow copy of the
    // wavefront's first (costliest) configuration -- same instructions, same data, no stores -- so
    // that the wavefront keeps more than 32 lanes active (see lane_work_index, lib_common.hpp).
    const long long first = (long long)blockIdx.x * 64;
    long long cfg = B.lds_order[first + threadIdx.x];
    const bool shadow = cfg < 0;
    if (shadow && B.diag == 3) return;      // development: PYSONIC_AMD_DIAG=3 runs without shadow lanes
    if (shadow) cfg = B.lds_order[first];
    if (cfg < 0) return;
    constexpr int NY = M::NY;
    constexpr int NCOL = NY + 3;

    const long long s0 = B.seg_off[cfg];
    Schedule S{B.seg_t0 + s0, B.seg_t1 + s0, B.seg_x + s0, B.seg_n + s0, B.seg_level + s0,
               (int)(B.seg_off[cfg + 1] - s0)};
    LevelGrid G{B.recs, B.n_cells, B.q0, B.qmax, B.inv_dq};

    double y0[NY];
#pragma unroll
    for (int i = 0; i < NY; i++) y0[M::out_perm(i)] = B.y0[i];

    double *rows = B.traces ? B.traces + B.row_off[cfg] * NCOL : nullptr;
    double qmin = INFINITY, qmax = -INFINITY, qlast = NAN;
    long long nrows = 0;

    SpikeTracker spk;
    spk.init(B.spk_cand + cfg * (long long)SPK_CAP * 5, B.spk_stack + cfg * (long long)SPK_CAP,
             SPK_CAP);

    auto emit = [&](long row, double t, double x, const double *y, double Vm) {
        if (shadow) return;
        const double q = y[0];
        spk.feed(t, q);
        qmin = fmin(qmin, q);
        qmax = fmax(qmax, q);
        qlast = q;
        nrows++;
        if (rows) {
            double *r = rows + row * NCOL;
            r[0] = t;
            r[1] = x;
#pragma unroll
            for (int i = 0; i < NY; i++) r[2 + i] = y[M::out_perm(i)];
            r[2 + NY] = Vm;
        }
    };

    // home cell: registers, or LDS for the models with many tables (see CellRec)
    constexpr bool CELL_LDS = M::NC > 1;          // TC, STN: the register file is over-subscribed
    __shared__ double cell_lds[CELL_LDS ? 2 * M::NT * 64 : 1];
    CellRec<M::NT, CELL_LDS> home;
    if constexpr (CELL_LDS) {
        home.v.p = cell_lds + threadIdx.x;
        home.s.p = cell_lds + M::NT * 64 + threadIdx.x;
    }
    int nsteps = 0, nrej = 0;
    StepCounts cnt;
    const int st = integrate_config<M>(P, G, S, y0, B.opts, emit, &nsteps, &nrej, home, &cnt);
    if (shadow) return;

    double *m = B.metrics + cfg * SONIC_NMETRICS;
    m[SONIC_M_NSTEPS] = (double)nsteps;
    m[SONIC_M_NREJ] = (double)nrej;
    m[SONIC_M_NROWS] = (double)nrows;
    m[SONIC_M_QMIN] = qmin;
    m[SONIC_M_QMAX] = qmax;
    m[SONIC_M_QLAST] = qlast;
    const SpikeSummary ss = spk.finish();
    m[SONIC_M_NSPIKES] = ss.nspikes;
    m[SONIC_M_TFIRST] = ss.t_first;
    m[SONIC_M_TLAST] = ss.t_last;
    m[SONIC_M_SUMINVISI] = ss.sum_inv_isi;
    m[SONIC_M_SPKFLAGS] = (double)ss.flags;
    m[SONIC_M_RESERVED] = 0.0;
    m[SONIC_M_NCAPPED] = (double)cnt.capped;
    m[SONIC_M_NREJ_NODE] = (double)cnt.over;
    m[SONIC_M_NCROSS] = (double)cnt.cross;
    m[SONIC_M_SPARE] = 0.0;
    B.status[cfg] = st;
}

// Level records of the quad kernel in LDS: the wavefront's copy of level 0 (A = 0) and of the ONE
// non-zero amplitude level its configurations use (the host groups slots by level), 2 x 25 KB for
// RS. Reloading the home cell after a node crossing is then a ~100-clock ds_read instead of an L2
// round trip whose latency grows with the number of wavefronts in flight (measured: 550 clocks
// with one wavefront per CU, 2500 with four).
extern __shared__ double quad_lds[];

struct TabLds {
    typedef int Ref;             // offset of the level inside quad_lds, in doubles
    int level_stride;
    __device__ __forceinline__ Ref level(int id) const { return id == 0 ? 0 : level_stride; }
    __device__ __forceinline__ void load(Ref lvl, int j, QuadCell<QuadOpsDev> &S) const
    {
        const double *r = quad_lds + lvl + j * QUAD_REC;
        const double2 a = *(const double2 *)r, b = *(const double2 *)(r + 2);
        S.xlo = a.x; S.xhi = a.y; S.vv = b.x; S.vs = b.y;
        QuadOpsDev::load_gate_lines(r, S.av, S.as, S.bv, S.bs);
    }
    __device__ __forceinline__ void vline(Ref lvl, int j, double &xlo, double &xhi, double &vv,
                                          double &vs) const
    {
        const double *r = quad_lds + lvl + j * QUAD_REC;
        const double2 a = *(const double2 *)r, b = *(const double2 *)(r + 2);
        xlo = a.x; xhi = a.y; vv = b.x; vs = b.y;
    }
};

// Quad-cooperative variant for the cortical RS / FS neurons (sonic_quad.hpp): one configuration
// per 4 adjacent lanes, up to 16 per wavefront. Every lane of a quad follows the same control flow
// (all control values are replicated), so DPP exchanges always see four active lanes.
//
// A wavefront issues the union of the paths its quads take (emitting rows, crossing a node,
// starting a segment ...), so a batch too small to fill the chip runs faster with FEWER
// configurations per wavefront on MORE SIMDs: the host decides how many of the B.qpw slots of each
// wavefront carry a configuration (quad_packing), the others hold -1.
#ifdef SONIC_QUAD_WAVES
#define SONIC_QUAD_OCCUPANCY __attribute__((amdgpu_waves_per_eu(SONIC_QUAD_WAVES, SONIC_QUAD_WAVES)))
#else
#define SONIC_QUAD_OCCUPANCY
#endif
// QUEUE = false: a launch without a work queue (every configuration already sits in a slot: the 4096-cell map on
// 908 wavefronts) -- the refill path and the state it keeps alive across the step loop are compiled out
template <bool LDS, bool QUEUE>
__global__ void __launch_bounds__(64) SONIC_QUAD_OCCUPANCY
sonic_integrate_quad_kernel(const BatchDev B, const CorticalParams P)
{
    const long long clk0 = clock64(), wall0 = wall_clock64();
    // A wavefront with 32 or fewer active lanes issues every instruction ~1.3x slower than one with
    // more (see lane_work_index, lib_common.hpp), so the quads without a configuration of their own
    // run shadow copies of the others: same instructions, same data, in lockstep with the original
    // -- and no stores.
    const int pos = threadIdx.x >> 2;
    const long long wave = blockIdx.x;
    const int level_stride = B.n_cells * QUAD_REC;
    const long long n_slots = B.n_slots;
    const int *slot_cfg = B.lds_order;
    long long cfg = -1;
    if (LDS) {
        // stage the two levels of this wavefront (all 64 lanes copy, 16 B each per trip)
        const int lvl1 = B.wave_level[wave];
        const double2 *src0 = (const double2 *)B.recs;
        const double2 *src1 = (const double2 *)(B.recs + (size_t)lvl1 * level_stride);
        double2 *dst = (double2 *)quad_lds;
        const int n2 = level_stride >> 1;
        for (int i = threadIdx.x; i < n2; i += 64) {
            dst[i] = src0[i];
            dst[n2 + i] = src1[i];
        }
        __syncthreads();
    }
    bool shadow = false;
    {
        const long long first = wave * B.qpw;
        long long slot = first + pos;
        if (pos < B.qpw && slot < n_slots) cfg = slot_cfg[slot];
        if (cfg < 0) {
            shadow = true;
            slot = first + pos % B.qpw;
            if (slot < n_slots) cfg = slot_cfg[slot];
            if (cfg < 0 && first < n_slots) cfg = slot_cfg[first];
        }
    }
    if (cfg < 0) return;                         // whole quads leave together
    constexpr int NCOL = 8;

    QuadGrid G{B.recs, B.n_cells, B.q0, B.qmax, B.inv_dq};
    // (a wavefront is full when its last slot holds a configuration: the host fills the slots from the front)
    const bool full = QUEUE && B.n_queue > 0 && slot_cfg[wave * B.qpw + B.qpw - 1] >= 0;
    KernelSource<QuadOpsDev, NCOL, QUEUE> src(B, cfg, shadow, full && !shadow, clk0, wall0);

    auto emit = [&](long row, double t, double x, double q, double g, double Vm) {
        if (src.shadow) return;
        src.spk.feed(t, q);
        src.qmin = fmin(src.qmin, q);
        src.qmax = fmax(src.qmax, q);
        src.qlast = q;
        src.nrows++;
        if (src.rows) QuadOpsDev::store_row(src.rows + row * NCOL, t, x, q, g, Vm);
    };

    if (LDS) {
        const TabLds T{level_stride};
        integrate_stream_quad<QUEUE, QuadOpsDev>(P, G, T, B.y0, B.opts, emit, src);
    } else {
        const TabGlobal<QuadOpsDev> T{B.recs, level_stride};
        integrate_stream_quad<QUEUE, QuadOpsDev>(P, G, T, B.y0, B.opts, emit, src);
    }
}

// Group-cooperative variant for LTS / IB / RE / TC / STN (sonic_group.hpp): one configuration per row of
// 16 adjacent lanes, up to B.qpw = 4 per wavefront. As in the quad kernel every lane of a group follows
// the same control flow, the host decides how many of the slots of each wavefront carry a configuration
// and the rows without one run a shadow copy (no stores) of one that has.
// QUEUE as in the quad kernel: a row of a full wavefront that ends its configuration takes the next of the batch's
// work queue (launches with more wavefronts than the chip holds at once).
//
// The queue needs every wavefront of the launch resident from the start (see the host side), i.e. the occupancy of
// the plain build: 2 wavefronts per SIMD for LTS / RE / STN (252 VGPRs), which the queue build is held to (left
// alone it comes out at 255 + 2 accumulation registers and drops to 1). TC (256 + ~100) and the data-driven models
// (254 + 14; held to 256 they spill to scratch) run one wavefront per SIMD either way.
template <class M> struct GroupQueueWaves { static constexpr int value = 1; };
template <> struct GroupQueueWaves<CorticalLTS> { static constexpr int value = 2; };
template <> struct GroupQueueWaves<ThalamicRE> { static constexpr int value = 2; };
template <> struct GroupQueueWaves<OtsukaSTN> { static constexpr int value = 2; };

template <class M, bool QUEUE>
__global__ void __launch_bounds__(64)
__attribute__((amdgpu_waves_per_eu(QUEUE ? GroupQueueWaves<M>::value : 1)))
sonic_integrate_group_kernel(const BatchDev B, const typename M::Params P)
{
    typedef GroupModel<M> GM;
    typedef GroupOpsDev O;
    const int pos = threadIdx.x >> 4;
    const long long first = (long long)blockIdx.x * B.qpw;
    long long cfg = -1;
    bool shadow = false;
    {
        long long slot = first + pos;
        if (pos < B.qpw && slot < B.n_slots) cfg = B.lds_order[slot];
        if (cfg < 0) {
            shadow = true;
            slot = first + pos % B.qpw;
            if (slot < B.n_slots) cfg = B.lds_order[slot];
            if (cfg < 0 && first < B.n_slots) cfg = B.lds_order[first];
        }
    }
    if (cfg < 0) return;                         // whole rows leave together
    constexpr int NCOL = GM::NCOL;

    QuadGrid G{B.recs, B.n_cells, B.q0, B.qmax, B.inv_dq};
    GroupConsts<O> C;
    O::load_consts(B.lanes, C);

    // (a wavefront is full when its last slot holds a configuration: the host fills the slots from the front)
    const bool full = QUEUE && B.n_queue > 0 && B.lds_order[first + B.qpw - 1] >= 0;
    KernelSource<O, NCOL, QUEUE> src(B, cfg, shadow, full && !shadow, 0, 0);

    auto emit = [&](long row, double t, double x, const double *z, double g, double Vm) {
        if (src.shadow) return;
        const double q = z[0];
        src.spk.feed(t, q);
        src.qmin = fmin(src.qmin, q);
        src.qmax = fmax(src.qmax, q);
        src.qlast = q;
        src.nrows++;
        if (src.rows) O::template store_row<GM::NC>(src.rows + row * NCOL, C, t, x, Vm, z, g);
    };

    const GroupTab<O, GM> T{B.recs, B.n_cells * GroupTab<O, GM>::REC};
    integrate_stream_group<QUEUE, O, GM>(P, C, G, T, B.y0, B.opts, emit, src);
}

// ------------------------------------------------------------------------------------------
// host-side objects
// ------------------------------------------------------------------------------------------
struct NeuronInfo {
    int nstates, ntables, nparams;
};

static bool neuron_info(int id, NeuronInfo &ni)
{
    switch (id) {
    case SONIC_NEURON_RS:
    case SONIC_NEURON_FS:
        ni = {4, 9, (int)(sizeof(CorticalParams) / sizeof(double))};
        return true;
    case SONIC_NEURON_LTS:
    case SONIC_NEURON_IB:
        ni = {6, 13, (int)(sizeof(LTSParams) / sizeof(double))};
        return true;
    case SONIC_NEURON_RE:
        ni = {5, 11, (int)(sizeof(REParams) / sizeof(double))};
        return true;
    case SONIC_NEURON_HH:
        ni = {3, 7, (int)(sizeof(GatedParams<3>) / sizeof(double))};
        return true;
    case SONIC_NEURON_SW:
        ni = {2, 5, (int)(sizeof(GatedParams<2>) / sizeof(double))};
        return true;
    case SONIC_NEURON_PAS:
        ni = {1, 3, (int)(sizeof(GatedParams<1>) / sizeof(double))};
        return true;
    case SONIC_NEURON_MRG:
    case SONIC_NEURON_SU:
    case SONIC_NEURON_FH:
        ni = {4, 9, (int)(sizeof(GatedParams<4>) / sizeof(double))};
        return true;
    case SONIC_NEURON_TC:
        ni = {9, 13, (int)(sizeof(TCParams) / sizeof(double))};
        return true;
    case SONIC_NEURON_STN:
        ni = {12, 19, (int)(sizeof(STNParams) / sizeof(double))};
        return true;
    default:
        return false;
    }
}

struct sonic_model {
    int device = 0;
    int neuron_id = 0;
    NeuronInfo ni{};
    std::vector<double> params;
    std::vector<double> tables;   // [n_tab][n_A][n_Q]
    std::vector<double> A_grid, Q_grid;
    int n_A = 0, n_Q = 0, n_tab = 0;
    int n_cu = 0;                 // compute units of the device
    // level cache: amplitude -> level index; device records grow geometrically
    std::map<double, int> level_of;
    std::vector<double> level_amp;
    double *d_recs = nullptr;
    size_t recs_capacity_levels = 0;
    std::mutex level_mutex;       // level bookkeeping: batches of one model may be prepared from
                                  // several host threads
    size_t rec_doubles() const { return (size_t)(n_Q - 1) * (2 + 2 * (size_t)n_tab); }
};


// ------------------------------------------------------------------------------------------
// Device memory of the batches: blocks are kept when a batch is destroyed and handed to the next one that
// fits. A sweep allocates the same few buffers again and again (0.5 GB of traces, 90 MB of spike scratch for
// the 4096-cell map), and hipMalloc / hipFree of such blocks cost 2 ms + 2 ms per batch -- a sixth of the
// kernel. sonic_release_device_memory() gives the idle blocks back.
// ------------------------------------------------------------------------------------------
namespace {
struct DevPool {
    std::mutex mu;
    std::multimap<size_t, void *> idle;          // capacity -> block
    std::unordered_map<void *, size_t> cap_of;   // every block this pool handed out
    size_t idle_bytes = 0;
};
constexpr int POOL_MAX_DEVICES = 16;
constexpr size_t POOL_MAX_IDLE = 24ull << 30;    // per device (288 GB of HBM)
DevPool g_pool[POOL_MAX_DEVICES];

void pool_release(int dev)
{
    DevPool &P = g_pool[dev % POOL_MAX_DEVICES];
    std::vector<void *> blocks;
    {
        std::lock_guard<std::mutex> lock(P.mu);
        for (auto &kv : P.idle) { blocks.push_back(kv.second); P.cap_of.erase(kv.second); }
        P.idle.clear();
        P.idle_bytes = 0;
    }
    for (void *b : blocks) (void)hipFree(b);
}

// the caller has selected the device
int pool_alloc(int dev, void **p, size_t bytes)
{
    bytes = (std::max<size_t>(bytes, 256) + 255) & ~(size_t)255;
    DevPool &P = g_pool[dev % POOL_MAX_DEVICES];
    {
        std::lock_guard<std::mutex> lock(P.mu);
        auto it = P.idle.lower_bound(bytes);
        if (it != P.idle.end() && it->first <= std::max<size_t>(bytes + bytes / 2, (size_t)1 << 20)) {
            *p = it->second;
            P.idle_bytes -= it->first;
            P.idle.erase(it);
            return SONIC_OK;
        }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        pool_release(dev);
        e = hipMalloc(p, bytes);
    }
    if (e != hipSuccess) return set_error(SONIC_EHIP, std::string("hipMalloc: ") + hipGetErrorString(e));
    std::lock_guard<std::mutex> lock(P.mu);
    P.cap_of[*p] = bytes;
    return SONIC_OK;
}

// nothing on the device may still use the block (the batch's streams are synchronised first)
void pool_free(int dev, void *p)
{
    if (!p) return;
    DevPool &P = g_pool[dev % POOL_MAX_DEVICES];
    {
        std::lock_guard<std::mutex> lock(P.mu);
        auto it = P.cap_of.find(p);
        if (it != P.cap_of.end() && P.idle_bytes + it->second <= POOL_MAX_IDLE) {
            P.idle.insert({it->second, p});
            P.idle_bytes += it->second;
            return;
        }
        if (it != P.cap_of.end()) P.cap_of.erase(it);
    }
    (void)hipFree(p);
}

// Streams with their two timing events are kept as well: creating and destroying a stream costs ~2 ms each.
struct StreamSet {
    hipStream_t stream = nullptr;
    hipEvent_t start = nullptr, stop = nullptr;
};
struct StreamPool {
    std::mutex mu;
    std::vector<StreamSet> idle;
};
StreamPool g_streams[POOL_MAX_DEVICES];

hipError_t stream_acquire(int dev, StreamSet &out)
{
    StreamPool &P = g_streams[dev % POOL_MAX_DEVICES];
    {
        std::lock_guard<std::mutex> lock(P.mu);
        if (!P.idle.empty()) { out = P.idle.back(); P.idle.pop_back(); return hipSuccess; }
    }
    hipError_t e = hipStreamCreateWithFlags(&out.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&out.start);
    if (e == hipSuccess) e = hipEventCreate(&out.stop);
    return e;
}

// the stream is idle (synchronised by the caller)
void stream_release(int dev, StreamSet &ss)
{
    if (!ss.stream) return;
    StreamPool &P = g_streams[dev % POOL_MAX_DEVICES];
    {
        std::lock_guard<std::mutex> lock(P.mu);
        if (P.idle.size() < 32 && ss.start && ss.stop) { P.idle.push_back(ss); ss = StreamSet{}; return; }
    }
    if (ss.start) (void)hipEventDestroy(ss.start);
    if (ss.stop) (void)hipEventDestroy(ss.stop);
    (void)hipStreamDestroy(ss.stream);
    ss = StreamSet{};
}
}  // namespace

struct sonic_batch {
    sonic_model *m = nullptr;
    long long n_cfg = 0, n_seg = 0, total_rows = 0;
    int ncol = 0;
    std::vector<long long> row_off;
    sonic_opts_t opts{};
    // device buffers
    double *d_seg_t0 = nullptr, *d_seg_t1 = nullptr, *d_seg_x = nullptr, *d_y0 = nullptr;
    int *d_seg_n = nullptr, *d_seg_level = nullptr, *d_status = nullptr;
    // quad kernel with LDS-resident tables (RS / FS): slots grouped by amplitude level
    int qss_gates = 0;                // quasi-steady-state gates (device gate order)
    int qpw = 0;
    bool lds_tables = false, quad_kernel = false, group_kernel = false;
    LaneSpec *d_lanes = nullptr;      // group kernel: lane roles of the model
    long long n_slots = 0;            // 0: grouping not possible, tables are read from HBM / L2
    int *d_lds_order = nullptr, *d_wave_level = nullptr;
    long long *d_seg_off = nullptr, *d_row_off = nullptr;
    double *d_traces = nullptr, *d_metrics = nullptr, *d_spk_cand = nullptr;
    int *d_spk_stack = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool launched = false;
    char *d_inputs = nullptr;         // ONE block holding the schedule arrays (d_seg_*, d_*_off, d_lds_order ...)
    int *d_queue = nullptr, *d_queue_head = nullptr;   // work queue of the quad / group kernels (BatchDev::queue)
    int n_queue = 0;
    // rows of configuration c: [row_start[c], row_start[c] + n_rows[c]) of the trace block. Queue order by default
    // (row_start = row_off); in a pipelined batch (opts.chunks > 1) the order of the slot list, i.e. of
    // descending estimated cost, so that the rows of a chunk of wavefronts are one contiguous range
    std::vector<long long> row_start, n_rows;
    struct Chunk {
        long long slot0 = 0, n_slots = 0, row0 = 0, n_rows = 0;
        hipStream_t stream = nullptr;
        hipEvent_t start = nullptr, stop = nullptr;
    };
    std::vector<Chunk> chunks;        // empty: one launch on `stream`
    double *host_traces = nullptr;    // pipelined launch: where each chunk's rows are copied as its kernel ends
};

// utils.isWithin (PySONIC/utils.py:321-348) for the amplitude projection (lookups.py:245-247)
static bool is_close_rel(double a, double b, double rel_tol)
{
    // math.isclose(a, b, rel_tol=rel_tol) with abs_tol = 0
    if (a == b) return true;
    if (std::isinf(a) || std::isinf(b)) return false;
    const double diff = std::fabs(b - a);
    return (diff <= std::fabs(rel_tol * b)) || (diff <= std::fabs(rel_tol * a));
}

static int snap_within(double &val, double lo, double hi)
{
    if (val >= lo && val <= hi) return SONIC_OK;
    if (val < lo && is_close_rel(val, lo, 1e-9)) { val = lo; return SONIC_OK; }
    if (val > hi && is_close_rel(val, hi, 1e-9)) { val = hi; return SONIC_OK; }
    char buf[160];
    snprintf(buf, sizeof buf, "A value (%.17g) out of [%.17g, %.17g] interval", val, lo, hi);
    return set_error(SONIC_ERANGE, buf);
}

// Lookup.project('A', value) (lookups.py:230-271) = scipy interp1d(kind='linear') along A:
//   idx = searchsorted(x, v) clipped to [1, n-1]; lo = idx - 1; hi = idx;
//   y = (y_hi - y_lo) / (x_hi - x_lo) * (v - x_lo) + y_lo
// then packed into per-cell records (sonic_integrator.hpp, CellRec).
static int build_level_records(const sonic_model *m, double amp, double *recs)
{
    double v = amp;
    int rc = snap_within(v, m->A_grid.front(), m->A_grid.back());
    if (rc != SONIC_OK) return rc;
    const int nA = m->n_A, nQ = m->n_Q, nT = m->n_tab;
    int idx = (int)(std::lower_bound(m->A_grid.begin(), m->A_grid.end(), v) - m->A_grid.begin());
    idx = std::min(std::max(idx, 1), nA - 1);
    const int lo = idx - 1, hi = idx;
    const double xlo = m->A_grid[lo], xhi = m->A_grid[hi];
    std::vector<double> t1d((size_t)nT * nQ);
    for (int k = 0; k < nT; k++) {
        const double *tab = m->tables.data() + (size_t)k * nA * nQ;
        for (int j = 0; j < nQ; j++) {
            const double ylo = tab[(size_t)lo * nQ + j], yhi = tab[(size_t)hi * nQ + j];
            const double slope = (yhi - ylo) / (xhi - xlo);
            t1d[(size_t)k * nQ + j] = slope * (v - xlo) + ylo;
        }
    }
    const size_t rd = 2 + 2 * (size_t)nT;
    for (int j = 0; j < nQ - 1; j++) {
        double *r = recs + (size_t)j * rd;
        const double qlo = m->Q_grid[j], qhi = m->Q_grid[j + 1];
        r[0] = qlo;
        r[1] = qhi;
        for (int k = 0; k < nT; k++) {
            const double f0 = t1d[(size_t)k * nQ + j], f1 = t1d[(size_t)k * nQ + j + 1];
            r[2 + 2 * k] = f0;
            r[3 + 2 * k] = (f1 - f0) / (qhi - qlo);   // np.interp's slope
        }
    }
    return SONIC_OK;
}

// Make sure every amplitude in `amps` has a level; upload new records. Level 0 is amplitude 0.
static int ensure_levels(sonic_model *m, const std::vector<double> &amps)
{
    std::vector<double> fresh;
    auto want = [&](double a) {
        if (!m->level_of.count(a)) {
            m->level_of[a] = (int)m->level_amp.size();
            m->level_amp.push_back(a);
            fresh.push_back(a);
        }
    };
    if (m->level_amp.empty()) want(0.0);
    for (double a : amps) want(a);
    if (fresh.empty()) return SONIC_OK;

    const size_t rd = m->rec_doubles();
    const size_t n_old = m->level_amp.size() - fresh.size();
    std::vector<double> host((size_t)fresh.size() * rd);
    for (size_t i = 0; i < fresh.size(); i++) {
        int rc = build_level_records(m, fresh[i], host.data() + i * rd);
        if (rc != SONIC_OK) {
            // roll back the bookkeeping of this call
            for (double a : fresh) m->level_of.erase(a);
            m->level_amp.resize(n_old);
            return rc;
        }
    }
    HIP_TRY(hipSetDevice(m->device));
    if (m->level_amp.size() > m->recs_capacity_levels) {
        size_t cap = std::max<size_t>(m->level_amp.size() * 2, 16);
        double *d_new = nullptr;
        HIP_TRY(hipMalloc(&d_new, cap * rd * sizeof(double)));
        if (m->d_recs && n_old)
            HIP_TRY(hipMemcpy(d_new, m->d_recs, n_old * rd * sizeof(double),
                              hipMemcpyDeviceToDevice));
        if (m->d_recs) HIP_TRY(hipFree(m->d_recs));
        m->d_recs = d_new;
        m->recs_capacity_levels = cap;
    }
    HIP_TRY(hipMemcpy(m->d_recs + n_old * rd, host.data(), host.size() * sizeof(double),
                      hipMemcpyHostToDevice));
    return SONIC_OK;
}

// ODESolver.getNSamples (solvers.py:77-87): max(int(np.round((tend - t0) / dt)), 2);
// np.round rounds half to even = nearbyint in the default rounding mode.
static inline long long n_samples(double t0, double tend, double dt)
{
    const long long n = (long long)std::nearbyint((tend - t0) / dt);
    return n > 2 ? n : 2;
}

// Parameter structs are plain arrays of doubles in the order of pneuron.device_params()
template <class M>
static void launch_model(const sonic_model *m, const BatchDev &B, unsigned grid, unsigned block,
                         hipStream_t stream)
{
    typename M::Params P;
    static_assert(sizeof(P) % sizeof(double) == 0, "params must be doubles");
    std::memcpy(&P, m->params.data(), sizeof(P));
    hipLaunchKernelGGL(sonic_integrate_kernel<M>, dim3(grid), dim3(block), 0, stream, B, P);
}

template <class M>
static void launch_group(const sonic_model *m, const BatchDev &B, unsigned grid, unsigned block,
                         hipStream_t stream)
{
    typename M::Params P;
    std::memcpy(&P, m->params.data(), sizeof(P));
    if (B.n_queue > 0)
        hipLaunchKernelGGL((sonic_integrate_group_kernel<M, true>), dim3(grid), dim3(block), 0, stream, B, P);
    else
        hipLaunchKernelGGL((sonic_integrate_group_kernel<M, false>), dim3(grid), dim3(block), 0, stream, B, P);
}

// Wavefronts of a kernel that one SIMD holds at once (its register budget decides): the number of wavefronts of a
// launch that are resident from the start is this x 4 SIMDs x the compute units of the device.
template <class K>
static int waves_per_simd(K kernel)
{
    int blocks = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kernel, 64, 0) != hipSuccess || blocks < 4) {
        (void)hipGetLastError();
        return 1;
    }
    return blocks / 4;
}

static int queue_kernel_waves_per_simd(const sonic_model *m, bool quad_kernel)
{
    static std::mutex mu;
    static std::map<int, int> cache;
    std::lock_guard<std::mutex> lock(mu);
    const int key = m->neuron_id * 2 + (quad_kernel ? 1 : 0);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    int w = 1;
    if (quad_kernel) w = waves_per_simd(sonic_integrate_quad_kernel<false, true>);
    else switch (m->neuron_id) {
        case SONIC_NEURON_LTS:
        case SONIC_NEURON_IB: w = waves_per_simd(sonic_integrate_group_kernel<CorticalLTS, true>); break;
        case SONIC_NEURON_RE: w = waves_per_simd(sonic_integrate_group_kernel<ThalamicRE, true>); break;
        case SONIC_NEURON_TC: w = waves_per_simd(sonic_integrate_group_kernel<ThalamoCortical, true>); break;
        case SONIC_NEURON_STN: w = waves_per_simd(sonic_integrate_group_kernel<OtsukaSTN, true>); break;
        case SONIC_NEURON_HH: w = waves_per_simd(sonic_integrate_group_kernel<GatedModel<3>, true>); break;
        case SONIC_NEURON_SW: w = waves_per_simd(sonic_integrate_group_kernel<GatedModel<2>, true>); break;
        case SONIC_NEURON_PAS: w = waves_per_simd(sonic_integrate_group_kernel<GatedModel<1>, true>); break;
        case SONIC_NEURON_MRG:
        case SONIC_NEURON_SU:
        case SONIC_NEURON_FH: w = waves_per_simd(sonic_integrate_group_kernel<GatedModel<4>, true>); break;
        default: break;
    }
    cache[key] = w;
    return w;
}

// development switches: see dev_switch() in lib_common.hpp
static bool use_quad_kernel() { return dev_switch("PYSONIC_AMD_QUAD", 1) != 0; }

static bool use_group_kernel() { return dev_switch("PYSONIC_AMD_GROUP", 1) != 0; }

// lane roles of a model of the group kernel (sonic_group.hpp); false if the neuron has none
static bool group_lane_specs(const sonic_model *m, std::vector<LaneSpec> &specs)
{
    specs.assign(GRP, lane_none());
    auto fill = [&](auto tag) {
        typedef decltype(tag) M;
        typename M::Params P;
        std::memcpy(&P, m->params.data(), sizeof(P));
        return GroupModel<M>::lanes(P, specs.data());
    };
    switch (m->neuron_id) {
    case SONIC_NEURON_LTS:
    case SONIC_NEURON_IB: return fill(CorticalLTS{});
    case SONIC_NEURON_RE: return fill(ThalamicRE{});
    case SONIC_NEURON_TC: return fill(ThalamoCortical{});
    case SONIC_NEURON_STN: return fill(OtsukaSTN{});
    // data-driven neurons: false if the currents do not fit one quad of lanes each (sonic_group.hpp)
    case SONIC_NEURON_HH: return fill(GatedModel<3>{});
    case SONIC_NEURON_SW: return fill(GatedModel<2>{});
    case SONIC_NEURON_PAS: return fill(GatedModel<1>{});
    case SONIC_NEURON_MRG:
    case SONIC_NEURON_SU:
    case SONIC_NEURON_FH: return fill(GatedModel<4>{});
    }
    return false;
}

// qss_mask (bit k = k-th state in PointNeuron.statesNames() order) -> device gate bits. Only
// voltage-gated states (the diagonal "gate" block of the model) can be quasi-steady-state.
template <class M>
static int qss_gate_bits(int mask, bool &ok)
{
    int bits = 0;
    ok = true;
    for (int k = 0; k < M::NY - 1; k++) {
        if (!(mask & (1 << k))) continue;
        const int slot = M::out_perm(1 + k);       // device index of reference column 1 + k
        if (slot < M::NC) { ok = false; continue; }
        bits |= 1 << (slot - M::NC);
    }
    if (mask >> (M::NY - 1)) ok = false;
    return bits;
}

static int qss_gate_bits_for(int neuron_id, int mask, bool &ok)
{
    switch (neuron_id) {
    case SONIC_NEURON_RS:
    case SONIC_NEURON_FS: return qss_gate_bits<CorticalRSFS>(mask, ok);
    case SONIC_NEURON_LTS:
    case SONIC_NEURON_IB: return qss_gate_bits<CorticalLTS>(mask, ok);
    case SONIC_NEURON_RE: return qss_gate_bits<ThalamicRE>(mask, ok);
    case SONIC_NEURON_TC: return qss_gate_bits<ThalamoCortical>(mask, ok);
    case SONIC_NEURON_STN: return qss_gate_bits<OtsukaSTN>(mask, ok);
    case SONIC_NEURON_HH: return qss_gate_bits<GatedModel<3>>(mask, ok);
    case SONIC_NEURON_SW: return qss_gate_bits<GatedModel<2>>(mask, ok);
    case SONIC_NEURON_PAS: return qss_gate_bits<GatedModel<1>>(mask, ok);
    case SONIC_NEURON_MRG:
    case SONIC_NEURON_SU:
    case SONIC_NEURON_FH: return qss_gate_bits<GatedModel<4>>(mask, ok);
    }
    ok = false;
    return 0;
}

// Development switch: PYSONIC_AMD_LDS=1 stages the level records of the quad kernel in LDS (batches of
// up to three wavefronts per CU). Off by default: once every wavefront keeps more than 32 lanes
// active (shadow quads, see the kernel) the L2-resident records are as fast, and they do not cap
// the number of resident wavefronts (profiles/r01g_lanes_and_packing.txt).
static bool use_lds_tables() { return dev_switch("PYSONIC_AMD_LDS", 0) == 1; }

// Packing of a batch into wavefronts of the quad kernel. A wavefront issues the union of the paths
// its quads take, so the fewer configurations share a wavefront the faster each advances: measured
// cost of one step of the slowest member with q configurations per wavefront (4096-configuration
// map, records in L2, profiles/r01g_lanes_and_packing.txt)
static const int    kPackQ[5] = {16, 8, 4, 2, 1};
static const double kPackC[5] = {1.40, 1.30, 1.18, 1.05, 0.95};    // us per step
// The lane-per-configuration kernels (LTS, RE, TC, STN: 3 - 15 us per step) lose little to divergence
// (a step costs 1.0 / 1.05 / 1.12 / 1.2 / 1.3 / 1.4 with 2 / 4 / 8 / 16 / 32 / 64 configurations per
// wavefront) but a lot to idle SIMDs: 2000 configurations are 32 full wavefronts on 1024 SIMDs. They
// are spread evenly, cost-sorted: q = ceil(n / n_simd) per wavefront (2000 configurations, 2 per
// wavefront instead of 64: LTS 100 -> 64 ms, RE 137 -> 88, TC 205 -> 148, STN 364 -> 293).
static std::vector<int> lane_packing(const sonic_model *m, long long n)
{
    long long q = dev_switch("PYSONIC_AMD_LPW", 0);
    if (q < 1 || q > 64) {
        const long long n_simd = 4LL * (m->n_cu > 0 ? m->n_cu : 256);
        q = std::min<long long>(64, std::max<long long>(1, (n + n_simd - 1) / n_simd));
    }
    std::vector<int> sizes;
    for (long long i = 0; i < n; i += q) sizes.push_back((int)std::min(q, n - i));
    return sizes;
}
// group kernel (16 lanes per configuration, up to 4 per wavefront): relative cost of a step of the slowest member
static const int    kGroupQ[3] = {4, 2, 1};
static const double kGroupC[3] = {1.15, 1.05, 1.0};
static const double kPackMargin = 0.8;
// Small batches are bound by their costliest configuration (a chain of ~10^4 dependent steps), large
// ones by the issue slots of the 4 x n_cu SIMDs. `order` lists the configurations by descending
// estimated cost. For a target time T (in units of cost x us-per-step), every wavefront takes as
// many configurations as keep its leader within T: q = max {q : cost[leader] c(q) <= T} (for q > 1
// within 0.8 T: the estimate is crude, and a shared wavefront that turns out costlier than
// estimated becomes the critical one); T starts
// at the best possible value, cost_max c(1), and grows by 10 % until the SIMD time the wavefronts
// consume fits: sum_w cost[leader_w] c(q_w) <= n_simd T. Returns the number of configurations of
// each wavefront. Measured (RS map and multiples of it, traces written): 1024 .. 16384 configurations
// 13.3 - 13.7 ms, 32768 14.0 ms, 65536 14.9 ms; the costliest configuration alone takes 12.9 ms. PYSONIC_AMD_QPW = q forces q per wavefront throughout (development).
static std::vector<int> pack_wavefronts(const sonic_model *m, const std::vector<int> &order,
                                        const std::vector<double> &cost, int nq, const int *Q,
                                        const double *C, const char *env)
{
    const long long n = (long long)order.size();
    std::vector<int> sizes;
    if (n == 0) return sizes;
    {
        const int v = dev_switch(env, 0);
        if (v >= 1 && v <= Q[0]) {
            for (long long i = 0; i < n; i += v) sizes.push_back((int)std::min<long long>(v, n - i));
            return sizes;
        }
    }
    const double n_simd = 4.0 * (m->n_cu > 0 ? m->n_cu : 256);
    if ((double)n <= n_simd) {
        // a SIMD for every configuration: sharing a wavefront can only cost (the members' steps diverge)
        sizes.assign((size_t)n, 1);
        return sizes;
    }
    const double cmax = std::max(cost[order[0]], 1e-300);
    for (double T = cmax * C[nq - 1];; T *= 1.1) {
        sizes.clear();
        double work = 0.0;
        for (long long i = 0; i < n;) {
            const double lead = std::max(cost[order[i]], 1e-300);
            int k = nq - 1;
            for (int j = 0; j < nq - 1; j++)
                if (lead * C[j] <= kPackMargin * T) { k = j; break; }
            const int q = (int)std::min<long long>(Q[k], n - i);
            sizes.push_back(q);
            work += lead * C[k];
            i += q;
        }
        if (work <= n_simd * T || T > 1e6 * cmax) return sizes;
    }
}

// slot list of a batch: `width` slots per wavefront, -1 where a wavefront carries fewer configurations
static std::vector<int> slot_list(const std::vector<int> &order, const std::vector<int> &sizes,
                                  int width, long long n_cfg, const char *what)
{
    std::vector<int> slots;
    std::map<int, int> hist;
    long long i = 0;
    for (int q : sizes) {
        for (int k = 0; k < width; k++) slots.push_back(k < q ? order[i + k] : -1);
        i += q;
        hist[q]++;
    }
    if (dev_switch("PYSONIC_AMD_DIAG", 0) == 2) {
            std::fprintf(stderr, "pysonic_amd: %s: %lld configurations in %zu wavefronts:", what, n_cfg,
                         sizes.size());
            for (auto &h : hist) std::fprintf(stderr, " %d x %d", h.second, h.first);
            std::fprintf(stderr, "\n");
        }
    return slots;
}

template <class T>
static int upload(T **dptr, const std::vector<T> &h)
{
    const size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
    HIP_TRY(hipMalloc((void **)dptr, bytes));
    if (!h.empty()) HIP_TRY(hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return SONIC_OK;
}

// rows of `ncol` doubles -> rows of `ld` doubles, the extra columns NaN: the table of an effective simulation as
// the reference returns it (columns Z and ng, nbls.py:432-434), assembled at HBM speed so that the copy to the
// host is ONE contiguous transfer and the host neither fills columns nor copies with a stride
__global__ void __launch_bounds__(256) pad_rows_kernel(const double *src, double *dst, long long n_rows, int ncol, int ld)
{
    const long long total = n_rows * ld;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / ld;
        const int c = (int)(i - r * ld);
        dst[i] = c < ncol ? src[r * ncol + c] : NAN;
    }
}

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

int sonic_abi_version(void) { return SONIC_ABI_VERSION; }

const char *sonic_last_error(void) { return last_error_string().c_str(); }

int sonic_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void sonic_default_opts(sonic_opts_t *o)
{
    o->chunks = 0;
    o->rtol = 0.0;      // 0: the kernel's own tolerance (sonic_batch_prepare)
    o->atol = 0.0;      // 0: 1e-8
    o->h0 = 1e-6;
    o->hmin = 1e-30;
    o->max_steps = 20000000;
    o->write_traces = 1;
    o->qss_mask = 0;
    o->idrive = 0.0;
}

int sonic_neuron_nstates(int id)
{
    NeuronInfo ni;
    return neuron_info(id, ni) ? ni.nstates : SONIC_EINVAL;
}
int sonic_neuron_ntables(int id)
{
    NeuronInfo ni;
    return neuron_info(id, ni) ? ni.ntables : SONIC_EINVAL;
}
int sonic_neuron_nparams(int id)
{
    NeuronInfo ni;
    return neuron_info(id, ni) ? ni.nparams : SONIC_EINVAL;
}

int sonic_model_create(int device, int neuron_id, const double *params, int n_params,
                       const double *tables, const double *A_grid, int n_A,
                       const double *Q_grid, int n_Q, int n_tab, sonic_model_t **out)
{
    if (!out || !params || !tables || !A_grid || !Q_grid)
        return set_error(SONIC_EINVAL, "sonic_model_create: null argument");
    NeuronInfo ni;
    if (!neuron_info(neuron_id, ni)) return set_error(SONIC_EINVAL, "unknown neuron id");
    if (n_params != ni.nparams || n_tab != ni.ntables)
        return set_error(SONIC_EINVAL, "parameter / table count does not match the neuron model");
    if (n_A < 2 || n_Q < 2) return set_error(SONIC_EINVAL, "lookup grids need >= 2 points");
    for (int i = 1; i < n_A; i++)
        if (!(A_grid[i] > A_grid[i - 1])) return set_error(SONIC_EINVAL, "A grid not ascending");
    for (int i = 1; i < n_Q; i++)
        if (!(Q_grid[i] > Q_grid[i - 1])) return set_error(SONIC_EINVAL, "Q grid not ascending");
    int ndev = sonic_device_count();
    if (ndev <= 0) return set_error(SONIC_ENODEV, "no HIP device available");
    if (device < 0 || device >= ndev) return set_error(SONIC_EINVAL, "device index out of range");
    sonic_model *m = new sonic_model;
    m->device = device;
    m->neuron_id = neuron_id;
    m->ni = ni;
    m->params.assign(params, params + n_params);
    m->tables.assign(tables, tables + (size_t)n_tab * n_A * n_Q);
    m->A_grid.assign(A_grid, A_grid + n_A);
    m->Q_grid.assign(Q_grid, Q_grid + n_Q);
    m->n_A = n_A;
    m->n_Q = n_Q;
    m->n_tab = n_tab;
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess)
            m->n_cu = ncu;
    }
    *out = m;
    return SONIC_OK;
}

void sonic_model_destroy(sonic_model_t *m)
{
    if (!m) return;
    if (m->d_recs) {
        (void)hipSetDevice(m->device);
        (void)hipFree(m->d_recs);
    }
    delete m;
}

int sonic_count_rows(const double *tstop, const double *dt, const double *ev_t,
                     const long long *ev_off, long long n_cfg, long long *n_rows)
{
    if (!tstop || !dt || !ev_off || !n_rows || n_cfg < 0)
        return set_error(SONIC_EINVAL, "sonic_count_rows: bad argument");
    for (long long c = 0; c < n_cfg; c++) {
        long long rows = 1;
        double tnow = 0.0;
        for (long long e = ev_off[c]; e < ev_off[c + 1]; e++) {
            if (ev_t[e] < tnow) return set_error(SONIC_EINVAL, "events must be sorted by time");
            rows += n_samples(tnow, ev_t[e], dt[c]);
            tnow = ev_t[e];
        }
        if (tnow > tstop[c])
            return set_error(SONIC_EINVAL, "all events must occur before stopping time");
        rows += n_samples(tnow, tstop[c], dt[c]);
        n_rows[c] = rows;
    }
    return SONIC_OK;
}

static void free_batch_buffers(sonic_batch *b)
{
    const int dev = b->m->device;
    (void)hipSetDevice(dev);
    // the blocks and streams go back to their pools: nothing may still be running on them
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    for (auto &c : b->chunks) {
        if (c.stream) (void)hipStreamSynchronize(c.stream);
        StreamSet ss{c.stream, c.start, c.stop};
        stream_release(dev, ss);
    }
    b->chunks.clear();
    void *ptrs[] = {b->d_inputs, b->d_status, b->d_traces, b->d_metrics, b->d_spk_cand, b->d_spk_stack};
    for (void *p : ptrs) pool_free(dev, p);
    StreamSet ss{b->stream, b->ev_start, b->ev_stop};
    stream_release(dev, ss);
    b->stream = nullptr; b->ev_start = b->ev_stop = nullptr;
}

void sonic_batch_destroy(sonic_batch_t *b)
{
    if (!b) return;
    free_batch_buffers(b);
    delete b;
}

int sonic_batch_prepare(sonic_model_t *m, const double *A, const double *tstop, const double *dt,
                        const double *ev_t, const double *ev_x, const long long *ev_off,
                        long long n_cfg, const double *y0, const sonic_opts_t *opts,
                        sonic_batch_t **out)
{
    if (!m || !A || !tstop || !dt || !ev_off || !y0 || !out || n_cfg < 0)
        return set_error(SONIC_EINVAL, "sonic_batch_prepare: bad argument");
    if (n_cfg > 0 && ev_off[n_cfg] > 0 && (!ev_t || !ev_x))
        return set_error(SONIC_EINVAL, "sonic_batch_prepare: null event arrays");
    sonic_opts_t o;
    if (opts) o = *opts; else sonic_default_opts(&o);
    bool qss_ok = true;
    const int qss_gates = o.qss_mask ? qss_gate_bits_for(m ? m->neuron_id : -1, o.qss_mask, qss_ok) : 0;
    if (!qss_ok)
        return set_error(SONIC_EINVAL, "qss_mask: only voltage-gated states can be quasi-steady-state");
    if (!(o.rtol >= 0) || !(o.atol >= 0) || !(o.h0 > 0) || !(o.hmin > 0) || o.max_steps <= 0)
        return set_error(SONIC_EINVAL, "sonic_batch_prepare: invalid solver options");

    // phase timings on stderr under PYSONIC_AMD_DIAG=2
    const bool diag = dev_switch("PYSONIC_AMD_DIAG", 0) == 2;
    auto clk = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!diag) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "pysonic_amd: prepare: %-28s %8.2f ms\n", what,
                     std::chrono::duration<double, std::milli>(now - clk).count());
        clk = now;
    };
    // ---- segment schedule (EventDrivenSolver.solve, solvers.py:445-480) ----
    std::vector<double> seg_t0, seg_t1, seg_x, seg_amp;
    std::vector<int> seg_n;
    std::vector<long long> seg_off(n_cfg + 1, 0), row_off(n_cfg + 1, 0);
    std::vector<double> cost(n_cfg, 0.0);
    // the arrays grow by one entry per segment: sized once (2 events per pulse + the closing segment)
    {
        const size_t nseg_guess = (size_t)(n_cfg > 0 ? ev_off[n_cfg] : 0) + (size_t)n_cfg;
        seg_t0.reserve(nseg_guess); seg_t1.reserve(nseg_guess); seg_x.reserve(nseg_guess);
        seg_amp.reserve(nseg_guess); seg_n.reserve(nseg_guess);
    }
    for (long long c = 0; c < n_cfg; c++) {
        if (!(dt[c] > 0)) return set_error(SONIC_EINVAL, "time step must be strictly positive");
        double tnow = 0.0, xcur = 0.0;
        long long rows = 1;
        double t_on = 0.0;
        bool too_long = false;
        auto push = [&](double te) {
            const long long n = n_samples(tnow, te, dt[c]);
            if (n > INT_MAX) too_long = true;
            seg_t0.push_back(tnow);
            seg_t1.push_back(te);
            seg_x.push_back(xcur);
            seg_amp.push_back(A[c] * xcur);   // drive.xvar * x (nbls.py:415)
            seg_n.push_back((int)n);
            rows += n;
            // stimulated time weighted by a saturating function of the pressure amplitude
            if (xcur != 0.0) t_on += (te - tnow) * (0.1 + std::fabs(A[c] * xcur) / (std::fabs(A[c] * xcur) + 40e3));
        };
        for (long long e = ev_off[c]; e < ev_off[c + 1]; e++) {
            if (ev_t[e] < tnow) return set_error(SONIC_EINVAL, "events must be sorted by time");
            if (ev_x[e] < 0.0)
                return set_error(SONIC_EINVAL,
                                 "Invalid time protocol: contains negative modulators");
            push(ev_t[e]);
            tnow = ev_t[e];
            xcur = ev_x[e];
        }
        if (tnow > tstop[c])
            return set_error(SONIC_EINVAL, "all events must occur before stopping time");
        push(tstop[c]);
        if (too_long)
            return set_error(SONIC_ERANGE, "a segment has more than 2^31 - 1 output rows");
        seg_off[c + 1] = (long long)seg_t0.size();
        row_off[c + 1] = row_off[c] + rows;
        // crude cost model (ordering only): the number of steps follows the spiking activity, which
        // grows with the stimulated time and saturates with the amplitude (4096-configuration map,
        // RS, DC = 1: 4 000 steps at 50 kPa, 10 000 at 80 kPa, 14 000 at 600 kPa; linear in DC), plus
        // ~60 steps per segment for the restart of the step size (PRF 1 kHz: 200 segments)
        cost[c] = t_on + 0.02 * tstop[c] + 4.4e-4 * (double)(seg_t0.size() - seg_off[c]);
    }
    // development: the costs from a file of n_cfg doubles (the step counts of an earlier run of the same batch: what
    // ordering, packing and the work queue would do with perfect knowledge -- tools/sat_probe.py: the 65 536-cell
    // sweep 20.8 -> 17.6 ms)
    if (const char *cf = std::getenv("PYSONIC_AMD_COST_FILE")) {
        if (FILE *fh = std::fopen(cf, "rb")) {
            std::vector<double> file_cost((size_t)n_cfg);
            if (std::fread(file_cost.data(), sizeof(double), (size_t)n_cfg, fh) == (size_t)n_cfg) cost = file_cost;
            std::fclose(fh);
        }
    }

    lap("segment schedule");
    // ---- levels ----
    std::vector<double> amps(seg_amp);
    std::sort(amps.begin(), amps.end());
    amps.erase(std::unique(amps.begin(), amps.end()), amps.end());
    std::vector<int> seg_level(seg_amp.size());
    int rc;
    {
        std::lock_guard<std::mutex> lock(m->level_mutex);
        rc = ensure_levels(m, amps);
        if (rc != SONIC_OK) return rc;
        // consecutive segments alternate between two amplitudes (0 and A): a two-entry memo in front
        // of the map lookup (10 000 configurations at PRF 1 kHz are 2e6 segments)
        double memo_a[2] = {NAN, NAN};
        int memo_l[2] = {0, 0};
        for (size_t i = 0; i < seg_amp.size(); i++) {
            const double a = seg_amp[i];
            if (a == memo_a[0]) { seg_level[i] = memo_l[0]; continue; }
            if (a == memo_a[1]) { seg_level[i] = memo_l[1]; continue; }
            const int l = m->level_of[a];
            memo_a[1] = memo_a[0]; memo_l[1] = memo_l[0];
            memo_a[0] = a; memo_l[0] = l;
            seg_level[i] = l;
        }
    }

    lap("level records");
    // ---- lane order: descending estimated cost, so a wavefront holds configs of similar cost
    std::vector<int> order(n_cfg);
    for (long long c = 0; c < n_cfg; c++) order[c] = (int)c;
    std::stable_sort(order.begin(), order.end(),
                     [&](int a, int b) { return cost[a] > cost[b]; });

    // ---- quad kernel (RS / FS): slots grouped by amplitude level for LDS-resident tables ----
    // Possible when every configuration uses level 0 plus at most ONE other level (true for the
    // reference's protocols: the modulation factor is 0 or 1) and two levels fit the 64 KB a
    // workgroup may claim. Groups are ordered by their costliest member and padded to whole
    // wavefronts with -1; more than 25 % padding falls back to the HBM / L2 path.
    const bool quad_neuron = m->neuron_id == SONIC_NEURON_RS || m->neuron_id == SONIC_NEURON_FS;
    // (an injected current is folded into the leak term of the quad kernel: needs gLeak != 0)
    const bool quad_kernel = quad_neuron && use_quad_kernel() && qss_gates == 0 &&
                             !(o.idrive != 0.0 && m->params[5] == 0.0);
    int qpw = 0;
    // LDS variant (development switch): two levels of records take 50 KB of the CU's 160 KB of
    // LDS, i.e. three wavefronts per CU
    int qpw_lds = 0;
    if (quad_neuron) {
        const long long max_waves = 3LL * (m->n_cu > 0 ? m->n_cu : 256);
        for (int q = 1; q <= 16; q <<= 1)
            if ((n_cfg + q - 1) / q <= max_waves) { qpw_lds = q; break; }
        {
            const int v = dev_switch("PYSONIC_AMD_QPW", 0);
            if (v >= 1 && v <= 16) qpw_lds = v;
        }
    }
    std::vector<int> lds_order, wave_level;
    if (quad_kernel && qpw_lds > 0 && use_lds_tables() &&
        2 * (size_t)(m->n_Q - 1) * (2 + 2 * (size_t)m->n_tab) * sizeof(double) <= 64 * 1024) {
        std::vector<int> cfg_level(n_cfg, 0);
        bool ok = true;
        for (long long c = 0; c < n_cfg && ok; c++)
            for (long long k = seg_off[c]; k < seg_off[c + 1]; k++) {
                const int l = seg_level[k];
                if (l == 0 || l == cfg_level[c]) continue;
                if (cfg_level[c] != 0) { ok = false; break; }
                cfg_level[c] = l;
            }
        if (ok) {
            std::map<int, std::vector<int>> groups;          // level -> configurations, cost-sorted
            for (int c : order) groups[cfg_level[c]].push_back(c);
            std::vector<std::pair<double, int>> rank;          // (max cost, level)
            for (auto &g : groups) rank.push_back({cost[g.second.front()], g.first});
            std::sort(rank.begin(), rank.end(),
                      [](const std::pair<double, int> &a, const std::pair<double, int> &b) {
                          return a.first > b.first || (a.first == b.first && a.second < b.second);
                      });
            for (auto &r : rank) {
                const std::vector<int> &g = groups[r.second];
                for (size_t i = 0; i < g.size(); i += qpw_lds) {
                    wave_level.push_back(r.second);
                    for (int k = 0; k < qpw_lds; k++)
                        lds_order.push_back(i + k < g.size() ? g[i + k] : -1);
                }
            }
            if (lds_order.size() > (size_t)(1.25 * n_cfg) + qpw_lds) {
                lds_order.clear();
                wave_level.clear();
            } else {
                qpw = qpw_lds;
            }
        }
    }

    // Tolerances left at 0 are the kernel's own. The quad kernel (ROS4, home cell with SONIC_OV_TARGET = 0.5 %) runs
    // at rtol 4e-6: with the second-order node predictor that is more accurate than the round-2 kernel was at 1e-6
    // (costliest golden, RS 600 kPa CW: 0.7e-8 against 1.7e-8 C/m2 RMS from the reference's converged run) in 26 %
    // fewer steps (9 956 against 13 471); its error is set by the slivers past the nodes as much as by the
    // tolerance. atol stays 1e-8: it is what controls the charge (|Qm| ~ 1e-4 C/m2) and the gates that sit near
    // zero (m at rest: 4.5e-4). The RODAS4 kernels keep 1e-6 / 1e-8.
    if (o.rtol == 0) o.rtol = quad_kernel ? 4e-6 : 1e-6;
    if (o.atol == 0) o.atol = 1e-8;

    // default: records in HBM / L2; wavefronts of 1 .. 16 configurations in cost order, 16 slots
    // each (the free quads of a wavefront run shadow copies of its first one, see the kernel)
    if (quad_kernel && lds_order.empty()) {
        qpw = 16;
        lds_order = slot_list(order, pack_wavefronts(m, order, cost, 5, kPackQ, kPackC, "PYSONIC_AMD_QPW"),
                              16, n_cfg, "quad kernel");
    }
    // group kernel (LTS / IB / RE / TC / STN without quasi-steady-state gates): 4 slots of 16 lanes per
    // wavefront, 1 .. 4 of them with a configuration, in cost order
    std::vector<LaneSpec> lane_specs;
    const bool group_kernel = !quad_neuron && use_group_kernel() && qss_gates == 0 && group_lane_specs(m, lane_specs);
    if (group_kernel) {
        qpw = 4;
        lds_order = slot_list(order, pack_wavefronts(m, order, cost, 3, kGroupQ, kGroupC, "PYSONIC_AMD_GPW"),
                              4, n_cfg, "group kernel");
    }
    // lane-per-configuration kernels: 64 slots per wavefront
    if (!quad_kernel && !group_kernel)
        lds_order = slot_list(order, lane_packing(m, n_cfg), 64, n_cfg, "lane kernel");

    // ---- work queue (quad and group kernels, records in L2) ----
    // A batch with more wavefronts than the chip holds at once keeps the first of them -- the costliest
    // configurations, packed as above -- and queues the configurations of the others: a quad (row) of a full
    // wavefront that ends its configuration takes the next of the queue, so the wavefronts stay full to the end
    // instead of waiting, masked, for their slowest member (a launch of 65 536 RS configurations spent half of its
    // wavefront-steps that way: mean 2 700 steps per configuration, 5 400 per wavefront). Every wavefront of the
    // launch must be resident from the start -- one that had to wait for a slot would start only when the queue is
    // empty, since the resident ones keep refilling -- so their number is the occupancy of the kernel (quad kernel:
    // 2 per SIMD at ~200 VGPRs; group kernel 2, TC and the data-driven models 1) x 4 SIMDs x the compute units.
    // Measured (profiles/r03j_sat_probe.txt): 65 536 RS configurations 24.9 ms without the queue, 21.0 ms with it;
    // 16 384 LTS configurations 25.3 -> 23.7 ms. A quad kernel squeezed to three wavefronts per SIMD spills and loses
    // (31 ms), raising the priority of the first wavefronts gains nothing (21.0 ms), and neither does giving the
    // costliest wavefronts a SIMD of their own -- a second kernel declared with 328 registers, so that no wavefront of
    // this one fits beside it: 19.3 - 22.2 ms even when told the true step counts (profiles/r03u_sat_probe.txt). Such a
    // launch is bound by issue throughput (a wavefront of 16 quads issues the union of their paths -- row output and
    // cell reloads in nearly every iteration), not by its costliest configuration; what does help is the ORDER: with
    // the measured step counts as the cost estimate 17.6 ms (PYSONIC_AMD_COST_FILE, development).
    // PYSONIC_AMD_WPS=n overrides the wavefronts per SIMD, 0 turns the queue off.
    std::vector<int> queue;
    if ((quad_kernel || group_kernel) && wave_level.empty() && o.chunks <= 1) {
        const long long n_waves = (long long)lds_order.size() / qpw;
        const long long n_simd = 4 * (long long)(m->n_cu > 0 ? m->n_cu : 256);
        long long wps = dev_switch("PYSONIC_AMD_WPS", -1);
        if (wps < 0) wps = n_waves > n_simd ? queue_kernel_waves_per_simd(m, quad_kernel) : 1;
        const long long w_max = wps * n_simd;
        if (wps > 0 && n_waves > w_max) {
            for (size_t i = (size_t)(w_max * qpw); i < lds_order.size(); i++)
                if (lds_order[i] >= 0) queue.push_back(lds_order[i]);
            lds_order.resize((size_t)(w_max * qpw));
        }
    }

    lap("ordering and packing");
    sonic_batch *b = new sonic_batch;
    b->m = m;
    b->n_cfg = n_cfg;
    b->qpw = qpw;
    b->lds_tables = !wave_level.empty();
    b->quad_kernel = quad_kernel;
    b->group_kernel = group_kernel;
    b->qss_gates = qss_gates;
    b->n_slots = (long long)lds_order.size();
    b->n_queue = (int)queue.size();
    b->n_seg = (long long)seg_t0.size();
    b->total_rows = row_off[n_cfg];
    b->ncol = m->ni.nstates + 4;
    b->row_off = row_off;
    b->opts = o;
    *out = nullptr;

    hipError_t e = hipSetDevice(m->device);
    if (e != hipSuccess) { delete b; return set_error(SONIC_EHIP, hipGetErrorString(e)); }
    std::vector<double> y0v(y0, y0 + 1 + m->ni.nstates);

    // ---- row layout and chunks ----
    // Default: rows in queue order, one launch. Pipelined (opts.chunks > 1, traces written): the rows follow
    // the slot list (descending estimated cost) and the wavefronts are cut into `chunks` launches of about
    // equal numbers of rows, each on a stream of its own, so that the rows of the cheap configurations --
    // whose kernels end first -- travel to the host while the costly ones still integrate.
    b->n_rows.resize(n_cfg);
    for (long long c = 0; c < n_cfg; c++) b->n_rows[c] = row_off[c + 1] - row_off[c];
    b->row_start.assign(row_off.begin(), row_off.begin() + n_cfg);
    const int width = quad_kernel ? qpw : (group_kernel ? 4 : 64);
    int n_chunks = (o.chunks > 1 && o.write_traces && wave_level.empty() && n_cfg > 0) ? o.chunks : 0;
    // (a process has 4 hardware queues by default: more streams than that take turns, and one is left to the copies)
    if (n_chunks > 3) n_chunks = 3;
    if (n_chunks) {
        long long run = 0;
        for (int c : lds_order)
            if (c >= 0) { b->row_start[c] = run; run += b->n_rows[c]; }
        const long long n_waves = (long long)lds_order.size() / width;
        long long w0 = 0, rows_done = 0, rows_chunk = 0;
        for (long long w = 0; w < n_waves; w++) {
            for (int k = 0; k < width; k++) {
                const int c = lds_order[w * width + k];
                if (c >= 0) rows_chunk += b->n_rows[c];
            }
            const int k_now = (int)b->chunks.size();
            const bool last_wave = w == n_waves - 1;
            if (last_wave || (k_now < n_chunks - 1 &&
                              (rows_done + rows_chunk) * n_chunks >= (long long)(k_now + 1) * b->total_rows)) {
                sonic_batch::Chunk ch;
                ch.slot0 = w0 * width; ch.n_slots = (w + 1 - w0) * width;
                ch.row0 = rows_done; ch.n_rows = rows_chunk;
                b->chunks.push_back(ch);
                rows_done += rows_chunk; rows_chunk = 0; w0 = w + 1;
            }
        }
        if (b->chunks.size() < 2) b->chunks.clear();
        if (b->chunks.empty()) b->row_start.assign(row_off.begin(), row_off.begin() + n_cfg);
    }
    std::vector<long long> row_start_dev(b->row_start);
    row_start_dev.push_back(b->total_rows);

    // ---- ONE block for the schedule arrays, ONE transfer ----
    {
        size_t off = 0;
        auto place = [&](size_t bytes) { const size_t at = off; off = (off + bytes + 255) & ~(size_t)255; return at; };
        const size_t o_t0 = place(seg_t0.size() * 8), o_t1 = place(seg_t1.size() * 8), o_x = place(seg_x.size() * 8),
                     o_n = place(seg_n.size() * 4), o_lv = place(seg_level.size() * 4),
                     o_so = place(seg_off.size() * 8), o_ro = place(row_start_dev.size() * 8),
                     o_ord = place(lds_order.size() * 4), o_wl = place(wave_level.size() * 4),
                     o_ln = place(group_kernel ? lane_specs.size() * sizeof(LaneSpec) : 0), o_y0 = place(y0v.size() * 8),
                     o_q = place(queue.size() * 4), o_qh = place(4);
        std::vector<char> host(std::max<size_t>(off, 256));
        auto put = [&](size_t at, const void *src, size_t bytes) { if (bytes) std::memcpy(host.data() + at, src, bytes); };
        put(o_t0, seg_t0.data(), seg_t0.size() * 8); put(o_t1, seg_t1.data(), seg_t1.size() * 8);
        put(o_x, seg_x.data(), seg_x.size() * 8); put(o_n, seg_n.data(), seg_n.size() * 4);
        put(o_lv, seg_level.data(), seg_level.size() * 4); put(o_so, seg_off.data(), seg_off.size() * 8);
        put(o_ro, row_start_dev.data(), row_start_dev.size() * 8); put(o_ord, lds_order.data(), lds_order.size() * 4);
        put(o_wl, wave_level.data(), wave_level.size() * 4);
        if (group_kernel) put(o_ln, lane_specs.data(), lane_specs.size() * sizeof(LaneSpec));
        put(o_y0, y0v.data(), y0v.size() * 8);
        put(o_q, queue.data(), queue.size() * 4);
        { const int zero = 0; put(o_qh, &zero, 4); }
        rc = pool_alloc(m->device, (void **)&b->d_inputs, host.size());
        if (rc == SONIC_OK) {
            e = hipMemcpy(b->d_inputs, host.data(), host.size(), hipMemcpyHostToDevice);
            if (e != hipSuccess) rc = set_error(SONIC_EHIP, std::string("hipMemcpy: ") + hipGetErrorString(e));
        }
        if (rc == SONIC_OK) {
            char *d = b->d_inputs;
            b->d_seg_t0 = (double *)(d + o_t0); b->d_seg_t1 = (double *)(d + o_t1); b->d_seg_x = (double *)(d + o_x);
            b->d_seg_n = (int *)(d + o_n); b->d_seg_level = (int *)(d + o_lv);
            b->d_seg_off = (long long *)(d + o_so); b->d_row_off = (long long *)(d + o_ro);
            b->d_lds_order = (int *)(d + o_ord);
            b->d_wave_level = b->lds_tables ? (int *)(d + o_wl) : nullptr;
            b->d_lanes = group_kernel ? (LaneSpec *)(d + o_ln) : nullptr;
            b->d_y0 = (double *)(d + o_y0);
            b->d_queue = (int *)(d + o_q);
            b->d_queue_head = (int *)(d + o_qh);
        }
    }
    lap("uploads");
    auto dmalloc = [&](void **p, size_t bytes) {
        if (rc == SONIC_OK) rc = pool_alloc(m->device, p, bytes);
    };
    if (rc == SONIC_OK && o.write_traces)
        dmalloc((void **)&b->d_traces, (size_t)b->total_rows * b->ncol * sizeof(double));
    if (rc == SONIC_OK) dmalloc((void **)&b->d_metrics, (size_t)n_cfg * SONIC_NMETRICS * sizeof(double));
    if (rc == SONIC_OK) dmalloc((void **)&b->d_status, (size_t)n_cfg * sizeof(int));
    if (rc == SONIC_OK)
        dmalloc((void **)&b->d_spk_cand, (size_t)n_cfg * SPK_CAP * 5 * sizeof(double));
    if (rc == SONIC_OK) dmalloc((void **)&b->d_spk_stack, (size_t)n_cfg * SPK_CAP * sizeof(int));
    if (rc == SONIC_OK) {
        StreamSet ss;
        hipError_t ee = stream_acquire(m->device, ss);
        b->stream = ss.stream; b->ev_start = ss.start; b->ev_stop = ss.stop;
        for (auto &ch : b->chunks) {
            if (ee != hipSuccess) break;
            StreamSet cs;
            ee = stream_acquire(m->device, cs);
            ch.stream = cs.stream; ch.start = cs.start; ch.stop = cs.stop;
        }
        if (ee != hipSuccess) rc = set_error(SONIC_EHIP, hipGetErrorString(ee));
    }
    lap("device allocations, stream");
    if (rc != SONIC_OK) {
        sonic_batch_destroy(b);
        return rc;
    }
    *out = b;
    return SONIC_OK;
}

long long sonic_batch_total_rows(const sonic_batch_t *b) { return b ? b->total_rows : -1; }

int sonic_batch_row_offsets(const sonic_batch_t *b, long long *row_off)
{
    if (!b || !row_off) return set_error(SONIC_EINVAL, "sonic_batch_row_offsets: null argument");
    if (!b->chunks.empty())
        return set_error(SONIC_EINVAL, "sonic_batch_row_offsets: pipelined batch, rows are not in queue order: "
                                       "use sonic_batch_row_blocks");
    std::memcpy(row_off, b->row_off.data(), b->row_off.size() * sizeof(long long));
    return SONIC_OK;
}

// kernel launch for the wavefronts of slots [slot0, slot0 + n_slots) on `stream`
static int launch_slots(sonic_batch_t *b, BatchDev B, long long slot0, long long n_slots, hipStream_t stream)
{
    sonic_model *m = b->m;
    B.lds_order = b->d_lds_order + slot0;
    B.n_slots = n_slots;
    const unsigned block = 64;
    const unsigned grid = (unsigned)(n_slots / block);   // lane kernels: 64 slots per wavefront
    if (b->group_kernel) {
        B.qpw = b->qpw;
        const unsigned nwaves = (unsigned)(n_slots / B.qpw);
        switch (m->neuron_id) {
        case SONIC_NEURON_LTS:
        case SONIC_NEURON_IB: launch_group<CorticalLTS>(m, B, nwaves, block, stream); break;
        case SONIC_NEURON_RE: launch_group<ThalamicRE>(m, B, nwaves, block, stream); break;
        case SONIC_NEURON_TC: launch_group<ThalamoCortical>(m, B, nwaves, block, stream); break;
        case SONIC_NEURON_STN: launch_group<OtsukaSTN>(m, B, nwaves, block, stream); break;
        case SONIC_NEURON_HH: launch_group<GatedModel<3>>(m, B, nwaves, block, stream); break;
        case SONIC_NEURON_SW: launch_group<GatedModel<2>>(m, B, nwaves, block, stream); break;
        case SONIC_NEURON_PAS: launch_group<GatedModel<1>>(m, B, nwaves, block, stream); break;
        case SONIC_NEURON_MRG:
        case SONIC_NEURON_SU:
        case SONIC_NEURON_FH: launch_group<GatedModel<4>>(m, B, nwaves, block, stream); break;
        default: return set_error(SONIC_EINVAL, "group kernel: neuron without lane roles");
        }
    } else
    switch (m->neuron_id) {
    case SONIC_NEURON_RS:
    case SONIC_NEURON_FS:
        if (b->quad_kernel) {
            CorticalParams P;
            std::memcpy(&P, m->params.data(), sizeof(P));
            B.qpw = b->qpw;
            const size_t lds_bytes = 2 * (size_t)B.n_cells * QUAD_REC * sizeof(double);
            const unsigned nwaves = (unsigned)(n_slots / B.qpw);
            if (b->lds_tables)
                hipLaunchKernelGGL((sonic_integrate_quad_kernel<true, false>), dim3(nwaves), dim3(block),
                                   lds_bytes, stream, B, P);
            else if (B.n_queue > 0 || dev_switch("PYSONIC_AMD_STREAM", 0) == 1)
                hipLaunchKernelGGL((sonic_integrate_quad_kernel<false, true>), dim3(nwaves), dim3(block),
                                   0, stream, B, P);
            else
                hipLaunchKernelGGL((sonic_integrate_quad_kernel<false, false>), dim3(nwaves), dim3(block),
                                   0, stream, B, P);
        } else {
            launch_model<CorticalRSFS>(m, B, grid, block, stream);
        }
        break;
    case SONIC_NEURON_LTS:
    case SONIC_NEURON_IB:
        launch_model<CorticalLTS>(m, B, grid, block, stream);
        break;
    case SONIC_NEURON_RE:
        launch_model<ThalamicRE>(m, B, grid, block, stream);
        break;
    case SONIC_NEURON_HH:
        launch_model<GatedModel<3>>(m, B, grid, block, stream);
        break;
    case SONIC_NEURON_SW:
        launch_model<GatedModel<2>>(m, B, grid, block, stream);
        break;
    case SONIC_NEURON_PAS:
        launch_model<GatedModel<1>>(m, B, grid, block, stream);
        break;
    case SONIC_NEURON_MRG:
    case SONIC_NEURON_SU:
    case SONIC_NEURON_FH:
        launch_model<GatedModel<4>>(m, B, grid, block, stream);
        break;
    case SONIC_NEURON_TC:
        launch_model<ThalamoCortical>(m, B, grid, block, stream);
        break;
    case SONIC_NEURON_STN:
        launch_model<OtsukaSTN>(m, B, grid, block, stream);
        break;
    default:
        return set_error(SONIC_EINVAL, "neuron model not implemented on device");
    }
    HIP_TRY(hipGetLastError());
    return SONIC_OK;
}

int sonic_batch_launch(sonic_batch_t *b)
{
    if (!b) return set_error(SONIC_EINVAL, "sonic_batch_launch: null batch");
    sonic_model *m = b->m;
    HIP_TRY(hipSetDevice(m->device));
    BatchDev B{};
    B.recs = m->d_recs;
    B.n_cells = m->n_Q - 1;
    B.q0 = m->Q_grid.front();
    B.qmax = m->Q_grid.back();
    B.inv_dq = (double)(m->n_Q - 1) / (m->Q_grid.back() - m->Q_grid.front());
    B.seg_t0 = b->d_seg_t0;
    B.seg_t1 = b->d_seg_t1;
    B.seg_x = b->d_seg_x;
    B.seg_n = b->d_seg_n;
    B.seg_level = b->d_seg_level;
    B.seg_off = b->d_seg_off;
    B.row_off = b->d_row_off;
    B.y0 = b->d_y0;
    B.traces = b->d_traces;
    B.spk_cand = b->d_spk_cand;
    B.spk_stack = b->d_spk_stack;
    B.metrics = b->d_metrics;
    B.status = b->d_status;
    B.n_cfg = b->n_cfg;
    B.wave_level = b->d_wave_level;
    B.lanes = b->d_lanes;
    B.queue = b->d_queue;
    B.queue_head = b->d_queue_head;
    B.n_queue = b->n_queue;
    B.opts = SolverOpts{b->opts.rtol, b->opts.atol, b->opts.h0, b->opts.hmin, b->opts.max_steps,
                        b->qss_gates, b->opts.idrive * 1e-3};

    B.diag = dev_switch("PYSONIC_AMD_DIAG", 0);
    if (!b->chunks.empty() && b->n_cfg > 0) {
        // pipelined: the costliest chunk first (its wavefronts are dispatched first), every chunk on its own
        // stream; the rows of a chunk leave for the host the moment its kernel ends
        for (auto &ch : b->chunks) {
            HIP_TRY(hipEventRecord(ch.start, ch.stream));
            int rc = launch_slots(b, B, ch.slot0, ch.n_slots, ch.stream);
            if (rc != SONIC_OK) return rc;
            HIP_TRY(hipEventRecord(ch.stop, ch.stream));
            if (b->host_traces && ch.n_rows > 0)
                HIP_TRY(hipMemcpyAsync(b->host_traces + ch.row0 * b->ncol, b->d_traces + ch.row0 * b->ncol,
                                       (size_t)ch.n_rows * b->ncol * sizeof(double), hipMemcpyDeviceToHost,
                                       ch.stream));
        }
        b->launched = true;
        return SONIC_OK;
    }
    if (b->n_queue > 0) HIP_TRY(hipMemsetAsync(b->d_queue_head, 0, sizeof(int), b->stream));
    HIP_TRY(hipEventRecord(b->ev_start, b->stream));
    if (b->n_cfg > 0) {
        int rc = launch_slots(b, B, 0, b->n_slots, b->stream);
        if (rc != SONIC_OK) return rc;
    }
    HIP_TRY(hipEventRecord(b->ev_stop, b->stream));
    if (b->host_traces && b->d_traces && b->total_rows > 0)
        HIP_TRY(hipMemcpyAsync(b->host_traces, b->d_traces, (size_t)b->total_rows * b->ncol * sizeof(double),
                               hipMemcpyDeviceToHost, b->stream));
    b->launched = true;
    return SONIC_OK;
}

int sonic_batch_launch_to_host(sonic_batch_t *b, double *host_traces)
{
    if (!b) return set_error(SONIC_EINVAL, "sonic_batch_launch_to_host: null batch");
    if (host_traces && !b->d_traces)
        return set_error(SONIC_EINVAL, "batch was prepared with write_traces = 0");
    b->host_traces = host_traces;
    const int rc = sonic_batch_launch(b);
    return rc;
}

int sonic_batch_row_blocks(const sonic_batch_t *b, long long *row_start, long long *n_rows)
{
    if (!b) return set_error(SONIC_EINVAL, "sonic_batch_row_blocks: null batch");
    if (row_start) std::memcpy(row_start, b->row_start.data(), b->row_start.size() * sizeof(long long));
    if (n_rows) std::memcpy(n_rows, b->n_rows.data(), b->n_rows.size() * sizeof(long long));
    return SONIC_OK;
}

int sonic_batch_n_chunks(const sonic_batch_t *b) { return b ? (int)b->chunks.size() : -1; }

int sonic_batch_tolerances(const sonic_batch_t *b, double *rtol, double *atol)
{
    if (!b) return set_error(SONIC_EINVAL, "sonic_batch_tolerances: null batch");
    if (rtol) *rtol = b->opts.rtol;
    if (atol) *atol = b->opts.atol;
    return SONIC_OK;
}

int sonic_release_device_memory(void)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) return SONIC_OK;
    for (int d = 0; d < ndev && d < POOL_MAX_DEVICES; d++) {
        if (hipSetDevice(d) != hipSuccess) continue;
        pool_release(d);
    }
    return SONIC_OK;
}

int sonic_batch_sync(sonic_batch_t *b, float *kernel_ms)
{
    if (!b) return set_error(SONIC_EINVAL, "sonic_batch_sync: null batch");
    HIP_TRY(hipSetDevice(b->m->device));
    HIP_TRY(hipStreamSynchronize(b->stream));
    for (auto &ch : b->chunks) HIP_TRY(hipStreamSynchronize(ch.stream));
    if (kernel_ms) {
        *kernel_ms = 0.f;
        if (b->launched && b->chunks.empty()) HIP_TRY(hipEventElapsedTime(kernel_ms, b->ev_start, b->ev_stop));
        if (b->launched && !b->chunks.empty()) {
            // the interval the integration kernels were at work: first start to last stop
            for (auto &ch : b->chunks) {
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, b->chunks.front().start, ch.stop));
                *kernel_ms = std::max(*kernel_ms, ms);
            }
        }
    }
    return SONIC_OK;
}

int sonic_batch_chunk_times(sonic_batch_t *b, float *kernel_ms, float *done_ms)
{
    if (!b || b->chunks.empty()) return set_error(SONIC_EINVAL, "sonic_batch_chunk_times: not a pipelined batch");
    HIP_TRY(hipSetDevice(b->m->device));
    for (size_t k = 0; k < b->chunks.size(); k++) {
        auto &ch = b->chunks[k];
        HIP_TRY(hipStreamSynchronize(ch.stream));
        if (kernel_ms) HIP_TRY(hipEventElapsedTime(&kernel_ms[k], ch.start, ch.stop));
        if (done_ms) HIP_TRY(hipEventElapsedTime(&done_ms[k], b->chunks.front().start, ch.stop));
    }
    return SONIC_OK;
}

int sonic_batch_fetch(sonic_batch_t *b, double *traces, double *metrics, int *status)
{
    if (!b) return set_error(SONIC_EINVAL, "sonic_batch_fetch: null batch");
    HIP_TRY(hipSetDevice(b->m->device));
    HIP_TRY(hipStreamSynchronize(b->stream));
    for (auto &ch : b->chunks) HIP_TRY(hipStreamSynchronize(ch.stream));
    if (traces) {
        if (!b->d_traces)
            return set_error(SONIC_EINVAL, "batch was prepared with write_traces = 0");
        HIP_TRY(hipMemcpy(traces, b->d_traces, (size_t)b->total_rows * b->ncol * sizeof(double),
                          hipMemcpyDeviceToHost));
    }
    if (metrics)
        HIP_TRY(hipMemcpy(metrics, b->d_metrics,
                          (size_t)b->n_cfg * SONIC_NMETRICS * sizeof(double),
                          hipMemcpyDeviceToHost));
    if (status)
        HIP_TRY(hipMemcpy(status, b->d_status, (size_t)b->n_cfg * sizeof(int),
                          hipMemcpyDeviceToHost));
    return SONIC_OK;
}

int sonic_batch_fetch_strided(sonic_batch_t *b, double *traces, long long row_stride, double *metrics,
                              int *status)
{
    if (!b) return set_error(SONIC_EINVAL, "sonic_batch_fetch_strided: null batch");
    if (traces && row_stride < b->ncol)
        return set_error(SONIC_EINVAL, "sonic_batch_fetch_strided: row_stride smaller than the row");
    if (!traces || row_stride == b->ncol) return sonic_batch_fetch(b, traces, metrics, status);
    HIP_TRY(hipSetDevice(b->m->device));
    HIP_TRY(hipStreamSynchronize(b->stream));
    for (auto &ch : b->chunks) HIP_TRY(hipStreamSynchronize(ch.stream));
    if (!b->d_traces) return set_error(SONIC_EINVAL, "batch was prepared with write_traces = 0");
    HIP_TRY(hipMemcpy2D(traces, (size_t)row_stride * sizeof(double), b->d_traces, (size_t)b->ncol * sizeof(double),
                        (size_t)b->ncol * sizeof(double), (size_t)b->total_rows, hipMemcpyDeviceToHost));
    return sonic_batch_fetch(b, nullptr, metrics, status);
}

int sonic_batch_fetch_padded(sonic_batch_t *b, double *traces, long long row_stride, double *metrics,
                             int *status)
{
    if (!b) return set_error(SONIC_EINVAL, "sonic_batch_fetch_padded: null batch");
    if (traces && row_stride < b->ncol)
        return set_error(SONIC_EINVAL, "sonic_batch_fetch_padded: row_stride smaller than the row");
    if (!traces || row_stride == b->ncol) return sonic_batch_fetch(b, traces, metrics, status);
    HIP_TRY(hipSetDevice(b->m->device));
    if (!b->d_traces) return set_error(SONIC_EINVAL, "batch was prepared with write_traces = 0");
    if (b->total_rows > 0) {
        double *d_wide = nullptr;
        const size_t bytes = (size_t)b->total_rows * (size_t)row_stride * sizeof(double);
        for (auto &ch : b->chunks) HIP_TRY(hipStreamSynchronize(ch.stream));
        { const int prc = pool_alloc(b->m->device, (void **)&d_wide, bytes); if (prc != SONIC_OK) return prc; }
        const long long total = b->total_rows * row_stride;
        const unsigned grid = (unsigned)std::min<long long>((total + 255) / 256, 256LL * 64);
        hipLaunchKernelGGL(pad_rows_kernel, dim3(grid), dim3(256), 0, b->stream, b->d_traces, d_wide, b->total_rows,
                           b->ncol, (int)row_stride);
        hipError_t e = hipStreamSynchronize(b->stream);
        if (e == hipSuccess) e = hipMemcpy(traces, d_wide, bytes, hipMemcpyDeviceToHost);
        pool_free(b->m->device, d_wide);
        if (e != hipSuccess) return set_error(SONIC_EHIP, hipGetErrorString(e));
    }
    return sonic_batch_fetch(b, nullptr, metrics, status);
}

int sonic_host_alloc(size_t bytes, void **out)
{
    if (!out) return set_error(SONIC_EINVAL, "sonic_host_alloc: null argument");
    *out = nullptr;
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return SONIC_OK;
}

int sonic_host_free(void *p)
{
    if (p) HIP_TRY(hipHostFree(p));
    return SONIC_OK;
}

int sonic_batch_device_ptrs(sonic_batch_t *b, void **traces, void **metrics, void **status)
{
    if (!b) return set_error(SONIC_EINVAL, "sonic_batch_device_ptrs: null batch");
    if (traces) *traces = b->d_traces;
    if (metrics) *metrics = b->d_metrics;
    if (status) *status = b->d_status;
    return SONIC_OK;
}

int sonic_batch_run(sonic_model_t *m, const double *A, const double *tstop, const double *dt,
                    const double *ev_t, const double *ev_x, const long long *ev_off,
                    long long n_cfg, const double *y0, const sonic_opts_t *opts,
                    double *traces, double *metrics, int *status)
{
    sonic_batch_t *b = nullptr;
    int rc = sonic_batch_prepare(m, A, tstop, dt, ev_t, ev_x, ev_off, n_cfg, y0, opts, &b);
    if (rc != SONIC_OK) return rc;
    rc = sonic_batch_launch(b);
    if (rc == SONIC_OK) rc = sonic_batch_sync(b, nullptr);
    if (rc == SONIC_OK) rc = sonic_batch_fetch(b, traces, metrics, status);
    sonic_batch_destroy(b);
    return rc;
}

}  // extern "C"
