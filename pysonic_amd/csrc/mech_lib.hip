// pysonic_amd/csrc/mech_lib.hip -- mech_* entry points of include/pysonic_amd.h:
// batched NeuronalBilayerSonophore.computeEffVars (PySONIC/core/nbls.py:153-222), i.e. the
// lookup-table generation of scripts/run_lookups.py:22-175 (BASELINE config 3).
//
// Kernel mapping: one (f, A, Qm) lookup cell per lane, 64-thread workgroups, cells ordered by
// descending amplitude (cost grows with A) so a wavefront holds cells of similar cost. The
// (U, Z, ng) state, the DOPRI5 stages and the dense-output coefficients live in registers; the
// only memory traffic is the per-cell ring of 999 (Z, ng) samples of the current cycle
// ([sample][cell] layout: lanes of a wavefront touch consecutive addresses) and 9-19 doubles of
// results per cell -- the kernel is FP64-transcendental bound (2 pow + 1 sin per right-hand side,
// ~10 exp per sample in the averaging pass), not HBM bound.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <vector>

#include "lib_common.hpp"
#include "mech_core.hpp"
#include "mech_coop.hpp"

using namespace sonic;

struct MechDev {
    const double *f, *A, *Q;     // [n] per cell
    const int *order;            // [n] lane -> cell
    const double *fs;            // [n_fs]
    int n_fs;
    double phi;
    double *zs, *ngs;            // [999][n] scratch
    double *effvars;             // [n][n_fs][1 + NR]
    int *ncycles, *status;       // [n]
    long long n;
    MechOpts opts;
    int n_ov;                    // charge overtones per cell (0: constant charge)
    const double *ov_A, *ov_phi; // [n][n_ov]
    double *ov_out;              // [n][n_fs][2 n_ov]
};

template <int NEURON>
__global__ void __launch_bounds__(64) mech_cycles_kernel(const MechDev D, const BLSParams p, const int per_wave)
{
    const long long lane = lane_work_index(D.n, per_wave);
    if (lane >= D.n) return;
    const long long c = D.order[lane];
    constexpr int NV = 1 + NeuronRates<NEURON>::NR;
    int st = 0;
    // scratch is indexed by LANE (not by cell) so that a wavefront's samples are contiguous
    const MechOvertones ov{D.n_ov, D.ov_A + c * D.n_ov, D.ov_phi + c * D.n_ov,
                           D.ov_out + c * (long long)D.n_fs * 2 * D.n_ov};
    const int nc = mech_cell<NEURON>(p, D.f[c], D.A[c], D.phi, D.Q[c], D.fs, D.n_fs, D.opts,
                                     D.zs + lane, D.ngs + lane, (long)D.n,
                                     D.effvars + c * (long long)D.n_fs * NV, &st, ov);
    D.ncycles[c] = nc;
    D.status[c] = st;
}

// Octet-cooperative kernel (constant charge; mech_coop.hpp) for the costliest cells of a batch:
// wavefront w carries the cells cells[w per_wave .. (w + 1) per_wave), one per octet of 8 lanes, the other
// octets run shadow copies (full_coop_kernel).
template <int NEURON>
__global__ void __launch_bounds__(64)
mech_coop_kernel(const MechDev D, const BLSParams p, const int *cells, const long long K, double *scratch,
                 const int per_wave)
{
    const int o = threadIdx.x >> 3;
    const long long first = (long long)blockIdx.x * per_wave;
    const long long left = K - first;
    const int cnt = (int)(left < per_wave ? left : per_wave);
    if (cnt <= 0) return;
    const long long slot = first + (o < cnt ? o : o % cnt);
    const bool store = o < cnt;
    const long long c = cells[slot];
    int st = 0;
    const int nc = mech_coop_cell<OctOpsDev, NEURON>(p, D.f[c], D.A[c], D.phi, D.Q[c], D.fs, D.n_fs, D.opts,
                                             scratch + slot * (long long)MECH_COOP_SCRATCH_DOUBLES,
                                             D.effvars + c * (long long)D.n_fs * (1 + NeuronRates<NEURON>::NR), &st, store);
    if (store && OctOpsDev::leader()) {
        D.ncycles[c] = nc;
        D.status[c] = st;
    }
}

template <int NEURON>
static void launch_mech(const MechDev &D, const BLSParams &p, unsigned grid, int per_wave, hipStream_t stream)
{
    hipLaunchKernelGGL(mech_cycles_kernel<NEURON>, dim3(grid), dim3(64), 0, stream, D, p, per_wave);
}

static int mech_nrates(int neuron_id)
{
    switch (neuron_id) {
    case 0: return NeuronRates<0>::NR;
    case 1: return NeuronRates<1>::NR;
    case 2: return NeuronRates<2>::NR;
    case 6: return NeuronRates<6>::NR;
    case 7: return NeuronRates<7>::NR;
    case 8: return NeuronRates<8>::NR;
    case 9: return NeuronRates<9>::NR;
    case 10: return NeuronRates<10>::NR;
    case 11: return NeuronRates<11>::NR;
    case 12: return NeuronRates<12>::NR;
    case 3: return NeuronRates<3>::NR;
    case 4: return NeuronRates<4>::NR;
    case 5: return NeuronRates<5>::NR;
    }
    return -1;
}

extern "C" {

void mech_default_opts(mech_opts_t *o)
{
    o->rtol = 1e-9;            /* Dormand-Prince 8(5,3): as close to the converged cycles as the 5(4) pair at 1e-10 */
    o->max_steps = 50000000;
    o->ncycles_max = 10;
    o->phi = 3.14159265358979323846;
}

int mech_neuron_nrates(int neuron_id) { int n = mech_nrates(neuron_id); return n < 0 ? SONIC_EINVAL : n; }

static int mech_run(int device, int neuron_id, const double *bls_params, int n_bls_params,
                    const double *f, const double *A, const double *Q, long long n,
                    const double *fs, int n_fs, int n_ov, const double *ov_A, const double *ov_phi,
                    const mech_opts_t *opts, double *effvars, double *ov_out, int *ncycles,
                    int *status, float *kernel_ms)
{
    const int NR = mech_nrates(neuron_id);
    if (NR < 0) return set_error(SONIC_EINVAL, "unknown neuron id");
    if (!bls_params || n_bls_params != (int)(sizeof(BLSParams) / sizeof(double)))
        return set_error(SONIC_EINVAL, "mech_batch_run: expected 9 sonophore parameters");
    if (n < 0 || n_fs < 1 || !fs || !effvars || (n > 0 && (!f || !A || !Q)))
        return set_error(SONIC_EINVAL, "mech_batch_run: bad argument");
    if (n_ov < 0 || n_ov > 8 || (n_ov > 0 && n > 0 && (!ov_A || !ov_phi || !ov_out)))
        return set_error(SONIC_EINVAL, "mech_batch_run_overtones: bad overtone arguments");
    mech_opts_t o;
    if (opts) o = *opts; else mech_default_opts(&o);
    if (!(o.rtol > 0) || o.max_steps <= 0 || o.ncycles_max < 1)
        return set_error(SONIC_EINVAL, "mech_batch_run: invalid options");
    for (long long i = 0; i < n; i++) {
        if (!(f[i] > 0)) return set_error(SONIC_EINVAL, "Invalid f (must be strictly positive)");
        if (A[i] < 0) return set_error(SONIC_EINVAL, "Invalid A (must be positive or null)");
        // CHARGE_RANGE (constants.py:35, bls.py:674-677)
        if (Q[i] < -300e-5 || Q[i] > 150e-5)
            return set_error(SONIC_EINVAL, "Invalid applied charge (outside CHARGE_RANGE)");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_error(SONIC_ENODEV, "no HIP device available");
    if (device < 0 || device >= ndev) return set_error(SONIC_EINVAL, "device index out of range");
    if (kernel_ms) *kernel_ms = 0.f;
    if (n == 0) return SONIC_OK;
    HIP_TRY(hipSetDevice(device));

    BLSParams p;
    std::memcpy(&p, bls_params, sizeof(p));
    const int NV = 1 + NR;

    // lane order: descending amplitude, then descending |Q| (larger excursions cost more steps)
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        if (A[a] != A[b]) return A[a] > A[b];
        return std::fabs(Q[a]) > std::fabs(Q[b]);
    });
    // A launch lasts as long as its slowest wavefront, and the cost of a cell grows with the length of its
    // acoustic period and with the amplitude (BASELINE config 3: the 20 kHz cells at 300 - 600 kPa take 7 s,
    // a whole 500 kHz slice 0.6 s; profiles/r02l_mech_probe.txt). Cells with a constant charge can run
    // on the octet-cooperative kernel, where a step costs a third of a lane's: a small batch goes there whole,
    // a large one sends its costliest cells (a quarter of the batch at most, 8 cells per wavefront, on a stream
    // of higher priority: the long chains start first) and keeps the rest --
    // throughput-bound -- one cell per lane. PYSONIC_AMD_MECH_COOP=0: lane kernel only.
    std::vector<int> coop_cells;
    if (n_ov == 0 && dev_switch("PYSONIC_AMD_MECH_COOP", 1) != 0) {
        auto cost = [&](int i) { return (1.0 + A[i] / 50e3) * (A[i] == 0.0 ? 4.0 : 1.0) / f[i]; };
        std::vector<int> by_cost(order);
        std::stable_sort(by_cost.begin(), by_cost.end(), [&](int a, int b) { return cost(a) > cost(b); });
        // (measured on config 3, three radii in flight: 2048 of 56 406 cells per radius 4.5 s, 4096 3.6 s,
        // 8192 3.2 - 3.9 s, 16 384 2.9 s; lane kernel alone 6.9 s)
        long long K = n <= 1024 ? n : std::min<long long>(n / 4, 16384);
        if (const int k_dev = dev_switch("PYSONIC_AMD_MECH_COOP", 1); k_dev > 1) K = std::min<long long>(n, k_dev);   // development: K cells
        // a uniform batch gains nothing from the split: the cooperative kernel is for the tail
        if (n > 1024 && cost(by_cost[0]) < 4.0 * cost(by_cost[n / 2])) K = 0;
        coop_cells.assign(by_cost.begin(), by_cost.begin() + K);
        if (K > 0) {
            std::vector<char> taken(n, 0);
            for (int c : coop_cells) taken[c] = 1;
            std::vector<int> rest;
            for (int c : order)
                if (!taken[c]) rest.push_back(c);
            order.swap(rest);
        }
    }
    const long long n_lane = (long long)order.size(), n_coop = (long long)coop_cells.size();

    double *d_f = nullptr, *d_A = nullptr, *d_Q = nullptr, *d_fs = nullptr, *d_zs = nullptr,
           *d_ngs = nullptr, *d_eff = nullptr, *d_ovA = nullptr, *d_ovphi = nullptr, *d_ovout = nullptr;
    int *d_order = nullptr, *d_nc = nullptr, *d_st = nullptr, *d_coop = nullptr;
    double *d_csc = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr, e2 = nullptr;
    hipStream_t stream = nullptr;   // private stream: calls from several host threads overlap on the GPU
    hipStream_t stream2 = nullptr;  // the cooperative kernel of the costliest cells, beside the lane kernel
    int rc = SONIC_OK;
    auto fail = [&](hipError_t e, const char *what) {
        rc = set_error(SONIC_EHIP, std::string(what) + ": " + hipGetErrorString(e));
    };
#define TRY_(expr) do { if (rc == SONIC_OK) { hipError_t _e = (expr); if (_e != hipSuccess) fail(_e, #expr); } } while (0)
    const size_t nb = (size_t)n * sizeof(double);
    TRY_(hipMalloc(&d_f, nb));
    TRY_(hipMalloc(&d_A, nb));
    TRY_(hipMalloc(&d_Q, nb));
    TRY_(hipMalloc(&d_fs, (size_t)n_fs * sizeof(double)));
    const size_t nbl = (size_t)std::max<long long>(n_lane, 1) * sizeof(double);
    TRY_(hipMalloc(&d_zs, nbl * (MECH_NPC - 1)));
    TRY_(hipMalloc(&d_ngs, nbl * (MECH_NPC - 1)));
    if (n_coop > 0) {
        TRY_(hipMalloc(&d_coop, (size_t)n_coop * sizeof(int)));
        TRY_(hipMalloc(&d_csc, (size_t)n_coop * MECH_COOP_SCRATCH_DOUBLES * sizeof(double)));
        TRY_(hipMemcpy(d_coop, coop_cells.data(), (size_t)n_coop * sizeof(int), hipMemcpyHostToDevice));
    }
    TRY_(hipMalloc(&d_eff, nb * n_fs * NV));
    if (n_ov > 0) {
        TRY_(hipMalloc(&d_ovA, nb * n_ov));
        TRY_(hipMalloc(&d_ovphi, nb * n_ov));
        TRY_(hipMalloc(&d_ovout, nb * n_fs * 2 * n_ov));
        TRY_(hipMemcpy(d_ovA, ov_A, nb * n_ov, hipMemcpyHostToDevice));
        TRY_(hipMemcpy(d_ovphi, ov_phi, nb * n_ov, hipMemcpyHostToDevice));
    }
    TRY_(hipMalloc(&d_order, (size_t)n * sizeof(int)));
    TRY_(hipMalloc(&d_nc, (size_t)n * sizeof(int)));
    TRY_(hipMalloc(&d_st, (size_t)n * sizeof(int)));
    TRY_(hipMemcpy(d_f, f, nb, hipMemcpyHostToDevice));
    TRY_(hipMemcpy(d_A, A, nb, hipMemcpyHostToDevice));
    TRY_(hipMemcpy(d_Q, Q, nb, hipMemcpyHostToDevice));
    TRY_(hipMemcpy(d_fs, fs, (size_t)n_fs * sizeof(double), hipMemcpyHostToDevice));
    if (n_lane > 0) TRY_(hipMemcpy(d_order, order.data(), (size_t)n_lane * sizeof(int), hipMemcpyHostToDevice));
    TRY_(hipEventCreate(&e0));
    TRY_(hipEventCreate(&e1));
    TRY_(hipEventCreate(&e2));
    if (rc == SONIC_OK) {
        MechDev D{d_f, d_A, d_Q, d_order, d_fs, n_fs, o.phi, d_zs, d_ngs, d_eff, d_nc, d_st, n_lane,
                  MechOpts{o.rtol, o.max_steps, o.ncycles_max}, n_ov, d_ovA, d_ovphi, d_ovout};
        int dev_id = 0;
        (void)hipGetDevice(&dev_id);
        const int per_wave = items_per_wave(std::max<long long>(n_lane, 1), dev_id);
        const int pw_abs = per_wave < 0 ? -per_wave : per_wave;
        const unsigned grid = (unsigned)((n_lane + pw_abs - 1) / pw_abs);
        TRY_(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        {
            int prio_low = 0, prio_high = 0;
            (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
            TRY_(hipStreamCreateWithPriority(&stream2, hipStreamNonBlocking, prio_high));
        }
        TRY_(hipEventRecord(e0, stream));
        if (n_coop > 0 && rc == SONIC_OK) {
            int ncu = 0;
            if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev_id) != hipSuccess || ncu <= 0)
                ncu = 256;
            const long long q = (n_coop + 4LL * ncu - 1) / (4LL * ncu);
            const int cpw = (int)std::min<long long>(8, std::max<long long>(1, q));
            const unsigned cgrid = (unsigned)((n_coop + cpw - 1) / cpw);
#define COOP_(N) case N: hipLaunchKernelGGL(mech_coop_kernel<N>, dim3(cgrid), dim3(64), 0, stream2, D, p, d_coop, n_coop, d_csc, cpw); break
            switch (neuron_id) {
                COOP_(0); COOP_(1); COOP_(2); COOP_(3); COOP_(4); COOP_(5); COOP_(6); COOP_(7); COOP_(8); COOP_(9);
                COOP_(10); COOP_(11); COOP_(12);
            }
#undef COOP_
            TRY_(hipGetLastError());
        }
        TRY_(hipEventRecord(e2, stream2));
        if (n_lane > 0)
        switch (neuron_id) {
        case 0: launch_mech<0>(D, p, grid, per_wave, stream); break;
        case 1: launch_mech<1>(D, p, grid, per_wave, stream); break;
        case 2: launch_mech<2>(D, p, grid, per_wave, stream); break;
        case 3: launch_mech<3>(D, p, grid, per_wave, stream); break;
        case 4: launch_mech<4>(D, p, grid, per_wave, stream); break;
        case 5: launch_mech<5>(D, p, grid, per_wave, stream); break;
        case 6: launch_mech<6>(D, p, grid, per_wave, stream); break;
        case 7: launch_mech<7>(D, p, grid, per_wave, stream); break;
        case 8: launch_mech<8>(D, p, grid, per_wave, stream); break;
        case 9: launch_mech<9>(D, p, grid, per_wave, stream); break;
        case 10: launch_mech<10>(D, p, grid, per_wave, stream); break;
        case 11: launch_mech<11>(D, p, grid, per_wave, stream); break;
        case 12: launch_mech<12>(D, p, grid, per_wave, stream); break;
        }
        TRY_(hipGetLastError());
        TRY_(hipEventRecord(e1, stream));
        TRY_(hipStreamSynchronize(stream));
        TRY_(hipStreamSynchronize(stream2));
        if (rc == SONIC_OK && kernel_ms) {
            // from the first launch to the end of the later of the two kernels
            float a_ms = 0.f, b_ms = 0.f;
            TRY_(hipEventElapsedTime(&a_ms, e0, e1));
            TRY_(hipEventElapsedTime(&b_ms, e0, e2));
            *kernel_ms = std::max(a_ms, b_ms);
            if (dev_switch("PYSONIC_AMD_DIAG", 0) == 2)
                std::fprintf(stderr, "pysonic_amd: mech: %lld cells on the lane kernel %.0f ms, %lld on the cooperative kernel %.0f ms\n",
                             n_lane, a_ms, n_coop, b_ms);
        }
        TRY_(hipMemcpy(effvars, d_eff, nb * n_fs * NV, hipMemcpyDeviceToHost));
        if (n_ov > 0) TRY_(hipMemcpy(ov_out, d_ovout, nb * n_fs * 2 * n_ov, hipMemcpyDeviceToHost));
        if (ncycles) TRY_(hipMemcpy(ncycles, d_nc, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
        if (status) TRY_(hipMemcpy(status, d_st, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    }
#undef TRY_
    void *ptrs[] = {d_f, d_A, d_Q, d_fs, d_zs, d_ngs, d_eff, d_order, d_nc, d_st, d_ovA, d_ovphi, d_ovout, d_coop,
                    d_csc};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e2) (void)hipEventDestroy(e2);
    if (stream) (void)hipStreamDestroy(stream);
    if (stream2) (void)hipStreamDestroy(stream2);
    return rc;
}

int mech_batch_run(int device, int neuron_id, const double *bls_params, int n_bls_params,
                   const double *f, const double *A, const double *Q, long long n,
                   const double *fs, int n_fs, const mech_opts_t *opts, double *effvars,
                   int *ncycles, int *status, float *kernel_ms)
{
    return mech_run(device, neuron_id, bls_params, n_bls_params, f, A, Q, n, fs, n_fs, 0, nullptr,
                    nullptr, opts, effvars, nullptr, ncycles, status, kernel_ms);
}

int mech_batch_run_overtones(int device, int neuron_id, const double *bls_params, int n_bls_params,
                             const double *f, const double *A, const double *Q, long long n,
                             const double *fs, int n_fs, int n_overtones, const double *ov_A,
                             const double *ov_phi, const mech_opts_t *opts, double *effvars,
                             double *ov_out, int *ncycles, int *status, float *kernel_ms)
{
    return mech_run(device, neuron_id, bls_params, n_bls_params, f, A, Q, n, fs, n_fs, n_overtones,
                    ov_A, ov_phi, opts, effvars, ov_out, ncycles, status, kernel_ms);
}

}  // extern "C"
