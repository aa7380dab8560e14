// pysonic_amd/csrc/full_lib.hip -- full_* entry points of include/pysonic_amd.h: batched
// NeuronalBilayerSonophore.simulate(method='full') (PySONIC/core/nbls.py:331-354): the detailed
// NICE model = bilayer-sonophore mechanics (BilayerSonophore.derivatives, bls.py:681-718) coupled
// to the point-neuron equations with the true rate functions and the deflection-dependent
// capacitance (NBLS.fullDerivatives, nbls.py:265-278; PointNeuron.derivatives,
// pneuron.py:485-505), integrated by EventDrivenSolver on a dense grid of 1000 points per
// acoustic period and resampled to 10 ns (solvers.py:184-191,213-221; constants.py:37).
//
// Kernel mapping: one configuration per lane, 64-thread workgroups. State
// y = [U, Z, ng | Qm, core..., gates...] in registers; explicit Dormand-Prince 5(4) (the system
// is not stiff at the ~1 ns steps the mechanics impose); the right-hand side re-uses the SONIC
// models' closed forms (sonic_models.hpp) fed with the TRUE rates at Vm = Qm / Cm_eff(Z)
// (mech_core.hpp: NeuronRates) instead of interpolated tables. The dense-grid solution is never
// stored: every dense point is evaluated from the step's continuous extension and consumed at
// once by the linear resampling (np.interp semantics) onto the 10 ns output grid, so HBM traffic
// is the resampled rows only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "lib_common.hpp"
#include "full_core.hpp"
#include "full_coop.hpp"
#include "hybrid_core.hpp"
#include "hybrid_coop.hpp"
#include "full_row_launch.hpp"

using namespace sonic;

template <class M, int NEURON>
__global__ void __launch_bounds__(64)
full_integrate_kernel(const FullDev D, const BLSParams p, const typename M::Params P, const int per_wave)
{
    const long long i = lane_work_index(D.n, per_wave);
    if (i >= D.n) return;
    full_config<M, NEURON>(D, p, P, D.sel ? D.sel[i] : i);
}

template <class M, int NEURON>
static void launch_full(const FullDev &D, const BLSParams &p, const std::vector<double> &params,
                        unsigned grid, int per_wave)
{
    typename M::Params P;
    std::memcpy(&P, params.data(), sizeof(P));
    hipLaunchKernelGGL((full_integrate_kernel<M, NEURON>), dim3(grid), dim3(64), 0, nullptr, D, p, P, per_wave);
}

// Octet-cooperative kernel (RS, FS): wavefront w carries the `per_wave` configurations
// [w per_wave, (w + 1) per_wave), one per octet of 8 lanes; the remaining octets run shadow copies
// (same instructions, same data, no stores) so that all 64 lanes stay active: DPP moves never read
// a disabled lane and the wavefront issues at the rate of one with more than 32 active lanes.
template <int NEURON, int METHOD>
__global__ void __launch_bounds__(64)
full_coop_kernel(const FullDev D, const BLSParams p, const CorticalParams P, const int per_wave)
{
    const int o = threadIdx.x >> 3;
    const long long first = (long long)blockIdx.x * per_wave;
    const long long left = D.n - first;
    const int cnt = (int)(left < per_wave ? left : per_wave);
    if (cnt <= 0) return;
    const long long c = first + (o < cnt ? o : o % cnt);
    full_coop_config<OctOpsDev, METHOD>(D, p, P, NEURON, c, o < cnt);
}

template <class M, int NEURON>
__global__ void __launch_bounds__(64)
hybrid_integrate_kernel(const HybridDev D, const BLSParams p, const typename M::Params P, const int per_wave)
{
    const long long i = lane_work_index(D.n, per_wave);
    if (i >= D.n) return;
    hybrid_config<M, NEURON>(D, p, P, D.sel ? D.sel[i] : i);
}

// octet-cooperative hybrid kernel (RS, FS): octets and shadow octets as in full_coop_kernel
template <int NEURON>
__global__ void __launch_bounds__(64)
hybrid_coop_kernel(const HybridDev D, const BLSParams p, const CorticalParams P, const int per_wave)
{
    const int o = threadIdx.x >> 3;
    const long long first = (long long)blockIdx.x * per_wave;
    const long long left = D.n - first;
    const int cnt = (int)(left < per_wave ? left : per_wave);
    if (cnt <= 0) return;
    const long long c = first + (o < cnt ? o : o % cnt);
    hybrid_coop_config<OctOpsDev>(D, p, P, NEURON, c, o < cnt);
}

template <class M, int NEURON>
static void launch_hybrid(const HybridDev &D, const BLSParams &p, const std::vector<double> &params,
                          unsigned grid, int per_wave)
{
    typename M::Params P;
    std::memcpy(&P, params.data(), sizeof(P));
    hipLaunchKernelGGL((hybrid_integrate_kernel<M, NEURON>), dim3(grid), dim3(64), 0, nullptr, D, p, P, per_wave);
}

static int full_nstates(int id)
{
    switch (id) {
    case 0: case 1: return 4;
    case 2: case 6: return 6;
    case 7: return 3;
    case 8: return 2;
    case 12: return 1;
    case 9: case 10: case 11: return 4;
    case 3: return 5;
    case 4: return 9;
    case 5: return 12;
    }
    return -1;
}
static size_t full_nparams(int id)
{
    switch (id) {
    case 0: case 1: return sizeof(CorticalParams) / 8;
    case 2: case 6: return sizeof(LTSParams) / 8;
    case 7: return sizeof(GatedParams<3>) / 8;
    case 8: return sizeof(GatedParams<2>) / 8;
    case 12: return sizeof(GatedParams<1>) / 8;
    case 9: case 10: case 11: return sizeof(GatedParams<4>) / 8;
    case 3: return sizeof(REParams) / 8;
    case 4: return sizeof(TCParams) / 8;
    case 5: return sizeof(STNParams) / 8;
    }
    return 0;
}

static inline long long n_samples_ll(double t0, double tend, double dt)
{
    const long long n = (long long)std::nearbyint((tend - t0) / dt);
    return n > 2 ? n : 2;
}

extern "C" {

void full_default_opts(full_opts_t *o)
{
    o->rtol = 0.0;             /* 0: the default of the integration method, see full_batch_run */
    o->max_steps = 0;          /* 0: budget proportional to the dense grid (full_core.hpp) */
    o->target_dt = 1e-8;       /* CLASSIC_TARGET_DT, constants.py:37 */
    o->phi = 3.14159265358979323846;
    o->idrive = 0.0;
    o->kernel = 0;
    o->stiff = 1;
}

int full_count_rows(const double *tstop, long long n_cfg, double target_dt, long long *n_rows)
{
    if (!tstop || !n_rows || n_cfg < 0 || !(target_dt > 0))
        return set_error(SONIC_EINVAL, "full_count_rows: bad argument");
    // ODESolver.resample -> getTimeVector(t[0] = 0, t[-1] = tstop, dt = target_dt)
    for (long long c = 0; c < n_cfg; c++) n_rows[c] = n_samples_ll(0.0, tstop[c], target_dt);
    return SONIC_OK;
}

int full_batch_run(int device, int neuron_id, const double *neuron_params, int n_params,
                   const double *bls_params, int n_bls_params, const double *f, const double *A,
                   const double *fs, const double *tstop, const double *ev_t, const double *ev_x,
                   const long long *ev_off, long long n_cfg, const double *y0,
                   const full_opts_t *opts, double *traces, int *status, int *nsteps,
                   float *kernel_ms)
{
    const int NS = full_nstates(neuron_id);
    if (NS < 0) return set_error(SONIC_EINVAL, "unknown neuron id");
    if (!neuron_params || (size_t)n_params != full_nparams(neuron_id))
        return set_error(SONIC_EINVAL, "full_batch_run: neuron parameter count mismatch");
    if (!bls_params || n_bls_params != (int)(sizeof(BLSParams) / 8))
        return set_error(SONIC_EINVAL, "full_batch_run: expected 9 sonophore parameters");
    if (n_cfg < 0 || !ev_off || !y0 || !traces || (n_cfg > 0 && (!f || !A || !fs || !tstop)))
        return set_error(SONIC_EINVAL, "full_batch_run: bad argument");
    full_opts_t o;
    if (opts) o = *opts; else full_default_opts(&o);
    if (!(o.rtol >= 0) || o.max_steps < 0 || !(o.target_dt > 0) || o.kernel < 0 || o.kernel > 3 || o.stiff < 0 ||
        o.stiff > 2)
        return set_error(SONIC_EINVAL, "full_batch_run: invalid options");
    const bool coop = o.kernel != 1 && (neuron_id == 0 || neuron_id == 1);
    // one configuration per row of 16 lanes (csrc/full_row.hpp, 8(5,3) pair): LTS, RE, TC, STN, IB
    const bool row = o.kernel != 1 && o.kernel != 3 && full_row_available(neuron_id) &&
                     full_row_usable(neuron_id, std::vector<double>(neuron_params, neuron_params + n_params));
    if (o.kernel >= 2 && !coop && !row)
        return set_error(SONIC_EINVAL, "full_batch_run: no cooperative kernel for this neuron");
    const bool dop853 = (coop && o.kernel != 3) || row;
    // rtol 0: the default of the method. The 8(5,3) pair at 1e-7 is as close to the converged solution as
    // the 5(4) pair at 1e-8 (RS golden: 4e-8 / 3e-8 of the deflection range) with half the right-hand sides. The row
    // kernel runs its 8(5,3) pair at 1e-8 with a per-state guard (full_row.hpp: row_dp8_attempt): the T-type calcium
    // gate of LTS / TC has a rate function that JUMPS at -80 mV, and how well the steps close in on the jump is what
    // sets that gate's error -- 0.2 - 0.3 of the golden bar at 1e-8 (the lane kernel: 0.15 - 0.36), 0.5 - 1.1 at 1e-7,
    // for 5 300 steps against 15 000 (tests/native/proto_row.py, reference goldens of LTS / TC / STN).
    const double rtol_lane = o.rtol == 0 ? 1e-8 : o.rtol;     // (configurations the row kernel hands to the lane kernel)
    if (o.rtol == 0) o.rtol = (dop853 && !row) ? 1e-7 : 1e-8;
    if (kernel_ms) *kernel_ms = 0.f;
    if (n_cfg == 0) return SONIC_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_error(SONIC_ENODEV, "no HIP device available");
    if (device < 0 || device >= ndev) return set_error(SONIC_EINVAL, "device index out of range");

    // dense-grid segments: dt = 1 / (1000 f) (drives.py:276-279; solvers.py:445-480)
    std::vector<double> seg_t0, seg_t1, seg_x;
    std::vector<int> seg_n;
    std::vector<long long> seg_off(n_cfg + 1, 0), row_off(n_cfg + 1, 0);
    for (long long c = 0; c < n_cfg; c++) {
        if (!(f[c] > 0)) return set_error(SONIC_EINVAL, "Invalid f (must be strictly positive)");
        if (A[c] < 0) return set_error(SONIC_EINVAL, "Invalid A (must be positive or null)");
        const double dt = 1.0 / (MECH_NPC * f[c]);
        double tnow = 0.0, xcur = 0.0;
        auto push = [&](double te) {
            seg_t0.push_back(tnow);
            seg_t1.push_back(te);
            seg_x.push_back(xcur);
            seg_n.push_back((int)n_samples_ll(tnow, te, dt));
        };
        for (long long e = ev_off[c]; e < ev_off[c + 1]; e++) {
            if (ev_t[e] < tnow) return set_error(SONIC_EINVAL, "events must be sorted by time");
            if (ev_x[e] < 0.0)
                return set_error(SONIC_EINVAL, "Invalid time protocol: contains negative modulators");
            push(ev_t[e]);
            tnow = ev_t[e];
            xcur = ev_x[e];
        }
        if (tnow > tstop[c])
            return set_error(SONIC_EINVAL, "all events must occur before stopping time");
        push(tstop[c]);
        seg_off[c + 1] = (long long)seg_t0.size();
        row_off[c + 1] = row_off[c] + n_samples_ll(0.0, tstop[c], o.target_dt);
    }
    const int NCOL = NS + 6;
    const long long total_rows = row_off[n_cfg];

    HIP_TRY(hipSetDevice(device));
    BLSParams p;
    std::memcpy(&p, bls_params, sizeof(p));
    std::vector<double> params(neuron_params, neuron_params + n_params);
    std::vector<double> y0v(y0, y0 + 1 + NS);

    double *d_f = nullptr, *d_A = nullptr, *d_fs = nullptr, *d_ts = nullptr, *d_t0 = nullptr,
           *d_t1 = nullptr, *d_x = nullptr, *d_y0 = nullptr, *d_tr = nullptr;
    int *d_n = nullptr, *d_st = nullptr, *d_ns = nullptr;
    long long *d_so = nullptr, *d_ro = nullptr, *d_sel = nullptr;
    void *d_specs = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = SONIC_OK;
    auto fail = [&](hipError_t e, const char *what) {
        rc = set_error(SONIC_EHIP, std::string(what) + ": " + hipGetErrorString(e));
    };
#define TRY_(expr) do { if (rc == SONIC_OK) { hipError_t _e = (expr); if (_e != hipSuccess) fail(_e, #expr); } } while (0)
#define UP_(dst, vec, T) do { TRY_(hipMalloc((void **)&dst, std::max<size_t>((vec).size(), 1) * sizeof(T))); \
    TRY_(hipMemcpy(dst, (vec).data(), (vec).size() * sizeof(T), hipMemcpyHostToDevice)); } while (0)
    const size_t nb = (size_t)n_cfg * sizeof(double);
    TRY_(hipMalloc(&d_f, nb)); TRY_(hipMemcpy(d_f, f, nb, hipMemcpyHostToDevice));
    TRY_(hipMalloc(&d_A, nb)); TRY_(hipMemcpy(d_A, A, nb, hipMemcpyHostToDevice));
    TRY_(hipMalloc(&d_fs, nb)); TRY_(hipMemcpy(d_fs, fs, nb, hipMemcpyHostToDevice));
    TRY_(hipMalloc(&d_ts, nb)); TRY_(hipMemcpy(d_ts, tstop, nb, hipMemcpyHostToDevice));
    UP_(d_t0, seg_t0, double);
    UP_(d_t1, seg_t1, double);
    UP_(d_x, seg_x, double);
    UP_(d_n, seg_n, int);
    UP_(d_so, seg_off, long long);
    UP_(d_ro, row_off, long long);
    UP_(d_y0, y0v, double);
    TRY_(hipMalloc(&d_tr, (size_t)total_rows * NCOL * sizeof(double)));
    TRY_(hipMalloc(&d_st, (size_t)n_cfg * sizeof(int)));
    TRY_(hipMalloc(&d_ns, (size_t)n_cfg * sizeof(int)));
    TRY_(hipEventCreate(&e0));
    TRY_(hipEventCreate(&e1));
    if (rc == SONIC_OK) {
        FullDev D{d_f, d_A, d_fs, d_ts, d_t0, d_t1, d_x, d_n, d_so, d_ro, d_y0, d_tr, d_st, d_ns,
                  n_cfg, o.phi, FullOpts{o.rtol, o.max_steps, o.idrive * 1e-3, o.stiff}};
        // The RODAS4 path of the row kernel runs at 30 x the tolerance of the explicit pair (3e-7 by default): on the
        // stiff goldens (STN 500 kPa, TC 600 kPa) 0.02 - 0.05 of the bar from the reference's converged run -- whose own
        // default-tolerance run is the bar's measure -- in 15 000 - 17 400 step attempts (in turns with the explicit
        // pair; the lane kernel at 1e-8: 83 000 - 98 000). 1e-6 would save a tenth of them, but leaves SWnode at
        // 400 kPa 4e-4 of its deflection range from its converged run (3e-7: 1.5e-5). The steps of the stiff stretches
        // are set by the mechanical half of the system, which wants a higher order than 4, not by the gates.
        D.opts.rtol_stiff = 30.0 * o.rtol;
        int dev_id = 0;
        (void)hipGetDevice(&dev_id);
        int per_wave = items_per_wave(n_cfg, dev_id);
        if (coop) {
            // octets per wavefront: as few as it takes to put one wavefront on every SIMD, 8 at most
            int ncu = 0;
            if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev_id) != hipSuccess || ncu <= 0)
                ncu = 256;
            const long long q = (n_cfg + 4LL * ncu - 1) / (4LL * ncu);
            per_wave = q > 8 ? 8 : (int)q;
        }
        const int pw_abs = per_wave < 0 ? -per_wave : per_wave;
        const unsigned grid = (unsigned)((n_cfg + pw_abs - 1) / pw_abs);
        auto launch_lane = [&](const FullDev &DD, unsigned g, int pw) {
            switch (neuron_id) {
            case 0: launch_full<CorticalRSFS, 0>(DD, p, params, g, pw); break;
            case 1: launch_full<CorticalRSFS, 1>(DD, p, params, g, pw); break;
            case 2: launch_full<CorticalLTS, 2>(DD, p, params, g, pw); break;
            case 3: launch_full<ThalamicRE, 3>(DD, p, params, g, pw); break;
            case 4: launch_full<ThalamoCortical, 4>(DD, p, params, g, pw); break;
            case 5: launch_full<OtsukaSTN, 5>(DD, p, params, g, pw); break;
            case 6: launch_full<CorticalLTS, 6>(DD, p, params, g, pw); break;
            case 7: launch_full<GatedModel<3>, 7>(DD, p, params, g, pw); break;
            case 8: launch_full<GatedModel<2>, 8>(DD, p, params, g, pw); break;
            case 9: launch_full<GatedModel<4>, 9>(DD, p, params, g, pw); break;
            case 10: launch_full<GatedModel<4>, 10>(DD, p, params, g, pw); break;
            case 11: launch_full<GatedModel<4>, 11>(DD, p, params, g, pw); break;
            case 12: launch_full<GatedModel<1>, 12>(DD, p, params, g, pw); break;
            }
        };
        TRY_(hipEventRecord(e0, nullptr));
        if (coop) {
            CorticalParams P;
            std::memcpy(&P, params.data(), sizeof(P));
            if (neuron_id == 0 && dop853) hipLaunchKernelGGL((full_coop_kernel<0, 8>), dim3(grid), dim3(64), 0, nullptr, D, p, P, per_wave);
            else if (neuron_id == 0) hipLaunchKernelGGL((full_coop_kernel<0, 5>), dim3(grid), dim3(64), 0, nullptr, D, p, P, per_wave);
            else if (dop853) hipLaunchKernelGGL((full_coop_kernel<1, 8>), dim3(grid), dim3(64), 0, nullptr, D, p, P, per_wave);
            else hipLaunchKernelGGL((full_coop_kernel<1, 5>), dim3(grid), dim3(64), 0, nullptr, D, p, P, per_wave);
        } else if (row) {
            // explicit pair first (stiff = 2: Rosenbrock from the start); the configurations it gives up as stiff
            // (FULL_ST_STIFF: STN above ~190 kPa, TC at 600 kPa) restart on the row kernel that holds both integrators (full_row.hpp:
            // row_switching_segment); one that fails there too (step budget) goes to the lane kernel as a last resort. stiff = 0: the explicit pair alone.
            auto flagged = [&](int mask, std::vector<long long> &sel) {
                std::vector<int> st((size_t)n_cfg);
                TRY_(hipMemcpy(st.data(), d_st, (size_t)n_cfg * sizeof(int), hipMemcpyDeviceToHost));   // (waits for the kernel)
                sel.clear();
                if (rc == SONIC_OK)
                    for (long long c = 0; c < n_cfg; c++)
                        if (st[(size_t)c] & mask) sel.push_back(c);
            };
            auto subset = [&](const std::vector<long long> &sel) {
                if (d_sel) { (void)hipFree(d_sel); d_sel = nullptr; }
                UP_(d_sel, sel, long long);
                FullDev D2 = D;
                D2.n = (long long)sel.size();
                D2.sel = d_sel;
                return D2;
            };
            const bool row_stiff = full_row_stiff_available(neuron_id);
            auto to_lane = [&](FullDev D2) {
                D2.opts.rtol = rtol_lane;
                const int pw2 = items_per_wave(D2.n, dev_id), pa = pw2 < 0 ? -pw2 : pw2;
                if (rc == SONIC_OK) launch_lane(D2, (unsigned)((D2.n + pa - 1) / pa), pw2);
            };
            if (o.stiff == 2 && !row_stiff) to_lane(D);
            else if (rc == SONIC_OK) rc = launch_full_row(neuron_id, D, p, params, dev_id, o.stiff == 2, &d_specs);
            TRY_(hipGetLastError());
            if (o.stiff == 1) {
                std::vector<long long> sel;
                flagged(FULL_ST_STIFF, sel);
                if (!sel.empty()) {
                    const FullDev D2 = subset(sel);
                    if (!row_stiff) to_lane(D2);
                    else if (rc == SONIC_OK) rc = launch_full_row(neuron_id, D2, p, params, dev_id, true, &d_specs);
                    TRY_(hipGetLastError());
                }
            }
            if (o.stiff != 0 && dev_switch("PYSONIC_AMD_ROW_NOFALLBACK", 0) == 0) {
                std::vector<long long> sel;
                flagged(4, sel);
                if (!sel.empty()) to_lane(subset(sel));
            }
        } else
            launch_lane(D, grid, per_wave);
        TRY_(hipGetLastError());
        TRY_(hipEventRecord(e1, nullptr));
        TRY_(hipDeviceSynchronize());
        if (rc == SONIC_OK && kernel_ms) TRY_(hipEventElapsedTime(kernel_ms, e0, e1));
        TRY_(hipMemcpy(traces, d_tr, (size_t)total_rows * NCOL * sizeof(double), hipMemcpyDeviceToHost));
        if (status) TRY_(hipMemcpy(status, d_st, (size_t)n_cfg * sizeof(int), hipMemcpyDeviceToHost));
        if (nsteps) TRY_(hipMemcpy(nsteps, d_ns, (size_t)n_cfg * sizeof(int), hipMemcpyDeviceToHost));
    }
#undef TRY_
#undef UP_
    void *ptrs[] = {d_f, d_A, d_fs, d_ts, d_t0, d_t1, d_x, d_y0, d_tr, d_n, d_st, d_ns, d_so, d_ro, d_sel, d_specs};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return rc;
}

// NeuronalBilayerSonophore.simulate(method='hybrid') (nbls.py:356-387, solvers.py:483-633) for a
// queue of configurations; same arguments and row layout as full_batch_run, plus the number of
// dense periods integrated per configuration.
int hybrid_batch_run(int device, int neuron_id, const double *neuron_params, int n_params,
                     const double *bls_params, int n_bls_params, const double *f, const double *A,
                     const double *fs, const double *tstop, const double *ev_t, const double *ev_x,
                     const long long *ev_off, long long n_cfg, const double *y0,
                     const full_opts_t *opts, double *traces, int *status, int *nsteps,
                     int *ncycles, float *kernel_ms)
{
    const int NS = full_nstates(neuron_id);
    if (NS < 0) return set_error(SONIC_EINVAL, "unknown neuron id");
    if (!neuron_params || (size_t)n_params != full_nparams(neuron_id))
        return set_error(SONIC_EINVAL, "hybrid_batch_run: neuron parameter count mismatch");
    if (!bls_params || n_bls_params != (int)(sizeof(BLSParams) / 8))
        return set_error(SONIC_EINVAL, "hybrid_batch_run: expected 9 sonophore parameters");
    if (n_cfg < 0 || !ev_off || !y0 || !traces || (n_cfg > 0 && (!f || !A || !fs || !tstop)))
        return set_error(SONIC_EINVAL, "hybrid_batch_run: bad argument");
    full_opts_t o;
    if (opts) o = *opts; else full_default_opts(&o);
    if (!(o.rtol >= 0) || o.max_steps < 0 || !(o.target_dt > 0) || o.stiff < 0 || o.stiff > 2)
        return set_error(SONIC_EINVAL, "hybrid_batch_run: invalid options");
    if (o.kernel < 0 || o.kernel > 2)
        return set_error(SONIC_EINVAL, "hybrid_batch_run: kernel must be 0 (automatic), 1 (lane) or 2 (cooperative)");
    // cooperative kernels (8(5,3) pair): RS and FS one configuration per eight lanes, LTS / IB / RE / TC / STN one per row
    // of sixteen (csrc/hybrid_row.hpp)
    const bool coop = o.kernel != 1 && (neuron_id == 0 || neuron_id == 1);
    const bool row = o.kernel != 1 && full_row_available(neuron_id) &&
                     full_row_usable(neuron_id, std::vector<double>(neuron_params, neuron_params + n_params));
    if (o.kernel == 2 && !coop && !row)
        return set_error(SONIC_EINVAL, "hybrid_batch_run: no cooperative kernel for this neuron");
    // 5e-8 on the cooperative kernel: the scheme decides discretely when a cycle has closed, so its error does not
    // fall smoothly with the tolerance -- at 1e-7 the charge of the FS golden lands between 5e-8 and 1.2e-7 of its
    // range depending on rounding, at 5e-8 below 6e-8 throughout (tools/hybrid_parity_probe.py); same run time
    if (o.rtol == 0) o.rtol = coop ? 5e-8 : 1e-8;      // (the row kernel: 1e-8 with the per-state guard, as for 'full')
    if (kernel_ms) *kernel_ms = 0.f;
    if (n_cfg == 0) return SONIC_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return set_error(SONIC_ENODEV, "no HIP device available");
    if (device < 0 || device >= ndev) return set_error(SONIC_EINVAL, "device index out of range");

    std::vector<long long> row_off(n_cfg + 1, 0);
    for (long long c = 0; c < n_cfg; c++) {
        if (!(f[c] > 0)) return set_error(SONIC_EINVAL, "Invalid f (must be strictly positive)");
        if (A[c] < 0) return set_error(SONIC_EINVAL, "Invalid A (must be positive or null)");
        double tnow = 0.0;
        for (long long e = ev_off[c]; e < ev_off[c + 1]; e++) {
            if (ev_t[e] < tnow) return set_error(SONIC_EINVAL, "events must be sorted by time");
            if (ev_x[e] < 0.0)
                return set_error(SONIC_EINVAL, "Invalid time protocol: contains negative modulators");
            tnow = ev_t[e];
        }
        if (tnow > tstop[c])
            return set_error(SONIC_EINVAL, "all events must occur before stopping time");
        row_off[c + 1] = row_off[c] + n_samples_ll(0.0, tstop[c], o.target_dt);
    }
    const int NCOL = NS + 6;
    const long long total_rows = row_off[n_cfg];
    const long long n_ev = ev_off[n_cfg];

    HIP_TRY(hipSetDevice(device));
    BLSParams p;
    std::memcpy(&p, bls_params, sizeof(p));
    std::vector<double> params(neuron_params, neuron_params + n_params);

    double *d_f = nullptr, *d_A = nullptr, *d_fs = nullptr, *d_ts = nullptr, *d_et = nullptr,
           *d_ex = nullptr, *d_y0 = nullptr, *d_tr = nullptr, *d_sc = nullptr;
    int *d_st = nullptr, *d_ns = nullptr, *d_nc = nullptr;
    long long *d_eo = nullptr, *d_ro = nullptr, *d_sel = nullptr;
    void *d_specs = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = SONIC_OK;
    auto fail = [&](hipError_t e, const char *what) {
        rc = set_error(SONIC_EHIP, std::string(what) + ": " + hipGetErrorString(e));
    };
#define TRY_(expr) do { if (rc == SONIC_OK) { hipError_t _e = (expr); if (_e != hipSuccess) fail(_e, #expr); } } while (0)
#define UPN_(dst, src, count, T) do { TRY_(hipMalloc((void **)&dst, std::max<size_t>((size_t)(count), 1) * sizeof(T))); \
    if ((count) > 0) TRY_(hipMemcpy(dst, src, (size_t)(count) * sizeof(T), hipMemcpyHostToDevice)); } while (0)
    UPN_(d_f, f, n_cfg, double);
    UPN_(d_A, A, n_cfg, double);
    UPN_(d_fs, fs, n_cfg, double);
    UPN_(d_ts, tstop, n_cfg, double);
    UPN_(d_et, ev_t, n_ev, double);
    UPN_(d_ex, ev_x, n_ev, double);
    UPN_(d_eo, ev_off, n_cfg + 1, long long);
    UPN_(d_ro, row_off.data(), n_cfg + 1, long long);
    UPN_(d_y0, y0, 1 + NS, double);
    TRY_(hipMalloc(&d_tr, (size_t)total_rows * NCOL * sizeof(double)));
    TRY_(hipMalloc(&d_sc, (size_t)n_cfg * HYB_SCRATCH_DOUBLES * sizeof(double)));
    TRY_(hipMalloc(&d_st, (size_t)n_cfg * sizeof(int)));
    TRY_(hipMalloc(&d_ns, (size_t)n_cfg * sizeof(int)));
    TRY_(hipMalloc(&d_nc, (size_t)n_cfg * sizeof(int)));
    TRY_(hipEventCreate(&e0));
    TRY_(hipEventCreate(&e1));
    if (rc == SONIC_OK) {
        HybridDev D{d_f, d_A, d_fs, d_ts, d_et, d_ex, d_eo, d_ro, d_y0, d_tr, d_sc, d_st, d_ns,
                    d_nc, n_cfg, o.phi, FullOpts{o.rtol, o.max_steps, o.idrive * 1e-3, o.stiff}};
        D.opts.rtol_stiff = 30.0 * o.rtol;         // (RODAS4 dense periods of the row kernel: as full_batch_run)
        int dev_id = 0;
        (void)hipGetDevice(&dev_id);
        // dense: the ring of the last two periods lives in HBM, indexed so that the lanes of a wavefront
        // touch neighbouring words; spread over 256 wavefronts the same batch ran 20 % slower
        int per_wave = -64;
        if (coop) {
            // as full_batch_run: as few octets per wavefront as it takes to give every SIMD one
            int ncu = 0;
            if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev_id) != hipSuccess || ncu <= 0)
                ncu = 256;
            const long long q = (n_cfg + 4LL * ncu - 1) / (4LL * ncu);
            per_wave = (int)std::min<long long>(8, std::max<long long>(1, q));
        }
        const int pw_abs = per_wave < 0 ? -per_wave : per_wave;
        const unsigned grid = (unsigned)((n_cfg + pw_abs - 1) / pw_abs);
        auto launch_lane = [&](const HybridDev &DD, unsigned g, int pw) {
            switch (neuron_id) {
            case 0: launch_hybrid<CorticalRSFS, 0>(DD, p, params, g, pw); break;
            case 1: launch_hybrid<CorticalRSFS, 1>(DD, p, params, g, pw); break;
            case 2: launch_hybrid<CorticalLTS, 2>(DD, p, params, g, pw); break;
            case 3: launch_hybrid<ThalamicRE, 3>(DD, p, params, g, pw); break;
            case 4: launch_hybrid<ThalamoCortical, 4>(DD, p, params, g, pw); break;
            case 5: launch_hybrid<OtsukaSTN, 5>(DD, p, params, g, pw); break;
            case 6: launch_hybrid<CorticalLTS, 6>(DD, p, params, g, pw); break;
            case 7: launch_hybrid<GatedModel<3>, 7>(DD, p, params, g, pw); break;
            case 8: launch_hybrid<GatedModel<2>, 8>(DD, p, params, g, pw); break;
            case 9: launch_hybrid<GatedModel<4>, 9>(DD, p, params, g, pw); break;
            case 10: launch_hybrid<GatedModel<4>, 10>(DD, p, params, g, pw); break;
            case 11: launch_hybrid<GatedModel<4>, 11>(DD, p, params, g, pw); break;
            case 12: launch_hybrid<GatedModel<1>, 12>(DD, p, params, g, pw); break;
            }
        };
        TRY_(hipEventRecord(e0, nullptr));
        if (coop) {
            CorticalParams P;
            std::memcpy(&P, params.data(), sizeof(P));
            if (neuron_id == 0)
                hipLaunchKernelGGL((hybrid_coop_kernel<0>), dim3(grid), dim3(64), 0, nullptr, D, p, P, per_wave);
            else
                hipLaunchKernelGGL((hybrid_coop_kernel<1>), dim3(grid), dim3(64), 0, nullptr, D, p, P, per_wave);
        } else if (row) {
            // explicit pair first (stiff = 2: RODAS4 dense periods from the start); the configurations whose dense periods
            // it gives up as stiff (FULL_ST_STIFF) restart on the RODAS4 build of the row kernel (a model without one: on
            // the lane kernel); one that runs out of its step budget there goes to the lane kernel as a last resort
            // (explicit 5(4) pair; sparse phase on RODAS4 there too). As full_batch_run.
            auto flagged = [&](int mask, std::vector<long long> &sel) {
                std::vector<int> st((size_t)n_cfg);
                TRY_(hipMemcpy(st.data(), d_st, (size_t)n_cfg * sizeof(int), hipMemcpyDeviceToHost));   // (waits for the kernel)
                sel.clear();
                if (rc == SONIC_OK)
                    for (long long c = 0; c < n_cfg; c++)
                        if (st[(size_t)c] & mask) sel.push_back(c);
            };
            auto subset = [&](const std::vector<long long> &sel) {
                if (d_sel) { (void)hipFree(d_sel); d_sel = nullptr; }
                UPN_(d_sel, sel.data(), sel.size(), long long);
                HybridDev D2 = D;
                D2.n = (long long)sel.size();
                D2.sel = d_sel;
                return D2;
            };
            const bool row_stiff = full_row_stiff_available(neuron_id);
            auto to_lane = [&](HybridDev D2) {
                const int pw2 = items_per_wave(D2.n, dev_id), pa = pw2 < 0 ? -pw2 : pw2;
                if (rc == SONIC_OK) launch_lane(D2, (unsigned)((D2.n + pa - 1) / pa), pw2);
            };
            if (o.stiff == 2 && !row_stiff) to_lane(D);
            else if (rc == SONIC_OK) rc = launch_hybrid_row(neuron_id, D, p, params, dev_id, o.stiff == 2, &d_specs);
            TRY_(hipGetLastError());
            if (o.stiff == 1) {
                std::vector<long long> sel;
                flagged(FULL_ST_STIFF, sel);
                if (!sel.empty()) {
                    const HybridDev D2 = subset(sel);
                    if (!row_stiff) to_lane(D2);
                    else if (rc == SONIC_OK) rc = launch_hybrid_row(neuron_id, D2, p, params, dev_id, true, &d_specs);
                    TRY_(hipGetLastError());
                }
            }
            if (o.stiff != 0 && dev_switch("PYSONIC_AMD_ROW_NOFALLBACK", 0) == 0) {
                std::vector<long long> sel;
                flagged(4, sel);
                if (!sel.empty()) to_lane(subset(sel));
            }
        } else
            launch_lane(D, grid, per_wave);
        TRY_(hipGetLastError());
        TRY_(hipEventRecord(e1, nullptr));
        TRY_(hipDeviceSynchronize());
        if (rc == SONIC_OK && kernel_ms) TRY_(hipEventElapsedTime(kernel_ms, e0, e1));
        TRY_(hipMemcpy(traces, d_tr, (size_t)total_rows * NCOL * sizeof(double), hipMemcpyDeviceToHost));
        if (status) TRY_(hipMemcpy(status, d_st, (size_t)n_cfg * sizeof(int), hipMemcpyDeviceToHost));
        if (nsteps) TRY_(hipMemcpy(nsteps, d_ns, (size_t)n_cfg * sizeof(int), hipMemcpyDeviceToHost));
        if (ncycles) TRY_(hipMemcpy(ncycles, d_nc, (size_t)n_cfg * sizeof(int), hipMemcpyDeviceToHost));
    }
#undef TRY_
#undef UPN_
    void *ptrs[] = {d_f, d_A, d_fs, d_ts, d_et, d_ex, d_y0, d_tr, d_sc, d_st, d_ns, d_nc, d_eo, d_ro, d_sel, d_specs};
    for (void *q : ptrs)
        if (q) (void)hipFree(q);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return rc;
}

}  // extern "C"
