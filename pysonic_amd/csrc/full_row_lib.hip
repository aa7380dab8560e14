// pysonic_amd/csrc/full_row_lib.hip -- the row-cooperative kernel of the detailed model (full_row.hpp) and its
// launcher, a translation unit of its own (compiled beside full_lib.hip, which calls launch_full_row).
#include <hip/hip_runtime.h>

#include <cstring>
#include <vector>

#include "lib_common.hpp"
#include "full_row.hpp"
#include "hybrid_row.hpp"
#include "full_row_launch.hpp"

using namespace sonic;

// Wavefront w carries the `per_wave` (1 .. 4) configurations [w per_wave, (w + 1) per_wave), one per row of 16
// lanes; the remaining rows run shadow copies (same instructions, same data, no stores), so that all 64 lanes stay
// active: DPP moves never read a disabled lane, and a wavefront with more than 32 active lanes issues faster.
// MODE 0: the explicit 8(5,3) pair (gives a stiff configuration up); MODE 1: the pair and RODAS4 in turns (or, with
// opts.stiff_mode 2, RODAS4 from the start), for the configurations D.sel lists (full_row.hpp: full_row_config)
template <class M, int MODE>
__global__ void __launch_bounds__(64)
full_row_kernel(const FullDev D, const BLSParams p, const typename M::Params P, const LaneSpec *gl,
                const RowLaneSpec *rl, const int per_wave)
{
    const int o = threadIdx.x >> 4;
    const long long first = (long long)blockIdx.x * per_wave;
    const long long left = D.n - first;
    const int cnt = (int)(left < per_wave ? left : per_wave);
    if (cnt <= 0) return;
    const long long i = first + (o < cnt ? o : o % cnt);
    full_row_config<GroupOpsDev, M, MODE>(D, p, P, gl, rl, D.sel ? D.sel[i] : i, o < cnt);
}

template <class M>
static int launch_row(int neuron_id, const FullDev &D, const BLSParams &p, const std::vector<double> &params,
                      int device, bool stiff, void **specs_out)
{
    typename M::Params P;
    std::memcpy(&P, params.data(), sizeof(P));
    LaneSpec gl[GRP];
    RowLaneSpec rl[GRP];
    if (!GroupModel<M>::lanes(P, gl) || !row_lane_specs<M>(neuron_id, gl, rl))
        return set_error(SONIC_EINVAL, "row kernel: no lane layout for this neuron");
    char *d = (char *)*specs_out;
    if (!d) {
        HIP_TRY(hipMalloc((void **)&d, sizeof(gl) + sizeof(rl)));
        *specs_out = d;
        HIP_TRY(hipMemcpy(d, gl, sizeof(gl), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d + sizeof(gl), rl, sizeof(rl), hipMemcpyHostToDevice));
    }
    // rows per wavefront: as few as it takes to put one wavefront on every SIMD, 4 at most
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ncu <= 0) ncu = 256;
    const long long q = (D.n + 4LL * ncu - 1) / (4LL * ncu);
    const int per_wave = q > 4 ? 4 : (q < 1 ? 1 : (int)q);
    const unsigned grid = (unsigned)((D.n + per_wave - 1) / per_wave);
    if (stiff) {
        if constexpr (RowModel<M>::DEVICE_STIFF)
            hipLaunchKernelGGL((full_row_kernel<M, 1>), dim3(grid), dim3(64), 0, nullptr, D, p, P, (const LaneSpec *)d,
                               (const RowLaneSpec *)(d + sizeof(gl)), per_wave);
        else
            return set_error(SONIC_EINVAL, "row kernel: no Rosenbrock kernel for this neuron");
    } else
        hipLaunchKernelGGL((full_row_kernel<M, 0>), dim3(grid), dim3(64), 0, nullptr, D, p, P, (const LaneSpec *)d,
                           (const RowLaneSpec *)(d + sizeof(gl)), per_wave);
    return SONIC_OK;
}

bool full_row_available(int neuron_id)
{
    return neuron_id >= 2 && neuron_id <= 11;
}

template <class M>
static bool row_layout_ok(int neuron_id, const std::vector<double> &params)
{
    typename M::Params P;
    if (params.size() * sizeof(double) < sizeof(P)) return false;
    std::memcpy(&P, params.data(), sizeof(P));
    LaneSpec gl[GRP];
    RowLaneSpec rl[GRP];
    return GroupModel<M>::lanes(P, gl) && row_lane_specs<M>(neuron_id, gl, rl);
}

bool full_row_usable(int neuron_id, const std::vector<double> &params)
{
    switch (neuron_id) {
    case 2: case 6: return row_layout_ok<CorticalLTS>(neuron_id, params);
    case 3: return row_layout_ok<ThalamicRE>(neuron_id, params);
    case 4: return row_layout_ok<ThalamoCortical>(neuron_id, params);
    case 5: return row_layout_ok<OtsukaSTN>(neuron_id, params);
    case 7: return row_layout_ok<GatedModel<3>>(neuron_id, params);
    case 8: return row_layout_ok<GatedModel<2>>(neuron_id, params);
    case 9: case 10: case 11: return row_layout_ok<GatedModel<4>>(neuron_id, params);
    }
    return false;
}

bool full_row_stiff_available(int neuron_id)
{
    switch (neuron_id) {
    case 2: case 6: return RowModel<CorticalLTS>::DEVICE_STIFF;
    case 3: return RowModel<ThalamicRE>::DEVICE_STIFF;
    case 4: return RowModel<ThalamoCortical>::DEVICE_STIFF;
    case 5: return RowModel<OtsukaSTN>::DEVICE_STIFF;
    case 7: case 8: case 9: case 10: case 11: return true;
    }
    return false;
}

int launch_full_row(int neuron_id, const FullDev &D, const BLSParams &p, const std::vector<double> &params,
                    int device, bool stiff, void **specs_out)
{
    switch (neuron_id) {
    case 2: case 6: return launch_row<CorticalLTS>(neuron_id, D, p, params, device, stiff, specs_out);
    case 3: return launch_row<ThalamicRE>(neuron_id, D, p, params, device, stiff, specs_out);
    case 4: return launch_row<ThalamoCortical>(neuron_id, D, p, params, device, stiff, specs_out);
    case 5: return launch_row<OtsukaSTN>(neuron_id, D, p, params, device, stiff, specs_out);
    case 7: return launch_row<GatedModel<3>>(neuron_id, D, p, params, device, stiff, specs_out);
    case 8: return launch_row<GatedModel<2>>(neuron_id, D, p, params, device, stiff, specs_out);
    case 9: case 10: case 11: return launch_row<GatedModel<4>>(neuron_id, D, p, params, device, stiff, specs_out);
    }
    return set_error(SONIC_EINVAL, "row kernel: neuron not covered");
}

// ---- hybrid scheme on rows (hybrid_row.hpp): rows and shadow rows as full_row_kernel ----
template <class M, bool STIFF>
__global__ void __launch_bounds__(64)
hybrid_row_kernel(const HybridDev D, const BLSParams p, const typename M::Params P, const LaneSpec *gl,
                  const RowLaneSpec *rl, const int per_wave)
{
    const int o = threadIdx.x >> 4;
    const long long first = (long long)blockIdx.x * per_wave;
    const long long left = D.n - first;
    const int cnt = (int)(left < per_wave ? left : per_wave);
    if (cnt <= 0) return;
    const long long i = first + (o < cnt ? o : o % cnt);
    hybrid_row_config<GroupOpsDev, M, STIFF>(D, p, P, gl, rl, D.sel ? D.sel[i] : i, o < cnt);
}

template <class M>
static int launch_hyb_row(int neuron_id, const HybridDev &D, const BLSParams &p, const std::vector<double> &params,
                          int device, bool stiff, void **specs_out)
{
    typename M::Params P;
    std::memcpy(&P, params.data(), sizeof(P));
    LaneSpec gl[GRP];
    RowLaneSpec rl[GRP];
    if (!GroupModel<M>::lanes(P, gl) || !row_lane_specs<M>(neuron_id, gl, rl))
        return set_error(SONIC_EINVAL, "row kernel: no lane layout for this neuron");
    char *d = (char *)*specs_out;
    if (!d) {
        HIP_TRY(hipMalloc((void **)&d, sizeof(gl) + sizeof(rl)));
        *specs_out = d;
        HIP_TRY(hipMemcpy(d, gl, sizeof(gl), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d + sizeof(gl), rl, sizeof(rl), hipMemcpyHostToDevice));
    }
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ncu <= 0) ncu = 256;
    const long long q = (D.n + 4LL * ncu - 1) / (4LL * ncu);
    const int per_wave = q > 4 ? 4 : (q < 1 ? 1 : (int)q);
    const unsigned grid = (unsigned)((D.n + per_wave - 1) / per_wave);
    if constexpr (RowModel<M>::DEVICE_STIFF) {
        if (stiff) {
            hipLaunchKernelGGL((hybrid_row_kernel<M, true>), dim3(grid), dim3(64), 0, nullptr, D, p, P, (const LaneSpec *)d,
                               (const RowLaneSpec *)(d + sizeof(gl)), per_wave);
            return SONIC_OK;
        }
    }
    if (stiff) return set_error(SONIC_EINVAL, "row kernel: no RODAS4 build for this neuron");
    hipLaunchKernelGGL((hybrid_row_kernel<M, false>), dim3(grid), dim3(64), 0, nullptr, D, p, P, (const LaneSpec *)d,
                       (const RowLaneSpec *)(d + sizeof(gl)), per_wave);
    return SONIC_OK;
}

int launch_hybrid_row(int neuron_id, const HybridDev &D, const BLSParams &p, const std::vector<double> &params,
                      int device, bool stiff, void **specs_out)
{
    switch (neuron_id) {
    case 2: case 6: return launch_hyb_row<CorticalLTS>(neuron_id, D, p, params, device, stiff, specs_out);
    case 3: return launch_hyb_row<ThalamicRE>(neuron_id, D, p, params, device, stiff, specs_out);
    case 4: return launch_hyb_row<ThalamoCortical>(neuron_id, D, p, params, device, stiff, specs_out);
    case 5: return launch_hyb_row<OtsukaSTN>(neuron_id, D, p, params, device, stiff, specs_out);
    case 7: return launch_hyb_row<GatedModel<3>>(neuron_id, D, p, params, device, stiff, specs_out);
    case 8: return launch_hyb_row<GatedModel<2>>(neuron_id, D, p, params, device, stiff, specs_out);
    case 9: case 10: case 11: return launch_hyb_row<GatedModel<4>>(neuron_id, D, p, params, device, stiff, specs_out);
    }
    return set_error(SONIC_EINVAL, "row kernel: neuron not covered");
}
