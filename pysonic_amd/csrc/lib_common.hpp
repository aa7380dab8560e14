// pysonic_amd/csrc/lib_common.hpp -- error plumbing shared by the translation units of
// libpysonic_amd.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <string>

#include "../../include/pysonic_amd.h"

inline std::string &last_error_string()
{
    static thread_local std::string s;
    return s;
}

inline int set_error(int code, const std::string &msg)
{
    last_error_string() = msg;
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return set_error(SONIC_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// Development / test switches of the library: environment variables read at prepare or launch time,
// ALL through dev_switch() below and nowhere else. They select between code paths that are each held
// to the same parity bars (tests/test_gpu_parity.py: test_golden_configs_kernel_variants,
// test_lane_kernel_packing, test_group_kernel) and exist for A/B measurements (tools/); unset = default.
//   PYSONIC_AMD_QUAD=0    RS / FS on the lane-per-configuration kernel instead of the quad kernel
//   PYSONIC_AMD_GROUP=0   LTS / IB / RE / TC / STN on the lane-per-configuration kernel instead of the group kernel
//   PYSONIC_AMD_MECH_COOP=0  lookup cells on the lane kernel only (no cooperative kernel for the costliest)
//   PYSONIC_AMD_LDS=1     quad kernel: level records staged in LDS instead of read from L2
//   PYSONIC_AMD_QPW=q     quad kernel: q configurations per wavefront (1 .. 16) instead of pack_wavefronts
//   PYSONIC_AMD_GPW=q     group kernel: q configurations per wavefront (1 .. 4)
//   PYSONIC_AMD_LPW=q     lane kernels: q configurations per wavefront (1 .. 64)
//   PYSONIC_AMD_IPW=q     mech / full / hybrid kernels: q items per wavefront (1 .. 64)
//   PYSONIC_AMD_SHADOW=1  mech / full / hybrid kernels: idle lanes run shadow copies
//   PYSONIC_AMD_WPS=n     quad / group kernels: n wavefronts per SIMD hold configurations at the start, the rest of
//                         the batch goes through the device-side work queue (0: no queue; unset: the kernel's occupancy)
//   PYSONIC_AMD_COST_FILE=path  sonic batches: cost estimates (ordering only) from a file of n_cfg doubles instead of the host's model
//   PYSONIC_AMD_STREAM=1  quad kernel: the work-queue build of the kernel even for a launch without a queue (A/B)
//   PYSONIC_AMD_ROW_NOFALLBACK=1  detailed model: a configuration the row kernels fail on is NOT rerun on the lane kernel
//   PYSONIC_AMD_DIAG=n    1: RESERVED metric = shader MHz; 2: print the packing; 3: lane kernels without shadow lanes
inline int dev_switch(const char *name, int unset)
{
    const char *e = std::getenv(name);
    return (e && e[0]) ? std::atoi(e) : unset;
}

// Work item of this lane in the one-item-per-lane kernels (blocks of 64 = one wavefront).
//
// Wavefront w carries the `per_wave` items [w per_wave, (w + 1) per_wave); lanes without an item of
// their own run a copy of one of them: a copy executes the same instructions on the same data in
// lockstep with the original and stores the same values to the same addresses. Two reasons:
//  * a wavefront with 32 or fewer ACTIVE lanes issues every instruction ~1.3x slower on gfx950
//    than one with 33 or more, wherever the active lanes sit (tools/micro/lanes_rate.hip: 8.5 vs
//    6.6 clocks per dependent FP64 instruction; 6.3 vs 4.7 with four independent chains);
//  * these kernels are latency-bound chains of ~10^4 .. 10^7 dependent steps, and a wavefront issues
//    the union of the paths its lanes take: a batch too small to fill the SIMDs runs faster spread
//    over more wavefronts (items_per_wave).
#if defined(__HIPCC__)
__device__ __forceinline__ long long lane_work_index(long long n, int per_wave)
{
    // per_wave < 0: |per_wave| items per wavefront and NO copies (the default of items_per_wave)
    const int q = per_wave < 0 ? -per_wave : per_wave;
    const long long first = (long long)blockIdx.x * q;
    const long long left = n - first;
    const int cnt = (int)(left < q ? left : q);
    if (cnt <= 0) return n;
    const int l = threadIdx.x & 63;
    if (per_wave < 0 && l >= cnt) return n;
    return first + (l < cnt ? l : l % cnt);
}
#endif

// Items per wavefront for a batch of n on `device`: as few as it takes to put one wavefront on
// every SIMD (4 per CU); full wavefronts once that would leave less than half of the lanes idle.
inline int items_per_wave(long long n, int device)
{
    const bool shadow = dev_switch("PYSONIC_AMD_SHADOW", 0) == 1;
    {
        const int v = dev_switch("PYSONIC_AMD_IPW", 0);
        if (v >= 1 && v <= 64) return shadow ? v : -v;
    }
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ncu <= 0)
        ncu = 256;
    const long long n_simd = 4LL * ncu;
    const long long q = (n + n_simd - 1) / n_simd;
    const int per_wave = q > 32 ? 64 : (q < 1 ? 1 : (int)q);
    // no copies in the idle lanes by default: these kernels keep their stage vectors in private memory
    // and store through HBM scratch, so 64 active lanes cost more memory traffic than the faster
    // issue of a wavefront with more than 32 active lanes gains (one 80 us run: full 3.07 s with
    // copies, 2.81 s without; hybrid 0.38 / 0.34 s). PYSONIC_AMD_SHADOW=1 turns them on.
    return shadow ? per_wave : -per_wave;
}
