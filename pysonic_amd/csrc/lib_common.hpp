// pysonic_amd/csrc/lib_common.hpp -- error plumbing shared by the translation units of
// libpysonic_amd.so.
#pragma once
#include <hip/hip_runtime.h>

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

// Work item of this lane in the one-item-per-lane kernels (blocks of 64 = one wavefront).
// A wavefront with 32 or fewer ACTIVE lanes issues every instruction ~1.3x slower on gfx950 than
// one with 33 or more, wherever the active lanes sit (tools/micro/lanes_rate.hip: 8.5 vs 6.6
// clocks per dependent FP64 instruction; 6.3 vs 4.7 with four independent chains). These kernels
// are latency-bound on small batches, so the lanes of a partially filled wavefront that have no
// item of their own run a copy of one of its items: a copy executes the same instructions on the
// same data in lockstep with the original and stores the same values to the same addresses.
// Returns n when the wavefront has no item at all.
#if defined(__HIPCC__)
__device__ __forceinline__ long long lane_work_index(long long n)
{
    const long long base = (long long)blockIdx.x * blockDim.x + (threadIdx.x & ~63);
    const int l = threadIdx.x & 63;
    if (base + l < n) return base + l;
    const long long m = n - base;            // items of this wavefront: 1..63 (or none)
    return m > 0 ? base + l % (int)m : n;
}
#endif
