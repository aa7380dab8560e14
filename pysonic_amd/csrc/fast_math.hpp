// pysonic_amd/csrc/fast_math.hpp
//
// Reciprocal, division, exp and log without the special-case handling of the library versions, for the
// integrator cores (device: hardware estimates + Newton steps / short polynomials; host build of the test
// harness: the libm functions). Results feed integrators with rtol >= 1e-10.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define SONIC_HD __host__ __device__ __forceinline__
#define SONIC_HD_CALL __host__ __device__ __attribute__((noinline))
#else
#define SONIC_HD inline
#define SONIC_HD_CALL inline
#endif

namespace sonic {

// 1/x without the IEEE-754 division expansion: hardware reciprocal estimate + two Newton steps
// (full double accuracy to within an ulp or two, which is all the W-matrix solve needs).
SONIC_HD double fast_rcp(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
#else
    return 1.0 / x;
#endif
}

// One Newton step: for the factors of W = I / (h gamma) - J, where a relative error of 1e-14 acts like
// a perturbation of the Jacobian far below its own accuracy.
SONIC_HD double fast_rcp1(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
#else
    return 1.0 / x;
#endif
}

// a / b without the IEEE-754 division expansion (~22 instructions per FP64 division on the GPU, and
// the mechanical right-hand sides hold some thirty of them): hardware reciprocal + two Newton steps,
// exact to an ulp or two.
SONIC_HD double qdiv(double a, double b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double r0 = __builtin_amdgcn_rcp(b);
    double r = fma(fma(-b, r0, 1.0), r0, r0);
    r = fma(fma(-b, r, 1.0), r, r);
    // b = +-inf or 0 (an overflowed exp() in a rate function): the refinement is 0 x inf = NaN,
    // the hardware estimate already is the IEEE result (0 or +-inf)
    return a * (r == r ? r : r0);
#else
    return a / b;
#endif
}

// exp and log (45 and 94 instructions in the device library on gfx950): arguments are finite and, for
// log, positive and normal; results within ~2 ulp.
//   exp: x = k ln2 + r, |r| <= ln2 / 2, Taylor polynomial of degree 12, ldexp
//   log: x = 2^e m, m in [sqrt(1/2), sqrt(2)), s = (m - 1) / (m + 1), log m = 2 s (1 + s^2/3 + ... + s^20/21)
SONIC_HD double fast_exp(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double k = __builtin_rint(x * 1.4426950408889634);
    double r = fma(-k, 6.93147180369123816490e-01, x);
    r = fma(-k, 1.90821492927058770002e-10, r);
    double q = 1.0 / 479001600.0;
    q = fma(q, r, 1.0 / 39916800.0);
    q = fma(q, r, 1.0 / 3628800.0);
    q = fma(q, r, 1.0 / 362880.0);
    q = fma(q, r, 1.0 / 40320.0);
    q = fma(q, r, 1.0 / 5040.0);
    q = fma(q, r, 1.0 / 720.0);
    q = fma(q, r, 1.0 / 120.0);
    q = fma(q, r, 1.0 / 24.0);
    q = fma(q, r, 1.0 / 6.0);
    q = fma(q, r, 0.5);
    q = fma(q, r, 1.0);
    q = fma(q, r, 1.0);
    // |x| beyond the range of int / ldexp: saturate the exponent (the result is 0 or inf either way)
    const double kc = fmin(fmax(k, -2000.0), 2000.0);
    return __builtin_amdgcn_ldexp(q, (int)kc);
#else
    return exp(x);
#endif
}

SONIC_HD double fast_log(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double m = __builtin_amdgcn_frexp_mant(x);            // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double f = m - 1.0;
    const double s = f * fast_rcp(2.0 + f);                // 2 + f in [1.7, 2.5): no guard
    const double z = s * s;
    double q = 2.0 / 21.0;
    q = fma(q, z, 2.0 / 19.0);
    q = fma(q, z, 2.0 / 17.0);
    q = fma(q, z, 2.0 / 15.0);
    q = fma(q, z, 2.0 / 13.0);
    q = fma(q, z, 2.0 / 11.0);
    q = fma(q, z, 2.0 / 9.0);
    q = fma(q, z, 2.0 / 7.0);
    q = fma(q, z, 2.0 / 5.0);
    q = fma(q, z, 2.0 / 3.0);
    const double ed = (double)e;
    // log x = e ln2 + 2 s + s z q
    return fma(ed, 6.93147180369123816490e-01, fma(s, 2.0, fma(s * z, q, ed * 1.90821492927058770002e-10)));
#else
    return log(x);
#endif
}

// sin for arguments of moderate size (|x| up to ~1e6: the phase of the drive, 2 pi f t - phi, over a few dozen
// periods): Cody-Waite reduction by pi / 2 in two pieces, Taylor polynomials of sin and cos on [-pi / 4, pi / 4]
// (truncation < 1e-19), quadrant selected without branches. The device library's sin spends ~150 instructions
// on an exact reduction this path does not need; absolute error here ~1e-16 (1 + |x| / 100).
SONIC_HD double fast_sin(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double k = __builtin_rint(x * 6.36619772367581382433e-01);
    double r = fma(-k, 1.57079632679489655800e+00, x);
    r = fma(-k, 6.12323399573676603587e-17, r);
    const double z = r * r;
    double ps = 1.0 / 355687428096000.0;              // 1 / 17!
    ps = fma(ps, z, -1.0 / 1307674368000.0);
    ps = fma(ps, z, 1.0 / 6227020800.0);
    ps = fma(ps, z, -1.0 / 39916800.0);
    ps = fma(ps, z, 1.0 / 362880.0);
    ps = fma(ps, z, -1.0 / 5040.0);
    ps = fma(ps, z, 1.0 / 120.0);
    ps = fma(ps, z, -1.0 / 6.0);
    const double sn = fma(r * z, ps, r);
    double pc = -1.0 / 6402373705728000.0;            // -1 / 18!
    pc = fma(pc, z, 1.0 / 20922789888000.0);
    pc = fma(pc, z, -1.0 / 87178291200.0);
    pc = fma(pc, z, 1.0 / 479001600.0);
    pc = fma(pc, z, -1.0 / 3628800.0);
    pc = fma(pc, z, 1.0 / 40320.0);
    pc = fma(pc, z, -1.0 / 720.0);
    pc = fma(pc, z, 1.0 / 24.0);
    pc = fma(pc, z, -0.5);
    const double cs = fma(pc, z, 1.0);
    const int n = (int)k;
    const double v = (n & 1) ? cs : sn;
    return (n & 2) ? -v : v;
#else
    return sin(x);
#endif
}

}  // namespace sonic
