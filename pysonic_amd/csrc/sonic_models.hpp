// pysonic_amd/csrc/sonic_models.hpp
//
// Point-neuron definitions for the SONIC effective system, as DATA + closed-form right-hand
// sides with ANALYTIC Jacobians (the reference defines them as Python lambdas whose source text
// is regex-rewritten at import time: PySONIC/core/translators.py:260-419 -- that mechanism is not
// reproduced; only its result is: which tables exist and what the effective RHS is).
//
// Every model exposes the same compile-time interface so the integrator (sonic_integrator.hpp)
// and the kernels are instantiated once per neuron type:
//
//   NG   number of first-order voltage-gated states  x' = alpha(Q) (1 - x) - beta(Q) x
//        ("gates": each depends on the charge Q and on itself only -> diagonal Jacobian block)
//   NC   number of "core" states: Q itself (index 0) followed by the states that are not pure
//        gates (Ca2+ concentration, iH regulation...). The core block is small and dense.
//   NT   number of lookup tables: table 0 is V_eff, tables 1.. are (alpha, beta) pairs in the
//        reference's effRates() order (translators.py:287-327).
//   State vector layout used on the device:  y = [core (NC) | gates (NG)].
//   `out_perm[i]` maps reference output column i (Qm, then `states` dict order) -> device index.
//
// Reference formulas: PySONIC/neurons/cortical.py:36-119 (+ subclasses 122-303),
// thalamic.py:31-114,164-366, stn.py:157-430; effective forms nbls.py:280-315,
// lookups.py:488-512 (tau = 1/(alpha+beta), xinf = alpha*tau  =>  (xinf - x)/tau == alpha -
// (alpha+beta) x up to rounding).
#pragma once

#if defined(__HIPCC__)
#define SONIC_HD __host__ __device__ __forceinline__
#else
#define SONIC_HD inline
#endif

namespace sonic {

// Jacobian pieces produced by Model::eval
//   fc[NC], fg[NG]            right-hand side
//   Jcc[NC][NC]               d fc / d core
//   Jcg[NC][NG]               d fc / d gates   (row 0 dense, other rows sparse)
//   Jgq[NG]                   d fg / d Q       (gates depend on core var 0 only)
//   Dg[NG]                    d fg_i / d g_i   (diagonal)
template <int NC, int NG>
struct Jac {
    double Jcc[NC][NC];
    double Jcg[NC][NG];
    double Jgq[NG];
    double Dg[NG];
};

// ---------------------------------------------------------------------------------------------
// Cortical regular-/fast-spiking neurons: states m h n p, currents iNa iKd iM iLeak
// (cortical.py:92-119, 122-201). Tables: V alpham betam alphah betah alphan betan alphap betap.
// ---------------------------------------------------------------------------------------------
struct CorticalParams {
    double gNabar, ENa, gKdbar, EK, gMbar, gLeak, ELeak;
};

struct CorticalRSFS {
    static constexpr int NG = 4;
    static constexpr int NC = 1;
    static constexpr int NT = 9;
    static constexpr int NY = NC + NG;
    typedef CorticalParams Params;
    // reference column order Qm m h n p -> device [Q | m h n p]
    SONIC_HD static int out_perm(int i) { return i; }

    // lk[k], dlk[k]: table value and dTable/dQ at the current charge
    template <bool WITH_JAC>
    SONIC_HD static void eval(const Params &P, const double *lk, const double *dlk,
                              const double *y, double *f, Jac<NC, NG> *J)
    {
        const double V = lk[0];
        const double m = y[1], h = y[2], n = y[3], p = y[4];
        const double m2 = m * m, m3 = m2 * m, n2 = n * n, n4 = n2 * n2;
        const double dNa = V - P.ENa, dK = V - P.EK;
        const double gNa = P.gNabar * m3 * h;
        const double gK = P.gKdbar * n4 + P.gMbar * p;
        // iNet (pneuron.py:288-296), dQ/dt = -iNet * 1e-3 (nbls.py:307)
        const double iNet = gNa * dNa + gK * dK + P.gLeak * (V - P.ELeak);
        f[0] = -1e-3 * iNet;
#pragma unroll
        for (int i = 0; i < NG; i++) {
            const double a = lk[1 + 2 * i], b = lk[2 + 2 * i];
            f[1 + i] = a - (a + b) * y[1 + i];
        }
        if (WITH_JAC) {
            J->Jcc[0][0] = -1e-3 * (gNa + gK + P.gLeak) * dlk[0];
            J->Jcg[0][0] = -1e-3 * (3.0 * P.gNabar * m2 * h) * dNa;
            J->Jcg[0][1] = -1e-3 * (P.gNabar * m3) * dNa;
            J->Jcg[0][2] = -1e-3 * (4.0 * P.gKdbar * n2 * n) * dK;
            J->Jcg[0][3] = -1e-3 * P.gMbar * dK;
#pragma unroll
            for (int i = 0; i < NG; i++) {
                const double a = lk[1 + 2 * i], b = lk[2 + 2 * i];
                const double da = dlk[1 + 2 * i], db = dlk[2 + 2 * i];
                J->Jgq[i] = da - (da + db) * y[1 + i];
                J->Dg[i] = -(a + b);
            }
        }
    }
};

}  // namespace sonic
