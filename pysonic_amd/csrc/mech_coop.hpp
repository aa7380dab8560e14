// pysonic_amd/csrc/mech_coop.hpp
//
// OCTET-COOPERATIVE lookup cell (NeuronalBilayerSonophore.computeEffVars, PySONIC/core/nbls.py:153-222 =
// BilayerSonophore.simCycles + PeriodicSolver, bls.py:749-789, solvers.py:224-365) for a constant imposed
// charge (any neuron: the mechanical system does not know the neuron, the averaging pass uses its rate functions): mech_cell (mech_core.hpp) on the layout of full_coop.hpp -- one cell
// per 8 lanes, the mechanical right-hand side spread over the lanes (coop_rhs<O, false>).
//
// Why: a launch of the lookup generation lasts as long as its slowest wavefront, and that is the handful of
// low-frequency / high-amplitude cells (BASELINE config 3: the 20 kHz cells at 300 - 600 kPa, ~1e5 steps per
// acoustic period; profiles/r02l_mech_probe.txt). mech_lib.hip gives those cells to this kernel, where a step
// costs a third of what it costs a single lane, and leaves the rest -- throughput-bound -- one cell per lane.
#pragma once
#include "full_coop.hpp"

namespace sonic {

constexpr int MECH_COOP_SCRATCH_DOUBLES = 4 * (MECH_NPC - 1);     // per cell: U, Z, ng, (unused) of a cycle's samples

template <class O, int NEURON>
SONIC_HD int mech_coop_cell(const BLSParams &p, double f, double A, double phi, double Qm0,
                            const double *fs, int n_fs, const MechOpts &o, double *scratch, double *effvars,
                            int *status_out, bool store)
{
    typedef typename O::V V;
    constexpr int NS = MECH_NPC - 1;                 // samples per cycle
    constexpr int NV = 1 + NeuronRates<NEURON>::NR;
    int status = 0;
    const double w = 2.0 * bls::PI * f;
    const double Tper = 1.0 / f;
    const double dt = 1.0 / (MECH_NPC * f);          // drives.py:276-279
    bool clamped = false;

    CorticalParams P0{};                             // membrane parameters are not used by the mechanical system
    const CoopConsts<O> C = coop_consts<O>(p, P0, 0, 0.0);            // (its rate-function constants are not used either)
    const CoopScalars<O> S = coop_scalars<O>(p, 1.0, 0.0);

    const double Pac_dt = A * sin(w * dt - phi);
    const double Zqs = bls_balancedefQS(p, p.ng0, Qm0, Pac_dt);
    if (!(Zqs == Zqs)) {
        if (store && O::leader())
            for (int i = 0; i < n_fs * NV; i++) effvars[i] = NAN;
        *status_out = 2;
        return 0;
    }
    V y = O::roles(0.0, Zqs, p.ng0, Qm0, 0.0, 0.0, 0.0, 0.0);
    V K[16];
    double t = 0.0, h = dt;
    int nsteps = 0, ncycles = 0;
    double z_last_start = Zqs;     // Z at the start of the last cycle run (= row before its samples)
    bool converged = false;
    // the error norm of the stepper is an RMS over the 8 lanes, 5 of which carry no equation here: the same
    // tolerance on (U, Z, ng) as mech_cell's RMS over 3 components
    const double rtol = o.rtol * sqrt(3.0 / 8.0);

    for (int cyc = 0; cyc <= o.nmax_cycles && !converged; cyc++) {
        V sse = O::splat(0.0), vmax = O::splat(-INFINITY), nvmin = O::splat(-INFINITY);
        z_last_start = O::first(O::template bcast<1>(y));
        int ks = 0;
        auto dense = [&](double, V yd) SONIC_COOP_INLINE {
            if (cyc > 0) {
                const V d = O::sub(yd, O::load4(scratch, NS, ks));
                sse = O::fma_(d, d, sse);
            }
            O::store4(scratch, NS, ks, yd);
            vmax = O::max_(vmax, yd);
            nvmin = O::max_(nvmin, O::neg(yd));
            ks++;
        };
        const int bad = coop_integrate_segment<O, 8, decltype(dense) &, false>(
            C, S, w, phi, rtol, A, t, t + Tper, MECH_NPC, dt, y, K, h, nsteps, o.max_steps, clamped, dense);
        if (bad) { status |= bad; break; }
        t = t + Tper;
        ncycles++;
        if (cyc >= 1) {
            // isPeriodicallyStable (solvers.py:317-330): rmse / ptp < MAX_RMSE_PTP_RATIO, on Z and ng
            const V ptp = O::add(vmax, nvmin);
            const double rz = sqrt(O::first(O::template bcast<1>(sse)) / NS) / O::first(O::template bcast<1>(ptp));
            const double rn = sqrt(O::first(O::template bcast<2>(sse)) / NS) / O::first(O::template bcast<2>(ptp));
            converged = (rz < 1e-4) && (rn < 1e-4);
        }
    }
    if (!converged) status |= 8;
    if (clamped) status |= 1;

    // cycle averages over the last 1000 rows (nbls.py:181-201), as mech_cell: the rate functions in their
    // closed forms, replicated on the lanes (the per-lane rational form of the integration kernels multiplies
    // zero coefficients by powers of an exponential that overflow at the potentials a cell far above the
    // lookup grid reaches -- -3 V at 5 MPa; this pass is 1e-3 of the cell's cost)
    constexpr int NR = NeuronRates<NEURON>::NR;
    for (int j = 0; j < n_fs; j++) {
        const double fsj = fs[j];
        double sumV = 0.0, sumR[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) sumR[r] = 0.0;
        for (int ks = 0; ks <= NS; ks++) {
            const double zv = (ks == 0) ? z_last_start : O::first(O::template bcast<1>(O::load4(scratch, NS, ks - 1)));
            const double Cm = bls_capacitance(p, zv);
            const double Vm = Qm0 / (fsj * Cm + (1.0 - fsj) * p.Cm0) * 1e3;     // nbls.py:148-151,188
            double rates[NR];
            NeuronRates<NEURON>::eval(Vm, rates);
            sumV += Vm;
#pragma unroll
            for (int r = 0; r < NR; r++) sumR[r] += rates[r];
        }
        if (store && O::leader()) {
            double *ev = effvars + (long)j * NV;
            ev[0] = sumV * (1.0 / MECH_NPC);
#pragma unroll
            for (int r = 0; r < NR; r++) ev[1 + r] = sumR[r] * (1.0 / MECH_NPC);
        }
    }
    *status_out = status;
    return ncycles;
}

}  // namespace sonic
