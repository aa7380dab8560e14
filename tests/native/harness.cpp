// tests/native/harness.cpp -- CPU build of the per-lane integrator core (development, numerics
// experiments and sanitizer runs ONLY; never loaded by pysonic_amd).
#include <cstring>
#include "../../pysonic_amd/csrc/sonic_integrator.hpp"

using namespace sonic;

extern "C" int harness_run_rsfs(const double *params7, const double *recs, int n_levels,
                                int n_cells, double q0, double qmax, double inv_dq,
                                const double *t0, const double *t1, const double *x,
                                const int *n, const int *level, int nseg, const double *y0,
                                double rtol, double atol, double h0, double hmin, int max_steps,
                                double *rows /* [N][8] */, int *nsteps, int *nrej)
{
    (void)n_levels;
    CorticalParams P{params7[0], params7[1], params7[2], params7[3], params7[4], params7[5],
                     params7[6]};
    LevelGrid G{recs, n_cells, q0, qmax, inv_dq};
    Schedule S{t0, t1, x, n, level, nseg};
    SolverOpts o{rtol, atol, h0, hmin, max_steps};
    auto emit = [&](long row, double t, double xs, const double *y, double Vm) {
        double *r = rows + row * 8;
        r[0] = t; r[1] = xs;
        for (int i = 0; i < 5; i++) r[2 + i] = y[i];
        r[7] = Vm;
    };
    return integrate_config<CorticalRSFS>(P, G, S, y0, o, emit, nsteps, nrej);
}
