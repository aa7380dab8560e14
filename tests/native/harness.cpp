// tests/native/harness.cpp -- CPU build of the per-lane integrator core (development, numerics
// experiments and sanitizer runs ONLY; never loaded by pysonic_amd).
#include <cstring>
#include <cstdlib>
#include "../../pysonic_amd/csrc/sonic_integrator.hpp"
#include "../../pysonic_amd/csrc/sonic_quad.hpp"
#include "../../pysonic_amd/csrc/sonic_group.hpp"
#include "../../pysonic_amd/csrc/mech_core.hpp"
#include "../../pysonic_amd/csrc/full_core.hpp"
#include "../../pysonic_amd/csrc/hybrid_core.hpp"

using namespace sonic;

template <class M>
static int run_model(const double *params, const LevelGrid &G, const Schedule &S, const double *y0ref,
                     const SolverOpts &o, double *rows, int *nsteps, int *nrej)
{
    typename M::Params P;
    std::memcpy(&P, params, sizeof(P));
    constexpr int NY = M::NY, NCOL = NY + 3;
    double y0[NY];
    for (int i = 0; i < NY; i++) y0[M::out_perm(i)] = y0ref[i];
    auto emit = [&](long row, double t, double xs, const double *y, double Vm) {
        double *r = rows + row * NCOL;
        r[0] = t; r[1] = xs;
        for (int i = 0; i < NY; i++) r[2 + i] = y[M::out_perm(i)];
        r[2 + NY] = Vm;
    };
    CellRec<M::NT> home;
    return integrate_config<M>(P, G, S, y0, o, emit, nsteps, nrej, home);
}

extern "C" int harness_run(int neuron_id, const double *params, const double *recs, int n_levels,
                           int n_cells, double q0, double qmax, double inv_dq,
                           const double *t0, const double *t1, const double *x,
                           const int *n, const int *level, int nseg, const double *y0,
                           double rtol, double atol, double h0, double hmin, int max_steps,
                           double *rows, int *nsteps, int *nrej)
{
    (void)n_levels;
    LevelGrid G{recs, n_cells, q0, qmax, inv_dq};
    Schedule S{t0, t1, x, n, level, nseg};
    SolverOpts o{rtol, atol, h0, hmin, max_steps, 0};
    switch (neuron_id) {
    case 0: case 1: return run_model<CorticalRSFS>(params, G, S, y0, o, rows, nsteps, nrej);
    case 2: case 6: return run_model<CorticalLTS>(params, G, S, y0, o, rows, nsteps, nrej);
    case 3: return run_model<ThalamicRE>(params, G, S, y0, o, rows, nsteps, nrej);
    case 4: return run_model<ThalamoCortical>(params, G, S, y0, o, rows, nsteps, nrej);
    case 5: return run_model<OtsukaSTN>(params, G, S, y0, o, rows, nsteps, nrej);
    case 7: return run_model<GatedModel<3>>(params, G, S, y0, o, rows, nsteps, nrej);
    case 8: return run_model<GatedModel<2>>(params, G, S, y0, o, rows, nsteps, nrej);
    case 9: case 10: case 11: return run_model<GatedModel<4>>(params, G, S, y0, o, rows, nsteps, nrej);
    }
    return -1;
}

// quad-cooperative RS/FS integrator with the 4-array emulation backend; recs in QUAD_REC layout
extern "C" int harness_run_quad(const double *params, const double *qrecs, int n_cells, double q0,
                                double qmax, double inv_dq, const double *t0, const double *t1,
                                const double *x, const int *n, const int *level, int nseg,
                                const double *y0, double rtol, double atol, double h0, double hmin,
                                int max_steps, double *rows, int *nsteps, int *nrej)
{
    CorticalParams P;
    std::memcpy(&P, params, sizeof(P));
    QuadGrid G{qrecs, n_cells, q0, qmax, inv_dq};
    Schedule S{t0, t1, x, n, level, nseg};
    SolverOpts o{rtol, atol, h0, hmin, max_steps, 0};
    auto emit = [&](long row, double t, double xs, double q, QuadOpsHost::V g, double Vm) {
        QuadOpsHost::store_row(rows + row * 8, t, xs, q, g, Vm);
    };
    TabGlobal<QuadOpsHost> T{qrecs, n_cells * QUAD_REC};
    return integrate_config_quad<QuadOpsHost>(P, G, T, S, y0, o, emit, nsteps, nrej);
}

#ifdef SONIC_QUAD_STATS
extern "C" void harness_quad_stats(long *out26, double *sums2, int reset)
{
    QuadStats &Q = quad_stats();
    const long v[8] = {Q.steps, Q.capped, Q.capped_acc, Q.errlim_acc, Q.last_acc, Q.rej_err, Q.rej_over, Q.cross};
    for (int i = 0; i < 8; i++) out26[i] = v[i];
    for (int i = 0; i < 8; i++) { out26[8 + i] = Q.err_hist_capped[i]; out26[16 + i] = Q.err_hist_free[i]; }
    sums2[0] = Q.sum_h_capped; sums2[1] = Q.sum_h_free;
    if (reset) Q = QuadStats{};
}
extern "C" void harness_quad_dom(long *out10) { for (int i = 0; i < 5; i++) { out10[i] = quad_stats().dom_free[i]; out10[5 + i] = quad_stats().dom_cap[i]; } }
extern "C" void harness_quad_log(double *buf, long cap) { quad_stats().log = buf; quad_stats().nlog = 0; quad_stats().caplog = cap; }
extern "C" long harness_quad_nlog() { return quad_stats().nlog; }
#endif

// group-cooperative LTS / RE / TC / STN integrator with the 16-array emulation backend; recs in the lane layout
template <class M>
static int run_group(const double *params, const QuadGrid &G, const Schedule &S, const double *y0,
                     const SolverOpts &o, double *rows, int *nsteps, int *nrej)
{
    typedef GroupModel<M> GM;
    typedef GroupOpsHost O;
    typename GM::Params P;
    std::memcpy(&P, params, sizeof(P));
    LaneSpec specs[GRP];
    if (!GM::lanes(P, specs)) return -2;            // currents that do not fit the lane layout
    GroupConsts<O> C;
    O::load_consts(specs, C);
    auto emit = [&](long row, double t, double xs, const double *z, O::V g, double Vm) {
        O::template store_row<GM::NC>(rows + row * GM::NCOL, C, t, xs, Vm, z, g);
    };
    GroupTab<O, GM> T{G.recs, G.n_cells * GroupTab<O, GM>::REC};
    return integrate_config_group<O, GM>(P, C, G, T, S, y0, o, emit, nsteps, nrej);
}

extern "C" int harness_run_group(int neuron_id, const double *params, const double *recs, int n_levels,
                                 int n_cells, double q0, double qmax, double inv_dq,
                                 const double *t0, const double *t1, const double *x,
                                 const int *n, const int *level, int nseg, const double *y0,
                                 double rtol, double atol, double h0, double hmin, int max_steps,
                                 double *rows, int *nsteps, int *nrej)
{
    (void)n_levels;
    QuadGrid G{recs, n_cells, q0, qmax, inv_dq};
    Schedule S{t0, t1, x, n, level, nseg};
    SolverOpts o{rtol, atol, h0, hmin, max_steps, 0};
    switch (neuron_id) {
    case 2: case 6: return run_group<CorticalLTS>(params, G, S, y0, o, rows, nsteps, nrej);
    case 3: return run_group<ThalamicRE>(params, G, S, y0, o, rows, nsteps, nrej);
    case 4: return run_group<ThalamoCortical>(params, G, S, y0, o, rows, nsteps, nrej);
    case 5: return run_group<OtsukaSTN>(params, G, S, y0, o, rows, nsteps, nrej);
    case 7: return run_group<GatedModel<3>>(params, G, S, y0, o, rows, nsteps, nrej);
    case 8: return run_group<GatedModel<2>>(params, G, S, y0, o, rows, nsteps, nrej);
    case 9: case 10: case 11: return run_group<GatedModel<4>>(params, G, S, y0, o, rows, nsteps, nrej);
    case 12: return run_group<GatedModel<1>>(params, G, S, y0, o, rows, nsteps, nrej);
    }
    return -1;
}

template <int NEURON>
static int run_mech(const BLSParams &p, double f, double A, double phi, double Q, const double *fs,
                    int n_fs, const MechOpts &o, double *zs, double *ngs, double *eff, int *status)
{
    return mech_cell<NEURON>(p, f, A, phi, Q, fs, n_fs, o, zs, ngs, 1, eff, status);
}

// RS only: computeEffVars with charge overtones
extern "C" int harness_mech_overtones(const double *bls9, double f, double A, double phi, double Q,
                                      const double *fs, int n_fs, int n_ov, const double *ovA,
                                      const double *ovphi, double rtol, int max_steps, double *zs,
                                      double *ngs, double *eff, double *ov_out, int *status)
{
    BLSParams p;
    std::memcpy(&p, bls9, sizeof(p));
    MechOpts o{rtol, max_steps, 10};
    return mech_cell<0>(p, f, A, phi, Q, fs, n_fs, o, zs, ngs, 1, eff, status,
                        MechOvertones{n_ov, ovA, ovphi, ov_out});
}

extern "C" int harness_mech(int neuron_id, const double *bls9, double f, double A, double phi, double Q,
                            const double *fs, int n_fs, double rtol, int max_steps,
                            double *zs /* [999] */, double *ngs /* [999] */, double *eff, int *status)
{
    BLSParams p;
    std::memcpy(&p, bls9, sizeof(p));
    MechOpts o{rtol, max_steps, 10};
    switch (neuron_id) {
    case 0: return run_mech<0>(p, f, A, phi, Q, fs, n_fs, o, zs, ngs, eff, status);
    case 1: return run_mech<1>(p, f, A, phi, Q, fs, n_fs, o, zs, ngs, eff, status);
    case 2: return run_mech<2>(p, f, A, phi, Q, fs, n_fs, o, zs, ngs, eff, status);
    case 3: return run_mech<3>(p, f, A, phi, Q, fs, n_fs, o, zs, ngs, eff, status);
    case 4: return run_mech<4>(p, f, A, phi, Q, fs, n_fs, o, zs, ngs, eff, status);
    case 5: return run_mech<5>(p, f, A, phi, Q, fs, n_fs, o, zs, ngs, eff, status);
    case 6: return run_mech<6>(p, f, A, phi, Q, fs, n_fs, o, zs, ngs, eff, status);
    case 7: return run_mech<7>(p, f, A, phi, Q, fs, n_fs, o, zs, ngs, eff, status);
    case 8: return run_mech<8>(p, f, A, phi, Q, fs, n_fs, o, zs, ngs, eff, status);
    case 9: return run_mech<9>(p, f, A, phi, Q, fs, n_fs, o, zs, ngs, eff, status);
    case 10: return run_mech<10>(p, f, A, phi, Q, fs, n_fs, o, zs, ngs, eff, status);
    case 11: return run_mech<11>(p, f, A, phi, Q, fs, n_fs, o, zs, ngs, eff, status);
    case 12: return run_mech<12>(p, f, A, phi, Q, fs, n_fs, o, zs, ngs, eff, status);
    }
    return -1;
}

template <class M, int NEURON>
static void run_full(const FullDev &D, const BLSParams &p, const double *params)
{
    typename M::Params P;
    std::memcpy(&P, params, sizeof(P));
    for (long long c = 0; c < D.n; c++) full_config<M, NEURON>(D, p, P, c);
}

// single configuration, arrays prepared by the caller exactly as full_batch_run does on the host
extern "C" void harness_full(int neuron_id, const double *params, const double *bls9, double f, double A,
                             double fs, double tstop, const double *seg_t0, const double *seg_t1,
                             const double *seg_x, const int *seg_n, int nseg, long long nrows,
                             const double *y0, double rtol, int max_steps, double *traces,
                             int *status, int *nsteps)
{
    BLSParams p;
    std::memcpy(&p, bls9, sizeof(p));
    long long seg_off[2] = {0, nseg}, row_off[2] = {0, nrows};
    // max_steps < 0: |max_steps| - 1 selects the stiff mode (0 explicit only, 1 automatic, 2 RODAS4), default budget
    const int stiff_mode = max_steps < 0 ? -max_steps - 1 : 1;
    FullDev D{&f, &A, &fs, &tstop, seg_t0, seg_t1, seg_x, seg_n, seg_off, row_off, y0, traces,
              status, nsteps, 1, 3.14159265358979323846, FullOpts{rtol, max_steps < 0 ? 0 : max_steps, 0.0, stiff_mode}};
    switch (neuron_id) {
    case 0: run_full<CorticalRSFS, 0>(D, p, params); break;
    case 1: run_full<CorticalRSFS, 1>(D, p, params); break;
    case 2: run_full<CorticalLTS, 2>(D, p, params); break;
    case 3: run_full<ThalamicRE, 3>(D, p, params); break;
    case 4: run_full<ThalamoCortical, 4>(D, p, params); break;
    case 5: run_full<OtsukaSTN, 5>(D, p, params); break;
    case 6: run_full<CorticalLTS, 6>(D, p, params); break;
    case 7: run_full<GatedModel<3>, 7>(D, p, params); break;
    case 8: run_full<GatedModel<2>, 8>(D, p, params); break;
    case 9: run_full<GatedModel<4>, 9>(D, p, params); break;
    case 10: run_full<GatedModel<4>, 10>(D, p, params); break;
    case 11: run_full<GatedModel<4>, 11>(D, p, params); break;
    }
}

// on-device spike tracker, fed with a (t, Qm) trace
extern "C" int harness_spikes(const double *t, const double *q, long n, double *out4, double *cand, int *stack, int cap)
{
    SpikeTracker s;
    s.init(cand, stack, cap);
    for (long i = 0; i < n; i++) s.feed(t[i], q[i]);
    SpikeSummary r = s.finish();
    out4[0] = r.nspikes; out4[1] = r.t_first; out4[2] = r.t_last; out4[3] = r.sum_inv_isi;
    return r.flags;
}


template <class M, int NEURON>
static void run_hybrid(const HybridDev &D, const BLSParams &p, const double *params)
{
    typename M::Params P;
    std::memcpy(&P, params, sizeof(P));
    for (long long c = 0; c < D.n; c++) hybrid_config<M, NEURON>(D, p, P, c);
}

// single configuration of the hybrid scheme; scratch: HYB_SCRATCH_DOUBLES doubles
extern "C" void harness_hybrid(int neuron_id, const double *params, const double *bls9, double f, double A,
                               double fs, double tstop, const double *ev_t, const double *ev_x, int nev,
                               long long nrows, const double *y0, double rtol, int max_steps,
                               double *traces, double *scratch, int *status, int *nsteps, int *ncycles)
{
    BLSParams p;
    std::memcpy(&p, bls9, sizeof(p));
    long long ev_off[2] = {0, nev}, row_off[2] = {0, nrows};
    HybridDev D{&f, &A, &fs, &tstop, ev_t, ev_x, ev_off, row_off, y0, traces, scratch, status, nsteps,
                ncycles, 1, 3.14159265358979323846, FullOpts{rtol, max_steps, 0.0, 0}};
    switch (neuron_id) {
    case 0: run_hybrid<CorticalRSFS, 0>(D, p, params); break;
    case 1: run_hybrid<CorticalRSFS, 1>(D, p, params); break;
    case 2: run_hybrid<CorticalLTS, 2>(D, p, params); break;
    case 3: run_hybrid<ThalamicRE, 3>(D, p, params); break;
    case 4: run_hybrid<ThalamoCortical, 4>(D, p, params); break;
    case 5: run_hybrid<OtsukaSTN, 5>(D, p, params); break;
    case 6: run_hybrid<CorticalLTS, 6>(D, p, params); break;
    case 7: run_hybrid<GatedModel<3>, 7>(D, p, params); break;
    case 8: run_hybrid<GatedModel<2>, 8>(D, p, params); break;
    case 9: run_hybrid<GatedModel<4>, 9>(D, p, params); break;
    case 10: run_hybrid<GatedModel<4>, 10>(D, p, params); break;
    case 11: run_hybrid<GatedModel<4>, 11>(D, p, params); break;
    }
}

extern "C" int harness_hybrid_scratch_doubles(void) { return HYB_SCRATCH_DOUBLES; }

// octet-cooperative detailed model (RS / FS) with the 8-array emulation backend; same arguments as
// harness_full
#include "../../pysonic_amd/csrc/full_coop.hpp"
#include "../../pysonic_amd/csrc/hybrid_coop.hpp"
#include "../../pysonic_amd/csrc/mech_coop.hpp"

// octet-cooperative lookup cell (RS / FS, constant charge); arguments as harness_mech, scratch [4][999]
extern "C" int harness_mech_coop(int neuron_id, const double *bls9, double f, double A, double phi, double Q,
                                 const double *fs, int n_fs, double rtol, int max_steps, double *scratch,
                                 double *eff, int *status)
{
    BLSParams p;
    std::memcpy(&p, bls9, sizeof(p));
    MechOpts o{rtol, max_steps, 10};
    if (neuron_id == 1) return mech_coop_cell<OctOpsHost, 1>(p, f, A, phi, Q, fs, n_fs, o, scratch, eff, status, true);
    return mech_coop_cell<OctOpsHost, 0>(p, f, A, phi, Q, fs, n_fs, o, scratch, eff, status, true);
}

// octet-cooperative hybrid integration (RS / FS); same arguments as harness_hybrid
extern "C" void harness_hybrid_coop(int neuron_id, const double *params, const double *bls9, double f, double A,
                                    double fs, double tstop, const double *ev_t, const double *ev_x, int nev,
                                    long long nrows, const double *y0, double rtol, int max_steps,
                                    double *traces, double *scratch, int *status, int *nsteps, int *ncycles)
{
    BLSParams p;
    std::memcpy(&p, bls9, sizeof(p));
    CorticalParams P;
    std::memcpy(&P, params, sizeof(P));
    long long ev_off[2] = {0, nev}, row_off[2] = {0, nrows};
    HybridDev D{&f, &A, &fs, &tstop, ev_t, ev_x, ev_off, row_off, y0, traces, scratch, status, nsteps,
                ncycles, 1, 3.14159265358979323846, FullOpts{rtol, max_steps, 0.0, 0}};
    hybrid_coop_config<OctOpsHost>(D, p, P, neuron_id, 0, true);
}

// one evaluation of the cooperative right-hand side (full_coop.hpp) at the acoustic pressure pac: y8, dy8 =
// (U, Z, ng, Qm, m, h, n, p); membrane = 0: the mechanical system alone (mech_coop.hpp)
extern "C" int harness_coop_rhs(int neuron_id, const double *params, const double *bls9, double fs, double pac,
                                int membrane, const double *y8, double *dy8)
{
    BLSParams p;
    std::memcpy(&p, bls9, sizeof(p));
    CorticalParams P;
    std::memcpy(&P, params, sizeof(P));
    typedef OctOpsHost O;
    const CoopConsts<O> C = coop_consts<O>(p, P, neuron_id, 0.0);
    const CoopScalars<O> S = coop_scalars<O>(p, fs, 0.0);
    O::V y, dy;
    for (int i = 0; i < OCT; i++) y.v[i] = y8[i];
    bool clamped = false;
    const O::V pterm = O::splat(S.p0r - pac * S.inv_rho);
    dy = membrane ? coop_rhs<O, true>(C, S, y, pterm, clamped) : coop_rhs<O, false>(C, S, y, pterm, clamped);
    for (int i = 0; i < OCT; i++) dy8[i] = dy.v[i];
    return clamped ? 1 : 0;
}

extern "C" void harness_full_coop(int neuron_id, const double *params, const double *bls9, double f, double A,
                                  double fs, double tstop, const double *seg_t0, const double *seg_t1,
                                  const double *seg_x, const int *seg_n, int nseg, long long nrows,
                                  const double *y0, double rtol, int max_steps, double *traces,
                                  int *status, int *nsteps)
{
    BLSParams p;
    std::memcpy(&p, bls9, sizeof(p));
    CorticalParams P;
    std::memcpy(&P, params, sizeof(P));
    long long seg_off[2] = {0, nseg}, row_off[2] = {0, nrows};
    FullDev D{&f, &A, &fs, &tstop, seg_t0, seg_t1, seg_x, seg_n, seg_off, row_off, y0, traces,
              status, nsteps, 1, 3.14159265358979323846, FullOpts{rtol, max_steps, 0.0, 0}};
    if (std::getenv("COOP_METHOD") && std::atoi(std::getenv("COOP_METHOD")) == 5)
        full_coop_config<OctOpsHost, 5>(D, p, P, neuron_id, 0, true);
    else
        full_coop_config<OctOpsHost, 8>(D, p, P, neuron_id, 0, true);
}


// ---- row-cooperative detailed model (full_row.hpp), emulated: one configuration on a row of 16 lanes ----
#include "../../pysonic_amd/csrc/full_row.hpp"

template <class M>
static bool row_setup(int neuron_id, const double *params, typename M::Params &P, LaneSpec *gl, RowLaneSpec *rl)
{
    std::memcpy(&P, params, sizeof(P));
    return GroupModel<M>::lanes(P, gl) && row_lane_specs<M>(neuron_id, gl, rl);
}

// one evaluation of the row right-hand side at the acoustic pressure pac; y, dy in the order of the lane kernel's
// state vector: U, Z, ng, Qm, then the states in reference column order
template <class M>
static int row_rhs_one(int neuron_id, const double *params, const BLSParams &p, double fs, double pac, const double *yin,
                       double *dyout)
{
    typedef GroupOpsHost O;
    typename M::Params P;
    LaneSpec gl[GRP];
    RowLaneSpec rl[GRP];
    if (!row_setup<M>(neuron_id, params, P, gl, rl)) return -1;
    GroupConsts<O> C;
    O::load_consts(gl, C);
    RowConsts<O> R;
    O::load_row_consts(rl, R);
    O::V y = O::splat(0.0);
    for (int i = 0; i < GRP; i++) {
        if (rl[i].v[RR_MU] != 0.0) y.v[i] = yin[0];
        else if (rl[i].col >= 2) y.v[i] = yin[rl[i].col - 1];      // Z = column 2 -> yin[1], ng, Qm, states
    }
    bool clamped = false;
    const O::V dy = row_rhs<O, M>(p, P, C, R, fs, 0.0, y, pac, clamped);
    for (int i = 0; i < GRP; i++) {
        if (rl[i].v[RR_MU] != 0.0) dyout[0] = dy.v[i];
        else if (rl[i].col >= 2) dyout[rl[i].col - 1] = dy.v[i];
    }
    return clamped ? 1 : 0;
}

extern "C" int harness_row_rhs(int neuron_id, const double *params, const double *bls9, double fs, double pac,
                               const double *y, double *dy)
{
    BLSParams p;
    std::memcpy(&p, bls9, sizeof(p));
    switch (neuron_id) {
    case 2: case 6: return row_rhs_one<CorticalLTS>(neuron_id, params, p, fs, pac, y, dy);
    case 3: return row_rhs_one<ThalamicRE>(neuron_id, params, p, fs, pac, y, dy);
    case 4: return row_rhs_one<ThalamoCortical>(neuron_id, params, p, fs, pac, y, dy);
    case 5: return row_rhs_one<OtsukaSTN>(neuron_id, params, p, fs, pac, y, dy);
    case 7: return row_rhs_one<GatedModel<3>>(neuron_id, params, p, fs, pac, y, dy);
    case 8: return row_rhs_one<GatedModel<2>>(neuron_id, params, p, fs, pac, y, dy);
    case 9: case 10: case 11: return row_rhs_one<GatedModel<4>>(neuron_id, params, p, fs, pac, y, dy);
    }
    return -1;
}

// the row Rosenbrock linear algebra against finite differences: with W = c0 I - df/dy (row_rhs_jac + row_factor),
// r = c0 k - (f(y + eps k) - f(y)) / eps for a given direction k, then W x = r (row_solve): x must give k back.
// y, k, x: U, Z, ng, Qm, then the states in reference column order (as harness_row_rhs).
template <class M>
static int row_jac_one(int neuron_id, const double *params, const BLSParams &p, double fs, double pac, const double *yin,
                       const double *kin, double c0, double eps, double *xout)
{
    typedef GroupOpsHost O;
    typename M::Params P;
    LaneSpec gl[GRP];
    RowLaneSpec rl[GRP];
    if (!row_setup<M>(neuron_id, params, P, gl, rl)) return -1;
    GroupConsts<O> C;
    O::load_consts(gl, C);
    RowConsts<O> R;
    O::load_row_consts(rl, R);
    auto to_lanes = [&](const double *v) {
        O::V y = O::splat(0.0);
        for (int i = 0; i < GRP; i++) {
            if (rl[i].v[RR_MU] != 0.0) y.v[i] = v[0];
            else if (rl[i].col >= 2) y.v[i] = v[rl[i].col - 1];
        }
        return y;
    };
    const O::V y = to_lanes(yin), k = to_lanes(kin);
    bool clamped = false;
    const MechDrive d{0.0, pac, -1.5707963267948966};
    RowJac<O, M, true> J;
    const O::V f0 = row_rhs_jac<O, M>(p, P, C, R, fs, 0.0, d, 0.0, y, J, clamped);
    const O::V f1 = row_rhs<O, M>(p, P, C, R, fs, 0.0, O::fma_(O::splat(eps), k, y), pac, clamped);
    O::V r = O::sub(O::mul(O::splat(c0), k), O::mul(O::sub(f1, f0), O::splat(1.0 / eps)));
    row_factor<O, M, true>(C, R, J, c0);
    row_solve<O, M, true>(C, R, J, r, 0.0);
    for (int i = 0; i < GRP; i++) {
        if (rl[i].v[RR_MU] != 0.0) xout[0] = r.v[i];
        else if (rl[i].col >= 2) xout[rl[i].col - 1] = r.v[i];
    }
    return 0;
}

extern "C" int harness_row_jac(int neuron_id, const double *params, const double *bls9, double fs, double pac,
                               const double *y, const double *k, double c0, double eps, double *x)
{
    BLSParams p;
    std::memcpy(&p, bls9, sizeof(p));
    switch (neuron_id) {
    case 2: case 6: return row_jac_one<CorticalLTS>(neuron_id, params, p, fs, pac, y, k, c0, eps, x);
    case 3: return row_jac_one<ThalamicRE>(neuron_id, params, p, fs, pac, y, k, c0, eps, x);
    case 4: return row_jac_one<ThalamoCortical>(neuron_id, params, p, fs, pac, y, k, c0, eps, x);
    case 5: return row_jac_one<OtsukaSTN>(neuron_id, params, p, fs, pac, y, k, c0, eps, x);
    case 7: return row_jac_one<GatedModel<3>>(neuron_id, params, p, fs, pac, y, k, c0, eps, x);
    case 8: return row_jac_one<GatedModel<2>>(neuron_id, params, p, fs, pac, y, k, c0, eps, x);
    case 9: case 10: case 11: return row_jac_one<GatedModel<4>>(neuron_id, params, p, fs, pac, y, k, c0, eps, x);
    }
    return -1;
}

template <class M>
static void run_full_row(int neuron_id, const FullDev &D, const BLSParams &p, const double *params)
{
    typename M::Params P;
    LaneSpec gl[GRP];
    RowLaneSpec rl[GRP];
    if (!row_setup<M>(neuron_id, params, P, gl, rl)) { D.status[0] = -1; return; }
    full_row_config<GroupOpsHost, M, 1>(D, p, P, gl, rl, 0, true);
}

// single configuration, arguments as harness_full
extern "C" void harness_full_row(int neuron_id, const double *params, const double *bls9, double f, double A,
                                 double fs, double tstop, const double *seg_t0, const double *seg_t1,
                                 const double *seg_x, const int *seg_n, int nseg, long long nrows,
                                 const double *y0, double rtol, int max_steps, double *traces,
                                 int *status, int *nsteps)
{
    BLSParams p;
    std::memcpy(&p, bls9, sizeof(p));
    long long seg_off[2] = {0, nseg}, row_off[2] = {0, nrows};
    // max_steps < 0: |max_steps| - 1 selects the stiff mode (0 explicit only, 1 automatic, 2 RODAS4), default budget;
    // ROW_RTOL_STIFF (environment): tolerance of the RODAS4 path
    const int stiff_mode = max_steps < 0 ? -max_steps - 1 : 0;
    FullOpts fo{rtol, max_steps < 0 ? 0 : max_steps, 0.0, stiff_mode};
    if (std::getenv("ROW_RTOL_STIFF")) fo.rtol_stiff = std::atof(std::getenv("ROW_RTOL_STIFF"));
    FullDev D{&f, &A, &fs, &tstop, seg_t0, seg_t1, seg_x, seg_n, seg_off, row_off, y0, traces,
              status, nsteps, 1, 3.14159265358979323846, fo};
    switch (neuron_id) {
    case 2: case 6: run_full_row<CorticalLTS>(neuron_id, D, p, params); break;
    case 3: run_full_row<ThalamicRE>(neuron_id, D, p, params); break;
    case 4: run_full_row<ThalamoCortical>(neuron_id, D, p, params); break;
    case 5: run_full_row<OtsukaSTN>(neuron_id, D, p, params); break;
    case 7: run_full_row<GatedModel<3>>(neuron_id, D, p, params); break;
    case 8: run_full_row<GatedModel<2>>(neuron_id, D, p, params); break;
    case 9: case 10: case 11: run_full_row<GatedModel<4>>(neuron_id, D, p, params); break;
    default: *status = -1;
    }
}

// ---- row-cooperative hybrid scheme (hybrid_row.hpp), emulated; arguments as harness_hybrid ----
#include "../../pysonic_amd/csrc/hybrid_row.hpp"

template <class M>
static void run_hybrid_row(int neuron_id, const HybridDev &D, const BLSParams &p, const double *params)
{
    typename M::Params P;
    LaneSpec gl[GRP];
    RowLaneSpec rl[GRP];
    if (!row_setup<M>(neuron_id, params, P, gl, rl)) { D.status[0] = -1; return; }
    if (D.opts.stiff_mode == 2) hybrid_row_config<GroupOpsHost, M, true>(D, p, P, gl, rl, 0, true);
    else hybrid_row_config<GroupOpsHost, M, false>(D, p, P, gl, rl, 0, true);
}

extern "C" void harness_hybrid_row(int neuron_id, const double *params, const double *bls9, double f, double A,
                                   double fs, double tstop, const double *ev_t, const double *ev_x, int nev,
                                   long long nrows, const double *y0, double rtol, int max_steps,
                                   double *traces, double *scratch, int *status, int *nsteps, int *ncycles)
{
    BLSParams p;
    std::memcpy(&p, bls9, sizeof(p));
    long long ev_off[2] = {0, nev}, row_off[2] = {0, nrows};
    HybridDev D{&f, &A, &fs, &tstop, ev_t, ev_x, ev_off, row_off, y0, traces, scratch, status, nsteps,
                ncycles, 1, 3.14159265358979323846, FullOpts{rtol, max_steps < 0 ? -max_steps : max_steps, 0.0, max_steps < 0 ? 2 : 0}};
    D.opts.rtol_stiff = 30.0 * rtol;                  // (max_steps < 0: the build with RODAS4 dense periods)
    switch (neuron_id) {
    case 2: case 6: run_hybrid_row<CorticalLTS>(neuron_id, D, p, params); break;
    case 3: run_hybrid_row<ThalamicRE>(neuron_id, D, p, params); break;
    case 4: run_hybrid_row<ThalamoCortical>(neuron_id, D, p, params); break;
    case 5: run_hybrid_row<OtsukaSTN>(neuron_id, D, p, params); break;
    case 7: run_hybrid_row<GatedModel<3>>(neuron_id, D, p, params); break;
    case 8: run_hybrid_row<GatedModel<2>>(neuron_id, D, p, params); break;
    case 9: case 10: case 11: run_hybrid_row<GatedModel<4>>(neuron_id, D, p, params); break;
    default: *status = -1;
    }
}
