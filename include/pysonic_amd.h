/* include/pysonic_amd.h -- C ABI of libpysonic_amd.so (MI355X / gfx950 HIP kernels).
 *
 * The reference (tjjlemaire/PySONIC) is pure Python and has no FFI: its data-parallel seam is
 *   Batch(func, queue).run(mpi=True)          PySONIC/core/batches.py:135-153
 * with func in { nbls.simulate, nbls.computeEffVars } executing one configuration per worker
 * process. These entry points replace what ONE WHOLE QUEUE of such calls computes:
 *
 *   sonic_*   <- NeuronalBilayerSonophore.simulate(method='sonic')      nbls.py:389-437, 513-536
 *                = EventDrivenSolver.solve over scipy odeint            solvers.py:150-170,445-480
 *                  of NBLS.effDerivatives                               nbls.py:280-315
 *   mech_*    <- NeuronalBilayerSonophore.computeEffVars                nbls.py:153-222
 *                = BilayerSonophore.simCycles / PeriodicSolver          bls.py:749-789,
 *                                                                       solvers.py:224-365
 *
 * Conventions: plain pointers + sizes, float64 everywhere, caller owns every host buffer,
 * the library owns device memory behind the opaque handles. Every function returns 0 on success
 * or a negative SONIC_E* code; sonic_last_error() gives the message (thread-local).
 * Host code binds this with ctypes (pysonic_amd/_native.py); see INTEGRATION.md.
 */
#ifndef PYSONIC_AMD_H
#define PYSONIC_AMD_H

#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SONIC_ABI_VERSION 6

/* error codes */
#define SONIC_OK 0
#define SONIC_EINVAL (-1)   /* bad argument                                                   */
#define SONIC_ERANGE (-2)   /* amplitude outside the lookup's A range (ValueError in the      */
                            /* reference: utils.py:321-348 via lookups.py:245-247)            */
#define SONIC_EHIP (-3)     /* HIP runtime error                                              */
#define SONIC_ENODEV (-4)   /* no usable GPU                                                  */

/* neuron ids (PySONIC/neurons: cortical.py, thalamic.py, stn.py) */
#define SONIC_NEURON_RS 0
#define SONIC_NEURON_FS 1
#define SONIC_NEURON_LTS 2
#define SONIC_NEURON_RE 3
#define SONIC_NEURON_TC 4
#define SONIC_NEURON_STN 5
#define SONIC_NEURON_IB 6      /* cortical intrinsically bursting: the six-gate cortical model of LTS */
#define SONIC_NEURON_HH 7      /* Hodgkin-Huxley segment, Sweeney node, MRG node, Sundt segment: data-driven gated model, */
#define SONIC_NEURON_SW 8      /* parameter block = gLeak, ELeak, g[4], E[4], ghk[4], Cin[4], Cout[4], exponents [4][n] */
#define SONIC_NEURON_MRG 9
#define SONIC_NEURON_SU 10     /* Sundt segment: same model */
#define SONIC_NEURON_FH 11     /* Frankenhaeuser-Huxley node: same model, Goldman-Hodgkin-Katz driving forces */
#define SONIC_NEURON_PAS 12    /* passive neuron (pas.py): the same model with ONE padding gate (rates and
                                  conductances 0) whose column the host strips */

/* per-configuration status bits written by sonic_batch_* */
#define SONIC_ST_Q_OUT_OF_RANGE 1  /* Qm left the lookup charge range: NaN rows, as np.interp's
                                      left/right = nan does in the reference (lookups.py:322)   */
#define SONIC_ST_STEP_UNDERFLOW 2  /* integrator step underflow                               */
#define SONIC_ST_MAX_STEPS 4       /* step budget exhausted                                    */

typedef struct sonic_model sonic_model_t;
typedef struct sonic_batch sonic_batch_t;

/* Integrator options. The device integrator is an adaptive Rosenbrock method of order 4(3) -- RODAS4,
 * or ROS4 with Shampine's parameters in the RS / FS kernel -- (it replaces LSODA); rtol/atol play the role of odeint's rtol/atol (solvers.py:167 uses scipy defaults). */
typedef struct {
    double rtol;       /* default 0 = the kernel's own: 4e-6 for the RS / FS kernel, 1e-6 for the others  */
    double atol;       /* default 0 = 1e-8 */
    double h0;         /* initial step at every segment start (s), default 1e-6 */
    double hmin;       /* step underflow threshold (s), default 1e-30 (see SolverOpts) */
    int max_steps;     /* per-configuration step budget, default 20 000 000 */
    int write_traces;  /* 1: write the full time series; 0: metrics only */
    int qss_mask;      /* bit k set: the k-th state (PointNeuron.statesNames() order) is a
                          quasi-steady-state variable, x = alpha / (alpha + beta) at the current
                          charge instead of a differential one (qss_vars of nbls.py:280-315,
                          389-437). Voltage-gated states only. Default 0. The output column of
                          such a state is not meaningful: the host fills it from the lookup like
                          the reference does after the integration (nbls.py:429-430). */
    double idrive;     /* constant injected current (mA/m2) of DrivenNeuronalBilayerSonophore:
                          dQm/dt += idrive 1e-3 (nbls.py:717-721). Default 0. */
    int chunks;        /* > 1 (with write_traces): pipelined batch. The wavefronts of the batch, in order of
                          descending estimated cost, are cut into this many launches of about equal numbers of
                          rows, each on a stream of its own, and the rows are laid out in that order
                          (sonic_batch_row_blocks) so that a chunk's rows are one contiguous range: with
                          sonic_batch_launch_to_host they travel to the host as its kernel ends, while the costly
                          configurations still integrate -- the counterpart of the reference's pool handing
                          results back as workers finish (batches.py:108-128). Default 0: rows in queue order,
                          one launch. At most 3 launches (the hardware queues
                          of a process). */
} sonic_opts_t;

/* metrics row layout ([n_cfg][SONIC_NMETRICS] float64) */
#define SONIC_NMETRICS 16
#define SONIC_M_NSTEPS 0     /* accepted + rejected step attempts */
#define SONIC_M_NREJ 1       /* rejected step attempts */
#define SONIC_M_NROWS 2      /* rows written */
#define SONIC_M_QMIN 3       /* min of Qm over the output rows (C/m2) */
#define SONIC_M_QMAX 4       /* max of Qm over the output rows (C/m2) */
#define SONIC_M_QLAST 5      /* Qm of the last row */
/* spike metrics = detectSpikes (postpro.py:263-284) evaluated on the device while rows are
 * produced: peaks of Qm with height >= 3e-5 and prominence >= 20e-5 C/m2 */
#define SONIC_M_NSPIKES 6    /* number of spikes */
#define SONIC_M_TFIRST 7     /* time of the first spike (s) = latency; NaN if none */
#define SONIC_M_TLAST 8      /* time of the last spike (s) */
#define SONIC_M_SUMINVISI 9  /* sum of 1 / inter-spike interval (Hz): mean FR = this / (n - 1),
                                FiringRateMap.xfunc (plt/actmap.py:119-127) */
#define SONIC_M_SPKFLAGS 10  /* 1: candidate buffer overflow; 2: two spikes closer than 0.5 ms
                                (the reference's distance rule would apply: re-check on host) */
#define SONIC_M_RESERVED 11  /* diagnostics, not a result: where the wavefront ran (quad kernel:
                                HW_ID + XCC_ID << 32) or 0 */
/* what set the steps (the rest of NSTEPS were sized by the error controller or ended a segment) */
#define SONIC_M_NCAPPED 12   /* accepted steps whose size was the node predictor's: they end just past a node of
                                the charge grid instead of where the error controller would have gone */
#define SONIC_M_NREJ_NODE 13 /* attempts rejected because they ended too far past the node (the other NREJ - this
                                were rejected by the error estimate) */
#define SONIC_M_NCROSS 14    /* cells of the charge grid crossed: a lower bound of the steps of this scheme */
#define SONIC_M_SPARE 15

int sonic_abi_version(void);
int sonic_device_count(void);
const char *sonic_last_error(void);
void sonic_default_opts(sonic_opts_t *opts);

/* Number of state variables (excluding Qm) / lookup tables (including V) / parameters of a
 * neuron model; negative on unknown id. Output columns are: t, stimstate, Qm, states..., Vm
 * i.e. n_states + 4 (the reference's Z / ng NaN columns are added by the host, nbls.py:432-434). */
int sonic_neuron_nstates(int neuron_id);
int sonic_neuron_ntables(int neuron_id);
int sonic_neuron_nparams(int neuron_id);

/* Create a model on `device`: uploads nothing yet, keeps a host copy of the 2-D lookup
 * (the (A, Q) projection of the reference's 5-D lookup at fixed a, f, fs: nbls.py:254-263).
 *   params   [n_params]            neuron parameters, order documented in pysonic_amd/neurons
 *   tables   [n_tab][n_A][n_Q]     row-major; table 0 = 'V', then effRates() order
 *   A_grid   [n_A] ascending (Pa); Q_grid [n_Q] ascending (C/m2) */
int sonic_model_create(int device, int neuron_id, const double *params, int n_params,
                       const double *tables, const double *A_grid, int n_A,
                       const double *Q_grid, int n_Q, int n_tab, sonic_model_t **out);
void sonic_model_destroy(sonic_model_t *m);

/* Number of output rows of each configuration, as EventDrivenSolver produces them
 * (solvers.py:77-127, 445-480): 1 + sum over segments of
 * max(round((t_e - t_now)/dt), 2), segments ending at each sorted event and at tstop.
 *   ev_t / ev_off: CSR event times, events of config i are [ev_off[i], ev_off[i+1]) sorted by t */
int sonic_count_rows(const double *tstop, const double *dt, const double *ev_t,
                     const long long *ev_off, long long n_cfg, long long *n_rows);

/* Prepare a batch: build segment schedules, project the lookup at every distinct amplitude
 * A[i]*ev_x[j] (plus 0), upload everything, allocate the device outputs.
 *   A      [n_cfg]  drive amplitude (Pa)                       AcousticDrive.A, drives.py:191-304
 *   tstop  [n_cfg]  protocol stop time (s)                     protocols.py:297-299
 *   dt     [n_cfg]  output time step (s)                       pneuron.py:481-483
 *   ev_t, ev_x, ev_off: CSR (time, modulation factor) events   protocols.py:386-391
 *   y0     [1 + n_states] initial conditions (Qm0, x_inf(Vm0)) nbls.py:408-411 */
int sonic_batch_prepare(sonic_model_t *m, const double *A, const double *tstop, const double *dt,
                        const double *ev_t, const double *ev_x, const long long *ev_off,
                        long long n_cfg, const double *y0, const sonic_opts_t *opts,
                        sonic_batch_t **out);
/* Total rows / per-config row offsets ([n_cfg + 1]) of a prepared batch. */
long long sonic_batch_total_rows(const sonic_batch_t *b);
int sonic_batch_row_offsets(const sonic_batch_t *b, long long *row_off);
/* Rows of configuration i: [row_start[i], row_start[i] + n_rows[i]) of the trace block ([n_cfg] each, either may
 * be NULL). Equal to the row offsets above unless the batch is pipelined (opts.chunks > 1), whose rows follow the
 * order in which the kernels work through the configurations; sonic_batch_row_offsets refuses such a batch. */
int sonic_batch_row_blocks(const sonic_batch_t *b, long long *row_start, long long *n_rows);
/* Number of launches of a pipelined batch (0: not pipelined). */
int sonic_batch_n_chunks(const sonic_batch_t *b);
/* The tolerances the batch integrates with (the kernel's own where the options left them at 0). */
int sonic_batch_tolerances(const sonic_batch_t *b, double *rtol, double *atol);
/* Launch the integration kernel(s) on the batch's stream(s) (asynchronous). */
int sonic_batch_launch(sonic_batch_t *b);
/* The same, and every launch is followed, on its stream, by the copy of its rows to host_traces
 * ([total_rows][n_states + 4], the layout of sonic_batch_row_blocks; page-locked memory of sonic_host_alloc for
 * the copies to overlap the kernels still running). sonic_batch_sync then waits for kernels AND copies;
 * metrics and status are fetched afterwards with sonic_batch_fetch(b, NULL, metrics, status). */
int sonic_batch_launch_to_host(sonic_batch_t *b, double *host_traces);
/* Pipelined batch, after completion: per launch, the duration of its kernel and the time from the first launch's
 * start to its end (ms; [sonic_batch_n_chunks] each, either may be NULL). */
int sonic_batch_chunk_times(sonic_batch_t *b, float *kernel_ms, float *done_ms);
/* Device memory of destroyed batches is kept for the next ones (a sweep re-allocates the same buffers again and
 * again): this gives the idle blocks back to the driver. */
int sonic_release_device_memory(void);
/* Wait for completion; *kernel_ms (may be NULL) receives the HIP-event duration of the last
 * launch's kernel on the batch's stream (pipelined batch: from the start of its first kernel to the end of
 * the last one to finish). */
int sonic_batch_sync(sonic_batch_t *b, float *kernel_ms);
/* Copy results to host buffers (any may be NULL):
 *   traces  [total_rows][n_states + 4]   metrics [n_cfg][SONIC_NMETRICS]   status [n_cfg] */
int sonic_batch_fetch(sonic_batch_t *b, double *traces, double *metrics, int *status);
/* The same with the rows of `traces` `row_stride` doubles apart (>= n_states + 4): the caller keeps extra
 * columns after the device's -- the reference appends the NaN columns Z and ng to the table of an effective
 * simulation (nbls.py:432-434) -- and fills them itself; the columns beyond n_states + 4 are not written. */
int sonic_batch_fetch_strided(sonic_batch_t *b, double *traces, long long row_stride, double *metrics,
                              int *status);
/* The same, and the columns beyond n_states + 4 are set to NaN: the padded table is assembled on the device and
 * copied in one contiguous transfer (fastest into a buffer of sonic_host_alloc). */
int sonic_batch_fetch_padded(sonic_batch_t *b, double *traces, long long row_stride, double *metrics,
                             int *status);
/* Page-locked host memory for the outputs (hipHostMalloc / hipHostFree): a transfer into it runs at the speed of
 * the link, without the staging copy and the page faults of a fresh pageable buffer. Optional: every fetch
 * function takes any host pointer. */
int sonic_host_alloc(size_t bytes, void **out);
int sonic_host_free(void *p);
/* Device addresses of the batch outputs (HBM-resident; valid until sonic_batch_destroy), for
 * consumers that stay on the GPU, e.g. an RCCL all-gather of the metric rows. Any may be NULL. */
int sonic_batch_device_ptrs(sonic_batch_t *b, void **traces, void **metrics, void **status);
void sonic_batch_destroy(sonic_batch_t *b);

/* One-shot convenience: prepare + launch + sync + fetch + destroy. */
int sonic_batch_run(sonic_model_t *m, const double *A, const double *tstop, const double *dt,
                    const double *ev_t, const double *ev_x, const long long *ev_off,
                    long long n_cfg, const double *y0, const sonic_opts_t *opts,
                    double *traces, double *metrics, int *status);

/* ---------------------------------------------------------------------------------------------
 * Mechanical lookup generation: NeuronalBilayerSonophore.computeEffVars (nbls.py:153-222) for a
 * whole queue of (drive, fs, Qm) items (scripts/run_lookups.py:99-148), one cell per GPU lane.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    double rtol;       /* relative tolerance of the Dormand-Prince 8(5,3) steps, default 1e-9  */
    int max_steps;     /* per-cell step budget, default 50 000 000                              */
    int ncycles_max;   /* NCYCLES_MAX (constants.py:34), default 10 => at most 11 cycles in all  */
    double phi;        /* drive phase (rad), default pi (drives.py:199)                         */
} mech_opts_t;

/* per-cell status bits */
#define MECH_ST_Z_CLAMPED 1      /* Z clamped at Zmin (bls.py:694-697 logs a warning)            */
#define MECH_ST_NO_QS_ROOT 2     /* quasi-static pressure does not change sign (ValueError)     */
#define MECH_ST_MAX_STEPS 4
#define MECH_ST_NOT_CONVERGED 8  /* stopped after ncycles_max + 1 cycles (solvers.py:361-365)    */

void mech_default_opts(mech_opts_t *opts);
/* number of effective rates of a neuron (effRates() of the reference); output rows hold
 * 1 + n_rates values: 'V' then the rates in lookup order */
int mech_neuron_nrates(int neuron_id);
/*   bls_params [9]: a, Cm0, Delta_eq, LJ x0, C, nrep, nattr, kA_tissue, ng0   (bls.py:115-137,44-77)
 *   f, A, Q    [n]: drive frequency (Hz), amplitude (Pa), imposed charge (C/m2) of every cell
 *   fs         [n_fs]: sonophore coverage fractions (nbls.py:148-151)
 *   effvars    [n][n_fs][1 + n_rates]; ncycles, status [n] (may be NULL); kernel_ms may be NULL */
int mech_batch_run(int device, int neuron_id, const double *bls_params, int n_bls_params,
                   const double *f, const double *A, const double *Q, long long n,
                   const double *fs, int n_fs, const mech_opts_t *opts, double *effvars,
                   int *ncycles, int *status, float *kernel_ms);
/* The same with Fourier overtones of the imposed charge: computeEffVars(drive, fs, Qm0,
 * Qm_overtones=[(A_1, phi_1), ...]) (nbls.py:153-222 with 169-178 and 194-201; the charge profile
 * imposed on BilayerSonophore.simCycles, bls.py:767-769).
 *   ov_A, ov_phi [n][n_overtones]: amplitude (C/m2) and phase (rad) of the overtones of every cell
 *   ov_out       [n][n_fs][2 n_overtones]: A_V1, phi_V1, A_V2, phi_V2 ... of the membrane potential */
int mech_batch_run_overtones(int device, int neuron_id, const double *bls_params, int n_bls_params,
                             const double *f, const double *A, const double *Q, long long n,
                             const double *fs, int n_fs, int n_overtones, const double *ov_A,
                             const double *ov_phi, const mech_opts_t *opts, double *effvars,
                             double *ov_out, int *ncycles, int *status, float *kernel_ms);

/* ---------------------------------------------------------------------------------------------
 * Detailed NICE model: NeuronalBilayerSonophore.simulate(method='full') (nbls.py:331-354) for a
 * queue of configurations sharing one sonophore and one neuron. Output rows are on the
 * reference's resampled grid np.linspace(0, tstop, round(tstop / target_dt)) with columns
 * t, stimstate, Z, ng, Qm, states..., Vm  (n_states + 6; 'U' is dropped as nbls.py:349 does).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    double rtol;       /* relative tolerance; 0 (default): 1e-8 for the 5(4) pair and the row kernel, 1e-7 for the 8(5,3)
                          pair of the RS / FS kernel
                          (hybrid_batch_run on the cooperative kernel: 5e-8) */
    int max_steps;     /* per-configuration step budget; 0 (default): 400 x dense points + 1e5 */
    double target_dt;  /* output resampling step (s), default CLASSIC_TARGET_DT = 1e-8         */
    double phi;        /* drive phase (rad), default pi                                        */
    double idrive;     /* injected current (mA/m2) of DrivenNeuronalBilayerSonophore: added to
                          dQm/dt of the detailed system (nbls.py:712-715), default 0. The sparse
                          phase of the hybrid scheme integrates pneuron.derivatives, without it. */
    int kernel;        /* full_batch_run: 0 (default) the cooperative 8(5,3) kernel where there is one (RS, FS:
                          one configuration per 8 lanes, csrc/full_coop.hpp; LTS, IB, RE, TC, STN, HHseg, SWnode, MRGnode, SUseg, FHnode: one per row
                          of 16 lanes, csrc/full_row.hpp), 1 one configuration
                          per lane for every neuron (5(4) pair), 2 cooperative 8(5,3) or SONIC_EINVAL,
                          3 cooperative 5(4) (RS, FS) or SONIC_EINVAL.
                          hybrid_batch_run: 0 (default) the cooperative 8(5,3) kernel where there is one (RS, FS:
                          csrc/hybrid_coop.hpp; LTS, IB, RE, TC, STN, HHseg, SWnode, MRGnode, SUseg, FHnode: csrc/hybrid_row.hpp), 1 one configuration
                          per lane (5(4) pair), 2 cooperative or SONIC_EINVAL                    */
    int stiff;         /* lane-per-configuration kernel of full_batch_run (the reference: LSODA's switch to BDF,
                          solvers.py:162-167): 1 (default) explicit 5(4) pair, handing a configuration over to
                          RODAS4 on the whole system once its steps are limited by stability (gates with rate
                          constants of 1e10 - 1e23 /s: STN above ~450 kPa, SUseg); 0 explicit pair only;
                          2 RODAS4 from the start. The row kernel does the same on its own Rosenbrock path (at 30 x
                          rtol); with stiff = 0 it gives such a configuration up with status bit 64.
                          hybrid_batch_run: the dense periods of the row kernel, likewise (the lane and octet
                          kernels integrate them explicitly whatever this says) */
} full_opts_t;

void full_default_opts(full_opts_t *opts);
int full_count_rows(const double *tstop, long long n_cfg, double target_dt, long long *n_rows);
/*   neuron_params: as sonic_model_create; bls_params [9]: as mech_batch_run
 *   f, A, fs, tstop [n_cfg]; events CSR as sonic_batch_prepare; y0 [1 + n_states]
 *   traces [sum n_rows][n_states + 6] (row counts from full_count_rows); status bits as mech_*;
 *   nsteps [n_cfg] integrator step attempts (may be NULL) */
int full_batch_run(int device, int neuron_id, const double *neuron_params, int n_params,
                   const double *bls_params, int n_bls_params, const double *f, const double *A,
                   const double *fs, const double *tstop, const double *ev_t, const double *ev_x,
                   const long long *ev_off, long long n_cfg, const double *y0,
                   const full_opts_t *opts, double *traces, int *status, int *nsteps,
                   float *kernel_ms);

/* ---------------------------------------------------------------------------------------------
 * Hybrid integration: NeuronalBilayerSonophore.simulate(method='hybrid') (nbls.py:356-387) =
 * HybridSolver.solve (solvers.py:483-633). Per interval of HYBRID_UPDATE_INTERVAL (or up to the
 * next event): whole acoustic periods of the detailed model until Z and ng are periodically stable
 * (PeriodicSolver, solvers.py:317-365), then (Qm, states) alone with U, Z, ng replayed from the
 * last period at 40 samples per period. Arguments, row layout, resampling (target_dt) and status
 * bits as full_batch_run; additional status bits:
 *   16  an interval shorter than two periods needs a dense phase (the reference asserts)
 *   32  a sparse phase without a complete dense period before it
 * ncycles [n_cfg] (may be NULL): dense periods integrated per configuration.
 * ------------------------------------------------------------------------------------------- */
int hybrid_batch_run(int device, int neuron_id, const double *neuron_params, int n_params,
                     const double *bls_params, int n_bls_params, const double *f, const double *A,
                     const double *fs, const double *tstop, const double *ev_t, const double *ev_x,
                     const long long *ev_off, long long n_cfg, const double *y0,
                     const full_opts_t *opts, double *traces, int *status, int *nsteps,
                     int *ncycles, float *kernel_ms);

#ifdef __cplusplus
}
#endif
#endif /* PYSONIC_AMD_H */
