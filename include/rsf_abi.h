/*
 * rsf_abi.h — C ABI of the MCMC-over-ODE hot path (rate-and-state friction).
 *
 * This is the drop-in boundary.  The reference (pure Python, no FFI) exposes the
 * path only through three classes; each entry point below names the reference
 * code it stands in for (paths relative to the reference tree):
 *
 *   RateStateModel.__init__/.evaluate   RateStateModel.py:109-186, 188-395
 *   RHS friction(t, y)                  RateStateModel.py:277-355
 *   MCMC.compute_initial_covariance     MCMC.py:206-266
 *   MCMC.SSqcalc / acceptreject         MCMC.py:335-389, 268-333
 *   MCMC.update_standard_deviation      MCMC.py:129-160
 *   MCMC.update_covariance_matrix       MCMC.py:162-204
 *   MCMC.sample (hot loop)              MCMC.py:391-544
 *   RSF.plot_dist (KDE of the samples)  RSF.py:717-746
 *
 * Two shared libraries implement this same header:
 *   - librsf_hip.so     (the product: hand-written gfx950 HIP kernels)
 *   - librsf_oracle.so  (oracle/: plain-C CPU restatement; TEST INFRASTRUCTURE ONLY)
 *
 * Conventions
 *   - plain C symbols, no C++ types, no exceptions across the boundary;
 *   - every function returns int status (RSF_OK == 0, negative == error);
 *     rsf_last_error() returns a thread-local message for the last failure;
 *   - all array arguments are caller-allocated and caller-owned; ctx->mem_space
 *     says whether they are host pointers (the shim stages them) or device
 *     pointers (e.g. torch.Tensor.data_ptr(), hipMalloc) used in place;
 *   - all floating point is IEEE float64;
 *   - a ctx is single-owner: one host thread at a time; distinct ctxs are
 *     independent; work is launched on the ctx stream and every call that
 *     returns results to HOST memory is synchronous on return; with DEVICE
 *     memory calls are stream-ordered and rsf_sync() waits for them.
 *
 * Parameter vector of a chain: q[0] = Dc (the reference's only parameter,
 * MCMC.py:381); for d == 3 (extension, BASELINE config 5) q = (Dc, a, b).
 */
#ifndef RSF_ABI_H
#define RSF_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RSF_ABI_VERSION 1

/* status codes */
#define RSF_OK 0
#define RSF_ERR_INVALID (-1)     /* bad argument */
#define RSF_ERR_DEVICE (-2)      /* HIP runtime failure / no device */
#define RSF_ERR_STATE (-3)       /* call order (model or chains not initialised) */
#define RSF_ERR_NOMEM (-4)
#define RSF_ERR_UNSUPPORTED (-5)
#define RSF_ERR_NOT_POSDEF (-6)  /* rsf_mcmc_adapt: the window's covariance has no Cholesky factor (np.linalg.LinAlgError) */

/* where caller buffers live */
#define RSF_MEM_HOST 0
#define RSF_MEM_DEVICE 1

/* rsf_model.flags */
#define RSF_FLAG_RADIATION_DAMPING 1u /* RateStateModel.RadiationDamping, RateStateModel.py:183 */
#define RSF_FLAG_DOP853 4u            /* integrate exactly like the reference: Hairer's DOP853, rtol 1e-6, atol 1e-10, one
                                         call per output interval with the step size carried between calls
                                         (RateStateModel.py:374-389 through scipy.integrate.ode); `substeps` is ignored */
#define RSF_FLAG_FP32_SOLVE 2u        /* integrate the ODE in float32 (BASELINE config 5 tolerance sweep); all
                                         interface arrays, the SSq accumulator and the sampler logic stay float64 */

/* rsf_mcmc_config.adapt_mode (MCMC.py:162-204, 523-527) */
#define RSF_ADAPT_NONE 0           /* list prior: adaptation raises and is swallowed => never adapts */
#define RSF_ADAPT_REFERENCE_DICT 1 /* dict prior: V <- chol(2.38^2/prior_len * np.cov(window)), then used AS covariance; np.cov in
                                      NumPy's own arithmetic (pairwise-summed mean), so that a window of identical samples
                                      fails or "succeeds" with a collapsed proposal exactly where the reference's does */
#define RSF_ADAPT_AM 2             /* corrected adaptive Metropolis (Haario et al. 2001): every adapt_interval iterations
                                      V <- 2.38^2/d * (cov(every sample since rsf_mcmc_init) + diag((1e-6 (hi - lo))^2)) */

#define RSF_MAX_PARAMS 3
#define RSF_DICT_MAX_INTERVAL 128 /* reference_dict adaptation keeps its window's samples: adapt_interval at most this (default 10) */

typedef struct rsf_ctx rsf_ctx;

typedef struct rsf_config {
  uint32_t size;          /* = sizeof(rsf_config) */
  uint32_t version;       /* = RSF_ABI_VERSION */
  int32_t device;         /* HIP device ordinal; -1 = current device */
  int32_t mem_space;      /* RSF_MEM_HOST or RSF_MEM_DEVICE for every array argument */
  void *stream;           /* hipStream_t to launch on; NULL = the device's default stream */
  uint32_t block_threads; /* workgroup size (multiple of 64); 0 = default */
  uint32_t cpu_threads;   /* oracle library only: OpenMP threads, 0 = all */
} rsf_config;

/* Mirrors the attributes of RateStateModel (RateStateModel.py:167-184). */
typedef struct rsf_model {
  uint32_t size;    /* = sizeof(rsf_model) */
  uint32_t flags;   /* RSF_FLAG_* */
  int32_t nsteps;   /* num_tsteps (number_time_steps) */
  int32_t substeps; /* RK4 steps per output interval delta_t (>= 1) */
  double t_start;
  double t_final;
  double mu_ref;
  double V_ref;
  double k1;
  double mu_t_zero;
  double a; /* used where no per-chain a is given */
  double b; /* used where no per-chain b is given */
} rsf_model;

typedef struct rsf_mcmc_config {
  uint32_t size;          /* = sizeof(rsf_mcmc_config) */
  int32_t n_params;       /* d: 1 (Dc) or 3 (Dc, a, b) */
  int64_t n_chains;       /* C: chains owned by this ctx */
  int64_t chain_offset;   /* global id of local chain 0 (multi-GPU sharding; keys the RNG) */
  uint64_t seed;          /* Philox4x32-10 key */
  double n0;              /* MCMC.n0 = 0.01, MCMC.py:97 */
  int32_t prior_len;      /* len(qpriors): the std2[0] divisor nout - len(qpriors), MCMC.py:261 (3 list / 2 dict; 0 => d),
                             and reference_dict's scale 2.38^2/len(qpriors.keys()), MCMC.py:200 (0 => 2) */
  int32_t adapt_mode;     /* RSF_ADAPT_* */
  int32_t adapt_interval; /* MCMC.adapt_interval, default 10 */
  int32_t n_groups;       /* observation series: 0/1 = one shared by all chains; G > 1 = data is [G][nout] and
                             chain i uses series i / (n_chains/G)  (RSF's dc_list sweep, RSF.py:874-882, in one launch) */
  double fd_rel_step;     /* 1e-6, MCMC.py:251 */
  double lo[RSF_MAX_PARAMS]; /* strict box prior, MCMC.py:318-320 */
  double hi[RSF_MAX_PARAMS];
} rsf_mcmc_config;

/* ---- library level ---------------------------------------------------------------- */
int rsf_version(void);
const char *rsf_backend(void);    /* "hip-gfx950" | "oracle-cpu" */
const char *rsf_build_id(void);   /* HIP library: first 16 hex digits of the SHA-256 of the kernel sources it was compiled from
                                     (csrc/Makefile) — what stored profiler evidence is keyed by (profiles/pmc_traffic.json,
                                     bench.py); the checker: "oracle" */
const char *rsf_last_error(void); /* thread-local, never NULL */
int rsf_device_count(void);       /* >= 0, or negative status */

int rsf_create(const rsf_config *cfg, rsf_ctx **out);
int rsf_destroy(rsf_ctx *ctx);
int rsf_sync(rsf_ctx *ctx);

/* ---- forward model: RateStateModel (RateStateModel.py:109-395) -------------------- */

/* Stores the model attributes and builds the chain-independent loading table
 * V_l(t) = V_ref (1 + exp(-t/20) sin(10 t)) (RateStateModel.py:327-329) at the
 * 2*substeps*(nout-1)+1 RK4 stage times.  Discards the chains of an earlier rsf_mcmc_init (their SSq,
 * sigma^2 and covariance belong to the previous model). */
int rsf_set_model(rsf_ctx *ctx, const rsf_model *model);

/* Length of the output series: int(floor((t_final-t_start)/delta_t)), RateStateModel.py:358. */
int rsf_model_nout(rsf_ctx *ctx, int32_t *nout);

/* Batched RateStateModel.evaluate()[1] (+ MCMC.SSqcalc, MCMC.py:381-387) for C
 * independent parameter sets.
 *   dc[C]              Dc per lane
 *   a[C], b[C]         optional (NULL => model.a / model.b)
 *   data[nout]         optional observation; needed iff ssq_out != NULL
 *   ssq_out[C]         optional: sum_k (acc_k - data_k)^2, k = 0..nout-1 (acc_0 = 0)
 *   acc_out[nout][C]   optional: clean acceleration series, TIME-MAJOR (row k = time k) */
int rsf_forward_batch(rsf_ctx *ctx, int64_t n_lanes, const double *dc, const double *a,
                      const double *b, const double *data, double *ssq_out, double *acc_out);

/* ---- sampler: MCMC (MCMC.py:4-544) ------------------------------------------------- */

/* MCMC.__init__ + compute_initial_covariance + the initial SSqcalc (MCMC.py:464-468),
 * per chain: std2_0 = SSq(q0)/(nout - prior_len); Vstart = std2_0 * (X^T X)^-1 with X the
 * forward-difference sensitivity (perturbed-Dc denominator quirk kept, MCMC.py:251,264).
 * For d == 3 (extension, BASELINE config 5) the sensitivity is taken per parameter and the box prior regularises the
 * covariance: with W = diag(hi - lo), M = W X^T X W / std2_0 + 12 I and Vstart = W M^-1 W — (X^T X)^-1 alone is no proposal
 * there, the series identifies Dc and a only through their product and b hardly at all; the prior supplies the rest, as a
 * Gaussian of the box's variance would (1/12 per unit interval).  A larger fd_rel_step (1e-4) is advisable for d == 3.
 *   q0[C][d]              start point per chain
 *   data[n_groups][nout]  observation series (kept by the ctx); n_chains must be a multiple of n_groups and,
 *                         in the HIP library, n_chains/n_groups a multiple of the workgroup size */
int rsf_mcmc_init(rsf_ctx *ctx, const rsf_mcmc_config *cfg, const double *q0, const double *data);

/* Read / overwrite the per-chain sampler state.  Any pointer may be NULL.
 *   q[C][d], ssq[C], std2[C], V[C][d][d] (proposal covariance "Vold", row-major) */
int rsf_mcmc_get_state(rsf_ctx *ctx, double *q, double *ssq, double *std2, double *V);
int rsf_mcmc_set_state(rsf_ctx *ctx, const double *q, const double *ssq, const double *std2,
                       const double *V);

/* n_iters iterations of the hot loop (MCMC.py:494-527) for every chain, fused in one
 * launch.  Draws come from Philox4x32-10 keyed by (seed; global chain id, iteration, slot).
 *   trace_q[n_iters][C][d]   optional: state after each iteration (qparams columns 1..)
 *   trace_std2[n_iters][C]   optional: sigma^2 after each iteration (std2[1:])
 *   trace_accept[n_iters][C] optional: 1 = accepted
 * RSF_MEM_HOST: a run whose trace exceeds ~32 MiB is cut into launches whose rows are copied to the caller's arrays
 * on a second stream while the next launch computes (same chain: the kernel continues from the stored state). */
int rsf_mcmc_run(rsf_ctx *ctx, int64_t n_iters, double *trace_q, double *trace_std2,
                 uint8_t *trace_accept);

/* Same iteration logic, consuming caller-supplied variates instead of Philox:
 *   z[n_iters][C][d] standard normals (proposal = q + chol(V) z; d == 1: q + sqrt(V) z)
 *   u[n_iters][C]    uniforms for the accept test (read only when the proposal is in bounds)
 *   g[n_iters][C]    standard Gamma(0.5 (n0 + nout)) variates for the sigma^2 update
 * This is how MCMC.sample() reproduces a np.random.seed chain: one call per proposal with the variates NumPy drew.
 * (RSF_MEM_HOST, n_iters = 1: the copies and the kernel are one pre-instantiated hipGraph launch.) */
int rsf_mcmc_replay(rsf_ctx *ctx, int64_t n_iters, const double *z, const double *u,
                    const double *g, double *trace_q, double *trace_std2, uint8_t *trace_accept);

/* Totals since rsf_mcmc_init over this ctx's chains: accepted proposals, proposals that
 * were in bounds (= forward solves), in-bounds proposals whose SSq was NaN/Inf, iterations
 * done per chain.  Any pointer may be NULL. */
int rsf_mcmc_stats(rsf_ctx *ctx, int64_t *n_accepted, int64_t *n_evaluated, int64_t *n_nonfinite,
                   int64_t *n_iters_done);

/* Launch statistics since rsf_mcmc_init, summed over this ctx's chains: out[k] for k < min(n, RSF_CNT_COUNT).
 * The first five are properties of the chains and identical in both libraries; the rest describe how the HIP kernels spent
 * their wave-steps (a wave = 64 chains advancing in lockstep) and are zero in the checker, which has no waves.  They
 * exist so that a data-dependent slowdown is measured, not inferred: wide proposal distributions (the reference's own
 * main.py: list prior, no adaptation, MCMC.py:524-527) leave lanes idle on out-of-bounds proposals (MCMC.py:318-322) and
 * put stiff small-Dc proposals next to ordinary ones. */
#define RSF_CNT_ACCEPTED 0       /* accepted proposals */
#define RSF_CNT_EVALUATED 1      /* proposals inside the prior box (each costs a forward solve, MCMC.py:322-324) */
#define RSF_CNT_NONFINITE 2      /* in-bounds proposals whose sum of squares was NaN/Inf when their solve ended */
#define RSF_CNT_OUT_OF_BOUNDS 3  /* proposals outside the box: rejected without a solve */
#define RSF_CNT_EARLY_REJECTED 4 /* HIP, float64 RK4: in-bounds proposals whose solve stopped before the end of the series
                                    because the running sum of squares already exceeded what this iteration's uniform
                                    could accept (partial sums of squares only grow; the decision is the one the full
                                    series would give).  0 in the checker, which always integrates to the end. */
#define RSF_CNT_WAVE_SOLVES 5    /* wave-proposals with at least one in-bounds lane (a forward solve was started) */
#define RSF_CNT_WAVE_SKIPS 6     /* wave-proposals with no in-bounds lane (no solve) */
#define RSF_CNT_STEPS_TIGHT 7    /* wave-steps (RK4 steps of one wave) integrated in each tier of the float64 RK4 step: */
#define RSF_CNT_STEPS_NARROW 8   /*   TIGHT / NARROW / WIDE = incremental evaluation with series of growing length,     */
#define RSF_CNT_STEPS_WIDE 9     /*   FULL = log / exp / reciprocal at every stage (stiff small-Dc lanes)                */
#define RSF_CNT_STEPS_FULL 10
#define RSF_CNT_STEPS_REDONE 11  /* wave-steps of incremental trips thrown away because a lane left the tier's guard region */
                                 /* float32 solve (RSF_FLAG_FP32_SOLVE, two chains per lane): STEPS_TIGHT = wave-steps of incremental
                                    trips, STEPS_FULL = of full-evaluation trips (a wave holding chains of both forms runs both),
                                    STEPS_REDONE = of trips replayed step by step because a chain left the incremental form in them;
                                    NARROW / WIDE / EARLY_REJECTED / LANE_STEPS stay 0 there */
#define RSF_CNT_LANE_STEPS 12    /* sum over wave-steps of the lanes still integrating: lane utilisation =
                                    LANE_STEPS / (64 * (STEPS_TIGHT + STEPS_NARROW + STEPS_WIDE + STEPS_FULL)) */
#define RSF_CNT_COUNT 13
int rsf_mcmc_counters(rsf_ctx *ctx, int64_t *out, int32_t n);

/* ---- the sampler as an operator over a caller-evaluated likelihood ------------------------------------------
 * The reference's sampler takes ANY model object with a settable .Dc and .evaluate() (MCMC.py:65-66, 127, 381-384).  These
 * three calls run its chain logic — proposal, strict box test, accept test, sigma^2 update, adaptation; MCMC.py:494-527 —
 * on the device for a likelihood the caller evaluates: no rsf_set_model, no observation, no forward solve in the library.
 * They also pin that logic to the reference with no integrator in the loop: replaying the reference's recorded variates
 * AND its recorded sums of squares must reproduce its chain to rounding.
 *
 *   rsf_mcmc_init_state   chains from an explicit state — q[C][d], ssq[C] (SSq at q), std2[C] (sigma^2, MCMC.py:261),
 *                         V[C][d][d] (proposal covariance, MCMC.py:266) — instead of rsf_mcmc_init's own initial solves.
 *                         cfg: n_params, n_chains, chain_offset, n0, prior_len, adapt_mode, adapt_interval, lo, hi are used.
 *                         Such chains advance by rsf_mcmc_replay_ssq only (rsf_mcmc_run / rsf_mcmc_replay: RSF_ERR_STATE).
 *   rsf_mcmc_propose      the proposal the NEXT iteration will make from z[C][d]: q_new[C][d] = q + chol(V) z, and
 *                         in_bounds[C] (1 = strictly inside the box, i.e. the caller owes its sum of squares).  No state changes.
 *   rsf_mcmc_replay_ssq   rsf_mcmc_replay with the proposals' sums of squares supplied: ssq_new[n_iters][C], read where
 *                         the proposal is in bounds.  Works on chains made by either init call. */
int rsf_mcmc_init_state(rsf_ctx *ctx, const rsf_mcmc_config *cfg, const double *q, const double *ssq, const double *std2,
                        const double *V);
int rsf_mcmc_propose(rsf_ctx *ctx, const double *z, double *q_new, uint8_t *in_bounds);
int rsf_mcmc_replay_ssq(rsf_ctx *ctx, int64_t n_iters, const double *z, const double *u, const double *g,
                        const double *ssq_new, double *trace_q, double *trace_std2, uint8_t *trace_accept);

/* ---- posterior post-processing on pooled samples (RSF.plot_dist, RSF.py:717-746) -------------- */

/* Moments of n samples x[i*stride] (stride in doubles selects one parameter of a [n][d] trace block):
 *   out[0] = n, out[1] = mean, out[2] = variance (ddof = 1, as np.cov / gaussian_kde), out[3] = min, out[4] = max.
 * `out` is a HOST array of 5 doubles in every mem_space; x follows the ctx mem_space. */
int rsf_pool_summary(rsf_ctx *ctx, int64_t n, const double *x, int64_t stride, double *out);

/* Gaussian kernel density estimate on m grid points, scipy.stats.gaussian_kde semantics (RSF.py:733-736):
 *   density[j] = 1/(n sqrt(2 pi c)) * sum_i exp(-(grid[j]-x_i)^2 / (2 c)),   c = var(x, ddof=1) * factor^2,
 * factor = n^(-1/5) (Scott's rule) when bw_factor <= 0, else bw_factor.  grid[m], density[m] follow mem_space. */
int rsf_pool_kde(rsf_ctx *ctx, int64_t n, const double *x, int64_t stride, int32_t m, const double *grid,
                 double bw_factor, double *density);

/* Fixed-bin histogram of n samples x[i*stride] over [lo, hi] (the summary path of SURVEY §8e: a few KB per GPU that
 * rsf_pool_allreduce_sum combines across ranks when the pool itself need not be materialised).  numpy.histogram
 * semantics, edge cases included: nbins equal bins between the edges np.linspace(lo, hi, nbins + 1), bin i = [edge_i,
 * edge_i+1), the last bin closed at hi — a sample exactly on an edge is counted where numpy counts it.
 *   counts[0] = samples below lo, counts[1 .. nbins] = the bins, counts[nbins+1] = samples above hi or NaN.
 * Counts are exact integers stored as doubles (so that the double all-reduce sums them exactly); `counts` follows the
 * ctx mem_space; 1 <= nbins <= 4096. */
int rsf_pool_histogram(rsf_ctx *ctx, int64_t n, const double *x, int64_t stride, int32_t nbins, double lo, double hi,
                       double *counts);

/* ---- multi-GPU posterior pool: one process per GPU, RCCL over xGMI -----------------
 * The reference runs its chains one after another in one process (RSF.py:1040-1046); here each rank
 * samples its own block of global chain ids with no exchange, and the kept rows are pooled ONCE.
 *   rank 0:      rsf_comm_unique_id(id), hands the 128 bytes to the other ranks by any channel
 *                (torch.distributed broadcast, MPI, a file ...);
 *   every rank:  rsf_comm_init(ctx, world, rank, id)   — collective, creates the RCCL communicator
 *                on the ctx device;
 *   then:        rsf_pool_allgather(ctx, send, count, recv): recv[r*count .. (r+1)*count) = rank r's
 *                send[0..count) on every rank; rsf_pool_allreduce_sum: element-wise sum in place
 *                (the summary path: sum q, sum q^2, counts — a few numbers instead of the pool).
 * Buffers are in the ctx memory space like everywhere else; the collectives run on the ctx stream.
 * RCCL is bound at run time (the copy already in the process — e.g. PyTorch's — else librccl.so.1;
 * RSF_RCCL_LIB overrides).  world = 1 with id = NULL needs no RCCL (copies); with an id it is a real one-rank
 * communicator.  The CPU oracle supports world > 1 only with RSF_RCCL_LIB naming a host-memory implementation of the
 * nccl* entry points (the test-suite's tests/c/fake_rccl.c). */
#define RSF_COMM_ID_BYTES 128
int rsf_comm_unique_id(uint8_t id[RSF_COMM_ID_BYTES]);
int rsf_comm_init(rsf_ctx *ctx, int32_t world, int32_t rank, const uint8_t id[RSF_COMM_ID_BYTES]);
int rsf_comm_destroy(rsf_ctx *ctx);
int rsf_pool_allgather(rsf_ctx *ctx, const double *send, int64_t count, double *recv);
int rsf_pool_allreduce_sum(rsf_ctx *ctx, double *buf, int64_t count);

/* The same exchange for a SINGLE process that owns every GPU of the node (SURVEY §8e's process model: "single process,
 * ncclCommInitAll over visible devices ... ncclGroupStart/End around the all-gather calls") — what a plain C caller, or any
 * caller without a process launcher, uses: one ctx per device, one host thread.
 *   rsf_comm_init_all(ctxs, n)       ncclCommInitAll over the ctxs' devices; ctxs[i] becomes rank i of world n.  Not
 *                                    collective across threads or processes (rsf_comm_init blocks until every rank has
 *                                    joined, so one thread holding all the ctxs could never finish it).
 *   rsf_pool_allgather_all(...)      one grouped all-gather: recv[i][r*count .. (r+1)*count) = send[r][0 .. count) for all i, r
 *   rsf_pool_allreduce_sum_all(...)  one grouped all-reduce: bufs[i][k] = sum over r of bufs[r][k], in place on every ctx
 * send[i] / recv[i] / bufs[i] live in ctxs[i]'s memory space (device buffers on ctxs[i]'s device); the collectives run on
 * each ctx's stream; host-space results are complete on return.  rsf_comm_destroy(ctxs[i]) ends each communicator. */
int rsf_comm_init_all(rsf_ctx *const *ctxs, int32_t n);
int rsf_pool_allgather_all(rsf_ctx *const *ctxs, int32_t n, const double *const *send, int64_t count, double *const *recv);
int rsf_pool_allreduce_sum_all(rsf_ctx *const *ctxs, int32_t n, double *const *bufs, int64_t count);

/* The Philox4x32-10 block function itself (known-answer tests; Random123 vectors). */
int rsf_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* The variates rsf_mcmc_run would draw for (seed, global chain id, iteration): d proposal
 * normals, the accept uniform and the Gamma(shape) variate.  Host-side, for tests. */
int rsf_mcmc_draws(uint64_t seed, int64_t chain, int64_t iteration, int32_t n_params, double shape,
                   double *z, double *u, double *g);

/* MCMC.update_covariance_matrix (MCMC.py:162-204) for ONE window of samples window[n][d] (the reference passes the last
 * adapt_interval columns of qparams), computed on the device by the sampler's own adaptation arithmetic.  V_out[d][d] is
 * what the reference's loop would assign to Vold (MCMC.py:525): RSF_ADAPT_REFERENCE_DICT — the Cholesky FACTOR of
 * 2.38^2/prior_len * cov(window) (the quirk: it is then used as a covariance; d = 1 only; prior_len 0 => 2);
 * RSF_ADAPT_AM — 2.38^2/d * cov(window).  RSF_ERR_NOT_POSDEF when that matrix is not positive definite (where
 * np.linalg.cholesky raises and the reference keeps its covariance, MCMC.py:524-527).  Host arrays; for callers that
 * compose the sampler's sub-steps themselves. */
int rsf_mcmc_adapt(int32_t n_params, int32_t n, const double *window, int32_t adapt_mode, int32_t prior_len, double *V_out);

#ifdef __cplusplus
}
#endif
#endif /* RSF_ABI_H */
