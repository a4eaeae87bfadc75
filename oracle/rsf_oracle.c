/*
 * rsf_oracle.c — CPU restatement of the MCMC-over-ODE hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path may load this library;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and only as the
 * checker / timed baseline.  It implements include/rsf_abi.h on the host (RSF_MEM_HOST only).
 *
 * What it restates (paths relative to the reference tree):
 *   friction()        RateStateModel.py:318-355   literal operation order, no FMA contraction
 *   evaluate()        RateStateModel.py:358-395   with the reference's adaptive dop853 replaced
 *                                                 by the fixed-step RK4 that BASELINE.json names
 *   SSqcalc           MCMC.py:381-387
 *   acceptreject      MCMC.py:318-333
 *   sigma^2 update    MCMC.py:158-160
 *   initial cov.      MCMC.py:244-266
 *   adaptation        MCMC.py:200-204, 523-527
 *   sample() loop     MCMC.py:494-527
 *
 * Parity pinning: the reference ships no tests or golden vectors.  This restatement is pinned
 * by tests/golden/ (captured from the live reference import by oracle/make_golden.py):
 *   - tier 2: RK4(substeps S) converges to the reference's dop853 trajectory at 16x per
 *     halving (tests/test_oracle_golden.py);
 *   - sampler logic: the Python twin oracle/rsf_oracle.py replays the reference's recorded
 *     draws exactly; this C file is checked against that twin.
 * The RNG (Philox4x32-10 + Box-Muller + Marsaglia-Tsang) has no reference counterpart (the
 * reference uses NumPy's global MT19937 stream); Philox is pinned by Random123's known-answer
 * vectors.
 */
#include "../include/rsf_abi.h"
#include "../include/rsf_dop853_tableau.h"

#include <dlfcn.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static __thread char g_err[512] = "";

static int fail(int code, const char *msg) {
  snprintf(g_err, sizeof g_err, "%s", msg);
  return code;
}

/* ------------------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11; Random123 constants)                             */
/* ------------------------------------------------------------------------------------ */
static void philox_block(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
  uint32_t k0 = key[0], k1 = key[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Draw slots of one (chain, iteration):
 *   slot 0: proposal normals z0, z1     slot 1: proposal normal z2
 *   slot 2: accept uniform              slot 8+2j / 9+2j: gamma attempt j (normal / uniform) */
enum { SLOT_Z01 = 0, SLOT_Z2 = 1, SLOT_U = 2, SLOT_GAMMA = 8 };

static void draw_words(uint64_t seed, uint64_t chain, uint32_t iter, uint32_t slot, uint32_t w[4]) {
  uint32_t ctr[4] = {(uint32_t)chain, (uint32_t)(chain >> 32), iter, slot};
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  philox_block(ctr, key, w);
}

/* 53-bit uniform in (0, 1] */
static double u53(uint32_t hi, uint32_t lo) {
  uint64_t k = (((uint64_t)hi << 32) | lo) >> 11;
  return (double)(k + 1) * 0x1.0p-53;
}

static void normal_pair(const uint32_t w[4], double *z0, double *z1) {
  double u1 = u53(w[0], w[1]), u2 = u53(w[2], w[3]);
  double r = sqrt(-2.0 * log(u1));
  double th = 6.283185307179586476925286766559 * u2;
  *z0 = r * cos(th);
  *z1 = r * sin(th);
}

/* Marsaglia & Tsang (2000), shape >= 1, log acceptance test only (no squeeze). */
static double gamma_draw(uint64_t seed, uint64_t chain, uint32_t iter, double shape) {
  double d = shape - 1.0 / 3.0;
  double c = 1.0 / sqrt(9.0 * d);
  for (uint32_t j = 0; j < 64; ++j) {
    uint32_t w[4];
    double x, unused, u, v;
    draw_words(seed, chain, iter, SLOT_GAMMA + 2 * j, w);
    normal_pair(w, &x, &unused);
    v = 1.0 + c * x;
    if (!(v > 0.0)) continue;
    v = v * v * v;
    draw_words(seed, chain, iter, SLOT_GAMMA + 2 * j + 1, w);
    u = u53(w[0], w[1]);
    if (log(u) < 0.5 * x * x + d * (1.0 - v + log(v))) return d * v;
  }
  return d; /* unreachable in practice (acceptance > 0.95 per attempt) */
}

/* ------------------------------------------------------------------------------------ */
/* context                                                                                */
/* ------------------------------------------------------------------------------------ */
struct rsf_ctx {
  rsf_config cfg;
  int have_model;
  rsf_model m;
  int32_t nout;
  double delta_t, h;
  /* sampler */
  int have_chains;
  rsf_mcmc_config mc;
  double *data;                 /* [n_groups][nout] */
  int32_t n_groups;
  double *q, *ssq, *std2, *V;   /* [C][d], [C], [C], [C][d][d] */
  double *wref, *wsum, *wsq;    /* adaptation window: shift [C][d], sums [C][d], [C][d][d] */
  int32_t *wn;                  /* [C] samples in window */
  double *wbuf;                 /* reference_dict: the window itself, [C][adapt_interval] (np.cov's own arithmetic needs the samples) */
  int64_t iters_done;
  int64_t n_acc, n_eval, n_nonfinite, n_oob;
  int external_chains;          /* made by rsf_mcmc_init_state: no observation, advanced by rsf_mcmc_replay_ssq only */
  int32_t world, rank;          /* world 0: rsf_comm_init not called */
  void *comm;                   /* world > 1: communicator of the nccl implementation named by RSF_RCCL_LIB (tests/c/fake_rccl.c) */
};

int rsf_version(void) { return RSF_ABI_VERSION; }
const char *rsf_backend(void) { return "oracle-cpu"; }
const char *rsf_build_id(void) { return "oracle"; }
const char *rsf_last_error(void) { return g_err; }
int rsf_device_count(void) { return 0; }

int rsf_create(const rsf_config *cfg, rsf_ctx **out) {
  if (!cfg || !out) return fail(RSF_ERR_INVALID, "rsf_create: NULL argument");
  if (cfg->size != sizeof(rsf_config) || cfg->version != RSF_ABI_VERSION)
    return fail(RSF_ERR_INVALID, "rsf_create: config size/version mismatch");
  if (cfg->mem_space != RSF_MEM_HOST)
    return fail(RSF_ERR_UNSUPPORTED, "rsf_create: the CPU oracle takes host buffers only");
  rsf_ctx *c = (rsf_ctx *)calloc(1, sizeof *c);
  if (!c) return fail(RSF_ERR_NOMEM, "rsf_create: out of memory");
  c->cfg = *cfg;
  *out = c;
  return RSF_OK;
}

static void free_chains(rsf_ctx *c) {
  free(c->data); free(c->q); free(c->ssq); free(c->std2); free(c->V);
  free(c->wref); free(c->wsum); free(c->wsq); free(c->wn); free(c->wbuf);
  c->data = c->q = c->ssq = c->std2 = c->V = c->wref = c->wsum = c->wsq = NULL;
  c->wn = NULL;
  c->wbuf = NULL;
  c->have_chains = 0;
}

int rsf_destroy(rsf_ctx *c) {
  if (!c) return RSF_OK;
  rsf_comm_destroy(c);
  free_chains(c);
  free(c);
  return RSF_OK;
}

int rsf_sync(rsf_ctx *c) { return c ? RSF_OK : fail(RSF_ERR_INVALID, "rsf_sync: NULL ctx"); }

static int nthreads(const rsf_ctx *c) {
#ifdef _OPENMP
  return c->cfg.cpu_threads ? (int)c->cfg.cpu_threads : omp_get_max_threads();
#else
  (void)c;
  return 1;
#endif
}

/* ------------------------------------------------------------------------------------ */
/* forward model                                                                          */
/* ------------------------------------------------------------------------------------ */
int rsf_set_model(rsf_ctx *c, const rsf_model *m) {
  if (!c || !m) return fail(RSF_ERR_INVALID, "rsf_set_model: NULL argument");
  if (m->size != sizeof(rsf_model)) return fail(RSF_ERR_INVALID, "rsf_set_model: struct size mismatch");
  if (m->nsteps < 2 || m->substeps < 1 || !(m->t_final > m->t_start))
    return fail(RSF_ERR_INVALID, "rsf_set_model: need nsteps >= 2, substeps >= 1, t_final > t_start");
  if ((m->flags & RSF_FLAG_FP32_SOLVE) && (m->flags & RSF_FLAG_DOP853))
    return fail(RSF_ERR_INVALID, "rsf_set_model: the dop853 integrator is float64 only");
  if (c->have_chains) free_chains(c);  /* chain state belongs to the previous model */
  c->m = *m;
  c->delta_t = (m->t_final - m->t_start) / m->nsteps;                   /* RateStateModel.py:176 */
  c->nout = (int32_t)floor((m->t_final - m->t_start) / c->delta_t);     /* RateStateModel.py:358 */
  c->h = c->delta_t / m->substeps;
  c->have_model = 1;
  return RSF_OK;
}

int rsf_model_nout(rsf_ctx *c, int32_t *nout) {
  if (!c || !nout) return fail(RSF_ERR_INVALID, "rsf_model_nout: NULL argument");
  if (!c->have_model) return fail(RSF_ERR_STATE, "rsf_model_nout: call rsf_set_model first");
  *nout = c->nout;
  return RSF_OK;
}

/* RateStateModel.py:318-355, literal operation order. */
static void friction(const rsf_model *m, double t, double dc, double a, double b, const double y[3],
                     double dydt[3]) {
  double V_ref = m->V_ref;
  double kprime = 1e-2 * 10 / dc;
  double a1 = 20, a2 = 10;
  double V_l = V_ref * (1 + exp(-t / a1) * sin(a2 * t));
  double temp = 1 / a * (y[0] - m->mu_ref - b * log(V_ref * y[1] / dc));
  double v = V_ref * exp(temp);
  dydt[1] = 1. - v * y[1] / dc;
  dydt[0] = kprime * V_l - kprime * v;
  dydt[2] = v / a * (dydt[0] - b / y[1] * dydt[1]);
  if (m->flags & RSF_FLAG_RADIATION_DAMPING) {
    dydt[0] = dydt[0] - m->k1 * dydt[2];
    dydt[2] = v / a * (dydt[0] - b / y[1] * dydt[1]);
  }
}

/* One forward solve: fixed-step classical RK4, `substeps` steps per output interval.
 * Stage times are t_start + j*(h/2) with integer j, so that the product's tabulated V_l
 * sees bit-identical arguments.  Returns SSq if data != NULL; writes acc[k*stride] if acc. */
static double solve_dop853(const rsf_ctx *c, double dc, double a, double b, const double *data, double *acc, int64_t stride);

/* RSF_FLAG_FP32_SOLVE (BASELINE config 5's float32 leg): the same fixed-step RK4 carried in IEEE float — plain `float`
 * arithmetic, one rounding per operation (fmaf where the formula is a fused multiply-add).  It restates the float32
 * FORMULATION of the product (csrc/rsf_device_f32.h), not an independent algorithm: the rescaled state ms = mu/k',
 * x = V_ref theta/Dc, the acceleration sample from the step's velocity increment (differencing two float velocities near
 * V_ref would lose four digits), float tables, the squares of float residuals summed in float over groups of eight samples
 * and the groups' sums in a pair of floats by the exact two-sum — and the product's rule for HOW a
 * chain takes a step (round 4), which is a function of the chain's own parameters and trajectory only:
 *   - incrementally (incr_step32): state (w, Rh = (h/2Dc)/x), every stage reached from the step's start by short series,
 *     no transcendental function — while the step's END increments satisfy |rho| < 2^-10 and |dlt| < 2^-7 (NaN passes);
 *   - from the first step that does not, to the end of the solve: full evaluations at every stage (full_step32: exp2f /
 *     log2f / division where the GPU has v_exp_f32 / v_log_f32 / v_rcp_f32), starting from (ms, x = hhd/Rh) at that
 *     step's start point.
 * In the incremental form GPU and restatement execute the same IEEE operations in the same order: they agree bit for bit.
 * In the full form they differ by the last-place behaviour of the hardware transcendentals against libm.  Either way every
 * constant and every term of the kernel is pinned far below the 1e-3 band that separates float32 from float64 results. */
typedef struct {
  float kia2, tc2, boa, beta, c3, kvk, vref, cv, hh, h, h6, hhd, hd, h6d; /* full evaluation */
  float khh, kh, kh6, nhboa, bh;                                         /* incremental step */
} lane32;

/* the RHS at (ms, x) in the product's regrouping (csrc/rsf_device_f32.h, rhs_full): w = v/V_ref = 2^(kia2 ms + tc2 - (b/a) log2 x),
 * d(ms)/dt = V_l - V_ref w, dtheta/dt = 1 - w x, and dV/dt = vk w g with g = (V_l - beta/x) + (beta - V_ref) w; the damping
 * pass (RateStateModel.py:349-353) subtracts (kvk w) g from d(ms)/dt and from g.  Returns w g. */
static float rhs32(const lane32 *L, int damp, float ms, float x, float vl, float *d0, float *d1) {
  float lg = log2f(x), rx = 1.0f / x;
  float w = exp2f(fmaf(-L->boa, lg, fmaf(ms, L->kia2, L->tc2)));
  float t1 = fmaf(-L->beta, rx, vl);
  float e0 = fmaf(-L->vref, w, vl);                                      /* RateStateModel.py:343, in units of k' */
  float e1 = fmaf(-w, x, 1.0f);                                          /* :340 */
  float g = fmaf(L->c3, w, t1);                                          /* :346, in units of vk w */
  if (damp) {                                                            /* :349-353 */
    float kw = L->kvk * w;
    e0 = fmaf(-kw, g, e0);
    g = fmaf(-kw, g, g);
  }
  *d0 = e0; *d1 = e1;
  return w * g;
}

/* one RK4 step by full evaluations (csrc/rsf_device_f32.h, rk4_full); returns k1 + 2 k2 + 2 k3 + k4 of dV/dt in units of vk */
static float full_step32(const lane32 *L, int damp, float *ms_io, float *x_io, float vl0, float vlm, float vl1) {
  float ms = *ms_io, x = *x_io, a0, a1, b0, b1, c0, c1, e0, e1;
  float wa = rhs32(L, damp, ms, x, vl0, &a0, &a1);
  float wb = rhs32(L, damp, fmaf(L->hh, a0, ms), fmaf(L->hhd, a1, x), vlm, &b0, &b1);
  float wc = rhs32(L, damp, fmaf(L->hh, b0, ms), fmaf(L->hhd, b1, x), vlm, &c0, &c1);
  float we = rhs32(L, damp, fmaf(L->h, c0, ms), fmaf(L->hd, c1, x), vl1, &e0, &e1);
  *ms_io = fmaf(L->h6, fmaf(2.0f, b0 + c0, a0 + e0), ms);
  *x_io = fmaf(L->h6d, fmaf(2.0f, b1 + c1, a1 + e1), x);
  return fmaf(2.0f, wb + wc, wa + we);
}

/* the RHS of the incremental form (csrc/rsf_device_f32.h, rhs_incr): theta derivatives scaled by Rh,
 * d1' = Rh (1 - w x) = Rh - w xr with xr = Rh x at the stage; d0 = V_l - V_ref w; g = d0 - brx d1' (+ the damping pass) */
static void rhs_incr32(const lane32 *L, int damp, float w, float xr, float Rh, float vl, float brx, float *d0, float *d1, float *g) {
  float e1 = fmaf(-w, xr, Rh);
  float e0 = fmaf(-L->vref, w, vl);
  float gg = fmaf(-brx, e1, e0);
  if (damp) {
    float kw = L->kvk * w;
    e0 = fmaf(-kw, gg, e0);
    gg = fmaf(-kw, gg, gg);
  }
  *d0 = e0; *d1 = e1; *g = gg;
}

/* (w', q) at the point reached from the step's start (w0, x) by rho = dx/x and d(mu)/a = kd d0 (csrc/rsf_device_f32.h, incr, in
 * its operation order): dlt = kd d0 - (b/a) log1p(rho) to rho^2/2, w' = w0 exp(dlt) to dlt^3/6 as
 * w0 (1 + dlt) + dlt^2 (w0/2 + (w0/6) dlt), 1/x' = (1/x)(1 + q) with q = -rho + rho^2 */
static void incr32(const lane32 *L, float rho, float kd, float d0, float w0, float w02, float w06, float *w, float *q, float *dlt_out) {
  float rP = rho * fmaf(rho, L->nhboa, L->boa);
  float dlt = fmaf(kd, d0, -rP);
  float d2 = dlt * dlt, A = fmaf(dlt, w0, w0), B = fmaf(dlt, w06, w02);
  *w = fmaf(d2, B, A);
  *q = fmaf(rho, rho, -rho);
  *dlt_out = dlt;
}

/* one RK4 step of the incremental form (csrc/rsf_device_f32.h, rk4_incr) */
static float incr_step32(const lane32 *L, int damp, float *w_io, float *Rh_io, float *ms_io, float vl0, float vlm, float vl1,
                         float *rho_end, float *dlt_end) {
  const float w0 = *w_io, Rh = *Rh_io, third = 1.0f / 3.0f;
  const float w02 = w0 * 0.5f, w06 = w0 * (1.0f / 6.0f);
  float a0, a1, ga, b0, b1, gb, c0, c1, gc, e0, e1, ge, w, q, dlt;
  rhs_incr32(L, damp, w0, L->hhd, Rh, vl0, L->bh, &a0, &a1, &ga);
  float sv = w0 * ga;
  incr32(L, a1, L->khh, a0, w0, w02, w06, &w, &q, &dlt);
  rhs_incr32(L, damp, w, fmaf(L->hhd, a1, L->hhd), Rh, vlm, fmaf(L->bh, q, L->bh), &b0, &b1, &gb);
  float sm = w * gb;
  incr32(L, b1, L->khh, b0, w0, w02, w06, &w, &q, &dlt);
  rhs_incr32(L, damp, w, fmaf(L->hhd, b1, L->hhd), Rh, vlm, fmaf(L->bh, q, L->bh), &c0, &c1, &gc);
  sm = fmaf(w, gc, sm);
  const float T0 = fmaf(2.0f, b0 + c0, a0), T13 = fmaf(2.0f, b1 + c1, a1) * third;
  incr32(L, c1 + c1, L->kh, c0, w0, w02, w06, &w, &q, &dlt);
  rhs_incr32(L, damp, w, fmaf(L->hd, c1, L->hhd), Rh, vl1, fmaf(L->bh, q, L->bh), &e0, &e1, &ge);
  sv = fmaf(w, ge, sv);
  const float t0 = T0 + e0;
  *rho_end = fmaf(e1, third, T13);
  incr32(L, *rho_end, L->kh6, t0, w0, w02, w06, &w, &q, dlt_end);
  *ms_io = fmaf(L->h6, t0, *ms_io);
  *w_io = w;
  *Rh_io = fmaf(Rh, q, Rh);
  return fmaf(2.0f, sm, sv);
}

static double solve_f32(const rsf_ctx *c, double dc, double a, double b, const double *data, double *acc, int64_t stride) {
  const rsf_model *m = &c->m;
  const int S = m->substeps, damp = (m->flags & RSF_FLAG_RADIATION_DAMPING) != 0;
  const double h = c->h, hh = 0.5 * c->h, h6 = c->h / 6.0, log2e = 1.4426950408889634074;
  const double inv_a = 1.0 / a, inv_dc = 1.0 / dc, kprime = (1e-2 * 10) / dc, vdc = m->V_ref * inv_dc;
  const double beta = b * m->V_ref * (1.0 / (1e-2 * 10));
  lane32 L;
  L.kia2 = (float)(kprime * inv_a * log2e); L.tc2 = (float)(-m->mu_ref * inv_a * log2e); L.boa = (float)(b * inv_a);
  L.beta = (float)beta; L.c3 = (float)(beta - m->V_ref);
  L.kvk = (float)(m->k1 * m->V_ref * inv_a); L.vref = (float)m->V_ref;
  L.cv = (float)((h6 * (1.0 / c->delta_t)) * (m->V_ref * inv_a * kprime));
  L.hh = (float)hh; L.h = (float)h; L.h6 = (float)h6;
  L.hhd = (float)(hh * vdc); L.hd = (float)(h * vdc); L.h6d = (float)(h6 * vdc);
  L.khh = (float)(kprime * inv_a * hh); L.kh = (float)(kprime * inv_a * h); L.kh6 = (float)(kprime * inv_a * h6);
  L.nhboa = (float)(-0.5 * (b * inv_a)); L.bh = (float)(beta / (hh * vdc));
  /* x(0) = 1: w(0) = exp((mu(0) - mu_ref)/a), evaluated in double and rounded once; Rh(0) = hhd */
  float ms = (float)(m->mu_t_zero / kprime), x = 1.0f, w = (float)exp((m->mu_t_zero - m->mu_ref) * inv_a), Rh = L.hhd;
  int full = 0;
  double ssq = 0.0;
  float s32 = 0.0f;
  int64_t j = 0;
  if (acc) acc[0] = 0.0;
  float hi = 0.0f, lo = 0.0f;
  if (data) { double d0 = (double)(float)data[0]; ssq = d0 * d0; hi = (float)ssq; lo = (float)(ssq - (double)hi); }
  for (int32_t k = 1; k < c->nout; ++k) {
    float dv = 0.0f;
    for (int s = 0; s < S; ++s, j += 2) {
      /* the float loading table of the product: V_l at t_start + j*h/2 evaluated in double, rounded to float */
      double t0 = m->t_start + (double)j * hh, tm = m->t_start + (double)(j + 1) * hh, t1 = m->t_start + (double)(j + 2) * hh;
      float vl0 = (float)(m->V_ref * (1 + exp(-t0 / 20) * sin(10 * t0)));
      float vlm = (float)(m->V_ref * (1 + exp(-tm / 20) * sin(10 * tm)));
      float vl1 = (float)(m->V_ref * (1 + exp(-t1 / 20) * sin(10 * t1)));
      float wsum = 0.0f;
      if (!full) {
        float w1 = w, Rh1 = Rh, ms1 = ms, rho, dlt;
        wsum = incr_step32(&L, damp, &w1, &Rh1, &ms1, vl0, vlm, vl1, &rho, &dlt);
        if (!(fabsf(rho) >= 0x1p-10f) && !(fabsf(dlt) >= 0x1p-7f)) { w = w1; Rh = Rh1; ms = ms1; }
        else { x = (float)((double)L.hhd / (double)Rh); full = 1; }  /* this step and every later one by full evaluations */
      }
      if (full) wsum = full_step32(&L, damp, &ms, &x, vl0, vlm, vl1);
      dv = dv + wsum;
    }
    float ak = dv * L.cv;                                                /* RateStateModel.py:388, from the increment */
    if (acc) acc[k * stride] = (double)ak;
    if (data) {
      /* float residuals, squared and summed in float within a group of eight samples (k = 1..8, 9..16, ...; the last may be
       * short); the total carried as an unevaluated sum of two floats, a group's sum added to it by the exact two-sum
       * (csrc/rsf_device_f32.h, Out32): no double arithmetic inside the solve */
      float r = ak - (float)data[k];
      s32 = fmaf(r, r, s32);
      if ((k & 7) == 0 || k == c->nout - 1) {
        float x = s32, sm = hi + x, bb = sm - hi, t = sm - bb;
        float e1 = hi - t, e2 = x - bb;
        lo = lo + (e1 + e2);
        hi = sm;
        s32 = 0.0f;
      }
    }
  }
  if (data) ssq = (double)hi + (double)lo;
  return ssq;
}

static double solve(const rsf_ctx *c, double dc, double a, double b, const double *data, double *acc,
                    int64_t stride) {
  const rsf_model *m = &c->m;
  if (m->flags & RSF_FLAG_DOP853) return solve_dop853(c, dc, a, b, data, acc, stride);
  if (m->flags & RSF_FLAG_FP32_SOLVE) return solve_f32(c, dc, a, b, data, acc, stride);
  const int S = m->substeps;
  const double h = c->h, hh = 0.5 * c->h;
  double y[3] = {m->mu_t_zero, dc / m->V_ref, m->V_ref}; /* RateStateModel.py:367-377 */
  double vprev = m->V_ref;
  double ssq = 0.0;
  int64_t j = 0; /* half-step index of the current step start */
  if (acc) acc[0] = 0.0;
  if (data) ssq = (0.0 - data[0]) * (0.0 - data[0]);
  for (int32_t k = 1; k < c->nout; ++k) {
    for (int s = 0; s < S; ++s, j += 2) {
      double k1[3], k2[3], k3[3], k4[3], ys[3];
      double t0 = m->t_start + (double)j * hh;
      double tm = m->t_start + (double)(j + 1) * hh;
      double t1 = m->t_start + (double)(j + 2) * hh;
      friction(m, t0, dc, a, b, y, k1);
      for (int i = 0; i < 3; ++i) ys[i] = y[i] + hh * k1[i];
      friction(m, tm, dc, a, b, ys, k2);
      for (int i = 0; i < 3; ++i) ys[i] = y[i] + hh * k2[i];
      friction(m, tm, dc, a, b, ys, k3);
      for (int i = 0; i < 3; ++i) ys[i] = y[i] + h * k3[i];
      friction(m, t1, dc, a, b, ys, k4);
      for (int i = 0; i < 3; ++i) y[i] = y[i] + h / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
    }
    double ak = (y[2] - vprev) / c->delta_t; /* RateStateModel.py:388 */
    vprev = y[2];
    if (acc) acc[k * stride] = ak;
    if (data) ssq += (ak - data[k]) * (ak - data[k]);
  }
  return ssq;
}

/* ------------------------------------------------------------------------------------ */
/* The reference's own integrator: Hairer's DOP853 as scipy.integrate.ode('dop853') drives it     */
/* (RateStateModel.py:374-389): rtol 1e-6, atol 1e-10, safety 0.9, step factors 0.3 .. 6, beta 0, */
/* at most 500 steps per call, one call per output interval, HMAX = interval length, and the     */
/* predicted step size carried from one call to the next through the work array (first call:     */
/* HINIT).  Restated from the published algorithm (Hairer, Norsett, Wanner: Solving ODEs I,      */
/* II.10 / dop853.f); the tableau comes from include/rsf_dop853_tableau.h.                        */
/* ------------------------------------------------------------------------------------ */
#define DP_RTOL 1e-6
#define DP_ATOL 1e-10

typedef struct {
  double h;      /* WORK(7): step size proposed by the previous call (0 => HINIT) */
  int failed;    /* r.successful() turned false: the reference stops integrating (trailing zeros) */
} dp_carry;

static double dp_hinit(const rsf_ctx *c, double dc, double a, double b, double x, const double y[3], double posneg,
                       const double f0[3], double hmax) {
  double dnf = 0.0, dny = 0.0, y1[3], f1[3], der2 = 0.0, h, h1, der12;
  for (int i = 0; i < 3; ++i) {
    double sk = DP_ATOL + DP_RTOL * fabs(y[i]);
    dnf += (f0[i] / sk) * (f0[i] / sk);
    dny += (y[i] / sk) * (y[i] / sk);
  }
  h = (dnf <= 1e-10 || dny <= 1e-10) ? 1.0e-6 : sqrt(dny / dnf) * 0.01;
  h = fmin(h, hmax);
  h = copysign(h, posneg);
  for (int i = 0; i < 3; ++i) y1[i] = y[i] + h * f0[i];
  friction(&c->m, x + h, dc, a, b, y1, f1);
  for (int i = 0; i < 3; ++i) {
    double sk = DP_ATOL + DP_RTOL * fabs(y[i]);
    der2 += ((f1[i] - f0[i]) / sk) * ((f1[i] - f0[i]) / sk);
  }
  der2 = sqrt(der2) / h;
  der12 = fmax(fabs(der2), sqrt(dnf));
  h1 = der12 <= 1e-15 ? fmax(1.0e-6, fabs(h) * 1.0e-3) : pow(0.01 / der12, 1.0 / 8.0);
  h = fmin(fmin(100 * fabs(h), h1), hmax);
  return copysign(h, posneg);
}

/* one call of dop853: integrate y from *x to xend; returns 1 on success (IDID = 1) */
static int dp_call(const rsf_ctx *c, double dc, double a, double b, double *x, double xend, double y[3], dp_carry *cw) {
  const double safe = 0.9, facc1 = 1.0 / 0.3, facc2 = 1.0 / 6.0, expo1 = 1.0 / 8.0, uround = 2.3e-16;
  const double posneg = copysign(1.0, xend - *x), hmax = fabs(xend - *x);
  double k[12][3], ys[3], h = cw->h, hnew;
  int last = 0, reject = 0, nstep = 0;
  friction(&c->m, *x, dc, a, b, y, k[0]);
  if (h == 0.0) h = dp_hinit(c, dc, a, b, *x, y, posneg, k[0], hmax);
  for (;;) {
    if (nstep > 500) return 0;                                     /* NMAX */
    if (0.1 * fabs(h) <= fabs(*x) * uround) return 0;              /* step size too small */
    if ((*x + 1.01 * h - xend) * posneg > 0.0) { h = xend - *x; last = 1; }
    ++nstep;
    for (int st = 1; st < 12; ++st) {                              /* the twelve stages */
      for (int i = 0; i < 3; ++i) {
        double s = 0.0;
        for (int j = 0; j < st; ++j) s += RSF_DP_A[st - 1][j] * k[j][i];
        ys[i] = y[i] + h * s;
      }
      friction(&c->m, st == 11 ? *x + h : *x + RSF_DP_C[st] * h, dc, a, b, ys, k[st]);
    }
    double k5[3], err = 0.0, err2 = 0.0, deno;
    for (int i = 0; i < 3; ++i) {
      double s = 0.0;
      for (int j = 0; j < 8; ++j) s += RSF_DP_B[j] * k[RSF_DP_W_STAGE[j]][i];
      k5[i] = y[i] + h * s;
    }
    for (int i = 0; i < 3; ++i) {                                  /* error estimation */
      double sk = DP_ATOL + DP_RTOL * fmax(fabs(y[i]), fabs(k5[i])), e3 = 0.0, e5 = 0.0;
      for (int j = 0; j < 8; ++j) {
        e3 += RSF_DP_E3[j] * k[RSF_DP_W_STAGE[j]][i];
        e5 += RSF_DP_E5[j] * k[RSF_DP_W_STAGE[j]][i];
      }
      err2 += (e3 / sk) * (e3 / sk);
      err += (e5 / sk) * (e5 / sk);
    }
    deno = err + 0.01 * err2;
    if (deno <= 0.0) deno = 1.0;
    err = fabs(h) * err * sqrt(1.0 / (3 * deno));
    double fac11 = pow(err, expo1), fac = fmax(facc2, fmin(facc1, fac11 / safe));   /* beta = 0 */
    hnew = h / fac;
    if (err <= 1.0) {                                              /* step accepted */
      friction(&c->m, *x + h, dc, a, b, k5, k[0]);                 /* first-same-as-last */
      for (int i = 0; i < 3; ++i) y[i] = k5[i];
      *x = *x + h;
      if (last) { cw->h = hnew; return 1; }
      if (fabs(hnew) > hmax) hnew = posneg * hmax;
      if (reject) hnew = posneg * fmin(fabs(hnew), fabs(h));
      reject = 0;
    } else {                                                       /* step rejected */
      hnew = h / fmin(facc1, fac11 / safe);
      reject = 1;
      last = 0;
    }
    h = hnew;
  }
}

static double solve_dop853(const rsf_ctx *c, double dc, double a, double b, const double *data, double *acc, int64_t stride) {
  const rsf_model *m = &c->m;
  double y[3] = {m->mu_t_zero, dc / m->V_ref, m->V_ref};           /* RateStateModel.py:377 */
  double x = m->t_start, vprev = m->V_ref, ssq = 0.0;
  dp_carry cw = {0.0, 0};
  if (acc) acc[0] = 0.0;
  if (data) ssq = (0.0 - data[0]) * (0.0 - data[0]);
  for (int32_t kk = 1; kk < c->nout; ++kk) {                       /* RateStateModel.py:380-389 */
    double ak = 0.0;                                               /* zero-initialised arrays keep 0 after a failure */
    if (!cw.failed) {
      if (dp_call(c, dc, a, b, &x, x + c->delta_t, y, &cw)) {
        ak = (y[2] - vprev) / c->delta_t;
        vprev = y[2];
      } else {
        /* the failing call still returns its partial state and the loop body runs once more before
         * r.successful() is tested: record that sample, then stop */
        ak = (y[2] - vprev) / c->delta_t;
        vprev = y[2];
        cw.failed = 1;
      }
    }
    if (acc) acc[kk * stride] = ak;
    if (data) ssq += (ak - data[kk]) * (ak - data[kk]);
  }
  return ssq;
}

int rsf_forward_batch(rsf_ctx *c, int64_t n, const double *dc, const double *a, const double *b,
                      const double *data, double *ssq_out, double *acc_out) {
  if (!c || !dc || n < 0) return fail(RSF_ERR_INVALID, "rsf_forward_batch: bad argument");
  if (!c->have_model) return fail(RSF_ERR_STATE, "rsf_forward_batch: call rsf_set_model first");
  if (ssq_out && !data) return fail(RSF_ERR_INVALID, "rsf_forward_batch: ssq_out needs data");
  int nt = nthreads(c);
  (void)nt;
#pragma omp parallel for schedule(static) num_threads(nt)
  for (int64_t i = 0; i < n; ++i) {
    double ai = a ? a[i] : c->m.a, bi = b ? b[i] : c->m.b;
    double s = solve(c, dc[i], ai, bi, ssq_out ? data : NULL, acc_out ? acc_out + i : NULL, n);
    if (ssq_out) ssq_out[i] = s;
  }
  return RSF_OK;
}

/* ------------------------------------------------------------------------------------ */
/* sampler                                                                                */
/* ------------------------------------------------------------------------------------ */
/* observation series of local chain i (RSF.py:874-882: one series per true Dc) */
static const double *data_of(const rsf_ctx *c, int64_t i) {
  return c->data + (i / (c->mc.n_chains / c->n_groups)) * (int64_t)c->nout;
}

static double ssq_of(const rsf_ctx *c, int64_t i, const double *q, int d) {
  double a = d == 3 ? q[1] : c->m.a, b = d == 3 ? q[2] : c->m.b;
  return solve(c, q[0], a, b, data_of(c, i), NULL, 0);
}

/* lower Cholesky factor of a d x d covariance; a failed pivot zeroes that column. */
static int chol_lower(const double *V, int d, double *L) {
  int ok = 1;
  memset(L, 0, sizeof(double) * d * d);
  for (int j = 0; j < d; ++j) {
    double s = V[j * d + j];
    for (int k = 0; k < j; ++k) s -= L[j * d + k] * L[j * d + k];
    if (!(s > 0.0)) { ok = 0; continue; }
    double ljj = sqrt(s);
    L[j * d + j] = ljj;
    for (int i = j + 1; i < d; ++i) {
      double t = V[i * d + j];
      for (int k = 0; k < j; ++k) t -= L[i * d + k] * L[j * d + k];
      L[i * d + j] = t / ljj;
    }
  }
  return ok;
}

/* inverse of a symmetric 3x3 (adjugate / determinant) or 1x1 */
static void sym_inverse(const double *A, int d, double *Ai) {
  if (d == 1) { Ai[0] = 1.0 / A[0]; return; }
  double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
  double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  double id = 1.0 / det;
  Ai[0] = c00 * id; Ai[1] = (A[2] * A[7] - A[1] * A[8]) * id; Ai[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  Ai[3] = c01 * id; Ai[4] = (A[0] * A[8] - A[2] * A[6]) * id; Ai[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  Ai[6] = c02 * id; Ai[7] = (A[1] * A[6] - A[0] * A[7]) * id; Ai[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

int rsf_mcmc_init(rsf_ctx *c, const rsf_mcmc_config *cfg, const double *q0, const double *data) {
  if (!c || !cfg || !q0 || !data) return fail(RSF_ERR_INVALID, "rsf_mcmc_init: NULL argument");
  if (!c->have_model) return fail(RSF_ERR_STATE, "rsf_mcmc_init: call rsf_set_model first");
  if (cfg->size != sizeof(rsf_mcmc_config)) return fail(RSF_ERR_INVALID, "rsf_mcmc_init: struct size mismatch");
  if (cfg->n_params != 1 && cfg->n_params != 3) return fail(RSF_ERR_UNSUPPORTED, "rsf_mcmc_init: n_params must be 1 or 3");
  if (cfg->n_chains < 1) return fail(RSF_ERR_INVALID, "rsf_mcmc_init: n_chains < 1");
  if (cfg->adapt_mode == RSF_ADAPT_REFERENCE_DICT && cfg->n_params != 1)
    return fail(RSF_ERR_UNSUPPORTED, "rsf_mcmc_init: reference_dict adaptation is defined for 1 parameter only");
  if (cfg->adapt_mode < 0 || cfg->adapt_mode > RSF_ADAPT_AM || (cfg->adapt_mode && cfg->adapt_interval < 2))
    return fail(RSF_ERR_INVALID, "rsf_mcmc_init: bad adapt_mode / adapt_interval");
  if (cfg->n_groups < 0 || (cfg->n_groups > 1 && cfg->n_chains % cfg->n_groups))
    return fail(RSF_ERR_INVALID, "rsf_mcmc_init: n_chains must be a multiple of n_groups");
  free_chains(c);
  c->mc = *cfg;
  c->n_groups = cfg->n_groups > 1 ? cfg->n_groups : 1;
  const int d = cfg->n_params;
  const int64_t C = cfg->n_chains;
  const int32_t N = c->nout;
  c->data = (double *)malloc(sizeof(double) * N * c->n_groups);
  c->q = (double *)malloc(sizeof(double) * C * d);
  c->ssq = (double *)malloc(sizeof(double) * C);
  c->std2 = (double *)malloc(sizeof(double) * C);
  c->V = (double *)malloc(sizeof(double) * C * d * d);
  c->wref = (double *)malloc(sizeof(double) * C * d);
  c->wsum = (double *)calloc(C * d, sizeof(double));
  c->wsq = (double *)calloc(C * d * d, sizeof(double));
  c->wn = (int32_t *)calloc(C, sizeof(int32_t));
  if (cfg->adapt_mode == RSF_ADAPT_REFERENCE_DICT) {
    if (cfg->adapt_interval > RSF_DICT_MAX_INTERVAL) { free_chains(c); return fail(RSF_ERR_UNSUPPORTED, "reference_dict adaptation keeps at most RSF_DICT_MAX_INTERVAL (128) samples per window"); }
    c->wbuf = (double *)calloc((size_t)C * cfg->adapt_interval, sizeof(double));
    if (!c->wbuf) { free_chains(c); return fail(RSF_ERR_NOMEM, "rsf_mcmc_init: out of memory"); }
  }
  if (!c->data || !c->q || !c->ssq || !c->std2 || !c->V || !c->wref || !c->wsum || !c->wsq || !c->wn) {
    free_chains(c);
    return fail(RSF_ERR_NOMEM, "rsf_mcmc_init: out of memory");
  }
  memcpy(c->data, data, sizeof(double) * N * c->n_groups);
  memcpy(c->q, q0, sizeof(double) * C * d);
  memcpy(c->wref, q0, sizeof(double) * C * d);
  const int plen = cfg->prior_len ? cfg->prior_len : d;
  const double fd = cfg->fd_rel_step;
  /* float32 mode, like the product: the forward-difference sensitivities (relative step 1e-6) and std2_0 stay float64;
   * only the initial SSq the sampler will compare against is the float32 solve's */
  rsf_ctx c64 = *c;
  c64.m.flags &= ~RSF_FLAG_FP32_SOLVE;
  const int f32 = (c->m.flags & RSF_FLAG_FP32_SOLVE) != 0;
  int nt = nthreads(c);
  (void)nt;
#pragma omp parallel num_threads(nt)
  {
    double *acc0 = (double *)malloc(sizeof(double) * N);
    double *accp = (double *)malloc(sizeof(double) * N * 3);
#pragma omp for schedule(static)
    for (int64_t i = 0; i < C; ++i) {
      const double *q = c->q + i * d;
      double a = d == 3 ? q[1] : c->m.a, b = d == 3 ? q[2] : c->m.b;
      double s0 = solve(&c64, q[0], a, b, data_of(c, i), acc0, 1);    /* MCMC.py:245-246, 468 */
      double qp[3], XtX[9], Xi[9];
      for (int p = 0; p < d; ++p) {                          /* MCMC.py:251-252 */
        double pq[3] = {q[0], a, b};
        pq[p] = pq[p] * (1 + fd);
        qp[p] = pq[p];
        solve(&c64, pq[0], pq[1], pq[2], NULL, accp + (int64_t)p * N, 1);
      }
      for (int p = 0; p < d; ++p)
        for (int r = 0; r < d; ++r) {
          double s = 0.0;
          for (int32_t k = 0; k < N; ++k) {                  /* MCMC.py:264-265; perturbed denominator */
            double xp = (accp[(int64_t)p * N + k] - acc0[k]) / (qp[p] * fd);
            double xr = (accp[(int64_t)r * N + k] - acc0[k]) / (qp[r] * fd);
            s += xp * xr;
          }
          XtX[p * d + r] = s;
        }
      double std2 = s0 / (double)(N - plen);                 /* MCMC.py:261 */
      if (d == 1) {
        sym_inverse(XtX, d, Xi);
        c->V[i] = std2 * Xi[0];                              /* MCMC.py:265-266 */
      } else {
        /* Three parameters (this build's extension): the reference's formula has no usable answer — the series depends on Dc
         * and a almost only through their product, hardly on b, so (X^T X)^-1 is astronomically wide along a ridge.  The box
         * prior enters as a Gaussian of equal variance in unit-cube coordinates: M = W X^T X W / sigma^2 + 12 I, V = W M^-1 W,
         * W = diag(hi - lo).  (csrc/rsf_kernels.h::initial_covariance has the full reasoning.) */
        double M[9], Mi[9], w[3];
        for (int p = 0; p < d; ++p) w[p] = cfg->hi[p] - cfg->lo[p];
        for (int p = 0; p < d; ++p)
          for (int r = 0; r < d; ++r) M[p * d + r] = (w[p] * XtX[p * d + r] * w[r]) * (1.0 / std2) + (p == r ? 12.0 : 0.0);
        sym_inverse(M, d, Mi);
        for (int p = 0; p < d; ++p)
          for (int r = 0; r < d; ++r) c->V[i * d * d + p * d + r] = w[p] * Mi[p * d + r] * w[r];
      }
      c->std2[i] = std2;
      c->ssq[i] = f32 ? solve(c, q[0], a, b, data_of(c, i), NULL, 0) : s0;
    }
    free(acc0);
    free(accp);
  }
  c->iters_done = 0;
  c->n_acc = c->n_eval = c->n_nonfinite = c->n_oob = 0;
  c->have_chains = 1;
  c->external_chains = 0;
  return RSF_OK;
}

int rsf_mcmc_get_state(rsf_ctx *c, double *q, double *ssq, double *std2, double *V) {
  if (!c) return fail(RSF_ERR_INVALID, "rsf_mcmc_get_state: NULL ctx");
  if (!c->have_chains) return fail(RSF_ERR_STATE, "rsf_mcmc_get_state: call rsf_mcmc_init first");
  const int d = c->mc.n_params;
  const int64_t C = c->mc.n_chains;
  if (q) memcpy(q, c->q, sizeof(double) * C * d);
  if (ssq) memcpy(ssq, c->ssq, sizeof(double) * C);
  if (std2) memcpy(std2, c->std2, sizeof(double) * C);
  if (V) memcpy(V, c->V, sizeof(double) * C * d * d);
  return RSF_OK;
}

int rsf_mcmc_set_state(rsf_ctx *c, const double *q, const double *ssq, const double *std2, const double *V) {
  if (!c) return fail(RSF_ERR_INVALID, "rsf_mcmc_set_state: NULL ctx");
  if (!c->have_chains) return fail(RSF_ERR_STATE, "rsf_mcmc_set_state: call rsf_mcmc_init first");
  const int d = c->mc.n_params;
  const int64_t C = c->mc.n_chains;
  if (q) memcpy(c->q, q, sizeof(double) * C * d);
  if (ssq) memcpy(c->ssq, ssq, sizeof(double) * C);
  if (std2) memcpy(c->std2, std2, sizeof(double) * C);
  if (V) memcpy(c->V, V, sizeof(double) * C * d * d);
  return RSF_OK;
}

/* np.add.reduce over n contiguous doubles as NumPy performs it (numpy/_core/src/umath/loops_utils.h.src, *_pairwise_sum):
 * fewer than 8 elements are added one by one from 0; up to 128 go through eight interleaved accumulators that are then
 * combined pairwise, the remainder added one by one; longer runs are halved (the first half a multiple of 8) recursively.
 * The ORDER is the point: whether the mean of a window of identical samples comes out as that sample — np.cov exactly 0,
 * np.linalg.cholesky raises, the reference keeps its covariance (MCMC.py:524-527) — or one ulp off — np.cov ~1e-31,
 * the Cholesky "succeeds" and the reference's proposal collapses to ~1e-8 until the next windows widen it again — depends on it. */
static double np_pairwise_sum(const double *a, int64_t n) {
  if (n < 8) {
    double res = 0.0;
    for (int64_t i = 0; i < n; ++i) res += a[i];
    return res;
  }
  if (n <= 128) {
    double r[8];
    int64_t i;
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int64_t n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

/* np.cov(x) of ONE variable with n observations, as numpy/lib/_function_base_impl.py::cov forms it: avg = sum / n (the sum
 * above), X -= avg, c = dot(X, X^T), c *= 1 / (n - 1).  (The dot product's own summation order is BLAS's and moves c by an
 * ulp at most; the mean's order decides zero against non-zero.)  This is the arithmetic behind MCMC.py:200-203 for the
 * reference's one-parameter chains; `reference_dict` adaptation uses it so that degenerate windows behave as they do there. */
static double np_cov_1d(const double *x, int32_t n) {
  const double avg = np_pairwise_sum(x, n) / (double)n;
  double c = 0.0;
  for (int32_t k = 0; k < n; ++k) { double dlt = x[k] - avg; c += dlt * dlt; }
  return c * (1.0 / (double)(n - 1));
}

/* One chain, n_iters iterations of MCMC.py:494-527. */
static void run_chain(rsf_ctx *c, int64_t i, int64_t n_iters, const double *zs, const double *us,
                      const double *gs, const double *sn, double *tq, double *ts, uint8_t *ta, int64_t *acc_cnt,
                      int64_t *eval_cnt, int64_t *nonfinite_cnt, int64_t *oob_cnt) {
  const rsf_mcmc_config *mc = &c->mc;
  const int d = mc->n_params;
  const int64_t C = mc->n_chains;
  const uint64_t gid = (uint64_t)(mc->chain_offset + i);
  const double shape = 0.5 * (mc->n0 + (double)c->nout);     /* MCMC.py:158 */
  double q[3], V[9], L[9];
  double ssq = c->ssq[i], std2 = c->std2[i];
  memcpy(q, c->q + i * d, sizeof(double) * d);
  memcpy(V, c->V + i * d * d, sizeof(double) * d * d);
  for (int64_t n = 0; n < n_iters; ++n) {
    const uint32_t it = (uint32_t)(c->iters_done + n);
    double z[4] = {0, 0, 0, 0}, qn[3];
    /* proposal, MCMC.py:497 (d == 1: q + sqrt(V) z) */
    if (zs) {
      for (int p = 0; p < d; ++p) z[p] = zs[(n * C + i) * d + p];
    } else {
      uint32_t w[4];
      draw_words(mc->seed, gid, it, SLOT_Z01, w);
      normal_pair(w, &z[0], &z[1]);
      if (d > 2) { draw_words(mc->seed, gid, it, SLOT_Z2, w); normal_pair(w, &z[2], &z[3]); }
    }
    chol_lower(V, d, L);
    for (int p = 0; p < d; ++p) {
      double s = q[p];
      for (int r = 0; r <= p; ++r) s += L[p * d + r] * z[r];
      qn[p] = s;
    }
    /* acceptreject, MCMC.py:318-333 */
    int inb = 1, accept = 0;
    for (int p = 0; p < d; ++p) inb = inb && (qn[p] > mc->lo[p]) && (qn[p] < mc->hi[p]);
    if (!inb) ++*oob_cnt;
    if (inb) {
      double ssqn = sn ? sn[n * C + i] : ssq_of(c, i, qn, d);   /* rsf_mcmc_replay_ssq: the caller evaluated the model */
      double u;
      if (us) u = us[n * C + i];
      else { uint32_t w[4]; draw_words(mc->seed, gid, it, SLOT_U, w); u = u53(w[0], w[1]); }
      double logalpha = 0.5 * (ssq - ssqn) / std2;
      if (logalpha > 0.0) logalpha = 0.0;                    /* np.clip(., -inf, 0) */
      accept = logalpha > log(u);                            /* NaN => False */
      ++*eval_cnt;
      if (!isfinite(ssqn)) ++*nonfinite_cnt;
      if (accept) { ssq = ssqn; memcpy(q, qn, sizeof(double) * d); ++*acc_cnt; }
    }
    /* update_standard_deviation, MCMC.py:158-160 (uses the post-accept SSq) */
    {
      double bval = 0.5 * (mc->n0 * std2 + ssq);
      double g = gs ? gs[n * C + i] : gamma_draw(mc->seed, gid, it, shape);
      std2 = 1.0 / (g * (1.0 / bval));
    }
    if (tq) memcpy(tq + (n * C + i) * d, q, sizeof(double) * d);
    if (ts) ts[n * C + i] = std2;
    if (ta) ta[n * C + i] = (uint8_t)accept;
    /* adaptation, MCMC.py:523-527 with update_covariance_matrix :200-204 */
    if (mc->adapt_mode != RSF_ADAPT_NONE) {
      double *wr = c->wref + i * d, *ws = c->wsum + i * d, *wq = c->wsq + i * d * d;
      for (int p = 0; p < d; ++p) {
        ws[p] += q[p] - wr[p];
        for (int r = 0; r < d; ++r) wq[p * d + r] += (q[p] - wr[p]) * (q[r] - wr[r]);
      }
      c->wn[i] += 1;
      if (c->wbuf) c->wbuf[i * mc->adapt_interval + (c->iters_done + n) % mc->adapt_interval] = q[0];
      if ((c->iters_done + n + 1) % mc->adapt_interval == 0) {
        const double nn = (double)c->wn[i];
        double cov[9], Vn[9], Ln[9];
        if (c->wn[i] >= 2) {
          for (int p = 0; p < d; ++p)
            for (int r = 0; r < d; ++r)
              cov[p * d + r] = (wq[p * d + r] - ws[p] * ws[r] / nn) / (nn - 1.0); /* np.cov, ddof=1 */
          if (mc->adapt_mode == RSF_ADAPT_REFERENCE_DICT) {
            /* d := len(qpriors.keys()) (MCMC.py:200; 2 for {1: lo, 2: hi}); the Cholesky FACTOR becomes the next covariance.
             * The window is the last adapt_interval samples in order (the ring is full and starts at slot 0 whenever an
             * adaptation is due), its covariance np.cov's own arithmetic. */
            Vn[0] = 2.38 * 2.38 / (double)(mc->prior_len > 0 ? mc->prior_len : 2) * np_cov_1d(c->wbuf + i * mc->adapt_interval, mc->adapt_interval);
            if (chol_lower(Vn, 1, Ln)) V[0] = Ln[0];
          } else {
            /* am: corrected adaptive Metropolis (Haario et al. 2001) — the covariance of the chain's WHOLE history (the sums are
             * never reset: adaptation diminishes, the chain keeps the posterior as its limit) plus eps_p = (1e-6 (hi_p - lo_p))^2
             * on the diagonal, which keeps it positive definite whatever the history holds */
            for (int p = 0; p < d; ++p)
              for (int r = 0; r < d; ++r) {
                double w = 1e-6 * (mc->hi[p] - mc->lo[p]);
                Vn[p * d + r] = 2.38 * 2.38 / (double)d * (cov[p * d + r] + (p == r ? w * w : 0.0));
              }
            if (chol_lower(Vn, d, Ln)) memcpy(V, Vn, sizeof(double) * d * d);
          }
        }
      }
    }
  }
  c->ssq[i] = ssq;
  c->std2[i] = std2;
  memcpy(c->q + i * d, q, sizeof(double) * d);
  memcpy(c->V + i * d * d, V, sizeof(double) * d * d);
}

static int run_all(rsf_ctx *c, int64_t n_iters, const double *z, const double *u, const double *g, const double *sn,
                   double *tq, double *ts, uint8_t *ta) {
  if (!c || n_iters < 0) return fail(RSF_ERR_INVALID, "rsf_mcmc_run: bad argument");
  if (!c->have_chains) return fail(RSF_ERR_STATE, "rsf_mcmc_run: call rsf_mcmc_init first");
  if (c->external_chains && !sn)
    return fail(RSF_ERR_STATE, "chains made by rsf_mcmc_init_state have no observation: advance them with rsf_mcmc_replay_ssq");
  int64_t a = 0, e = 0, nf = 0, ob = 0;
  int nt = nthreads(c);
  (void)nt;
#pragma omp parallel for schedule(static) reduction(+ : a, e, nf, ob) num_threads(nt)
  for (int64_t i = 0; i < c->mc.n_chains; ++i) run_chain(c, i, n_iters, z, u, g, sn, tq, ts, ta, &a, &e, &nf, &ob);
  c->n_acc += a; c->n_eval += e; c->n_nonfinite += nf; c->n_oob += ob;
  c->iters_done += n_iters;
  return RSF_OK;
}

int rsf_mcmc_run(rsf_ctx *c, int64_t n_iters, double *tq, double *ts, uint8_t *ta) {
  return run_all(c, n_iters, NULL, NULL, NULL, NULL, tq, ts, ta);
}

int rsf_mcmc_replay(rsf_ctx *c, int64_t n_iters, const double *z, const double *u, const double *g,
                    double *tq, double *ts, uint8_t *ta) {
  if (!z || !u || !g) return fail(RSF_ERR_INVALID, "rsf_mcmc_replay: z, u and g are required");
  return run_all(c, n_iters, z, u, g, NULL, tq, ts, ta);
}

int rsf_mcmc_replay_ssq(rsf_ctx *c, int64_t n_iters, const double *z, const double *u, const double *g, const double *ssq_new,
                        double *tq, double *ts, uint8_t *ta) {
  if (!z || !u || !g || !ssq_new) return fail(RSF_ERR_INVALID, "rsf_mcmc_replay_ssq: z, u, g and ssq_new are required");
  return run_all(c, n_iters, z, u, g, ssq_new, tq, ts, ta);
}

/* chains from an explicit state: MCMC.py:464-468 done by the caller (any model object), MCMC.py:494-527 here */
int rsf_mcmc_init_state(rsf_ctx *c, const rsf_mcmc_config *cfg, const double *q, const double *ssq, const double *std2, const double *V) {
  if (!c || !cfg || !q || !ssq || !std2 || !V) return fail(RSF_ERR_INVALID, "rsf_mcmc_init_state: NULL argument");
  if (cfg->size != sizeof(rsf_mcmc_config)) return fail(RSF_ERR_INVALID, "rsf_mcmc_init_state: struct size mismatch");
  if (cfg->n_params != 1 && cfg->n_params != 3) return fail(RSF_ERR_UNSUPPORTED, "rsf_mcmc_init_state: n_params must be 1 or 3");
  if (cfg->n_chains < 1) return fail(RSF_ERR_INVALID, "rsf_mcmc_init_state: n_chains < 1");
  if (cfg->adapt_mode == RSF_ADAPT_REFERENCE_DICT && cfg->n_params != 1)
    return fail(RSF_ERR_UNSUPPORTED, "rsf_mcmc_init_state: reference_dict adaptation is defined for 1 parameter only");
  if (cfg->adapt_mode < 0 || cfg->adapt_mode > RSF_ADAPT_AM || (cfg->adapt_mode && cfg->adapt_interval < 2))
    return fail(RSF_ERR_INVALID, "rsf_mcmc_init_state: bad adapt_mode / adapt_interval");
  free_chains(c);
  c->mc = *cfg;
  c->n_groups = 1;
  const int d = cfg->n_params;
  const int64_t C = cfg->n_chains;
  c->q = (double *)malloc(sizeof(double) * C * d);
  c->ssq = (double *)malloc(sizeof(double) * C);
  c->std2 = (double *)malloc(sizeof(double) * C);
  c->V = (double *)malloc(sizeof(double) * C * d * d);
  c->wref = (double *)malloc(sizeof(double) * C * d);
  c->wsum = (double *)calloc(C * d, sizeof(double));
  c->wsq = (double *)calloc(C * d * d, sizeof(double));
  c->wn = (int32_t *)calloc(C, sizeof(int32_t));
  if (cfg->adapt_mode == RSF_ADAPT_REFERENCE_DICT) {
    if (cfg->adapt_interval > RSF_DICT_MAX_INTERVAL) { free_chains(c); return fail(RSF_ERR_UNSUPPORTED, "reference_dict adaptation keeps at most RSF_DICT_MAX_INTERVAL (128) samples per window"); }
    c->wbuf = (double *)calloc((size_t)C * cfg->adapt_interval, sizeof(double));
    if (!c->wbuf) { free_chains(c); return fail(RSF_ERR_NOMEM, "rsf_mcmc_init_state: out of memory"); }
  }
  if (!c->q || !c->ssq || !c->std2 || !c->V || !c->wref || !c->wsum || !c->wsq || !c->wn) {
    free_chains(c);
    return fail(RSF_ERR_NOMEM, "rsf_mcmc_init_state: out of memory");
  }
  memcpy(c->q, q, sizeof(double) * C * d);
  memcpy(c->wref, q, sizeof(double) * C * d);
  memcpy(c->ssq, ssq, sizeof(double) * C);
  memcpy(c->std2, std2, sizeof(double) * C);
  memcpy(c->V, V, sizeof(double) * C * d * d);
  c->iters_done = 0;
  c->n_acc = c->n_eval = c->n_nonfinite = c->n_oob = 0;
  c->have_chains = 1;
  c->external_chains = 1;
  return RSF_OK;
}

/* the proposal the next iteration makes from z (MCMC.py:497) and its box test (MCMC.py:318-320); nothing changes */
int rsf_mcmc_propose(rsf_ctx *c, const double *z, double *q_new, uint8_t *in_bounds) {
  if (!c || !z || !q_new || !in_bounds) return fail(RSF_ERR_INVALID, "rsf_mcmc_propose: NULL argument");
  if (!c->have_chains) return fail(RSF_ERR_STATE, "rsf_mcmc_propose: call rsf_mcmc_init or rsf_mcmc_init_state first");
  const rsf_mcmc_config *mc = &c->mc;
  const int d = mc->n_params;
  for (int64_t i = 0; i < mc->n_chains; ++i) {
    double L[9];
    int inb = 1;
    chol_lower(c->V + i * d * d, d, L);
    for (int p = 0; p < d; ++p) {
      double s = c->q[i * d + p];
      for (int r = 0; r <= p; ++r) s += L[p * d + r] * z[i * d + r];
      q_new[i * d + p] = s;
      inb = inb && (s > mc->lo[p]) && (s < mc->hi[p]);
    }
    in_bounds[i] = (uint8_t)inb;
  }
  return RSF_OK;
}

int rsf_mcmc_stats(rsf_ctx *c, int64_t *n_acc, int64_t *n_eval, int64_t *n_nonfinite, int64_t *n_done) {
  if (!c) return fail(RSF_ERR_INVALID, "rsf_mcmc_stats: NULL ctx");
  if (!c->have_chains) return fail(RSF_ERR_STATE, "rsf_mcmc_stats: call rsf_mcmc_init first");
  if (n_acc) *n_acc = c->n_acc;
  if (n_eval) *n_eval = c->n_eval;
  if (n_nonfinite) *n_nonfinite = c->n_nonfinite;
  if (n_done) *n_done = c->iters_done;
  return RSF_OK;
}

/* the counters that are properties of the chains; the wave-level ones describe the HIP kernels and are zero here */
int rsf_mcmc_counters(rsf_ctx *c, int64_t *out, int32_t n) {
  if (!c || !out || n < 0) return fail(RSF_ERR_INVALID, "rsf_mcmc_counters: bad argument");
  if (!c->have_chains) return fail(RSF_ERR_STATE, "rsf_mcmc_counters: call rsf_mcmc_init first");
  int64_t v[RSF_CNT_COUNT] = {0};
  v[RSF_CNT_ACCEPTED] = c->n_acc; v[RSF_CNT_EVALUATED] = c->n_eval; v[RSF_CNT_NONFINITE] = c->n_nonfinite;
  v[RSF_CNT_OUT_OF_BOUNDS] = c->n_oob;
  for (int32_t k = 0; k < n && k < RSF_CNT_COUNT; ++k) out[k] = v[k];
  return RSF_OK;
}

/* ------------------------------------------------------------------------------------ */
/* posterior post-processing (RSF.plot_dist, RSF.py:717-746)                              */
/* ------------------------------------------------------------------------------------ */
int rsf_pool_summary(rsf_ctx *c, int64_t n, const double *x, int64_t stride, double *out) {
  if (!c || !x || !out || n < 1 || stride < 1) return fail(RSF_ERR_INVALID, "rsf_pool_summary: bad argument");
  double mean = 0.0, mn = x[0], mx = x[0], ss = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    double v = x[i * stride];
    mean += v;
    if (v < mn) mn = v;
    if (v > mx) mx = v;
  }
  mean /= (double)n;
  for (int64_t i = 0; i < n; ++i) { double dlt = x[i * stride] - mean; ss += dlt * dlt; }  /* two-pass, like np.cov */
  out[0] = (double)n; out[1] = mean; out[2] = n > 1 ? ss / (double)(n - 1) : 0.0; out[3] = mn; out[4] = mx;
  return RSF_OK;
}

int rsf_pool_kde(rsf_ctx *c, int64_t n, const double *x, int64_t stride, int32_t m, const double *grid, double bw_factor,
                 double *density) {
  if (!c || !x || !grid || !density || n < 2 || m < 1 || stride < 1) return fail(RSF_ERR_INVALID, "rsf_pool_kde: bad argument");
  double s[5];
  int rc = rsf_pool_summary(c, n, x, stride, s);
  if (rc) return rc;
  double factor = bw_factor > 0.0 ? bw_factor : pow((double)n, -1.0 / 5.0);  /* scipy scotts_factor, d = 1 */
  double cov = s[2] * factor * factor;
  if (!(cov > 0.0)) return fail(RSF_ERR_INVALID, "rsf_pool_kde: the samples have zero variance (singular KDE)");
  double norm = 1.0 / ((double)n * sqrt(2.0 * 3.14159265358979323846 * cov)), inv2c = 0.5 / cov;
  int nt = nthreads(c);
  (void)nt;
#pragma omp parallel for schedule(static) num_threads(nt)
  for (int32_t j = 0; j < m; ++j) {
    double acc = 0.0;
    for (int64_t i = 0; i < n; ++i) { double dlt = grid[j] - x[i * stride]; acc += exp(-dlt * dlt * inv2c); }
    density[j] = acc * norm;
  }
  return RSF_OK;
}

/* Fixed-bin histogram, numpy.histogram semantics (include/rsf_abi.h): counts[0] below lo, counts[1..nbins], counts[nbins+1]
 * above hi or NaN.  numpy's rule restated (numpy/lib/_histograms_impl.py, uniform bins): a first index from
 * (x - lo)/(hi - lo) * nbins, then the correction against the bin edges np.linspace(lo, hi, nbins + 1) = b*step + lo
 * (last edge = hi): one down if x < edge[b], one up if x >= edge[b+1] (except in the last bin, closed at hi). */
static double hist_edge(int32_t b, double lo, double hi, double step, int32_t nbins) {
  return b == nbins ? hi : (double)b * step + lo;   /* -ffp-contract=off: two roundings, like numpy */
}

int rsf_pool_histogram(rsf_ctx *c, int64_t n, const double *x, int64_t stride, int32_t nbins, double lo, double hi, double *counts) {
  if (!c || !x || !counts || n < 1 || stride < 1 || nbins < 1 || nbins > 4096 || !(hi > lo) || !isfinite(hi - lo))
    return fail(RSF_ERR_INVALID, "rsf_pool_histogram: bad argument (1 <= nbins <= 4096, finite lo < hi)");
  const double step = (hi - lo) / (double)nbins;
  for (int32_t b = 0; b < nbins + 2; ++b) counts[b] = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    const double v = x[i * stride];
    int32_t b;
    if (v < lo) b = 0;
    else if (!(v <= hi)) b = nbins + 1;
    else {
      b = (int32_t)(((v - lo) / (hi - lo)) * (double)nbins);
      if (b >= nbins) b = nbins - 1;
      if (v < hist_edge(b, lo, hi, step, nbins)) --b;
      if (b != nbins - 1 && v >= hist_edge(b + 1, lo, hi, step, nbins)) ++b;
      b += 1;
    }
    counts[b] += 1.0;
  }
  return RSF_OK;
}

/* Pool collectives (include/rsf_abi.h).  The checker itself is one process on host memory: world = 1 is a copy.  For
 * world > 1 it binds — exactly like the product, through RSF_RCCL_LIB — an implementation of the nccl* entry points that
 * works on host pointers (the test-suite's tests/c/fake_rccl.c; the real RCCL needs device memory and is never loaded
 * here), so that the same multi-rank host code can be run against both libraries. */
typedef struct { char internal[RSF_COMM_ID_BYTES]; } oracle_uid;
static struct {
  void *h;
  int (*get_unique_id)(oracle_uid *);
  int (*comm_init_rank)(void **, int, oracle_uid, int);
  int (*comm_init_all)(void **, int, const int *);
  int (*comm_destroy)(void *);
  int (*group_start)(void);
  int (*group_end)(void);
  int (*all_gather)(const void *, void *, size_t, int, void *, void *);
  int (*all_reduce)(const void *, void *, size_t, int, int, void *, void *);
} g_nccl;
static pthread_once_t g_nccl_once = PTHREAD_ONCE_INIT;
enum { NCCL_FLOAT64 = 8, NCCL_SUM = 0 };

static void bind_nccl(void) {
  const char *path = getenv("RSF_RCCL_LIB");
  if (!path || !*path) return;
  void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!h) return;
  *(void **)&g_nccl.get_unique_id = dlsym(h, "ncclGetUniqueId");
  *(void **)&g_nccl.comm_init_rank = dlsym(h, "ncclCommInitRank");
  *(void **)&g_nccl.comm_init_all = dlsym(h, "ncclCommInitAll");
  *(void **)&g_nccl.comm_destroy = dlsym(h, "ncclCommDestroy");
  *(void **)&g_nccl.group_start = dlsym(h, "ncclGroupStart");
  *(void **)&g_nccl.group_end = dlsym(h, "ncclGroupEnd");
  *(void **)&g_nccl.all_gather = dlsym(h, "ncclAllGather");
  *(void **)&g_nccl.all_reduce = dlsym(h, "ncclAllReduce");
  if (g_nccl.get_unique_id && g_nccl.comm_init_rank && g_nccl.comm_init_all && g_nccl.comm_destroy && g_nccl.group_start &&
      g_nccl.group_end && g_nccl.all_gather && g_nccl.all_reduce)
    g_nccl.h = h;
}

static int have_nccl(void) {
  pthread_once(&g_nccl_once, bind_nccl);
  return g_nccl.h != NULL;
}

int rsf_comm_unique_id(uint8_t id[RSF_COMM_ID_BYTES]) {
  if (!id) return fail(RSF_ERR_INVALID, "rsf_comm_unique_id: NULL argument");
  memset(id, 0, RSF_COMM_ID_BYTES);
  if (have_nccl()) {
    oracle_uid u;
    if (g_nccl.get_unique_id(&u)) return fail(RSF_ERR_DEVICE, "rsf_comm_unique_id: ncclGetUniqueId failed");
    memcpy(id, u.internal, RSF_COMM_ID_BYTES);
  }
  return RSF_OK;
}

int rsf_comm_init(rsf_ctx *c, int32_t world, int32_t rank, const uint8_t id[RSF_COMM_ID_BYTES]) {
  if (!c || world < 1 || rank < 0 || rank >= world) return fail(RSF_ERR_INVALID, "rsf_comm_init: bad argument");
  if (c->world) return fail(RSF_ERR_STATE, "rsf_comm_init: this ctx already has a communicator (rsf_comm_destroy first)");
  if (world > 1) {
    if (!have_nccl())
      return fail(RSF_ERR_UNSUPPORTED, "rsf_comm_init: the CPU oracle is single-process (world = 1 only) unless RSF_RCCL_LIB "
                                       "names a host-memory implementation of the nccl entry points");
    if (!id) return fail(RSF_ERR_INVALID, "rsf_comm_init: world > 1 needs the id from rsf_comm_unique_id on rank 0");
    oracle_uid u;
    memcpy(u.internal, id, RSF_COMM_ID_BYTES);
    if (g_nccl.comm_init_rank(&c->comm, world, u, rank)) return fail(RSF_ERR_DEVICE, "rsf_comm_init: ncclCommInitRank failed");
  }
  c->world = world;
  c->rank = rank;
  return RSF_OK;
}

int rsf_comm_destroy(rsf_ctx *c) {
  if (!c) return fail(RSF_ERR_INVALID, "rsf_comm_destroy: NULL ctx");
  if (c->comm) { g_nccl.comm_destroy(c->comm); c->comm = NULL; }
  c->world = 0;
  c->rank = 0;
  return RSF_OK;
}

int rsf_pool_allgather(rsf_ctx *c, const double *send, int64_t count, double *recv) {
  if (!c || !send || !recv || count < 1) return fail(RSF_ERR_INVALID, "rsf_pool_allgather: bad argument");
  if (!c->world) return fail(RSF_ERR_STATE, "rsf_pool_allgather: call rsf_comm_init first");
  if (c->comm) {
    if (g_nccl.all_gather(send, recv, (size_t)count, NCCL_FLOAT64, c->comm, NULL)) return fail(RSF_ERR_DEVICE, "rsf_pool_allgather: ncclAllGather failed");
  } else if (recv != send) {
    memmove(recv, send, (size_t)count * sizeof(double));
  }
  return RSF_OK;
}

int rsf_pool_allreduce_sum(rsf_ctx *c, double *buf, int64_t count) {
  if (!c || !buf || count < 1) return fail(RSF_ERR_INVALID, "rsf_pool_allreduce_sum: bad argument");
  if (!c->world) return fail(RSF_ERR_STATE, "rsf_pool_allreduce_sum: call rsf_comm_init first");
  if (c->comm && g_nccl.all_reduce(buf, buf, (size_t)count, NCCL_FLOAT64, NCCL_SUM, c->comm, NULL))
    return fail(RSF_ERR_DEVICE, "rsf_pool_allreduce_sum: ncclAllReduce failed");
  return RSF_OK;
}

static int check_group(rsf_ctx *const *ctxs, int32_t n, const char *who, int need_comm) {
  char msg[160];
  if (!ctxs || n < 1) { snprintf(msg, sizeof msg, "%s: bad argument", who); return fail(RSF_ERR_INVALID, msg); }
  for (int32_t i = 0; i < n; ++i) {
    if (!ctxs[i]) { snprintf(msg, sizeof msg, "%s: ctxs[%d] is NULL", who, i); return fail(RSF_ERR_INVALID, msg); }
    for (int32_t j = 0; j < i; ++j)
      if (ctxs[j] == ctxs[i]) { snprintf(msg, sizeof msg, "%s: ctxs[%d] and ctxs[%d] are the same ctx", who, j, i); return fail(RSF_ERR_INVALID, msg); }
    if (need_comm && (ctxs[i]->world != n || ctxs[i]->rank != i || !ctxs[i]->comm)) {
      snprintf(msg, sizeof msg, "%s: ctxs[%d] is not rank %d of a %d-rank group made by rsf_comm_init_all", who, i, i, n);
      return fail(RSF_ERR_STATE, msg);
    }
  }
  return RSF_OK;
}

int rsf_comm_init_all(rsf_ctx *const *ctxs, int32_t n) {
  int rc = check_group(ctxs, n, "rsf_comm_init_all", 0);
  if (rc) return rc;
  for (int32_t i = 0; i < n; ++i)
    if (ctxs[i]->world) return fail(RSF_ERR_STATE, "rsf_comm_init_all: a ctx already has a communicator (rsf_comm_destroy first)");
  if (!have_nccl())
    return fail(RSF_ERR_UNSUPPORTED, "rsf_comm_init_all: the CPU oracle needs RSF_RCCL_LIB (a host-memory implementation of the nccl entry points)");
  void *comms[64];
  int devs[64] = {0};
  if (n > 64) return fail(RSF_ERR_INVALID, "rsf_comm_init_all: at most 64 ctxs");
  if (g_nccl.comm_init_all(comms, n, devs)) return fail(RSF_ERR_DEVICE, "rsf_comm_init_all: ncclCommInitAll failed");
  for (int32_t i = 0; i < n; ++i) { ctxs[i]->comm = comms[i]; ctxs[i]->world = n; ctxs[i]->rank = i; }
  return RSF_OK;
}

int rsf_pool_allgather_all(rsf_ctx *const *ctxs, int32_t n, const double *const *send, int64_t count, double *const *recv) {
  int rc = check_group(ctxs, n, "rsf_pool_allgather_all", 1);
  if (rc) return rc;
  if (!send || !recv || count < 1) return fail(RSF_ERR_INVALID, "rsf_pool_allgather_all: bad argument");
  for (int32_t i = 0; i < n; ++i)
    if (!send[i] || !recv[i]) return fail(RSF_ERR_INVALID, "rsf_pool_allgather_all: a send / recv pointer is NULL");
  g_nccl.group_start();
  for (int32_t i = 0; i < n; ++i)
    if (g_nccl.all_gather(send[i], recv[i], (size_t)count, NCCL_FLOAT64, ctxs[i]->comm, NULL)) {
      g_nccl.group_end();
      return fail(RSF_ERR_DEVICE, "rsf_pool_allgather_all: ncclAllGather failed");
    }
  if (g_nccl.group_end()) return fail(RSF_ERR_DEVICE, "rsf_pool_allgather_all: ncclGroupEnd failed");
  return RSF_OK;
}

int rsf_pool_allreduce_sum_all(rsf_ctx *const *ctxs, int32_t n, double *const *bufs, int64_t count) {
  int rc = check_group(ctxs, n, "rsf_pool_allreduce_sum_all", 1);
  if (rc) return rc;
  if (!bufs || count < 1) return fail(RSF_ERR_INVALID, "rsf_pool_allreduce_sum_all: bad argument");
  for (int32_t i = 0; i < n; ++i)
    if (!bufs[i]) return fail(RSF_ERR_INVALID, "rsf_pool_allreduce_sum_all: a buffer pointer is NULL");
  g_nccl.group_start();
  for (int32_t i = 0; i < n; ++i)
    if (g_nccl.all_reduce(bufs[i], bufs[i], (size_t)count, NCCL_FLOAT64, NCCL_SUM, ctxs[i]->comm, NULL)) {
      g_nccl.group_end();
      return fail(RSF_ERR_DEVICE, "rsf_pool_allreduce_sum_all: ncclAllReduce failed");
    }
  if (g_nccl.group_end()) return fail(RSF_ERR_DEVICE, "rsf_pool_allreduce_sum_all: ncclGroupEnd failed");
  return RSF_OK;
}

/* update_covariance_matrix, MCMC.py:200-204, for one window (the adaptation block of run_chain on a given set of samples) */
int rsf_mcmc_adapt(int32_t d, int32_t n, const double *window, int32_t adapt_mode, int32_t prior_len, double *V_out) {
  if ((d != 1 && d != 3) || n < 1 || !window || !V_out || (adapt_mode != RSF_ADAPT_REFERENCE_DICT && adapt_mode != RSF_ADAPT_AM))
    return fail(RSF_ERR_INVALID, "rsf_mcmc_adapt: bad argument");
  if (adapt_mode == RSF_ADAPT_REFERENCE_DICT && d != 1)
    return fail(RSF_ERR_UNSUPPORTED, "rsf_mcmc_adapt: reference_dict adaptation is defined for 1 parameter only");
  double mean[3] = {0, 0, 0}, cov[9], Vn[9], Ln[9];
  for (int32_t k = 0; k < n; ++k)
    for (int p = 0; p < d; ++p) mean[p] += window[k * d + p];
  for (int p = 0; p < d; ++p) mean[p] /= (double)n;
  for (int p = 0; p < d; ++p)
    for (int r = 0; r < d; ++r) {
      double s = 0.0;
      for (int32_t k = 0; k < n; ++k) s += (window[k * d + p] - mean[p]) * (window[k * d + r] - mean[r]);
      cov[p * d + r] = n > 1 ? s / (double)(n - 1) : 0.0;  /* np.cov, ddof = 1 */
    }
  const double scale = adapt_mode == RSF_ADAPT_REFERENCE_DICT ? 2.38 * 2.38 / (double)(prior_len > 0 ? prior_len : 2) : 2.38 * 2.38 / (double)d;
  if (adapt_mode == RSF_ADAPT_REFERENCE_DICT) {
    if (n > RSF_DICT_MAX_INTERVAL) return fail(RSF_ERR_UNSUPPORTED, "rsf_mcmc_adapt: reference_dict windows hold at most RSF_DICT_MAX_INTERVAL (128) samples");
    if (n > 1) cov[0] = np_cov_1d(window, n);  /* np.cov's own arithmetic, like the sampler's (run_chain) */
  }
  for (int e = 0; e < d * d; ++e) Vn[e] = scale * cov[e];
  if (n < 2 || !chol_lower(Vn, d, Ln)) return fail(RSF_ERR_NOT_POSDEF, "rsf_mcmc_adapt: the window's covariance is not positive definite");
  memcpy(V_out, adapt_mode == RSF_ADAPT_REFERENCE_DICT ? Ln : Vn, sizeof(double) * d * d);
  return RSF_OK;
}

int rsf_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  if (!ctr || !key || !out) return fail(RSF_ERR_INVALID, "rsf_philox4x32_10: NULL argument");
  philox_block(ctr, key, out);
  return RSF_OK;
}

int rsf_mcmc_draws(uint64_t seed, int64_t chain, int64_t iteration, int32_t d, double shape, double *z,
                   double *u, double *g) {
  if (d < 1 || d > 3) return fail(RSF_ERR_INVALID, "rsf_mcmc_draws: n_params out of range");
  uint32_t w[4];
  double zz[4] = {0, 0, 0, 0};
  draw_words(seed, (uint64_t)chain, (uint32_t)iteration, SLOT_Z01, w);
  normal_pair(w, &zz[0], &zz[1]);
  if (d > 2) { draw_words(seed, (uint64_t)chain, (uint32_t)iteration, SLOT_Z2, w); normal_pair(w, &zz[2], &zz[3]); }
  if (z) for (int p = 0; p < d; ++p) z[p] = zz[p];
  if (u) { draw_words(seed, (uint64_t)chain, (uint32_t)iteration, SLOT_U, w); *u = u53(w[0], w[1]); }
  if (g) *g = gamma_draw(seed, (uint64_t)chain, (uint32_t)iteration, shape);
  return RSF_OK;
}
