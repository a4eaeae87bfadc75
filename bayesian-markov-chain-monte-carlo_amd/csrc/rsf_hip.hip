// rsf_hip.hip — gfx950 (MI355X / CDNA4) implementation of include/rsf_abi.h: the host side of the C ABI.
// The kernels are in rsf_kernels.h (device building blocks: rsf_device*.h, rsf_math.h).
//
// There is no host fallback in this file: every entry point either runs on the GPU or fails.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is bound with dlopen (see struct Rccl)
#include <dlfcn.h>

#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>

#include "../../include/rsf_abi.h"
#include "rsf_kernels.h"

using rsf::Consts;
using namespace rsfk;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) return fail(RSF_ERR_DEVICE, "%s -> %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
};

struct DeviceGuard {  // run on the ctx device, restore the caller's current device afterwards
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
    if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

struct rsf_ctx {
  rsf_config cfg{};
  int device = 0;
  hipStream_t stream = nullptr;
  int block = kMaxBlock;
  // model
  bool have_model = false;
  rsf_model m{};
  int32_t nout = 0;
  double delta_t = 0, h = 0;
  int32_t kc = 0, nchunks = 0;
  int32_t kc32 = 0, nchunks32 = 0;  // the float32 SAMPLER's own chunking: its tables are floats, twice as many fit the budget
  size_t lds_bytes = 0;
  DevBuf vl;
  // chains
  bool have_chains = false;
  bool external_chains = false;  // made by rsf_mcmc_init_state: no observation, advanced by rsf_mcmc_replay_ssq only
  rsf_mcmc_config mc{};
  DevBuf data, q, ssq, std2, V, wref, wsum, wsq, wn, wbuf, stats;
  int64_t group_chains = 0;  // chains per observation group (0: one series)
  int64_t iters_done = 0;
  // staging for RSF_MEM_HOST callers
  DevBuf stage[8];
  // drain pipeline of rsf_mcmc_run for RSF_MEM_HOST callers: the trace of launch k is copied out on its own stream
  // while launch k+1 computes
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_done[2] = {nullptr, nullptr};
  // one-proposal replay as a captured graph (the drop-in single-chain MCMC.sample() is launch-bound)
  struct ReplayGraph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipGraphNode_t kernel = nullptr;
    void *host = nullptr;   // pinned: [z C*d][u C][g C] | [tq C*d][ts C][ta C bytes]
    void *dev = nullptr;
    int64_t C = 0;
    int d = 0;
    const void *fn = nullptr;
    size_t lds = 0;
    int block = 0;
  } rg;
  // posterior-pool communicator (one process per GPU)
  int32_t world = 0, rank = 0;  // world 0: rsf_comm_init not called
  ncclComm_t comm = nullptr;
  DevBuf pool;  // workspace of the posterior post-processing kernels
};

namespace {

int ensure(DevBuf &b, size_t bytes) {
  if (bytes <= b.cap && b.p) return RSF_OK;
  if (b.p) { HIP_TRY(hipFree(b.p)); b.p = nullptr; b.cap = 0; }
  if (bytes == 0) bytes = 8;
  HIP_TRY(hipMalloc(&b.p, bytes));
  b.cap = bytes;
  return RSF_OK;
}

void release(DevBuf &b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

bool host_mem(const rsf_ctx *c) { return c->cfg.mem_space == RSF_MEM_HOST; }

// input array: device pointer the kernels may read (staged copy for host callers)
int stage_in(rsf_ctx *c, int slot, const void *src, size_t bytes, const void **dev) {
  if (!src) { *dev = nullptr; return RSF_OK; }
  if (!host_mem(c)) { *dev = src; return RSF_OK; }
  int rc = ensure(c->stage[slot], bytes);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(c->stage[slot].p, src, bytes, hipMemcpyHostToDevice, c->stream));
  *dev = c->stage[slot].p;
  return RSF_OK;
}

// output array: device pointer the kernels may write
int stage_out(rsf_ctx *c, int slot, void *dst, size_t bytes, void **dev) {
  if (!dst) { *dev = nullptr; return RSF_OK; }
  if (!host_mem(c)) { *dev = dst; return RSF_OK; }
  int rc = ensure(c->stage[slot], bytes);
  if (rc) return rc;
  *dev = c->stage[slot].p;
  return RSF_OK;
}

int copy_back(rsf_ctx *c, int slot, void *dst, size_t bytes) {
  if (!dst || !host_mem(c)) return RSF_OK;
  HIP_TRY(hipMemcpyAsync(dst, c->stage[slot].p, bytes, hipMemcpyDeviceToHost, c->stream));
  return RSF_OK;
}

int finish(rsf_ctx *c) {  // host callers get synchronous semantics
  HIP_TRY(hipGetLastError());
  if (host_mem(c)) HIP_TRY(hipStreamSynchronize(c->stream));
  return RSF_OK;
}

Consts make_consts(const rsf_ctx *c, const double *data) {
  Consts K{};
  K.mu_ref = c->m.mu_ref; K.V_ref = c->m.V_ref; K.k1 = c->m.k1; K.mu0 = c->m.mu_t_zero;
  K.a_def = c->m.a; K.b_def = c->m.b;
  K.h = c->h; K.hh = 0.5 * c->h; K.h6 = c->h / 6.0;
  K.inv_dt = 1.0 / c->delta_t;
  K.cacc = K.h6 * K.inv_dt;
  K.inv_vref = 1.0 / c->m.V_ref;
  K.t0 = c->m.t_start;
  K.dt = c->delta_t;
  K.vl = (const double *)c->vl.p;
  K.data = data;
  K.nout = c->nout; K.S = c->m.substeps; K.kc = c->kc; K.nchunks = c->nchunks;
  K.group_chains = 0;
  return K;
}

unsigned grid_for(const rsf_ctx *c, int64_t n) { return (unsigned)((n + c->block - 1) / c->block); }

int mode_of(const rsf_ctx *c) {
  return (c->m.flags & RSF_FLAG_DOP853) ? DOP853 : ((c->m.flags & RSF_FLAG_FP32_SOLVE) ? RK4_F32 : RK4_F64);
}

// chains a lane of the sampler kernel carries: two in the float32 mode (mcmc_f32x2_kernel), else one
int chains_per_lane(const rsf_ctx *c) { return mode_of(c) == RK4_F32 ? 2 : 1; }

// LDS of a sampler launch: the table chunk, and behind it the per-lane Cholesky factors of a three-parameter chain
// (six doubles per lane, mcmc_kernel)
// the table chunk as the sampler kernel stages it: doubles, or floats in the float32 sampler — with a chunk length of its own
// (kc32, rsf_set_model): nsteps 4000 is ONE chunk of 48 KB there, resident for the whole launch, where the shared length kc
// (sized for doubles) made it two, staged — with two workgroup barriers each — for every proposal
size_t mcmc_table_bytes(const rsf_ctx *c) {
  if (mode_of(c) != RK4_F32) return c->lds_bytes;
  const size_t floats = 2 * (size_t)c->m.substeps * (size_t)c->kc32 + 1 + (size_t)c->kc32 + 1;
  return (floats * sizeof(float) + 15) & ~(size_t)15;
}

// the sampler's kernel constants: make_consts with the chunking the sampler kernel of this mode uses
Consts make_sampler_consts(const rsf_ctx *c, const double *data) {
  Consts K = make_consts(c, data);
  if (mode_of(c) == RK4_F32) { K.kc = c->kc32; K.nchunks = c->nchunks32; }
  return K;
}

size_t mcmc_lds_bytes(const rsf_ctx *c) {
  // per lane behind the table chunk: the float64 RK4 sampler parks the chain's point, sigma^2, SSq and log u there across
  // the forward solve (kParkSlots: 4 doubles for one parameter; 12 for three, whose first six hold the chain's Cholesky
  // factor); the other samplers keep only the factor of a three-parameter chain (two chains per lane in float32)
  const int d = c->mc.n_params;
  size_t slots = d == 3 ? (mode_of(c) == RK4_F32 ? 12 : 6) : 0;
  if (mode_of(c) == RK4_F64) slots = d == 3 ? kParkSlots<3> : kParkSlots<1>;
  return mcmc_table_bytes(c) + slots * sizeof(double) * (size_t)c->block;
}

// workgroups of a sampler launch over n chains
unsigned mcmc_grid(const rsf_ctx *c, int64_t n) {
  const int64_t per = (int64_t)c->block * chains_per_lane(c);
  return (unsigned)((n + per - 1) / per);
}

// [n][d] (the C ABI's layout) <-> [d][n] (the kernels' structure of arrays); both device pointers, on the ctx stream
int transpose(rsf_ctx *c, int64_t n, int d, const double *src, double *dst, bool to_soa) {
  if (d == 1) {
    if (src != dst) HIP_TRY(hipMemcpyAsync(dst, src, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    return RSF_OK;
  }
  hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((n + kMaxBlock - 1) / kMaxBlock)), dim3(kMaxBlock), 0, c->stream, n, d, src, dst, to_soa);
  return RSF_OK;
}

template <int D, bool DAMP, int MODE>
int launch_mcmc(rsf_ctx *c, const Consts &K, const McmcArgs &A, bool replay) {
  const dim3 grid(mcmc_grid(c, A.C)), block(c->block);
  if constexpr (MODE == RK4_F32) {
    if (replay)
      hipLaunchKernelGGL((mcmc_f32x2_kernel<D, DAMP, true>), grid, block, mcmc_lds_bytes(c), c->stream, K, A);
    else
      hipLaunchKernelGGL((mcmc_f32x2_kernel<D, DAMP, false>), grid, block, mcmc_lds_bytes(c), c->stream, K, A);
  } else {
    if (replay)
      hipLaunchKernelGGL((mcmc_kernel<D, DAMP, true, MODE>), grid, block, mcmc_lds_bytes(c), c->stream, K, A);
    else
      hipLaunchKernelGGL((mcmc_kernel<D, DAMP, false, MODE>), grid, block, mcmc_lds_bytes(c), c->stream, K, A);
  }
  return RSF_OK;
}

template <int D, bool DAMP>
int launch_mcmc_m(rsf_ctx *c, const Consts &K, const McmcArgs &A, bool replay) {
  switch (mode_of(c)) {
    case RK4_F32: return launch_mcmc<D, DAMP, RK4_F32>(c, K, A, replay);
    case DOP853: return launch_mcmc<D, DAMP, DOP853>(c, K, A, replay);
    default: return launch_mcmc<D, DAMP, RK4_F64>(c, K, A, replay);
  }
}

template <int D>
int launch_mcmc_d(rsf_ctx *c, const Consts &K, const McmcArgs &A, bool replay) {
  return (c->m.flags & RSF_FLAG_RADIATION_DAMPING) ? launch_mcmc_m<D, true>(c, K, A, replay) : launch_mcmc_m<D, false>(c, K, A, replay);
}

// RSF_MEM_HOST callers with a long run: launches of `per` iterations write their trace rows into one of two device
// staging sets; while launch k+1 computes, the rows of launch k go to the caller's arrays on a second stream.  The
// chain is the same as with one launch (the kernel continues from iter_base; tests: continuation == single launch).
// trace bytes per launch (cfg1: ~30 iterations, ~5 ms of compute); RSF_DRAIN_BYTES overrides it (tests use a tiny value)
size_t drain_bytes() {
  const char *e = std::getenv("RSF_DRAIN_BYTES");
  const long long v = e ? std::atoll(e) : 0;
  return v > 0 ? (size_t)v : (size_t)32 << 20;
}

int run_mcmc_drained(rsf_ctx *c, const Consts &K, McmcArgs A, int64_t per, double *tq, double *ts, uint8_t *ta) {
  const int d = c->mc.n_params;
  const size_t C = (size_t)A.C;
  const int64_t n_iters = A.n_iters;
  if (!c->copy_stream) {
    HIP_TRY(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (auto &e : c->ev_done) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  const int slot_q[2] = {3, 0}, slot_s[2] = {4, 1}, slot_a[2] = {5, 2};
  int rc;
  for (int b = 0; b < 2; ++b) {
    if (tq && (rc = ensure(c->stage[slot_q[b]], (size_t)per * C * d * sizeof(double)))) return rc;
    if (ts && (rc = ensure(c->stage[slot_s[b]], (size_t)per * C * sizeof(double)))) return rc;
    if (ta && (rc = ensure(c->stage[slot_a[b]], (size_t)per * C))) return rc;
  }
  auto drain = [&](int b, int64_t first, int64_t n) -> int {
    const size_t r0 = (size_t)first * C, rn = (size_t)n * C;
    HIP_TRY(hipStreamWaitEvent(c->copy_stream, c->ev_done[b], 0));
    if (tq) HIP_TRY(hipMemcpyAsync(tq + r0 * d, c->stage[slot_q[b]].p, rn * d * sizeof(double), hipMemcpyDeviceToHost, c->copy_stream));
    if (ts) HIP_TRY(hipMemcpyAsync(ts + r0, c->stage[slot_s[b]].p, rn * sizeof(double), hipMemcpyDeviceToHost, c->copy_stream));
    if (ta) HIP_TRY(hipMemcpyAsync(ta + r0, c->stage[slot_a[b]].p, rn, hipMemcpyDeviceToHost, c->copy_stream));
    HIP_TRY(hipStreamSynchronize(c->copy_stream));  // the staging set is free again, the rows are in the caller's arrays
    return RSF_OK;
  };
  const int64_t base = A.iter_base;
  int64_t done = 0, prev_first = 0, prev_n = 0;
  int b = 0;
  while (done < n_iters) {
    const int64_t n = std::min(per, n_iters - done);
    A.n_iters = n; A.iter_base = base + done;
    A.tq = tq ? (double *)c->stage[slot_q[b]].p : nullptr;
    A.ts = ts ? (double *)c->stage[slot_s[b]].p : nullptr;
    A.ta = ta ? (uint8_t *)c->stage[slot_a[b]].p : nullptr;
    rc = d == 1 ? launch_mcmc_d<1>(c, K, A, false) : launch_mcmc_d<3>(c, K, A, false);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(c->ev_done[b], c->stream));
    if (prev_n && (rc = drain(b ^ 1, prev_first, prev_n))) return rc;
    prev_first = done; prev_n = n;
    done += n;
    b ^= 1;
  }
  if ((rc = drain(b ^ 1, prev_first, prev_n))) return rc;
  c->iters_done += n_iters;
  return finish(c);
}

constexpr int64_t kReplayGraphMaxChains = 4096;  // beyond this the copies dominate and the plain path is as good

template <int D, bool DAMP>
const void *replay_kernel_m(const rsf_ctx *c) {
  switch (mode_of(c)) {
    case RK4_F32: return (const void *)mcmc_f32x2_kernel<D, DAMP, true>;
    case DOP853: return (const void *)mcmc_kernel<D, DAMP, true, DOP853>;
    default: return (const void *)mcmc_kernel<D, DAMP, true, RK4_F64>;
  }
}

const void *replay_kernel(const rsf_ctx *c) {
  const bool damp = c->m.flags & RSF_FLAG_RADIATION_DAMPING;
  if (c->mc.n_params == 1) return damp ? replay_kernel_m<1, true>(c) : replay_kernel_m<1, false>(c);
  return damp ? replay_kernel_m<3, true>(c) : replay_kernel_m<3, false>(c);
}

void release_replay_graph(rsf_ctx *c) {
  auto &g = c->rg;
  if (g.exec) (void)hipGraphExecDestroy(g.exec);
  if (g.graph) (void)hipGraphDestroy(g.graph);
  if (g.host) (void)hipHostFree(g.host);
  if (g.dev) (void)hipFree(g.dev);
  g = rsf_ctx::ReplayGraph{};
}

// ONE replayed proposal per call from host memory — what the drop-in MCMC.sample() does a thousand times, each call
// otherwise being three small H2D copies, a 0.1 ms kernel, three D2H copies and a synchronise.  The sequence is a
// three-node hipGraph (H2D of one pinned input block, the kernel, D2H of one pinned output block) instantiated once per
// (chains, parameters, kernel) and relaunched with fresh kernel arguments: one runtime call per proposal instead of seven.
int run_replay_graph(rsf_ctx *c, const Consts &K, McmcArgs A, const double *z, const double *u, const double *g, double *tq,
                     double *ts, uint8_t *ta) {
  auto &G = c->rg;
  const int d = c->mc.n_params;
  const size_t C = (size_t)A.C;
  const size_t in_bytes = (C * d + 2 * C) * sizeof(double), out_bytes = (C * d + C) * sizeof(double) + C;
  const size_t out_off = (in_bytes + 255) & ~(size_t)255, total = out_off + ((out_bytes + 255) & ~(size_t)255);
  const void *fn = replay_kernel(c);
  char *hb = (char *)G.host, *db = (char *)G.dev;
  const bool rebuild = !G.exec || G.C != A.C || G.d != d || G.fn != fn || G.lds != mcmc_lds_bytes(c) || G.block != c->block;
  if (rebuild) {
    release_replay_graph(c);
    HIP_TRY(hipHostMalloc(&G.host, total, hipHostMallocDefault));
    HIP_TRY(hipMalloc(&G.dev, total));
    hb = (char *)G.host; db = (char *)G.dev;
  }
  A.z = (const double *)db; A.u = A.z + C * d; A.g = A.u + C;
  A.tq = tq ? (double *)(db + out_off) : nullptr;
  A.ts = ts ? (double *)(db + out_off) + C * d : nullptr;
  A.ta = ta ? (uint8_t *)((double *)(db + out_off) + C * d + C) : nullptr;
  Consts Kc = K;
  void *params[2] = {&Kc, &A};
  hipKernelNodeParams kp{};
  kp.func = const_cast<void *>(fn);
  kp.gridDim = dim3(mcmc_grid(c, A.C)); kp.blockDim = dim3(c->block);
  kp.sharedMemBytes = (unsigned)mcmc_lds_bytes(c);
  kp.kernelParams = params;
  kp.extra = nullptr;
  if (rebuild) {
    hipGraphNode_t h2d, d2h;
    HIP_TRY(hipGraphCreate(&G.graph, 0));
    HIP_TRY(hipGraphAddMemcpyNode1D(&h2d, G.graph, nullptr, 0, db, hb, in_bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipGraphAddKernelNode(&G.kernel, G.graph, &h2d, 1, &kp));
    HIP_TRY(hipGraphAddMemcpyNode1D(&d2h, G.graph, &G.kernel, 1, hb + out_off, db + out_off, out_bytes, hipMemcpyDeviceToHost));
    HIP_TRY(hipGraphInstantiate(&G.exec, G.graph, nullptr, nullptr, 0));
    G.C = A.C; G.d = d; G.fn = fn; G.lds = mcmc_lds_bytes(c); G.block = c->block;
  } else {
    HIP_TRY(hipGraphExecKernelNodeSetParams(G.exec, G.kernel, &kp));
  }
  double *hz = (double *)hb;
  std::memcpy(hz, z, C * d * sizeof(double));
  std::memcpy(hz + C * d, u, C * sizeof(double));
  std::memcpy(hz + C * d + C, g, C * sizeof(double));
  HIP_TRY(hipGraphLaunch(G.exec, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  const double *ho = (const double *)(hb + out_off);
  if (tq) std::memcpy(tq, ho, C * d * sizeof(double));
  if (ts) std::memcpy(ts, ho + C * d, C * sizeof(double));
  if (ta) std::memcpy(ta, (const uint8_t *)(ho + C * d + C), C);
  c->iters_done += 1;
  return RSF_OK;
}

// the chain logic alone on caller-supplied sums of squares (rsf_mcmc_replay_ssq): no tables, no solve, one chain per lane
template <int D>
int launch_mcmc_inject(rsf_ctx *c, const Consts &K, McmcArgs A) {
  A.lc_off = 0;  // nothing is staged: the per-lane slots of a three-parameter chain start at the base of LDS
  const size_t lds = kParkSlots<D> * sizeof(double) * (size_t)c->block;
  hipLaunchKernelGGL((mcmc_kernel<D, false, true, RK4_F64, true>), dim3(grid_for(c, A.C)), dim3(c->block), lds, c->stream, K, A);
  return RSF_OK;
}

int run_mcmc(rsf_ctx *c, int64_t n_iters, const double *z, const double *u, const double *g, const double *ssq_new, double *tq,
             double *ts, uint8_t *ta, bool replay) {
  if (!c || n_iters < 0) return fail(RSF_ERR_INVALID, "rsf_mcmc_run: bad argument");
  if (!c->have_chains) return fail(RSF_ERR_STATE, "rsf_mcmc_run: call rsf_mcmc_init first");
  if (c->external_chains && !ssq_new)
    return fail(RSF_ERR_STATE, "chains made by rsf_mcmc_init_state have no observation: advance them with rsf_mcmc_replay_ssq");
  if (n_iters > INT32_MAX) return fail(RSF_ERR_INVALID, "at most 2^31 - 1 iterations per call (every lane counts its own)");
  if (n_iters == 0) return RSF_OK;
  DeviceGuard guard(c->device);
  if (!guard.ok) return fail(RSF_ERR_DEVICE, "rsf_mcmc_run: cannot select device %d", c->device);
  const int d = c->mc.n_params;
  const int64_t C = c->mc.n_chains;
  const size_t rows = (size_t)n_iters * (size_t)C;
  McmcArgs A{};
  A.C = C; A.chain_offset = c->mc.chain_offset; A.n_iters = n_iters; A.iter_base = c->iters_done;
  A.seed = c->mc.seed; A.n0 = c->mc.n0; A.shape = 0.5 * (c->mc.n0 + (double)c->nout);  // MCMC.py:158
  A.gd = A.shape - 1.0 / 3.0; A.gc = 1.0 / std::sqrt(9.0 * A.gd);
  for (int p = 0; p < RSF_MAX_PARAMS; ++p) {
    A.lo[p] = c->mc.lo[p]; A.hi[p] = c->mc.hi[p];
    const double w = 1e-6 * (c->mc.hi[p] - c->mc.lo[p]);
    A.am_eps[p] = w * w;
  }
  A.adapt_mode = c->mc.adapt_mode; A.adapt_interval = c->mc.adapt_interval > 0 ? c->mc.adapt_interval : 1;
  A.dict_scale = 2.38 * 2.38 / (double)(c->mc.prior_len > 0 ? c->mc.prior_len : 2);
  A.lc_off = (int32_t)(mcmc_table_bytes(c) / sizeof(double));
  A.q = (double *)c->q.p; A.ssq = (double *)c->ssq.p; A.std2 = (double *)c->std2.p; A.V = (double *)c->V.p;
  A.wref = (double *)c->wref.p; A.wsum = (double *)c->wsum.p; A.wsq = (double *)c->wsq.p; A.wn = (int32_t *)c->wn.p;
  A.wbuf = (double *)c->wbuf.p;
  A.stats = (unsigned long long *)c->stats.p;
  int rc;
  if (replay && !ssq_new && host_mem(c) && n_iters == 1 && C <= kReplayGraphMaxChains) {
    Consts Kg = make_sampler_consts(c, (const double *)c->data.p);
    Kg.group_chains = c->group_chains;
    return run_replay_graph(c, Kg, A, z, u, g, tq, ts, ta);
  }
  const void *dz = nullptr, *du = nullptr, *dg = nullptr;
  void *dtq = nullptr, *dts = nullptr, *dta = nullptr;
  if ((rc = stage_in(c, 0, z, rows * d * sizeof(double), &dz))) return rc;
  if ((rc = stage_in(c, 1, u, rows * sizeof(double), &du))) return rc;
  if ((rc = stage_in(c, 2, g, rows * sizeof(double), &dg))) return rc;
  A.z = (const double *)dz; A.u = (const double *)du; A.g = (const double *)dg;
  const void *dsn = nullptr;
  if ((rc = stage_in(c, 6, ssq_new, rows * sizeof(double), &dsn))) return rc;
  A.ssq_new = (const double *)dsn;
  Consts K = make_sampler_consts(c, (const double *)c->data.p);
  K.group_chains = c->group_chains;
  if (host_mem(c) && !replay) {
    const size_t row_bytes = (size_t)C * ((tq ? d * sizeof(double) : 0) + (ts ? sizeof(double) : 0) + (ta ? 1 : 0));
    const int64_t per = row_bytes ? std::max<int64_t>(1, (int64_t)(drain_bytes() / row_bytes)) : n_iters;
    if (per < n_iters) return run_mcmc_drained(c, K, A, per, tq, ts, ta);
  }
  if ((rc = stage_out(c, 3, tq, rows * d * sizeof(double), &dtq))) return rc;
  if ((rc = stage_out(c, 4, ts, rows * sizeof(double), &dts))) return rc;
  if ((rc = stage_out(c, 5, ta, rows, &dta))) return rc;
  A.tq = (double *)dtq; A.ts = (double *)dts; A.ta = (uint8_t *)dta;
  if (ssq_new) rc = d == 1 ? launch_mcmc_inject<1>(c, K, A) : launch_mcmc_inject<3>(c, K, A);
  else rc = d == 1 ? launch_mcmc_d<1>(c, K, A, replay) : launch_mcmc_d<3>(c, K, A, replay);
  if (rc) return rc;
  if ((rc = copy_back(c, 3, tq, rows * d * sizeof(double)))) return rc;
  if ((rc = copy_back(c, 4, ts, rows * sizeof(double)))) return rc;
  if ((rc = copy_back(c, 5, ta, rows))) return rc;
  c->iters_done += n_iters;
  return finish(c);
}

// RCCL, bound at run time: a process that already holds a copy (PyTorch links its own) must not get a second one,
// and a caller that never pools across GPUs needs none at all.
struct Rccl {
  void *h = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommInitAll) comm_init_all = nullptr;
  decltype(&ncclGroupStart) group_start = nullptr;
  decltype(&ncclGroupEnd) group_end = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
};

void bind_rccl(Rccl &r) {
  const char *env = std::getenv("RSF_RCCL_LIB");
  const char *names[] = {"librccl.so", "librccl.so.1"};
  if (env && *env) r.h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
  for (const char *n : names) if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);  // the copy already in the process
  if (!r.h) {  // a copy PyTorch loaded by path is found through one of its symbols
    Dl_info info;
    void *sym = dlsym(RTLD_DEFAULT, "ncclGetUniqueId");
    if (sym && dladdr(sym, &info) && info.dli_fname) r.h = dlopen(info.dli_fname, RTLD_NOW | RTLD_NOLOAD);
  }
  for (const char *n : names) if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
  if (!r.h) r.h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!r.h) return;
  r.get_unique_id = (decltype(r.get_unique_id))dlsym(r.h, "ncclGetUniqueId");
  r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(r.h, "ncclCommInitRank");
  r.comm_init_all = (decltype(r.comm_init_all))dlsym(r.h, "ncclCommInitAll");
  r.group_start = (decltype(r.group_start))dlsym(r.h, "ncclGroupStart");
  r.group_end = (decltype(r.group_end))dlsym(r.h, "ncclGroupEnd");
  r.comm_destroy = (decltype(r.comm_destroy))dlsym(r.h, "ncclCommDestroy");
  r.all_gather = (decltype(r.all_gather))dlsym(r.h, "ncclAllGather");
  r.all_reduce = (decltype(r.all_reduce))dlsym(r.h, "ncclAllReduce");
  r.error_string = (decltype(r.error_string))dlsym(r.h, "ncclGetErrorString");
  if (!r.get_unique_id || !r.comm_init_rank || !r.comm_init_all || !r.group_start || !r.group_end || !r.comm_destroy || !r.all_gather ||
      !r.all_reduce || !r.error_string)
    r.h = nullptr;
}

const Rccl *rccl() {  // bound once, whichever thread asks first
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, bind_rccl, std::ref(r));
  return r.h ? &r : nullptr;
}

#define RCCL_TRY(R, expr)                                                                          \
  do {                                                                                             \
    ncclResult_t e_ = (expr);                                                                      \
    if (e_ != ncclSuccess) return fail(RSF_ERR_DEVICE, "%s -> %s", #expr, (R)->error_string(e_));  \
  } while (0)

void free_chains(rsf_ctx *c) {
  release(c->data); release(c->q); release(c->ssq); release(c->std2); release(c->V);
  release(c->wref); release(c->wsum); release(c->wsq); release(c->wn); release(c->wbuf); release(c->stats);
  c->have_chains = false;
  c->external_chains = false;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" {

int rsf_version(void) { return RSF_ABI_VERSION; }
const char *rsf_backend(void) { return "hip-gfx950"; }
#ifndef RSF_BUILD_ID  // csrc/Makefile passes the SHA-256 prefix of the kernel sources
#define RSF_BUILD_ID "unknown"
#endif
const char *rsf_build_id(void) { return RSF_BUILD_ID; }
const char *rsf_last_error(void) { return g_err; }

int rsf_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { fail(RSF_ERR_DEVICE, "hipGetDeviceCount -> %s", hipGetErrorString(e)); return 0; }
  return n;
}

int rsf_create(const rsf_config *cfg, rsf_ctx **out) {
  if (!cfg || !out) return fail(RSF_ERR_INVALID, "rsf_create: NULL argument");
  if (cfg->size != sizeof(rsf_config) || cfg->version != RSF_ABI_VERSION)
    return fail(RSF_ERR_INVALID, "rsf_create: config size/version mismatch");
  if (cfg->mem_space != RSF_MEM_HOST && cfg->mem_space != RSF_MEM_DEVICE)
    return fail(RSF_ERR_INVALID, "rsf_create: bad mem_space");
  int block = cfg->block_threads ? (int)cfg->block_threads : kMaxBlock;
  if (block % 64 != 0 || block < 64 || block > kMaxBlock)
    return fail(RSF_ERR_INVALID, "rsf_create: block_threads must be a multiple of 64 in [64, %d]", kMaxBlock);
  int n = 0;
  HIP_TRY(hipGetDeviceCount(&n));
  if (n <= 0) return fail(RSF_ERR_DEVICE, "rsf_create: no HIP device visible (this library has no CPU fallback)");
  int dev = cfg->device;
  if (dev < 0) HIP_TRY(hipGetDevice(&dev));
  if (dev >= n) return fail(RSF_ERR_INVALID, "rsf_create: device %d out of range (%d visible)", dev, n);
  rsf_ctx *c = new (std::nothrow) rsf_ctx();
  if (!c) return fail(RSF_ERR_NOMEM, "rsf_create: out of memory");
  c->cfg = *cfg;
  c->device = dev;
  c->stream = (hipStream_t)cfg->stream;  // NULL = the device's default stream
  c->block = block;
  *out = c;
  return RSF_OK;
}

int rsf_destroy(rsf_ctx *c) {
  if (!c) return RSF_OK;
  {
    DeviceGuard guard(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_chains(c);
    release(c->vl);
    for (auto &s : c->stage) release(s);
    release(c->pool);
    release_replay_graph(c);
    if (c->comm) { const Rccl *R = rccl(); if (R) (void)R->comm_destroy(c->comm); }
    for (auto &e : c->ev_done) if (e) (void)hipEventDestroy(e);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  }
  delete c;
  return RSF_OK;
}

int rsf_sync(rsf_ctx *c) {
  if (!c) return fail(RSF_ERR_INVALID, "rsf_sync: NULL ctx");
  DeviceGuard guard(c->device);
  HIP_TRY(hipStreamSynchronize(c->stream));
  return RSF_OK;
}

int rsf_set_model(rsf_ctx *c, const rsf_model *m) {
  if (!c || !m) return fail(RSF_ERR_INVALID, "rsf_set_model: NULL argument");
  if (m->size != sizeof(rsf_model)) return fail(RSF_ERR_INVALID, "rsf_set_model: struct size mismatch");
  if (m->nsteps < 2 || m->substeps < 1 || !(m->t_final > m->t_start))
    return fail(RSF_ERR_INVALID, "rsf_set_model: need nsteps >= 2, substeps >= 1, t_final > t_start");
  if ((m->flags & RSF_FLAG_DOP853) && (m->flags & RSF_FLAG_FP32_SOLVE))
    return fail(RSF_ERR_INVALID, "rsf_set_model: the dop853 integrator is float64 only");
  DeviceGuard guard(c->device);
  if (!guard.ok) return fail(RSF_ERR_DEVICE, "rsf_set_model: cannot select device %d", c->device);
  const double delta_t = (m->t_final - m->t_start) / m->nsteps;                  // RateStateModel.py:176
  const int32_t nout = (int32_t)std::floor((m->t_final - m->t_start) / delta_t); // RateStateModel.py:358
  if (nout < 2) return fail(RSF_ERR_INVALID, "rsf_set_model: fewer than 2 output samples");
  const int S = m->substeps;
  const double h = delta_t / S, hh = 0.5 * h;
  const bool dop = m->flags & RSF_FLAG_DOP853;
  const size_t words = kLdsBudget / sizeof(double) - 2 * rsf::kLdsPad;
  int64_t kc;
  std::vector<double> vl;
  if (dop) {
    // DOP853 mode: per output interval the 12 loading values of the standard step (first step clipped to the
    // interval).  x is accumulated like scipy's r.t (x <- x + h with h = fl(fl(x + dt) - x)), so the stage times
    // are bit-identical to the ones the kernel — and the reference — use.
    kc = (int64_t)words / (rsf::dp::kTab + 1);
    if (kc > nout - 1) kc = nout - 1;
    vl.resize((size_t)rsf::dp::kTab * (size_t)(nout - 1));
    double x = m->t_start;
    for (int32_t k = 0; k < nout - 1; ++k) {
      const double xend = x + delta_t, hk = xend - x;
      for (int st = 0; st < rsf::dp::kTab; ++st) {
        const double t = st == 0 ? x : (st == 11 ? x + hk : x + RSF_DP_C[st] * hk);
        vl[(size_t)rsf::dp::kTab * k + st] = m->V_ref * (1 + std::exp(-t / 20) * std::sin(10 * t));
      }
      x = x + hk;
    }
  } else {
    // LDS chunking: kc output intervals need (2*S*kc + 1) loading values + kc observations + the observation's sample 0
    kc = ((int64_t)words - 2) / (2 * (int64_t)S + 1);
    if (kc < 1) return fail(RSF_ERR_UNSUPPORTED, "rsf_set_model: substeps=%d does not fit the LDS staging budget", S);
    if (kc > nout - 1) kc = nout - 1;
    // float32 solve: residuals are summed in float32 over groups of eight samples (k = 1..8, 9..16, ...; rsf_device_f32.h,
    // Out32) and its assembly trip covers one group: chunks begin on a group boundary
    if ((m->flags & RSF_FLAG_FP32_SOLVE) && kc < nout - 1 && kc >= 8) kc &= ~(int64_t)7;
    // chain-independent loading velocity at every RK4 stage time, RateStateModel.py:327-329
    vl.resize(2 * (size_t)S * (size_t)(nout - 1) + 1);
    for (size_t j = 0; j < vl.size(); ++j) {
      const double t = m->t_start + (double)j * hh;
      vl[j] = m->V_ref * (1 + std::exp(-t / 20) * std::sin(10 * t));
    }
  }
  const size_t nvl = vl.size();
  int rc = ensure(c->vl, nvl * sizeof(double));
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(c->vl.p, vl.data(), nvl * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));  // vl (host vector) goes out of scope
  if (c->have_chains) free_chains(c);  // chain state (SSq, sigma^2, covariance) belongs to the previous model
  c->m = *m;
  c->delta_t = delta_t; c->h = h; c->nout = nout;
  c->kc = (int32_t)kc;
  c->nchunks = (int32_t)((nout - 1 + kc - 1) / kc);
  c->kc32 = c->kc; c->nchunks32 = c->nchunks;
  if ((m->flags & RSF_FLAG_FP32_SOLVE) && !dop) {
    // the float32 sampler stages floats: the same budget in bytes holds twice the entries (chunks on a group boundary, as above)
    int64_t k32 = ((int64_t)(kLdsBudget / sizeof(float)) - 2) / (2 * (int64_t)S + 1);
    if (k32 > nout - 1) k32 = nout - 1;
    if (k32 < nout - 1 && k32 >= 8) k32 &= ~(int64_t)7;
    c->kc32 = (int32_t)k32;
    c->nchunks32 = (int32_t)((nout - 1 + k32 - 1) / k32);
  }
  c->lds_bytes = (size_t)((dop ? rsf::dp::kTab * kc : 2 * S * kc + 1) + kc + 1 + 2 * rsf::kLdsPad) * sizeof(double);
  c->have_model = true;
  return RSF_OK;
}

int rsf_model_nout(rsf_ctx *c, int32_t *nout) {
  if (!c || !nout) return fail(RSF_ERR_INVALID, "rsf_model_nout: NULL argument");
  if (!c->have_model) return fail(RSF_ERR_STATE, "rsf_model_nout: call rsf_set_model first");
  *nout = c->nout;
  return RSF_OK;
}

int rsf_forward_batch(rsf_ctx *c, int64_t n, const double *dc, const double *a, const double *b,
                      const double *data, double *ssq_out, double *acc_out) {
  if (!c || !dc || n < 0) return fail(RSF_ERR_INVALID, "rsf_forward_batch: bad argument");
  if (!c->have_model) return fail(RSF_ERR_STATE, "rsf_forward_batch: call rsf_set_model first");
  if (ssq_out && !data) return fail(RSF_ERR_INVALID, "rsf_forward_batch: ssq_out needs data");
  if (n == 0) return RSF_OK;
  DeviceGuard guard(c->device);
  if (!guard.ok) return fail(RSF_ERR_DEVICE, "rsf_forward_batch: cannot select device %d", c->device);
  const size_t nb = (size_t)n * sizeof(double);
  const void *ddc, *da, *db, *ddata;
  void *dssq, *dacc;
  int rc;
  if ((rc = stage_in(c, 0, dc, nb, &ddc))) return rc;
  if ((rc = stage_in(c, 1, a, nb, &da))) return rc;
  if ((rc = stage_in(c, 2, b, nb, &db))) return rc;
  if ((rc = stage_in(c, 3, ssq_out ? data : nullptr, (size_t)c->nout * sizeof(double), &ddata))) return rc;
  if ((rc = stage_out(c, 4, ssq_out, nb, &dssq))) return rc;
  if ((rc = stage_out(c, 5, acc_out, nb * (size_t)c->nout, &dacc))) return rc;
  const Consts K = make_consts(c, (const double *)ddata);
  const dim3 grid(grid_for(c, n)), block(c->block);
  const bool damp = c->m.flags & RSF_FLAG_RADIATION_DAMPING;
#define RSF_LAUNCH_FWD_M(DAMP, SSQ, ACC, MODE)                                                              \
  hipLaunchKernelGGL((forward_kernel<DAMP, SSQ, ACC, MODE>), grid, block, c->lds_bytes, c->stream, K, n,      \
                     (const double *)ddc, (const double *)da, (const double *)db, (double *)dssq, (double *)dacc)
#define RSF_LAUNCH_FWD(DAMP, SSQ, ACC)                                                                      \
  do {                                                                                                      \
    switch (mode_of(c)) {                                                                                   \
      case RK4_F32: RSF_LAUNCH_FWD_M(DAMP, SSQ, ACC, RK4_F32); break;                                       \
      case DOP853: RSF_LAUNCH_FWD_M(DAMP, SSQ, ACC, DOP853); break;                                         \
      default: RSF_LAUNCH_FWD_M(DAMP, SSQ, ACC, RK4_F64); break;                                            \
    }                                                                                                       \
  } while (0)
  const int sel = (damp ? 4 : 0) | (ssq_out ? 2 : 0) | (acc_out ? 1 : 0);
  switch (sel) {
    case 0: case 4: break;  // nothing requested
    case 1: RSF_LAUNCH_FWD(false, false, true); break;
    case 2: RSF_LAUNCH_FWD(false, true, false); break;
    case 3: RSF_LAUNCH_FWD(false, true, true); break;
    case 5: RSF_LAUNCH_FWD(true, false, true); break;
    case 6: RSF_LAUNCH_FWD(true, true, false); break;
    case 7: RSF_LAUNCH_FWD(true, true, true); break;
  }
#undef RSF_LAUNCH_FWD
#undef RSF_LAUNCH_FWD_M
  if ((rc = copy_back(c, 4, ssq_out, nb))) return rc;
  if ((rc = copy_back(c, 5, acc_out, nb * (size_t)c->nout))) return rc;
  return finish(c);
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
  const int G = cfg->n_groups > 1 ? cfg->n_groups : 1;
  // a workgroup's chains share one observation series: a group must be whole workgroups' worth of chains
  const int wg_chains = c->block * chains_per_lane(c);
  if (cfg->n_groups < 0 || cfg->n_chains % G || (G > 1 && (cfg->n_chains / G) % wg_chains))
    return fail(RSF_ERR_INVALID, "rsf_mcmc_init: n_chains/n_groups must be a whole multiple of a workgroup's chains (%d%s)", wg_chains,
                chains_per_lane(c) == 2 ? ": the float32 sampler carries two chains per lane" : "");
  DeviceGuard guard(c->device);
  if (!guard.ok) return fail(RSF_ERR_DEVICE, "rsf_mcmc_init: cannot select device %d", c->device);
  const int d = cfg->n_params;
  const int64_t C = cfg->n_chains;
  const size_t cb = (size_t)C * sizeof(double);
  const size_t data_bytes = (size_t)G * (size_t)c->nout * sizeof(double);
  int rc;
  if ((rc = ensure(c->data, data_bytes))) return rc;
  if ((rc = ensure(c->q, cb * d))) return rc;
  if ((rc = ensure(c->ssq, cb))) return rc;
  if ((rc = ensure(c->std2, cb))) return rc;
  if ((rc = ensure(c->V, cb * d * d))) return rc;
  if ((rc = ensure(c->wref, cb * d))) return rc;
  if ((rc = ensure(c->wsum, cb * d))) return rc;
  if ((rc = ensure(c->wsq, cb * d * d))) return rc;
  if ((rc = ensure(c->wn, (size_t)C * sizeof(int32_t)))) return rc;
  if (cfg->adapt_mode == RSF_ADAPT_REFERENCE_DICT) {  // the window's samples themselves: np.cov's own arithmetic needs them
    if (cfg->adapt_interval > RSF_DICT_MAX_INTERVAL)
      return fail(RSF_ERR_UNSUPPORTED, "reference_dict adaptation keeps at most %d samples per window", RSF_DICT_MAX_INTERVAL);
    if ((rc = ensure(c->wbuf, cb * (size_t)cfg->adapt_interval))) return rc;
    HIP_TRY(hipMemsetAsync(c->wbuf.p, 0, cb * (size_t)cfg->adapt_interval, c->stream));
  }
  if ((rc = ensure(c->stats, RSF_CNT_COUNT * sizeof(unsigned long long)))) return rc;
  const hipMemcpyKind kind = host_mem(c) ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  HIP_TRY(hipMemcpyAsync(c->data.p, data, data_bytes, kind, c->stream));
  const void *dq0;
  if ((rc = stage_in(c, 0, q0, cb * d, &dq0))) return rc;
  if ((rc = transpose(c, C, d, (const double *)dq0, (double *)c->q.p, true))) return rc;
  HIP_TRY(hipMemcpyAsync(c->wref.p, c->q.p, cb * d, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(hipMemsetAsync(c->wsum.p, 0, cb * d, c->stream));
  HIP_TRY(hipMemsetAsync(c->wsq.p, 0, cb * d * d, c->stream));
  HIP_TRY(hipMemsetAsync(c->wn.p, 0, (size_t)C * sizeof(int32_t), c->stream));
  HIP_TRY(hipMemsetAsync(c->stats.p, 0, RSF_CNT_COUNT * sizeof(unsigned long long), c->stream));
  InitArgs A{};
  A.C = C;
  A.fd = cfg->fd_rel_step;
  A.inv_dof = 1.0 / (double)(c->nout - (cfg->prior_len ? cfg->prior_len : d));
  for (int p = 0; p < RSF_MAX_PARAMS; ++p) A.width[p] = cfg->hi[p] - cfg->lo[p];
  A.q0 = (const double *)c->q.p;
  A.ssq = (double *)c->ssq.p; A.std2 = (double *)c->std2.p; A.V = (double *)c->V.p;
  c->group_chains = G > 1 ? C / G : 0;
  Consts K = make_consts(c, (const double *)c->data.p);
  K.group_chains = c->group_chains;
  const dim3 grid(grid_for(c, C)), block(c->block);
  // the init kernels run one lane per TRAJECTORY: 1 + d adjacent lanes per chain (rsf_kernels.h, InitGroup)
  const dim3 igrid((unsigned)((C * (d + 1) + c->block - 1) / c->block));
  const bool damp = c->m.flags & RSF_FLAG_RADIATION_DAMPING;
  if (c->m.flags & RSF_FLAG_DOP853) {
    if (d == 1) {
      if (damp) hipLaunchKernelGGL((init_dp_kernel<1, true>), igrid, block, c->lds_bytes, c->stream, K, A);
      else hipLaunchKernelGGL((init_dp_kernel<1, false>), igrid, block, c->lds_bytes, c->stream, K, A);
    } else {
      if (damp) hipLaunchKernelGGL((init_dp_kernel<3, true>), igrid, block, c->lds_bytes, c->stream, K, A);
      else hipLaunchKernelGGL((init_dp_kernel<3, false>), igrid, block, c->lds_bytes, c->stream, K, A);
    }
  } else if (d == 1) {
    if (damp) hipLaunchKernelGGL((init_kernel<1, true>), igrid, block, c->lds_bytes, c->stream, K, A);
    else hipLaunchKernelGGL((init_kernel<1, false>), igrid, block, c->lds_bytes, c->stream, K, A);
  } else {
    if (damp) hipLaunchKernelGGL((init_kernel<3, true>), igrid, block, c->lds_bytes, c->stream, K, A);
    else hipLaunchKernelGGL((init_kernel<3, false>), igrid, block, c->lds_bytes, c->stream, K, A);
  }
  if (c->m.flags & RSF_FLAG_FP32_SOLVE) {
    if (d == 1) {
      if (damp) hipLaunchKernelGGL((ssq32_kernel<1, true>), grid, block, c->lds_bytes, c->stream, K, C, A.q0, A.ssq);
      else hipLaunchKernelGGL((ssq32_kernel<1, false>), grid, block, c->lds_bytes, c->stream, K, C, A.q0, A.ssq);
    } else {
      if (damp) hipLaunchKernelGGL((ssq32_kernel<3, true>), grid, block, c->lds_bytes, c->stream, K, C, A.q0, A.ssq);
      else hipLaunchKernelGGL((ssq32_kernel<3, false>), grid, block, c->lds_bytes, c->stream, K, C, A.q0, A.ssq);
    }
  }
  c->mc = *cfg;
  c->iters_done = 0;
  c->have_chains = true;
  c->external_chains = false;
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));  // q0/data may be host buffers the caller reuses
  return RSF_OK;
}

int rsf_mcmc_get_state(rsf_ctx *c, double *q, double *ssq, double *std2, double *V) {
  if (!c) return fail(RSF_ERR_INVALID, "rsf_mcmc_get_state: NULL ctx");
  if (!c->have_chains) return fail(RSF_ERR_STATE, "rsf_mcmc_get_state: call rsf_mcmc_init first");
  DeviceGuard guard(c->device);
  const int d = c->mc.n_params;
  const size_t cb = (size_t)c->mc.n_chains * sizeof(double);
  const hipMemcpyKind kind = host_mem(c) ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  const int64_t C = c->mc.n_chains;
  void *dq, *dV;
  int rc;
  if ((rc = stage_out(c, 0, q, cb * d, &dq))) return rc;
  if ((rc = stage_out(c, 1, V, cb * d * d, &dV))) return rc;
  if (q && (rc = transpose(c, C, d, (const double *)c->q.p, (double *)dq, false))) return rc;
  if (V && (rc = transpose(c, C, d * d, (const double *)c->V.p, (double *)dV, false))) return rc;
  if ((rc = copy_back(c, 0, q, cb * d))) return rc;
  if ((rc = copy_back(c, 1, V, cb * d * d))) return rc;
  if (ssq) HIP_TRY(hipMemcpyAsync(ssq, c->ssq.p, cb, kind, c->stream));
  if (std2) HIP_TRY(hipMemcpyAsync(std2, c->std2.p, cb, kind, c->stream));
  return finish(c);
}

int rsf_mcmc_set_state(rsf_ctx *c, const double *q, const double *ssq, const double *std2, const double *V) {
  if (!c) return fail(RSF_ERR_INVALID, "rsf_mcmc_set_state: NULL ctx");
  if (!c->have_chains) return fail(RSF_ERR_STATE, "rsf_mcmc_set_state: call rsf_mcmc_init first");
  DeviceGuard guard(c->device);
  const int d = c->mc.n_params;
  const size_t cb = (size_t)c->mc.n_chains * sizeof(double);
  const hipMemcpyKind kind = host_mem(c) ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  const int64_t C = c->mc.n_chains;
  const void *dq, *dV;
  int rc;
  if ((rc = stage_in(c, 0, q, cb * d, &dq))) return rc;
  if ((rc = stage_in(c, 1, V, cb * d * d, &dV))) return rc;
  if (q && (rc = transpose(c, C, d, (const double *)dq, (double *)c->q.p, true))) return rc;
  if (V && (rc = transpose(c, C, d * d, (const double *)dV, (double *)c->V.p, true))) return rc;
  if (ssq) HIP_TRY(hipMemcpyAsync(c->ssq.p, ssq, cb, kind, c->stream));
  if (std2) HIP_TRY(hipMemcpyAsync(c->std2.p, std2, cb, kind, c->stream));
  return finish(c);
}

int rsf_mcmc_run(rsf_ctx *c, int64_t n_iters, double *tq, double *ts, uint8_t *ta) {
  return run_mcmc(c, n_iters, nullptr, nullptr, nullptr, nullptr, tq, ts, ta, false);
}

int rsf_mcmc_replay(rsf_ctx *c, int64_t n_iters, const double *z, const double *u, const double *g,
                    double *tq, double *ts, uint8_t *ta) {
  if (!z || !u || !g) return fail(RSF_ERR_INVALID, "rsf_mcmc_replay: z, u and g are required");
  return run_mcmc(c, n_iters, z, u, g, nullptr, tq, ts, ta, true);
}

int rsf_mcmc_replay_ssq(rsf_ctx *c, int64_t n_iters, const double *z, const double *u, const double *g, const double *ssq_new,
                        double *tq, double *ts, uint8_t *ta) {
  if (!z || !u || !g || !ssq_new) return fail(RSF_ERR_INVALID, "rsf_mcmc_replay_ssq: z, u, g and ssq_new are required");
  return run_mcmc(c, n_iters, z, u, g, ssq_new, tq, ts, ta, true);
}

int rsf_mcmc_init_state(rsf_ctx *c, const rsf_mcmc_config *cfg, const double *q, const double *ssq, const double *std2, const double *V) {
  if (!c || !cfg || !q || !ssq || !std2 || !V) return fail(RSF_ERR_INVALID, "rsf_mcmc_init_state: NULL argument");
  if (cfg->size != sizeof(rsf_mcmc_config)) return fail(RSF_ERR_INVALID, "rsf_mcmc_init_state: struct size mismatch");
  if (cfg->n_params != 1 && cfg->n_params != 3) return fail(RSF_ERR_UNSUPPORTED, "rsf_mcmc_init_state: n_params must be 1 or 3");
  if (cfg->n_chains < 1) return fail(RSF_ERR_INVALID, "rsf_mcmc_init_state: n_chains < 1");
  if (cfg->adapt_mode == RSF_ADAPT_REFERENCE_DICT && cfg->n_params != 1)
    return fail(RSF_ERR_UNSUPPORTED, "rsf_mcmc_init_state: reference_dict adaptation is defined for 1 parameter only");
  if (cfg->adapt_mode < 0 || cfg->adapt_mode > RSF_ADAPT_AM || (cfg->adapt_mode && cfg->adapt_interval < 2))
    return fail(RSF_ERR_INVALID, "rsf_mcmc_init_state: bad adapt_mode / adapt_interval");
  DeviceGuard guard(c->device);
  if (!guard.ok) return fail(RSF_ERR_DEVICE, "rsf_mcmc_init_state: cannot select device %d", c->device);
  const int d = cfg->n_params;
  const int64_t C = cfg->n_chains;
  const size_t cb = (size_t)C * sizeof(double);
  int rc;
  if ((rc = ensure(c->q, cb * d))) return rc;
  if ((rc = ensure(c->ssq, cb))) return rc;
  if ((rc = ensure(c->std2, cb))) return rc;
  if ((rc = ensure(c->V, cb * d * d))) return rc;
  if ((rc = ensure(c->wref, cb * d))) return rc;
  if ((rc = ensure(c->wsum, cb * d))) return rc;
  if ((rc = ensure(c->wsq, cb * d * d))) return rc;
  if ((rc = ensure(c->wn, (size_t)C * sizeof(int32_t)))) return rc;
  if (cfg->adapt_mode == RSF_ADAPT_REFERENCE_DICT) {  // the window's samples themselves: np.cov's own arithmetic needs them
    if (cfg->adapt_interval > RSF_DICT_MAX_INTERVAL)
      return fail(RSF_ERR_UNSUPPORTED, "reference_dict adaptation keeps at most %d samples per window", RSF_DICT_MAX_INTERVAL);
    if ((rc = ensure(c->wbuf, cb * (size_t)cfg->adapt_interval))) return rc;
    HIP_TRY(hipMemsetAsync(c->wbuf.p, 0, cb * (size_t)cfg->adapt_interval, c->stream));
  }
  if ((rc = ensure(c->stats, RSF_CNT_COUNT * sizeof(unsigned long long)))) return rc;
  const hipMemcpyKind kind = host_mem(c) ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  const void *dq, *dV;
  if ((rc = stage_in(c, 0, q, cb * d, &dq))) return rc;
  if ((rc = stage_in(c, 1, V, cb * d * d, &dV))) return rc;
  if ((rc = transpose(c, C, d, (const double *)dq, (double *)c->q.p, true))) return rc;
  if ((rc = transpose(c, C, d * d, (const double *)dV, (double *)c->V.p, true))) return rc;
  HIP_TRY(hipMemcpyAsync(c->ssq.p, ssq, cb, kind, c->stream));
  HIP_TRY(hipMemcpyAsync(c->std2.p, std2, cb, kind, c->stream));
  HIP_TRY(hipMemcpyAsync(c->wref.p, c->q.p, cb * d, hipMemcpyDeviceToDevice, c->stream));
  HIP_TRY(hipMemsetAsync(c->wsum.p, 0, cb * d, c->stream));
  HIP_TRY(hipMemsetAsync(c->wsq.p, 0, cb * d * d, c->stream));
  HIP_TRY(hipMemsetAsync(c->wn.p, 0, (size_t)C * sizeof(int32_t), c->stream));
  HIP_TRY(hipMemsetAsync(c->stats.p, 0, RSF_CNT_COUNT * sizeof(unsigned long long), c->stream));
  c->mc = *cfg;
  c->group_chains = 0;
  c->iters_done = 0;
  c->have_chains = true;
  c->external_chains = true;
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));  // the arguments may be host buffers the caller reuses
  return RSF_OK;
}

int rsf_mcmc_propose(rsf_ctx *c, const double *z, double *q_new, uint8_t *in_bounds) {
  if (!c || !z || !q_new || !in_bounds) return fail(RSF_ERR_INVALID, "rsf_mcmc_propose: NULL argument");
  if (!c->have_chains) return fail(RSF_ERR_STATE, "rsf_mcmc_propose: call rsf_mcmc_init or rsf_mcmc_init_state first");
  DeviceGuard guard(c->device);
  if (!guard.ok) return fail(RSF_ERR_DEVICE, "rsf_mcmc_propose: cannot select device %d", c->device);
  const int d = c->mc.n_params;
  const int64_t C = c->mc.n_chains;
  const size_t cb = (size_t)C * sizeof(double);
  const void *dz;
  void *dqn, *dinb;
  int rc;
  if ((rc = stage_in(c, 0, z, cb * d, &dz))) return rc;
  if ((rc = stage_out(c, 3, q_new, cb * d, &dqn))) return rc;
  if ((rc = stage_out(c, 5, in_bounds, (size_t)C, &dinb))) return rc;
  ProposeArgs A{};
  A.C = C; A.q = (const double *)c->q.p; A.V = (const double *)c->V.p; A.z = (const double *)dz;
  for (int p = 0; p < RSF_MAX_PARAMS; ++p) { A.lo[p] = c->mc.lo[p]; A.hi[p] = c->mc.hi[p]; }
  A.qn = (double *)dqn; A.inb = (uint8_t *)dinb;
  const dim3 grid((unsigned)((C + kMaxBlock - 1) / kMaxBlock)), block(kMaxBlock);
  if (d == 1) hipLaunchKernelGGL(propose_kernel<1>, grid, block, 0, c->stream, A);
  else hipLaunchKernelGGL(propose_kernel<3>, grid, block, 0, c->stream, A);
  if ((rc = copy_back(c, 3, q_new, cb * d))) return rc;
  if ((rc = copy_back(c, 5, in_bounds, (size_t)C))) return rc;
  return finish(c);
}

int rsf_mcmc_stats(rsf_ctx *c, int64_t *n_acc, int64_t *n_eval, int64_t *n_nonfinite, int64_t *n_done) {
  if (!c) return fail(RSF_ERR_INVALID, "rsf_mcmc_stats: NULL ctx");
  if (!c->have_chains) return fail(RSF_ERR_STATE, "rsf_mcmc_stats: call rsf_mcmc_init first");
  DeviceGuard guard(c->device);
  unsigned long long s[3];
  HIP_TRY(hipMemcpyAsync(s, c->stats.p, sizeof s, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (n_acc) *n_acc = (int64_t)s[RSF_CNT_ACCEPTED];
  if (n_eval) *n_eval = (int64_t)s[RSF_CNT_EVALUATED];
  if (n_nonfinite) *n_nonfinite = (int64_t)s[RSF_CNT_NONFINITE];
  if (n_done) *n_done = c->iters_done;
  return RSF_OK;
}

int rsf_mcmc_counters(rsf_ctx *c, int64_t *out, int32_t n) {
  if (!c || !out || n < 0) return fail(RSF_ERR_INVALID, "rsf_mcmc_counters: bad argument");
  if (!c->have_chains) return fail(RSF_ERR_STATE, "rsf_mcmc_counters: call rsf_mcmc_init first");
  DeviceGuard guard(c->device);
  unsigned long long s[RSF_CNT_COUNT];
  HIP_TRY(hipMemcpyAsync(s, c->stats.p, sizeof s, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  for (int32_t k = 0; k < n && k < RSF_CNT_COUNT; ++k) out[k] = (int64_t)s[k];
  return RSF_OK;
}

namespace {

// moments of x[i*stride] with x already a device pointer; result on the host
int pool_moments(rsf_ctx *c, int64_t n, const double *dx, int64_t stride, double out[5]) {
  int rc = ensure(c->pool, sizeof(PoolPartial) * kPoolBlocks);
  if (rc) return rc;
  double shift = 0.0;
  HIP_TRY(hipMemcpyAsync(&shift, dx, sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  const int blocks = (int)std::min<int64_t>(kPoolBlocks, (n + kMaxBlock - 1) / kMaxBlock);
  hipLaunchKernelGGL(pool_moments_kernel, dim3(blocks), dim3(kMaxBlock), 0, c->stream, n, dx, stride, shift, (PoolPartial *)c->pool.p);
  std::vector<PoolPartial> h(blocks);
  HIP_TRY(hipMemcpyAsync(h.data(), c->pool.p, sizeof(PoolPartial) * blocks, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  double cnt = 0, sum = 0, sumsq = 0, mn = INFINITY, mx = -INFINITY;
  for (const auto &p : h) { cnt += p.cnt; sum += p.sum; sumsq += p.sumsq; mn = std::fmin(mn, p.mn); mx = std::fmax(mx, p.mx); }
  const double mean_s = sum / cnt;
  out[0] = cnt; out[1] = shift + mean_s;
  out[2] = cnt > 1 ? (sumsq - cnt * mean_s * mean_s) / (cnt - 1) : 0.0;
  out[3] = mn; out[4] = mx;
  return RSF_OK;
}

}  // namespace

int rsf_pool_summary(rsf_ctx *c, int64_t n, const double *x, int64_t stride, double *out) {
  if (!c || !x || !out || n < 1 || stride < 1) return fail(RSF_ERR_INVALID, "rsf_pool_summary: bad argument");
  DeviceGuard guard(c->device);
  if (!guard.ok) return fail(RSF_ERR_DEVICE, "rsf_pool_summary: cannot select device %d", c->device);
  const void *dx;
  int rc = stage_in(c, 0, x, (size_t)((n - 1) * stride + 1) * sizeof(double), &dx);
  if (rc) return rc;
  return pool_moments(c, n, (const double *)dx, stride, out);
}

int rsf_pool_kde(rsf_ctx *c, int64_t n, const double *x, int64_t stride, int32_t m, const double *grid, double bw_factor,
                 double *density) {
  if (!c || !x || !grid || !density || n < 2 || m < 1 || stride < 1) return fail(RSF_ERR_INVALID, "rsf_pool_kde: bad argument");
  DeviceGuard guard(c->device);
  if (!guard.ok) return fail(RSF_ERR_DEVICE, "rsf_pool_kde: cannot select device %d", c->device);
  const void *dx, *dg;
  void *dd;
  int rc;
  if ((rc = stage_in(c, 0, x, (size_t)((n - 1) * stride + 1) * sizeof(double), &dx))) return rc;
  if ((rc = stage_in(c, 1, grid, (size_t)m * sizeof(double), &dg))) return rc;
  if ((rc = stage_out(c, 2, density, (size_t)m * sizeof(double), &dd))) return rc;
  double s[5];
  if ((rc = pool_moments(c, n, (const double *)dx, stride, s))) return rc;
  const double factor = bw_factor > 0.0 ? bw_factor : std::pow((double)n, -1.0 / 5.0);  // scipy scotts_factor, d = 1
  const double cov = s[2] * factor * factor;
  if (!(cov > 0.0)) return fail(RSF_ERR_INVALID, "rsf_pool_kde: the samples have zero variance (singular KDE)");
  const int blocks = (int)std::min<int64_t>(kPoolBlocks, (n + kKdeTile - 1) / kKdeTile);
  DevBuf &ws = c->stage[7];
  if ((rc = ensure(ws, (size_t)blocks * (size_t)m * sizeof(double)))) return rc;
  hipLaunchKernelGGL(pool_kde_kernel, dim3(blocks), dim3(kMaxBlock), 0, c->stream, n, (const double *)dx, stride, (int)m,
                     (const double *)dg, 0.5 / cov, (double *)ws.p);
  hipLaunchKernelGGL(pool_kde_reduce_kernel, dim3((m + kMaxBlock - 1) / kMaxBlock), dim3(kMaxBlock), 0, c->stream, blocks, (int)m,
                     (const double *)ws.p, 1.0 / ((double)n * std::sqrt(2.0 * 3.14159265358979323846 * cov)), (double *)dd);
  if ((rc = copy_back(c, 2, density, (size_t)m * sizeof(double)))) return rc;
  return finish(c);
}

int rsf_pool_histogram(rsf_ctx *c, int64_t n, const double *x, int64_t stride, int32_t nbins, double lo, double hi, double *counts) {
  if (!c || !x || !counts || n < 1 || stride < 1 || nbins < 1 || nbins > kHistMaxBins || !(hi > lo) || !std::isfinite(hi - lo))
    return fail(RSF_ERR_INVALID, "rsf_pool_histogram: bad argument (1 <= nbins <= %d, finite lo < hi)", kHistMaxBins);
  DeviceGuard guard(c->device);
  if (!guard.ok) return fail(RSF_ERR_DEVICE, "rsf_pool_histogram: cannot select device %d", c->device);
  const void *dx;
  void *dout;
  int rc;
  const int nb = nbins + 2;
  if ((rc = stage_in(c, 0, x, (size_t)((n - 1) * stride + 1) * sizeof(double), &dx))) return rc;
  if ((rc = stage_out(c, 2, counts, (size_t)nb * sizeof(double), &dout))) return rc;
  DevBuf &ws = c->stage[7];
  if ((rc = ensure(ws, (size_t)nb * sizeof(unsigned long long)))) return rc;
  HIP_TRY(hipMemsetAsync(ws.p, 0, (size_t)nb * sizeof(unsigned long long), c->stream));
  const int blocks = (int)std::min<int64_t>(kPoolBlocks, (n + kMaxBlock - 1) / kMaxBlock);
  hipLaunchKernelGGL(pool_hist_kernel, dim3(blocks), dim3(kMaxBlock), (size_t)nb * sizeof(unsigned int), c->stream, n, (const double *)dx,
                     stride, (int)nbins, lo, hi, (double)nbins / (hi - lo), (hi - lo) / (double)nbins, (unsigned long long *)ws.p);
  hipLaunchKernelGGL(pool_hist_finish_kernel, dim3((nb + kMaxBlock - 1) / kMaxBlock), dim3(kMaxBlock), 0, c->stream, nb,
                     (const unsigned long long *)ws.p, (double *)dout);
  if ((rc = copy_back(c, 2, counts, (size_t)nb * sizeof(double)))) return rc;
  return finish(c);
}

int rsf_comm_unique_id(uint8_t id[RSF_COMM_ID_BYTES]) {
  if (!id) return fail(RSF_ERR_INVALID, "rsf_comm_unique_id: NULL argument");
  static_assert(sizeof(ncclUniqueId) == RSF_COMM_ID_BYTES, "RCCL unique id size");
  const Rccl *R = rccl();
  if (!R) return fail(RSF_ERR_UNSUPPORTED, "rsf_comm_unique_id: RCCL (librccl.so) could not be loaded: %s", dlerror());
  ncclUniqueId u;
  RCCL_TRY(R, R->get_unique_id(&u));
  std::memcpy(id, u.internal, RSF_COMM_ID_BYTES);
  return RSF_OK;
}

int rsf_comm_init(rsf_ctx *c, int32_t world, int32_t rank, const uint8_t id[RSF_COMM_ID_BYTES]) {
  if (!c || world < 1 || rank < 0 || rank >= world) return fail(RSF_ERR_INVALID, "rsf_comm_init: bad argument");
  if (c->world) return fail(RSF_ERR_STATE, "rsf_comm_init: this ctx already has a communicator (rsf_comm_destroy first)");
  if (world > 1 && !id) return fail(RSF_ERR_INVALID, "rsf_comm_init: world > 1 needs the id from rsf_comm_unique_id on rank 0");
  if (id) {  // (world = 1 with an id makes a real one-rank communicator: the single-GPU test of the RCCL binding)
    const Rccl *R = rccl();
    if (!R) return fail(RSF_ERR_UNSUPPORTED, "rsf_comm_init: RCCL (librccl.so) could not be loaded");
    DeviceGuard guard(c->device);
    if (!guard.ok) return fail(RSF_ERR_DEVICE, "rsf_comm_init: cannot select device %d", c->device);
    ncclUniqueId u;
    std::memcpy(u.internal, id, RSF_COMM_ID_BYTES);
    RCCL_TRY(R, R->comm_init_rank(&c->comm, world, u, rank));
  }
  c->world = world;
  c->rank = rank;
  return RSF_OK;
}

int rsf_comm_destroy(rsf_ctx *c) {
  if (!c) return fail(RSF_ERR_INVALID, "rsf_comm_destroy: NULL ctx");
  ncclComm_t comm = c->comm;
  c->comm = nullptr;  // the ctx is out of its group whatever RCCL says about the teardown
  c->world = 0;
  c->rank = 0;
  if (comm) {
    DeviceGuard guard(c->device);
    (void)hipStreamSynchronize(c->stream);
    const Rccl *R = rccl();
    if (R) RCCL_TRY(R, R->comm_destroy(comm));
  }
  return RSF_OK;
}

int rsf_pool_allgather(rsf_ctx *c, const double *send, int64_t count, double *recv) {
  if (!c || !send || !recv || count < 1) return fail(RSF_ERR_INVALID, "rsf_pool_allgather: bad argument");
  if (!c->world) return fail(RSF_ERR_STATE, "rsf_pool_allgather: call rsf_comm_init first");
  DeviceGuard guard(c->device);
  if (!guard.ok) return fail(RSF_ERR_DEVICE, "rsf_pool_allgather: cannot select device %d", c->device);
  const size_t bytes = (size_t)count * sizeof(double);
  const void *ds;
  void *dr;
  int rc;
  if ((rc = stage_in(c, 0, send, bytes, &ds))) return rc;
  if ((rc = stage_out(c, 1, recv, bytes * (size_t)c->world, &dr))) return rc;
  if (!c->comm) {
    if (dr != ds) HIP_TRY(hipMemcpyAsync(dr, ds, bytes, hipMemcpyDeviceToDevice, c->stream));
  } else {
    const Rccl *R = rccl();
    RCCL_TRY(R, R->all_gather(ds, dr, (size_t)count, ncclFloat64, c->comm, c->stream));
  }
  if ((rc = copy_back(c, 1, recv, bytes * (size_t)c->world))) return rc;
  return finish(c);
}

int rsf_pool_allreduce_sum(rsf_ctx *c, double *buf, int64_t count) {
  if (!c || !buf || count < 1) return fail(RSF_ERR_INVALID, "rsf_pool_allreduce_sum: bad argument");
  if (!c->world) return fail(RSF_ERR_STATE, "rsf_pool_allreduce_sum: call rsf_comm_init first");
  if (!c->comm) return RSF_OK;
  DeviceGuard guard(c->device);
  if (!guard.ok) return fail(RSF_ERR_DEVICE, "rsf_pool_allreduce_sum: cannot select device %d", c->device);
  const size_t bytes = (size_t)count * sizeof(double);
  const void *ds;
  int rc;
  if ((rc = stage_in(c, 0, buf, bytes, &ds))) return rc;
  const Rccl *R = rccl();
  RCCL_TRY(R, R->all_reduce(ds, (void *)ds, (size_t)count, ncclFloat64, ncclSum, c->comm, c->stream));
  if (host_mem(c)) HIP_TRY(hipMemcpyAsync(buf, ds, bytes, hipMemcpyDeviceToHost, c->stream));
  return finish(c);
}

// ---- single-process form: one ctx per device, one host thread drives them all (ncclCommInitAll + grouped calls) ----
namespace {

int check_group(rsf_ctx *const *ctxs, int32_t n, const char *who, bool need_comm) {
  if (!ctxs || n < 1) return fail(RSF_ERR_INVALID, "%s: bad argument", who);
  for (int32_t i = 0; i < n; ++i) {
    if (!ctxs[i]) return fail(RSF_ERR_INVALID, "%s: ctxs[%d] is NULL", who, i);
    for (int32_t j = 0; j < i; ++j)
      if (ctxs[j] == ctxs[i]) return fail(RSF_ERR_INVALID, "%s: ctxs[%d] and ctxs[%d] are the same ctx", who, j, i);
    if (need_comm && (ctxs[i]->world != n || ctxs[i]->rank != i || !ctxs[i]->comm))
      return fail(RSF_ERR_STATE, "%s: ctxs[%d] is not rank %d of a %d-rank group made by rsf_comm_init_all", who, i, i, n);
  }
  return RSF_OK;
}

}  // namespace

int rsf_comm_init_all(rsf_ctx *const *ctxs, int32_t n) {
  int rc = check_group(ctxs, n, "rsf_comm_init_all", false);
  if (rc) return rc;
  for (int32_t i = 0; i < n; ++i)
    if (ctxs[i]->world) return fail(RSF_ERR_STATE, "rsf_comm_init_all: ctxs[%d] already has a communicator (rsf_comm_destroy first)", i);
  const Rccl *R = rccl();
  if (!R) return fail(RSF_ERR_UNSUPPORTED, "rsf_comm_init_all: RCCL (librccl.so) could not be loaded");
  std::vector<int> devs(n);
  std::vector<ncclComm_t> comms(n, nullptr);
  for (int32_t i = 0; i < n; ++i) devs[i] = ctxs[i]->device;
  RCCL_TRY(R, R->comm_init_all(comms.data(), n, devs.data()));
  for (int32_t i = 0; i < n; ++i) { ctxs[i]->comm = comms[i]; ctxs[i]->world = n; ctxs[i]->rank = i; }
  return RSF_OK;
}

int rsf_pool_allgather_all(rsf_ctx *const *ctxs, int32_t n, const double *const *send, int64_t count, double *const *recv) {
  int rc = check_group(ctxs, n, "rsf_pool_allgather_all", true);
  if (rc) return rc;
  if (!send || !recv || count < 1) return fail(RSF_ERR_INVALID, "rsf_pool_allgather_all: bad argument");
  for (int32_t i = 0; i < n; ++i)
    if (!send[i] || !recv[i]) return fail(RSF_ERR_INVALID, "rsf_pool_allgather_all: send[%d] / recv[%d] is NULL", i, i);
  const Rccl *R = rccl();
  const size_t bytes = (size_t)count * sizeof(double);
  std::vector<const void *> ds(n);
  std::vector<void *> dr(n);
  for (int32_t i = 0; i < n; ++i) {
    DeviceGuard guard(ctxs[i]->device);
    if (!guard.ok) return fail(RSF_ERR_DEVICE, "rsf_pool_allgather_all: cannot select device %d", ctxs[i]->device);
    if ((rc = stage_in(ctxs[i], 0, send[i], bytes, &ds[i]))) return rc;
    if ((rc = stage_out(ctxs[i], 1, recv[i], bytes * (size_t)n, &dr[i]))) return rc;
  }
  RCCL_TRY(R, R->group_start());
  for (int32_t i = 0; i < n; ++i) {
    DeviceGuard guard(ctxs[i]->device);
    const ncclResult_t e = R->all_gather(ds[i], dr[i], (size_t)count, ncclFloat64, ctxs[i]->comm, ctxs[i]->stream);
    if (e != ncclSuccess) { (void)R->group_end(); return fail(RSF_ERR_DEVICE, "ncclAllGather (rank %d) -> %s", i, R->error_string(e)); }
  }
  RCCL_TRY(R, R->group_end());
  for (int32_t i = 0; i < n; ++i) {
    DeviceGuard guard(ctxs[i]->device);
    if ((rc = copy_back(ctxs[i], 1, recv[i], bytes * (size_t)n))) return rc;
    if ((rc = finish(ctxs[i]))) return rc;
  }
  return RSF_OK;
}

int rsf_pool_allreduce_sum_all(rsf_ctx *const *ctxs, int32_t n, double *const *bufs, int64_t count) {
  int rc = check_group(ctxs, n, "rsf_pool_allreduce_sum_all", true);
  if (rc) return rc;
  if (!bufs || count < 1) return fail(RSF_ERR_INVALID, "rsf_pool_allreduce_sum_all: bad argument");
  for (int32_t i = 0; i < n; ++i)
    if (!bufs[i]) return fail(RSF_ERR_INVALID, "rsf_pool_allreduce_sum_all: bufs[%d] is NULL", i);
  const Rccl *R = rccl();
  const size_t bytes = (size_t)count * sizeof(double);
  std::vector<const void *> ds(n);
  for (int32_t i = 0; i < n; ++i) {
    DeviceGuard guard(ctxs[i]->device);
    if (!guard.ok) return fail(RSF_ERR_DEVICE, "rsf_pool_allreduce_sum_all: cannot select device %d", ctxs[i]->device);
    if ((rc = stage_in(ctxs[i], 0, bufs[i], bytes, &ds[i]))) return rc;
  }
  RCCL_TRY(R, R->group_start());
  for (int32_t i = 0; i < n; ++i) {
    DeviceGuard guard(ctxs[i]->device);
    const ncclResult_t e = R->all_reduce(ds[i], (void *)ds[i], (size_t)count, ncclFloat64, ncclSum, ctxs[i]->comm, ctxs[i]->stream);
    if (e != ncclSuccess) { (void)R->group_end(); return fail(RSF_ERR_DEVICE, "ncclAllReduce (rank %d) -> %s", i, R->error_string(e)); }
  }
  RCCL_TRY(R, R->group_end());
  for (int32_t i = 0; i < n; ++i) {
    DeviceGuard guard(ctxs[i]->device);
    if (host_mem(ctxs[i])) HIP_TRY(hipMemcpyAsync(bufs[i], ds[i], bytes, hipMemcpyDeviceToHost, ctxs[i]->stream));
    if ((rc = finish(ctxs[i]))) return rc;
  }
  return RSF_OK;
}

int rsf_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  if (!ctr || !key || !out) return fail(RSF_ERR_INVALID, "rsf_philox4x32_10: NULL argument");
  uint32_t *d = nullptr;
  HIP_TRY(hipMalloc(&d, 4 * sizeof(uint32_t)));
  hipLaunchKernelGGL(probe_philox_kernel, dim3(1), dim3(64), 0, nullptr, ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], d);
  hipError_t e = hipMemcpy(out, d, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(RSF_ERR_DEVICE, "rsf_philox4x32_10: %s", hipGetErrorString(e));
  return RSF_OK;
}

int rsf_mcmc_draws(uint64_t seed, int64_t chain, int64_t iteration, int32_t d, double shape, double *z, double *u,
                   double *g) {
  if (d < 1 || d > 3) return fail(RSF_ERR_INVALID, "rsf_mcmc_draws: n_params out of range");
  double *dev = nullptr, h[5];
  HIP_TRY(hipMalloc(&dev, sizeof h));
  hipLaunchKernelGGL(probe_draws_kernel, dim3(1), dim3(64), 0, nullptr, seed, (uint64_t)chain, (uint32_t)iteration, d, shape, dev);
  hipError_t e = hipMemcpy(h, dev, sizeof h, hipMemcpyDeviceToHost);
  (void)hipFree(dev);
  if (e != hipSuccess) return fail(RSF_ERR_DEVICE, "rsf_mcmc_draws: %s", hipGetErrorString(e));
  if (z) for (int p = 0; p < d; ++p) z[p] = h[p];
  if (u) *u = h[3];
  if (g) *g = h[4];
  return RSF_OK;
}

int rsf_mcmc_adapt(int32_t d, int32_t n, const double *window, int32_t adapt_mode, int32_t prior_len, double *V_out) {
  if ((d != 1 && d != 3) || n < 1 || !window || !V_out || (adapt_mode != RSF_ADAPT_REFERENCE_DICT && adapt_mode != RSF_ADAPT_AM))
    return fail(RSF_ERR_INVALID, "rsf_mcmc_adapt: bad argument");
  if (adapt_mode == RSF_ADAPT_REFERENCE_DICT && d != 1)
    return fail(RSF_ERR_UNSUPPORTED, "rsf_mcmc_adapt: reference_dict adaptation is defined for 1 parameter only");
  if (adapt_mode == RSF_ADAPT_REFERENCE_DICT && n > RSF_DICT_MAX_INTERVAL)
    return fail(RSF_ERR_UNSUPPORTED, "rsf_mcmc_adapt: reference_dict windows hold at most %d samples", RSF_DICT_MAX_INTERVAL);
  double *dev = nullptr, h[10];
  const size_t wb = (size_t)n * d * sizeof(double);
  HIP_TRY(hipMalloc(&dev, wb + sizeof h));
  hipError_t e = hipMemcpy(dev + 10, window, wb, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(probe_adapt_kernel, dim3(1), dim3(64), 0, nullptr, (int)d, (int)n, (const double *)(dev + 10), (int)adapt_mode,
                       2.38 * 2.38 / (double)(prior_len > 0 ? prior_len : 2), dev);
    e = hipMemcpy(h, dev, sizeof h, hipMemcpyDeviceToHost);
  }
  (void)hipFree(dev);
  if (e != hipSuccess) return fail(RSF_ERR_DEVICE, "rsf_mcmc_adapt: %s", hipGetErrorString(e));
  if (h[d * d] == 0.0) return fail(RSF_ERR_NOT_POSDEF, "rsf_mcmc_adapt: the window's covariance is not positive definite");
  for (int i = 0; i < d * d; ++i) V_out[i] = h[i];
  return RSF_OK;
}

}  // extern "C"
