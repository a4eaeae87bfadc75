// rsf_kernels.h — the gfx950 kernels of the hot path (included by rsf_hip.hip, and by tools/ that build one kernel alone).
//
// One lane = one chain, wave64 = 64 independent chains, fp64 VALU bound; no MFMA — the path is an elementwise ODE
// recurrence plus per-lane reductions, not a contraction:
//   forward_kernel  K1  batched RateStateModel.evaluate + SSq        (RateStateModel.py:188-395, MCMC.py:381-387)
//   init_kernel     K4  compute_initial_covariance + initial SSq     (MCMC.py:244-266, 468)
//   mcmc_kernel     K2  n_iters fused Metropolis iterations          (MCMC.py:494-527)
//   pool_*          posterior post-processing of the pooled draws    (RSF.py:717-746)
//   probe_*         K3  Philox / variate self-test entry points
// The chain-independent tables (loading velocity V_l at the RK4 stage times, observation) are staged through LDS once per
// workgroup (or per chunk when they exceed the LDS budget) and read as wave-wide broadcasts; per-chain state lives in
// registers for the whole launch and touches HBM only at launch start/end plus one coalesced trace row per iteration.
//
// Per-chain state in HBM is STRUCTURE OF ARRAYS — q[p][C], V[e][C], window sums likewise — so that lane i of a wave reads
// element i of a contiguous 512-byte run whatever the number of parameters (the C ABI's [C][d] layout is transposed at
// rsf_mcmc_init / get_state / set_state, rsf_hip.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/rsf_abi.h"
#include "rsf_device.h"
#include "rsf_device_dop853.h"
#include "rsf_device_f32.h"

namespace rsfk {

using rsf::Consts;

enum Mode : int { RK4_F64 = 0, RK4_F32 = 1, DOP853 = 2 };  // how the ODE is integrated (rsf_model.flags)

constexpr int kMaxBlock = 256;  // 4 waves: one per SIMD of a CU
// Register budget of the sampler kernels: at least this many workgroups per CU, i.e. waves per SIMD (2 => at most 256 of
// the 512 unified registers per lane).  cfg2 runs 4 waves per SIMD worth of chains, so a kernel that drifts above 256
// registers would run it in four rounds instead of two; the DOP853 sampler is held to the same budget (unbounded it took
// 300 registers and ran one wave per SIMD whatever the chain count: profiles/r02/dop853_occupancy_ab.log).
constexpr int kMinBlocks = 2;
// three-parameter sampler: TIGHT loop trips of kD3Trip * kTightUnroll steps like the one-parameter sampler's 2 * (+2.3 %
// over 1 at 131 072 chains x nsteps 4000; the few spills it costs all lie outside the loops)
constexpr int kD3Trip = 2;
// LDS per workgroup for the loading table + observation chunk.  Two workgroups per CU (kMinBlocks) at 56 KiB each fit the
// CU's 160 KiB next to the samplers' per-lane slots (up to 24 KiB: Cholesky factors, parked chain state); nsteps 2000
// (48 KB) stays resident for the whole launch instead of being staged twice per proposal (+1.3 % at cfg2).  The chunk LENGTH
// kc is sized for tables of doubles (rsf_set_model); the float32 SAMPLER, whose tables are floats, has its own (kc32: nsteps
// 4000 is one resident chunk of 48 KB there), the other float32 kernels share kc with the float64 init kernel of that mode.
constexpr size_t kLdsBudget = 56 * 1024;
// per-lane LDS slots (doubles) of the float64 RK4 sampler behind the table chunk: D = 3: the Cholesky factor's six; then the
// chain's point, sigma^2, SSq and log u parked across the forward solve (mcmc_kernel)
template <int D>
constexpr int kParkSlots = (D == 3 ? 6 : 0) + D + 3;

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
template <bool DAMP, bool WANT_SSQ, bool WANT_ACC, int MODE>
__global__ void __launch_bounds__(kMaxBlock)
forward_kernel(Consts K, int64_t n, const double *__restrict__ dc, const double *__restrict__ a,
               const double *__restrict__ b, double *__restrict__ ssq_out, double *__restrict__ acc_out) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < n;
  const double dci = active ? dc[i] : 1.0;
  const double ai = (active && a) ? a[i] : K.a_def;
  const double bi = (active && b) ? b[i] : K.b_def;
  double *acc_i = WANT_ACC ? acc_out + i : nullptr;
  const bool resident = K.nchunks == 1;
  double ssq;
  if constexpr (MODE == RK4_F32) {
    float *lds32 = reinterpret_cast<float *>(lds);
    if (resident) rsf::f32::stage_chunk32(lds32, K, 1, K.nout - 1);
    ssq = rsf::f32::solve32<DAMP, WANT_SSQ, WANT_ACC>(lds32, K, resident, active, dci, ai, bi, acc_i, n);
  } else {
    if constexpr (MODE == DOP853) {
      if (resident) rsf::dp::stage_chunk_dp(lds, K, 1, K.nout - 1);
      ssq = rsf::dp::solve<DAMP, WANT_SSQ, WANT_ACC>(lds, K, resident, active, dci, ai, bi, acc_i, n);
    } else {
      if (resident) rsf::stage_chunk(lds, K, 1, K.nout - 1);
      rsf::Wave W;  // every lane's result is wanted: no early rejection (thr = +inf), statistics unused
      ssq = rsf::solve<DAMP, WANT_SSQ, WANT_ACC, 2 * rsf::kTightUnroll>(lds, K, resident, active, dci, ai, bi, INFINITY, acc_i, n, W);
    }
  }
  if (WANT_SSQ && active) ssq_out[i] = ssq;
}

struct InitArgs {
  int64_t C;
  double fd;       // forward-difference relative step, MCMC.py:251
  double inv_dof;  // 1 / (nout - len(qpriors)), MCMC.py:261
  double width[RSF_MAX_PARAMS];  // hi - lo of the prior box (three-parameter chains: initial_covariance)
  const double *q0;  // [d][C]
  double *ssq, *std2, *V;  // [C], [C], [d*d][C]
};

// The initial proposal covariance from the sensitivities' Gram matrix X^T X and sigma^2_0.
// One parameter — the reference's sampler: Vstart = sigma^2 (X^T X)^-1, MCMC.py:265-266, as it stands.
// Three parameters (Dc, a, b) — this build's extension (BASELINE config 5), where that formula does not give a proposal:
// the series depends on Dc and a almost only through their product (relative sensitivities equal to five digits, correlation
// eigenvalue 2e-11) and hardly at all on b (3000 times smaller), so (X^T X)^-1 is astronomically wide along a ridge — and
// what little it says there is forward-difference rounding.  The data do not identify those directions; the PRIOR does.  So
// the box prior enters the way a Gaussian of the same variance would, in coordinates u_p = (q_p - lo_p) / w_p that make the box a
// unit cube:        M = W (X^T X) W / sigma^2 + 12 I,      V = W M^-1 W,      W = diag(w_p = hi_p - lo_p)
// (a uniform variable on a unit interval has variance 1/12).  M is symmetric positive definite with every eigenvalue >= 12
// (condition number ~4e3 at the BASELINE problem): no guard, no fallback.  Identified directions get their Gauss-Newton
// width, unidentified ones the width of the box.
template <int D>
__device__ __forceinline__ void initial_covariance(const double *xtx, double std2, const double *width, double *V) {
  if constexpr (D == 1) {
    V[0] = std2 * (1.0 / xtx[0]);
  } else {
    double M[D * D], Mi[D * D];
    const double is2 = 1.0 / std2;
#pragma unroll
    for (int p = 0; p < D; ++p)
#pragma unroll
      for (int r = 0; r < D; ++r) M[p * D + r] = (width[p] * xtx[p * D + r] * width[r]) * is2 + (p == r ? 12.0 : 0.0);
    rsf::sym_inverse<D>(M, Mi);
#pragma unroll
    for (int p = 0; p < D; ++p)
#pragma unroll
      for (int r = 0; r < D; ++r) V[p * D + r] = width[p] * Mi[p * D + r] * width[r];
  }
}

// compute_initial_covariance + the initial SSq (MCMC.py:244-266, 468): ONE LANE PER TRAJECTORY.  A chain owns a group of
// G = D + 1 adjacent lanes (a pair for one parameter, a quad for three): lane 0 of the group integrates the chain's start
// point, lane p + 1 the point with parameter p moved by the forward-difference step (MCMC.py:251) — each with the sampler's
// own straight-line tier code and nothing but the forward kernel's registers.  Where an output sample completes, the
// group's lanes exchange their acceleration samples by lane shuffles: every lane forms its sensitivity against lane 0's
// sample (perturbed value in the denominator, MCMC.py:264), lane 0 collects them and accumulates the residual and X^T X,
// sample by sample, without storing trajectories.  (Until round 4 ONE lane carried all 1 + D trajectories in lockstep: four
// sets of lane constants and states, 256 VGPRs + 42-120 AGPRs of spills, one wave per SIMD.)
// The acceleration sample is cv * (sum of the interval's weighted V-derivative sums), like the sampler's (rsf::emit_incr):
// the initial SSq is the value the sampler computes for the same point to rounding, and the difference of two trajectories'
// samples — which a relative step of 1e-6 amplifies a million-fold — does not go through two velocities near V_ref.
template <int D>
struct InitGroup {
  static constexpr int G = D + 1;  // lanes per chain: 2 or 4, a power of two, so a group never straddles a wave
  const unsigned t = threadIdx.x;
  const int tr = (int)(t & (G - 1));                                              // which trajectory of its chain this lane integrates
  const int64_t chain = (int64_t)blockIdx.x * (blockDim.x / G) + (t / G);
  const int lane0 = (int)((t & 63) & ~(unsigned)(G - 1));                         // the group's first lane within the wave
  double xtx[D * D], ssq = 0.0;

  __device__ __forceinline__ InitGroup() {
#pragma unroll
    for (int e = 0; e < D * D; ++e) xtx[e] = 0.0;
  }
  // the observation series of the workgroup's chain group (all of a workgroup's chains belong to one)
  __device__ __forceinline__ void select_group(Consts &K) const {
    if (K.group_chains > 0) K.data += (((int64_t)blockIdx.x * (blockDim.x / G)) / K.group_chains) * K.nout;
  }
  // this lane's parameter vector (Dc, a, b) and, for a perturbed trajectory, 1 / (perturbed value * step)
  __device__ __forceinline__ void parameters(const Consts &K, const InitArgs &A, bool active, double (&pq)[3], double &inv_den) const {
    pq[0] = 1000.0; pq[1] = K.a_def; pq[2] = K.b_def;
    if (active) {
      pq[0] = A.q0[chain];
      if (D == 3) { pq[1] = A.q0[A.C + chain]; pq[2] = A.q0[2 * A.C + chain]; }
    }
    inv_den = 0.0;
#pragma unroll
    for (int p = 0; p < D; ++p)
      if (tr == p + 1) {
        pq[p] = pq[p] * (1 + A.fd);
        inv_den = 1.0 / (pq[p] * A.fd);  // perturbed value in the denominator, MCMC.py:264
      }
  }
  // the value lane `SRC` of this lane's group holds.  A group is an aligned pair or quad of lanes, so this is a DPP quad_perm move
  // (two, for the two halves of a double) in the VALU — no trip through the LDS pipe as a general shuffle (ds_bpermute) takes
  template <int SRC>
  static __device__ __forceinline__ double from_lane(double v) {
    constexpr int ctrl = G == 4 ? (SRC | SRC << 2 | SRC << 4 | SRC << 6) : (SRC | SRC << 2 | (2 + SRC) << 4 | (2 + SRC) << 6);
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
  }
  template <int P>
  __device__ __forceinline__ void gather(double x, double (&xs)[D]) const {
    if constexpr (P < D) {
      xs[P] = from_lane<P + 1>(x);
      gather<P + 1>(x, xs);
    }
  }
  // an output sample is complete: ak = this lane's acceleration sample, obs the observation (every lane of the group calls, in
  // converged control flow).  Only the upper triangle of X^T X is accumulated; finish() mirrors it
  __device__ __forceinline__ void sample(double ak, double obs, double inv_den) {
    const double ak0 = from_lane<0>(ak);
    const double x = (ak - ak0) * inv_den;  // lane p + 1: the sensitivity to parameter p; lane 0: 0
    double xs[D];
    gather<0>(x, xs);
    const double r = ak - obs;  // meaningful in lane 0 (the others accumulate values nobody reads)
    ssq = __builtin_fma(r, r, ssq);
#pragma unroll
    for (int p = 0; p < D; ++p)
#pragma unroll
      for (int r2 = p; r2 < D; ++r2) xtx[p * D + r2] = __builtin_fma(xs[p], xs[r2], xtx[p * D + r2]);
  }
  __device__ __forceinline__ void finish(const InitArgs &A, bool active) const {
    if (active && tr == 0) {
      const double std2 = ssq * A.inv_dof;
      double V[D * D], M[D * D];
#pragma unroll
      for (int p = 0; p < D; ++p)
#pragma unroll
        for (int r2 = 0; r2 < D; ++r2) M[p * D + r2] = xtx[p <= r2 ? p * D + r2 : r2 * D + p];
      initial_covariance<D>(M, std2, A.width, V);
#pragma unroll
      for (int e = 0; e < D * D; ++e) A.V[e * A.C + chain] = V[e];  // MCMC.py:266
      A.std2[chain] = std2;
      A.ssq[chain] = ssq;
    }
  }
};

template <int D, bool DAMP>
__global__ void __launch_bounds__(kMaxBlock, kMinBlocks) init_kernel(Consts K, InitArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int NU = 8;
  static_assert(rsf::kResync % NU == 0, "the resync test looks at the first step of a trip");
  InitGroup<D> grp;
  grp.select_group(K);
  const bool active = grp.chain < A.C;
  double pq[3], inv_den;
  grp.parameters(K, A, active, pq, inv_den);
  const rsf::Lane L = rsf::make_lane(pq[0], pq[1], pq[2], K);
  rsf::State st = rsf::initial_state(pq[0], L, K);
  double dsum = 0.0;
  if (active) { const double d0 = K.data[0]; grp.ssq = d0 * d0; }
  const double *ld = lds + rsf::lds_data_offset(K);
  int phase = 0;  // RK4 steps since the last output sample (wave-uniform)
  for (int k0 = 1; k0 < K.nout; k0 += K.kc) {
    const int kn = min(K.kc, K.nout - k0);
    rsf::stage_chunk(lds, K, k0, kn);
    const int nsteps = K.S * kn;
    int ko = 0;
    auto emit = [&]() {
      grp.sample(dsum * L.cv, ld[ko], inv_den);
      dsum = 0.0;
      ++ko;
    };
    // (every lane of the wave integrates — lanes past the last chain carry a harmless Dc = 1000 — so that the shuffles
    // and the wave-uniform tier decisions see whole waves)
    int tier = rsf::start_tier(L, K);
    int r = 0;
    auto trip = [&](auto tier_tag, auto nu_tag) {  // one trip of tier T, NUT steps; a tripped guard: that lane redoes it in full
      constexpr int T = decltype(tier_tag)::value, NUT = decltype(nu_tag)::value;
      const double *v = lds + 2 * r;
      rsf::Lane Lt = L;
      rsf::set_tier<T>(Lt);
      const rsf::State save = st;
      double dv[NUT];
      rsf::tier_enter<T>(st, Lt);
      const bool bad = rsf::trip_fast<DAMP, T, NUT>(v, Lt, K, st, dv);
      rsf::tier_leave<T>(st, Lt);
      const bool any_bad = rsf::ballot(bad) != 0;
      if (__builtin_expect(any_bad, 0)) {
        if (bad) {  // back to the trip's start (the plain state: saved before tier_enter) and through it with full evaluations
          st = save;
          rsf::trip_cold_plain<DAMP, NUT>(v, L, K, st, dv);
        }
      }
#pragma unroll
      for (int j = 0; j < NUT; ++j) {
        dsum += dv[j];
        if (++phase == K.S) { phase = 0; emit(); }
      }
      return any_bad;
    };
    for (; r + NU <= nsteps; r += NU) {
      if ((r & (rsf::kResync - 1)) == 0) rsf::eval_full(st.ms, st.x, L, K, st.w, st.rx);
      bool any_bad;
      if (tier == rsf::TIGHT) any_bad = trip(std::integral_constant<int, rsf::TIGHT>{}, std::integral_constant<int, NU>{});
      else if (tier == rsf::NARROW) any_bad = trip(std::integral_constant<int, rsf::NARROW>{}, std::integral_constant<int, NU>{});
      else any_bad = trip(std::integral_constant<int, rsf::WIDE>{}, std::integral_constant<int, NU>{});
      if (any_bad && tier < rsf::WIDE) ++tier;
    }
    for (; r < nsteps; ++r) {  // fewer than NU steps left in the chunk: one at a time
      if ((r & (rsf::kResync - 1)) == 0) rsf::eval_full(st.ms, st.x, L, K, st.w, st.rx);
      trip(std::integral_constant<int, rsf::WIDE>{}, std::integral_constant<int, 1>{});
    }
  }
  grp.finish(A, active);
}

// compute_initial_covariance + initial SSq in the reference's DOP853 scheme, one lane per trajectory like init_kernel: every
// lane takes its own dop853 calls interval by interval (its own carried step size); the group's lanes meet at every
// output sample.
template <int D, bool DAMP>
__global__ void __launch_bounds__(kMaxBlock, kMinBlocks) init_dp_kernel(Consts K, InitArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  InitGroup<D> grp;
  grp.select_group(K);
  const bool active = grp.chain < A.C;
  double pq[3], inv_den;
  grp.parameters(K, A, active, pq, inv_den);
  const rsf::dp::LaneD L = rsf::dp::make_lane_dp(pq[0], pq[1], pq[2]);
  rsf::dp::Carry cw = rsf::dp::fresh_carry();
  double y[3] = {K.mu0, pq[0] / K.V_ref, K.V_ref}, x = K.t0, vprev = K.V_ref;
  bool failed = false;
  if (active) { const double d0 = K.data[0]; grp.ssq = d0 * d0; }
  const double *ld = lds + rsf::dp::lds_data_offset_dp(K);
  const double delta_t = K.dt;
  for (int k0 = 1; k0 < K.nout; k0 += K.kc) {
    const int kn = min(K.kc, K.nout - k0);
    rsf::dp::stage_chunk_dp(lds, K, k0, kn);
    for (int kk = 0; kk < kn; ++kk) {
      double ak = 0.0;  // a trajectory whose integrator failed leaves zeros, like the reference (RateStateModel.py:361-366, 381)
      if (!failed) {
        failed = !rsf::dp::call<DAMP>(K, L, lds + rsf::dp::kTab * kk, x, x + delta_t, y, cw, true);
        ak = (y[2] - vprev) * K.inv_dt;
        vprev = y[2];
      }
      grp.sample(ak, ld[kk], inv_den);
    }
  }
  grp.finish(A, active);
}

// float32 mode: the sampler compares sums of squares from float32 solves, so the initial SSq (computed by the
// float64 init kernel together with the float64-only sensitivities) is replaced by its float32 value.
template <int D, bool DAMP>
__global__ void __launch_bounds__(kMaxBlock) ssq32_kernel(Consts K, int64_t C, const double *q, double *ssq) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  rsf::select_group(K);
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < C;
  const double dc = active ? q[i] : 1.0;
  const double a = (active && D == 3) ? q[C + i] : K.a_def, b = (active && D == 3) ? q[2 * C + i] : K.b_def;
  const double s = rsf::f32::solve32<DAMP, true, false>(reinterpret_cast<float *>(lds), K, false, active, dc, a, b, nullptr, 0);
  if (active) ssq[i] = s;
}

// The accept test of MCMC.py:327-331: log alpha = clip(0.5 (SSq_prev - SSq_new) / sigma^2, -inf, 0) > log u.  np.clip keeps a
// NaN, and NaN > log u is False: a proposal whose series blew up (a stiff small-Dc lane under fixed-step RK4: Inf - Inf) is
// REJECTED.  fmin(x, 0) would not do: IEEE minNum returns the operand that is not NaN, i.e. 0 > log u, accepted — which is
// what this kernel did until round 4, unnoticed because no test before the wide-proposal ones produced a non-finite sum.
__device__ __forceinline__ bool accept_test(double ratio, double log_u) {
  const double logalpha = ratio > 0.0 ? 0.0 : ratio;  // NaN > 0 is false: NaN stays NaN
  return logalpha > log_u;                             // NaN compares false => reject
}

// The proposal of MCMC.py:497 from the chain's point, the lower Cholesky factor of its proposal covariance (row-major
// lower triangle, D (D + 1) / 2 entries) and D standard normals — one definition, so that rsf_mcmc_propose announces
// exactly the point the sampler kernel will evaluate.
template <int D, typename F>
__device__ __forceinline__ void propose(const double (&q)[D], F factor, const double *z, double (&qn)[D]) {
  int e = 0;
#pragma unroll
  for (int p = 0; p < D; ++p) {
    double s = q[p];
#pragma unroll
    for (int r = 0; r <= p; ++r) s = __builtin_fma(factor(e++), z[r], s);
    qn[p] = s;
  }
}

// (ARGS: a kernel-argument struct with lo[] / hi[] members, indexed in place — a pointer INTO the argument block would make
// these per-lane flat loads, on the vector-memory counter the trace stores sit on, instead of scalar loads)
template <int D, typename ARGS>
__device__ __forceinline__ bool in_box(const double (&qn)[D], const ARGS &A) {
  bool inb = true;
#pragma unroll
  for (int p = 0; p < D; ++p) inb = inb && (qn[p] > A.lo[p]) && (qn[p] < A.hi[p]);  // strict box, MCMC.py:318-320
  return inb;
}

// Per-wave statistics of a sampler launch: wave-uniform 32-bit accumulators (scalar registers), added to the ctx totals
// (McmcArgs::stats, 64-bit) by the wave's first lane.  Index = RSF_CNT_* of rsf_abi.h.
constexpr int kFlushEvery = 16;  // rounds between flushes: 16 x 4000 x 8 sub-steps x 64 lanes < 2^32
// tries a lane gets per forward solve to come up with a proposal inside the prior box (mcmc_kernel): with four, at the 39 %
// out-of-bounds rate of the reference's main.py problem 2 % of the lanes still enter a solve idle, for three short rounds
constexpr int kProposalTries = 4;
struct WaveCounters {
  uint32_t accepted, evaluated, nonfinite, oob, early, wave_solves, wave_skips;
  rsf::Wave W;  // the running solve's control, and steps per tier / redone / lane_steps accumulated over the solves
  __device__ __forceinline__ void reset() {
    accepted = evaluated = nonfinite = oob = early = wave_solves = wave_skips = 0;
    W.steps[0] = W.steps[1] = W.steps[2] = W.steps[3] = W.redone = W.lane_steps = 0;
  }
  __device__ __forceinline__ void flush(unsigned long long *stats) {
    if ((threadIdx.x & 63) == 0) {
      const uint32_t v[RSF_CNT_COUNT] = {accepted, evaluated, nonfinite, oob, early, wave_solves, wave_skips,
                                         W.steps[0], W.steps[1], W.steps[2], W.steps[3], W.redone, W.lane_steps};
#pragma unroll
      for (int k = 0; k < RSF_CNT_COUNT; ++k)
        if (v[k]) atomicAdd(&stats[k], (unsigned long long)v[k]);
    }
    reset();
  }
};

struct McmcArgs {
  int64_t C, chain_offset, n_iters, iter_base;
  uint64_t seed;
  double n0, shape;
  double gd, gc;  // Marsaglia-Tsang constants of Gamma(shape): d = shape - 1/3, c = 1/sqrt(9 d)
  double lo[RSF_MAX_PARAMS], hi[RSF_MAX_PARAMS];
  int32_t adapt_mode, adapt_interval;
  int32_t lc_off;     // D = 3: offset (in doubles) of the per-lane Cholesky factors behind the table chunk in LDS
  double dict_scale;  // 2.38^2 / len(qpriors.keys()), MCMC.py:200 (reference_dict mode)
  double am_eps[RSF_MAX_PARAMS];  // am mode: (1e-6 (hi - lo))^2 added to the history's variances (rsf::window_covariance)
  double *q, *ssq, *std2, *V;           // per-chain state: q[d][C], ssq[C], std2[C], V[d*d][C]
  double *wref, *wsum, *wsq;            // adaptation window (shifted sums): [d][C], [d][C], [d*d][C]
  int32_t *wn;
  double *wbuf;                         // reference_dict: the window's samples themselves, [adapt_interval][C] (rsf::np_cov_1d)
  unsigned long long *stats;            // [RSF_CNT_COUNT] totals since rsf_mcmc_init (rsf_abi.h: rsf_mcmc_counters)
  const double *z, *u, *g;              // replay variates (REPLAY only): z[n][C][d], u[n][C], g[n][C]
  const double *ssq_new;                // INJECT only: the proposals' sums of squares, [n][C] (rsf_mcmc_replay_ssq)
  double *tq, *ts;                      // traces, iteration-major: tq[n][C][d] (the ABI's layout), ts[n][C]
  uint8_t *ta;
};

// INJECT (rsf_mcmc_replay_ssq): the proposals' sums of squares come from the caller — the chain logic alone, no tables, no solve.
template <int D, bool DAMP, bool REPLAY, int MODE, bool INJECT = false>
__global__ void __launch_bounds__(kMaxBlock, kMinBlocks) mcmc_kernel(Consts K, McmcArgs A) {
  static_assert(MODE == RK4_F64 || MODE == DOP853, "the float32 sampler is mcmc_f32x2_kernel (two chains per lane)");
  static_assert(!INJECT || REPLAY, "supplied sums of squares come with supplied variates");
  extern __shared__ __attribute__((aligned(16))) double lds[];
  rsf::select_group(K);
  // Per-chain arrays are addressed as (wave-uniform row pointer)[threadIdx.x]: the row pointer — array + element * C + the
  // workgroup's first chain — is scalar arithmetic, and the lane's share is one small 32-bit offset, so no access keeps a
  // 64-bit per-lane address alive across the forward solve (with plain [e * C + i] indexing the compiler hoisted two dozen
  // of them out of the iteration loop: 224 B of scratch per lane in the three-parameter kernel).
  const int64_t blk = (int64_t)blockIdx.x * blockDim.x;
  const unsigned t = threadIdx.x;
  const int64_t i = blk + t;
  const bool valid = i < A.C;
  auto at = [&](auto *base, int e) { return base + ((int64_t)e * A.C + blk); };  // wave-uniform
  const bool resident = K.nchunks == 1;

  double q[D];
  double ssq = 0.0, std2 = 1.0;
#pragma unroll
  for (int p = 0; p < D; ++p) q[p] = 1.0;
  // What a proposal needs of the covariance V is its lower Cholesky factor (MCMC.py:497).  D = 1: one double, sqrt(V), kept
  // in a register.  D = 3: the factor's six doubles live in LDS, one slot per lane behind the table chunk (lc[e][lane]:
  // conflict-free), formed from V once per launch and again when the chain adapts.  The adaptation window (D + D + D^2
  // doubles of shifted sums) stays in its HBM arrays and is read-modified-written once per proposal when the chain adapts at
  // all — registers across the forward solve belong to the integrator.  (Until round 4 the one-parameter window sat in
  // registers for the launch: seven of them held across every solve, also when nothing adapts — the BASELINE case — and
  // with them the kernel spilled loop-invariant values whose reload, at the top of every solve, waited for the trace row
  // just stored: vector stores and scratch loads share one counter.)
  constexpr bool kWinRegs = false;
  double *lcs = lds + A.lc_off + t;  // D = 3: element e of this lane's factor at lcs[e * blockDim.x]
  double V1 = 0.0;                             // D = 1: the proposal variance
  double wr[D], ws[D], wq[D * D];
  int32_t wn = 0;
  auto store_factor = [&](const double *Lf) {  // row-major lower triangle
    int e = 0;
#pragma unroll
    for (int p = 0; p < D; ++p)
#pragma unroll
      for (int r = 0; r <= p; ++r) lcs[(e++) * blockDim.x] = Lf[p * D + r];
  };
  if (valid) {
#pragma unroll
    for (int p = 0; p < D; ++p) q[p] = at(A.q, p)[t];
    ssq = at(A.ssq, 0)[t];
    std2 = at(A.std2, 0)[t];
  }
  if constexpr (D == 1) {
    if (valid) V1 = at(A.V, 0)[t];
  } else {
    double V[D * D], Lf[D * D];
#pragma unroll
    for (int e = 0; e < D * D; ++e) V[e] = valid ? at(A.V, e)[t] : 0.0;
    rsf::chol_lower<D>(V, Lf);
    store_factor(Lf);
  }
  auto load_window = [&](unsigned t) {
#pragma unroll
    for (int p = 0; p < D; ++p) { wr[p] = at(A.wref, p)[t]; ws[p] = at(A.wsum, p)[t]; }
#pragma unroll
    for (int e = 0; e < D * D; ++e) wq[e] = at(A.wsq, e)[t];
    wn = at(A.wn, 0)[t];
  };
  auto store_window = [&](unsigned t) {
#pragma unroll
    for (int p = 0; p < D; ++p) { at(A.wref, p)[t] = wr[p]; at(A.wsum, p)[t] = ws[p]; }
#pragma unroll
    for (int e = 0; e < D * D; ++e) at(A.wsq, e)[t] = wq[e];
    at(A.wn, 0)[t] = wn;
  };
  if (kWinRegs && A.adapt_mode != RSF_ADAPT_NONE && valid) load_window(t);
  // statistics (rsf_mcmc_counters): wave-uniform popcounts and step counts — scalar registers, nothing per lane — added to
  // the ctx totals by one lane every kFlushEvery proposals (32-bit accumulators cannot overflow in between)
  WaveCounters cnt;
  cnt.reset();

  if (resident && !INJECT) {
    if constexpr (MODE == DOP853) rsf::dp::stage_chunk_dp(lds, K, 1, K.nout - 1);
    else rsf::stage_chunk(lds, K, 1, K.nout - 1);
  }

  // The chain's current point, sigma^2, SSq and the logarithm of the accept test's uniform wait out the forward solve in LDS
  // (per-lane slots behind the table chunk; D = 3: behind the Cholesky factor's six) instead of in registers the
  // integrator needs — the spills per proposal these kernels had otherwise.
  constexpr bool kPark = MODE == RK4_F64;  // (the DOP853 kernel allocates worse with it: measured, tools/one_kernel.sh)
  constexpr int kSlotQ = D == 3 ? 6 : 0, kSlotStd2 = kSlotQ + D, kSlotSsq = kSlotStd2 + 1, kSlotLu = kSlotSsq + 1;
  static_assert(kSlotLu + 1 == kParkSlots<D>, "rsf_hip.hip sizes the launch's LDS with kParkSlots");

  // Every lane walks its OWN chain through iterations 0 .. n_iters-1 (nl: the lane's next one).  A round of the loop below
  // gives every lane that has no proposal in hand its next one; a proposal outside the prior box is a finished iteration
  // as it stands — rejected without a solve and without a uniform (MCMC.py:318-322), sigma^2 updated, trace row written —
  // so while some lane of the wave came out of bounds and tries are left, the wave closes those iterations and goes round
  // again instead of taking them through a forward solve as idle lanes: with the proposal as wide as the reference's own
  // main.py leaves it (list prior: never adapted, MCMC.py:524-527) four proposals in ten are out of bounds, and a lane
  // that ran ahead this way does a solve's worth of work in every solve.  Lanes that already hold an in-bounds proposal
  // wait out those short rounds.  Nothing about a chain changes: its variates are keyed by (chain, iteration), whichever
  // round draws them.  With replayed variates, or with tables staged chunk by chunk behind workgroup barriers (every
  // wave must then take the same number of solves), there is one try: one iteration per lane and round, as before round 4.
  // The chain state loaded above is complete before the loop starts (the asm reads and redefines the registers): otherwise
  // the loads stay "pending" on the loop's no-solve path as far as the compiler's wait-count bookkeeping can tell, and it
  // guards the first use of sigma^2 and SSq in every round with a wait for ALL vector memory — which at run time is the
  // trace row stored a few instructions earlier (stores and loads share the counter): a store's round trip per round.
#pragma unroll
  for (int p = 0; p < D; ++p) asm volatile("" : "+v"(q[p]));
  asm volatile("" : "+v"(ssq), "+v"(std2), "+v"(V1));
  const int32_t n_iters = (int32_t)A.n_iters;
  const int max_tries = (REPLAY || !resident) ? 1 : kProposalTries;
  int32_t nl = valid ? 0 : n_iters;
  int32_t to_adapt = A.adapt_interval - (int32_t)(A.iter_base % A.adapt_interval);  // closes until the next adaptation is due
  bool have = false;  // an in-bounds proposal (qn, lu, thr) waits for the solve
  double qn[D], lu = 0.0, thr = INFINITY;
#pragma unroll
  for (int p = 0; p < D; ++p) qn[p] = 1.0;
  int tries = 0;
  for (int32_t round = 0;; ++round) {
    const bool todo = nl < n_iters;
    if (resident ? !__any(todo) : round == n_iters) break;
    const uint32_t it = (uint32_t)(A.iter_base + nl);
    // the lane's offset as this round sees it: opaque, so that the addresses built from it are formed where they are
    // used instead of being hoisted out of the loop and kept (or spilled) across every forward solve
    unsigned tl = t;
    asm volatile("" : "+v"(tl));
    const uint64_t gid = (uint64_t)(A.chain_offset + blk + tl);  // RNG is keyed by the GLOBAL chain id
    const int64_t row = (int64_t)nl * A.C + blk + tl;  // this lane's row of the traces / replayed variates
    // ---- proposal, MCMC.py:497 ----
    if (todo && !have) {
      double z[4] = {0.0, 0.0, 0.0, 0.0};
      if (REPLAY) {
#pragma unroll
        for (int p = 0; p < D; ++p) z[p] = A.z[row * D + p];
      } else {
        uint32_t w[4];
        rsf::draw_words(A.seed, gid, it, rsf::SLOT_Z01, w);
        rsf::normal_pair(w, z[0], z[1]);
        if (D > 2) {
          rsf::draw_words(A.seed, gid, it, rsf::SLOT_Z2, w);
          rsf::normal_pair(w, z[2], z[3]);
        }
      }
      if constexpr (D == 1) {
        double Lc;
        rsf::chol_lower<1>(&V1, &Lc);  // sqrt(V), or 0 where V is not positive
        propose<1>(q, [&](int) { return Lc; }, z, qn);
      } else {
        propose<D>(q, [&](int e) { return lcs[e * blockDim.x]; }, z, qn);
      }
      have = in_box<D>(qn, A);
      // ---- the accept test's uniform, drawn before the solve: with it the largest sum of squares that could still be
      // accepted is known, and a lane whose running sum passes it stops holding its wave (rsf::Wave).  thr is that bound
      // widened by 1e-9 (rounding in the test itself is ~1e-16): a lane inside the margin simply integrates to the end.
      lu = 0.0;
      thr = INFINITY;
      if (have) {
        double u;
        if (REPLAY) {
          u = A.u[row];
        } else {
          uint32_t w[4];
          rsf::draw_words(A.seed, gid, it, rsf::SLOT_U, w);
          u = rsf::u53(w[0], w[1]);
        }
        // (replaying recorded variates follows the reference's arithmetic to the last bit: IEEE division, libm-grade log;
        //  the sampler proper uses the kernel's own reciprocal and log — the same value to ~1 ulp)
        lu = REPLAY ? log(u) : rsf::rng_log(u);
        if constexpr (MODE == RK4_F64 && !INJECT) {
          const double t0 = __builtin_fma(-2.0 * std2, lu, ssq);  // accept iff ssqn < ssq - 2 std2 log u, MCMC.py:327-331
          thr = __builtin_fma(1e-9, __builtin_fabs(t0), t0);       // NaN (a chain whose state is not finite): never stops early
        }
      }
    }
    const bool oob = todo && !have;  // this lane's iteration is finished without a solve
    // solve now, unless a lane that just closed an out-of-bounds iteration can still come back with a proposal
    const bool solve_now = tries + 1 >= max_tries || !__any(oob);
    tries = solve_now ? 0 : tries + 1;
    double ssqn = 0.0;
    if (solve_now) {
      const unsigned long long inbmask = rsf::ballot(have);
      cnt.evaluated += (uint32_t)__builtin_popcountll(inbmask);
      if constexpr (!INJECT) {  // (with supplied sums of squares there is no solve to count)
        if (inbmask != 0) ++cnt.wave_solves;
        else ++cnt.wave_skips;
      }
      if constexpr (kPark) {
#pragma unroll
        for (int p = 0; p < D; ++p) lcs[(kSlotQ + p) * blockDim.x] = q[p];
        lcs[kSlotStd2 * blockDim.x] = std2;
        lcs[kSlotSsq * blockDim.x] = ssq;
        lcs[kSlotLu * blockDim.x] = lu;
      }
      // ---- likelihood: forward solve only for in-bounds proposals, MCMC.py:322-324 ----
      double an = K.a_def, bn = K.b_def;
      if constexpr (D == 3) { an = qn[1]; bn = qn[2]; }
      // (a wave with no in-bounds lane skips the solve when the tables are resident: no barrier inside)
      if constexpr (INJECT) {
        if (have) ssqn = A.ssq_new[row];
      } else if (!resident || inbmask != 0) {
        if constexpr (MODE == DOP853) {
          ssqn = rsf::dp::solve<DAMP, true, false>(lds, K, resident, have, qn[0], an, bn, nullptr, 0);
        } else {
          ssqn = rsf::solve<DAMP, true, false, (D == 1 ? 2 : kD3Trip) * rsf::kTightUnroll>(lds, K, resident, have, qn[0], an, bn, thr, nullptr, 0, cnt.W);
          cnt.early += (uint32_t)__builtin_popcountll(inbmask & ~cnt.W.alive);
        }
      }
      if constexpr (kPark) {
        // opaque: the values are re-read, not carried across the solve.  The OFFSET is laundered, not the pointer: a pointer
        // that went through the asm has lost its address space and is read back with flat loads, which sit on the
        // vector-memory counter too — the wait for them then also waits for the trace row stored just before.
        unsigned relaunder = 0;
        asm volatile("" : "+v"(relaunder));
        const double *back = lcs + relaunder;
#pragma unroll
        for (int p = 0; p < D; ++p) q[p] = back[(kSlotQ + p) * blockDim.x];
        std2 = back[kSlotStd2 * blockDim.x];
        ssq = back[kSlotSsq * blockDim.x];
        lu = back[kSlotLu * blockDim.x];
      }
    }
    // ---- close the iteration: of the lanes that were solved, and of the lanes whose proposal was out of bounds ----
    const bool solved = solve_now && have;
    // accept / reject, MCMC.py:327-333
    bool accept = false;
    if (solved) {
      accept = accept_test(REPLAY ? 0.5 * (ssq - ssqn) / std2 : (0.5 * (ssq - ssqn)) * rsf::fm::rcp(std2), lu);
      if (accept) {
        ssq = ssqn;
#pragma unroll
        for (int p = 0; p < D; ++p) q[p] = qn[p];
      }
    }
    cnt.accepted += (uint32_t)__builtin_popcountll(rsf::ballot(accept));
    cnt.nonfinite += (uint32_t)__builtin_popcountll(rsf::ballot(solved && !isfinite(ssqn)));
    cnt.oob += (uint32_t)__builtin_popcountll(rsf::ballot(oob));
    if (solved || oob) {
      // sigma^2 Gibbs update with the post-accept SSq, MCMC.py:158-160
      const double bval = 0.5 * (A.n0 * std2 + ssq);
      const double g = REPLAY ? A.g[row] : rsf::gamma_draw(A.seed, gid, it, A.gd, A.gc);
      std2 = REPLAY ? bval / g : bval * rsf::fm::rcp(g);
      if (A.tq) {
#pragma unroll
        for (int p = 0; p < D; ++p) A.tq[row * D + p] = q[p];
      }
      if (A.ts) A.ts[row] = std2;
      if (A.ta) A.ta[row] = accept ? 1 : 0;
      // adaptation, MCMC.py:200-204, 523-527
      if (A.adapt_mode != RSF_ADAPT_NONE) {
        if (!kWinRegs) load_window(tl);
#pragma unroll
        for (int p = 0; p < D; ++p) {
          ws[p] += q[p] - wr[p];
#pragma unroll
          for (int r = 0; r < D; ++r) wq[p * D + r] += (q[p] - wr[p]) * (q[r] - wr[r]);
        }
        ++wn;
        if (A.adapt_mode == RSF_ADAPT_REFERENCE_DICT) at(A.wbuf, A.adapt_interval - to_adapt)[tl] = q[0];  // slot (iteration % interval)
        if (--to_adapt == 0) {
          to_adapt = A.adapt_interval;
          if (wn >= 2) {
            const double nn = (double)wn;
            double Vn[D * D], Ln[D * D];
            if (A.adapt_mode == RSF_ADAPT_REFERENCE_DICT) {
              // d := len(qpriors.keys()) (2 for {1: lo, 2: hi}), and the Cholesky FACTOR becomes the next covariance
              // (one-parameter chains only: rsf_mcmc_init refuses the mode for D = 3).  The window's covariance in np.cov's own
              // arithmetic, from the samples kept in wbuf (full and in order whenever an adaptation is due).
              Vn[0] = A.dict_scale * rsf::np_cov_1d([&](int k) { return at(A.wbuf, k)[tl]; }, A.adapt_interval);
              if (rsf::chol_lower<1>(Vn, Ln)) V1 = Ln[0];
            } else if (rsf::window_covariance<D>(ws, wq, nn, 2.38 * 2.38 / (double)D, A.am_eps, Vn, Ln)) {
              if constexpr (D == 1) {
                V1 = Vn[0];
              } else {
#pragma unroll
                for (int e = 0; e < D * D; ++e) at(A.V, e)[tl] = Vn[e];
                store_factor(Ln);
              }
            }
          }
          // (am keeps its sums: the covariance of the whole history.  reference_dict forms its window from wbuf and ignores them.)
        }
        if (!kWinRegs) store_window(tl);
      }
      ++nl;
      have = false;
    }
    if ((round & (kFlushEvery - 1)) == kFlushEvery - 1) cnt.flush(A.stats);
  }

  if (valid) {
#pragma unroll
    for (int p = 0; p < D; ++p) at(A.q, p)[t] = q[p];
    if constexpr (D == 1) at(A.V, 0)[t] = V1;
    at(A.ssq, 0)[t] = ssq;
    at(A.std2, 0)[t] = std2;
    if (kWinRegs && A.adapt_mode != RSF_ADAPT_NONE) store_window(t);
  }
  cnt.flush(A.stats);
}

// rsf_mcmc_propose: the proposal the next iteration of mcmc_kernel will make from z, and whether it is inside the box
struct ProposeArgs {
  int64_t C;
  const double *q, *V;  // [d][C], [d*d][C]
  const double *z;      // [C][d]
  double lo[RSF_MAX_PARAMS], hi[RSF_MAX_PARAMS];
  double *qn;           // [C][d]
  uint8_t *inb;         // [C]
};

template <int D>
__global__ void __launch_bounds__(kMaxBlock) propose_kernel(ProposeArgs A) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= A.C) return;
  double q[D], V[D * D], Lf[D * D], z[D], qn[D];
#pragma unroll
  for (int p = 0; p < D; ++p) { q[p] = A.q[p * A.C + i]; z[p] = A.z[i * D + p]; }
#pragma unroll
  for (int e = 0; e < D * D; ++e) V[e] = A.V[e * A.C + i];
  rsf::chol_lower<D>(V, Lf);
  double tri[D * (D + 1) / 2];
  int e = 0;
#pragma unroll
  for (int p = 0; p < D; ++p)
#pragma unroll
    for (int r = 0; r <= p; ++r) tri[e++] = Lf[p * D + r];
  propose<D>(q, [&](int k) { return tri[k]; }, z, qn);
#pragma unroll
  for (int p = 0; p < D; ++p) A.qn[i * D + p] = qn[p];
  A.inb[i] = in_box<D>(qn, A) ? 1 : 0;
}

// The float32 sampler (RSF_FLAG_FP32_SOLVE): the same iteration as mcmc_kernel, with TWO chains per lane, because its
// solve advances two chains per packed instruction (rsf_device_f32.h, solve32x2).  Chain slot s of lane t of workgroup w
// is chain w * 2 * blockDim + s * blockDim + t, so that both slots read and write coalesced runs.  (A second kernel
// rather than a chains-per-lane parameter of mcmc_kernel: written that way, the float64 kernels — which sit at 252-256
// registers — came out with 12-44 B of scratch per lane.)  The sampler logic itself stays float64.
// Per-chain sampler state of one slot:
template <int D>
struct Chain {
  double q[D], ssq, std2;
  double V1;                   // D = 1: the proposal variance
  double wr[D], ws[D], wq[D * D];  // adaptation window (D = 1 only: held in registers for the launch)
  int32_t wn;
  bool valid;
  uint64_t gid;                // RNG is keyed by the GLOBAL chain id
};

template <int D, bool DAMP, bool REPLAY>
__global__ void __launch_bounds__(kMaxBlock, kMinBlocks) __attribute__((amdgpu_num_vgpr(RSF_F32_TRIP_COMPILER_VGPRS / 2)))
mcmc_f32x2_kernel(Consts K, McmcArgs A) {  // (the registers above that count: the solve's private file, rsf_device_f32.h trip32)
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int NC = 2;  // chains per lane
  // Per-chain arrays are addressed as (wave-uniform row pointer)[threadIdx.x]: the row pointer — array + element * C + the
  // slot's first chain — is scalar arithmetic, and the lane's share is one small 32-bit offset, so no access keeps a
  // 64-bit per-lane address alive across the forward solve (with plain [e * C + i] indexing the compiler hoisted two dozen
  // of them out of the iteration loop: 224 B of scratch per lane in the three-parameter kernel).
  const int64_t blk = (int64_t)blockIdx.x * blockDim.x * NC;  // first chain of this workgroup
  if (K.group_chains > 0) K.data += (blk / K.group_chains) * K.nout;  // the observation series of the workgroup's chain group
  const unsigned t = threadIdx.x;
  auto at = [&](auto *base, int e, int s) { return base + ((int64_t)e * A.C + blk + (int64_t)s * blockDim.x); };  // wave-uniform
  const bool resident = K.nchunks == 1;

  // What a proposal needs of the covariance V is its lower Cholesky factor (MCMC.py:497).  D = 1: one double, sqrt(V), kept
  // in a register with the three doubles of the adaptation window.  D = 3: the factor's six doubles live in LDS, one slot
  // per chain behind the table chunk (lc[e][slot][lane]: conflict-free), formed from V once per launch and again when the
  // chain adapts; the window (3 + 3 + 9 doubles of shifted sums) stays in its HBM arrays and is read-modified-written once
  // per proposal when the chain adapts at all — registers across the forward solve belong to the integrator.
  constexpr bool kWinRegs = D == 1;
  double *lcs = lds + A.lc_off + t;  // D = 3: element e of slot s's factor at lcs[(e * NC + s) * blockDim.x]
  Chain<D> ch[NC];
  auto store_factor = [&](int s, const double *Lf) {  // row-major lower triangle
    int e = 0;
#pragma unroll
    for (int p = 0; p < D; ++p)
#pragma unroll
      for (int r = 0; r <= p; ++r) lcs[((e++) * NC + s) * blockDim.x] = Lf[p * D + r];
  };
  auto load_window = [&](int s, unsigned t) {
    Chain<D> &c = ch[s];
#pragma unroll
    for (int p = 0; p < D; ++p) { c.wr[p] = at(A.wref, p, s)[t]; c.ws[p] = at(A.wsum, p, s)[t]; }
#pragma unroll
    for (int e = 0; e < D * D; ++e) c.wq[e] = at(A.wsq, e, s)[t];
    c.wn = at(A.wn, 0, s)[t];
  };
  auto store_window = [&](int s, unsigned t) {
    const Chain<D> &c = ch[s];
#pragma unroll
    for (int p = 0; p < D; ++p) { at(A.wref, p, s)[t] = c.wr[p]; at(A.wsum, p, s)[t] = c.ws[p]; }
#pragma unroll
    for (int e = 0; e < D * D; ++e) at(A.wsq, e, s)[t] = c.wq[e];
    at(A.wn, 0, s)[t] = c.wn;
  };
#pragma unroll
  for (int s = 0; s < NC; ++s) {
    Chain<D> &c = ch[s];
    const int64_t i = blk + (int64_t)s * blockDim.x + t;
    c.valid = i < A.C;
    c.gid = (uint64_t)(A.chain_offset + i);
    c.ssq = 0.0; c.std2 = 1.0; c.V1 = 0.0; c.wn = 0;
#pragma unroll
    for (int p = 0; p < D; ++p) c.q[p] = 1.0;
    if (c.valid) {
#pragma unroll
      for (int p = 0; p < D; ++p) c.q[p] = at(A.q, p, s)[t];
      c.ssq = at(A.ssq, 0, s)[t];
      c.std2 = at(A.std2, 0, s)[t];
    }
    if constexpr (D == 1) {
      if (c.valid) c.V1 = at(A.V, 0, s)[t];
    } else {
      double V[D * D], Lf[D * D];
#pragma unroll
      for (int e = 0; e < D * D; ++e) V[e] = c.valid ? at(A.V, e, s)[t] : 0.0;
      rsf::chol_lower<D>(V, Lf);
      store_factor(s, Lf);
    }
    if (kWinRegs && A.adapt_mode != RSF_ADAPT_NONE && c.valid) load_window(s, t);
  }
  uint32_t n_acc = 0, n_eval = 0, n_nonfinite = 0, n_oob = 0, n_solves = 0;
  rsf::f32::Trips32 trips;  // wave-uniform: what the wave's solves ran (steps_tight = incremental, steps_full, steps_redone)

  float *lds32 = reinterpret_cast<float *>(lds);
  if (resident) rsf::f32::stage_chunk32(lds32, K, 1, K.nout - 1);

  for (int64_t n = 0; n < A.n_iters; ++n) {
    const uint32_t it = (uint32_t)(A.iter_base + n);
    // the lane's offset as this iteration sees it: opaque, so that the addresses built from it are formed where they are
    // used instead of being hoisted out of the loop and kept (or spilled) across every forward solve
    unsigned tl = t;
    asm volatile("" : "+v"(tl));
    double qn[NC][D], ssqn[NC];
    bool inb[NC];
    // ---- proposal, MCMC.py:497 ----
#pragma unroll
    for (int s = 0; s < NC; ++s) {
      const Chain<D> &c = ch[s];
      const int64_t row0 = n * A.C + blk + (int64_t)s * blockDim.x;  // trace row of the slot's first chain (wave-uniform)
      double z[4] = {0.0, 0.0, 0.0, 0.0};
      if (REPLAY) {
        if (c.valid) {
#pragma unroll
          for (int p = 0; p < D; ++p) z[p] = (A.z + row0 * D)[tl * D + p];
        }
      } else {
        uint32_t w[4];
        rsf::draw_words(A.seed, c.gid, it, rsf::SLOT_Z01, w);
        rsf::normal_pair(w, z[0], z[1]);
        if (D > 2) {
          rsf::draw_words(A.seed, c.gid, it, rsf::SLOT_Z2, w);
          rsf::normal_pair(w, z[2], z[3]);
        }
      }
      if constexpr (D == 1) {
        double Lc;
        rsf::chol_lower<1>(&c.V1, &Lc);  // sqrt(V), or 0 where V is not positive
        qn[s][0] = c.q[0] + Lc * z[0];
      } else {
        int e = 0;
#pragma unroll
        for (int p = 0; p < D; ++p) {
          double acc = c.q[p];
#pragma unroll
          for (int r = 0; r <= p; ++r) acc += lcs[((e++) * NC + s) * blockDim.x] * z[r];
          qn[s][p] = acc;
        }
      }
      inb[s] = c.valid;
#pragma unroll
      for (int p = 0; p < D; ++p) inb[s] = inb[s] && (qn[s][p] > A.lo[p]) && (qn[s][p] < A.hi[p]);  // strict box, MCMC.py:318-320
      if (c.valid && !inb[s]) ++n_oob;
      ssqn[s] = 0.0;
    }
    // ---- likelihood: forward solve only for in-bounds proposals, MCMC.py:322-324 ----
    // (a wave with no in-bounds lane skips the solve when the tables are resident: no barrier inside)
    if (!resident || __any(inb[0] || inb[1])) {
      double dcn[2], an[2], bn[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) { dcn[s] = qn[s][0]; an[s] = D == 3 ? qn[s][D - 2] : K.a_def; bn[s] = D == 3 ? qn[s][D - 1] : K.b_def; }
      rsf::f32::solve32x2<DAMP>(lds32, K, resident, inb, dcn, an, bn, ssqn, trips);
      ++n_solves;
    }
#pragma unroll
    for (int s = 0; s < NC; ++s) {
      Chain<D> &c = ch[s];
      const int64_t row0 = n * A.C + blk + (int64_t)s * blockDim.x;
      // ---- accept / reject, MCMC.py:327-333 ----
      bool accept = false;
      if (inb[s]) {
        double u;
        if (REPLAY) {
          u = (A.u + row0)[tl];
        } else {
          uint32_t w[4];
          rsf::draw_words(A.seed, c.gid, it, rsf::SLOT_U, w);
          u = rsf::u53(w[0], w[1]);
        }
        // (replaying recorded variates follows the reference's arithmetic to the last bit: IEEE division, libm-grade log;
        //  the sampler proper uses the kernel's own reciprocal and log — the same value to ~1 ulp)
        accept = accept_test(REPLAY ? 0.5 * (c.ssq - ssqn[s]) / c.std2 : (0.5 * (c.ssq - ssqn[s])) * rsf::fm::rcp(c.std2),
                             REPLAY ? log(u) : rsf::rng_log(u));
        ++n_eval;
        if (!isfinite(ssqn[s])) ++n_nonfinite;
        if (accept) {
          c.ssq = ssqn[s];
#pragma unroll
          for (int p = 0; p < D; ++p) c.q[p] = qn[s][p];
          ++n_acc;
        }
      }
      // ---- sigma^2 Gibbs update with the post-accept SSq, MCMC.py:158-160 ----
      if (c.valid) {
        const double bval = 0.5 * (A.n0 * c.std2 + c.ssq);
        const double g = REPLAY ? (A.g + row0)[tl] : rsf::gamma_draw(A.seed, c.gid, it, A.gd, A.gc);
        c.std2 = REPLAY ? bval / g : bval * rsf::fm::rcp(g);
        if (A.tq) {
#pragma unroll
          for (int p = 0; p < D; ++p) (A.tq + row0 * D)[tl * D + p] = c.q[p];
        }
        if (A.ts) (A.ts + row0)[tl] = c.std2;
        if (A.ta) (A.ta + row0)[tl] = accept ? 1 : 0;
      }
      // ---- adaptation, MCMC.py:200-204, 523-527 ----
      if (A.adapt_mode != RSF_ADAPT_NONE && c.valid) {
        if (!kWinRegs) load_window(s, tl);
#pragma unroll
        for (int p = 0; p < D; ++p) {
          c.ws[p] += c.q[p] - c.wr[p];
#pragma unroll
          for (int r = 0; r < D; ++r) c.wq[p * D + r] += (c.q[p] - c.wr[p]) * (c.q[r] - c.wr[r]);
        }
        ++c.wn;
        if (A.adapt_mode == RSF_ADAPT_REFERENCE_DICT) at(A.wbuf, (int)((A.iter_base + n) % A.adapt_interval), s)[tl] = c.q[0];
        if ((A.iter_base + n + 1) % A.adapt_interval == 0) {
          if (c.wn >= 2) {
            const double nn = (double)c.wn;
            double Vn[D * D], Ln[D * D];
            if (A.adapt_mode == RSF_ADAPT_REFERENCE_DICT) {
              // d := len(qpriors.keys()) (2 for {1: lo, 2: hi}), and the Cholesky FACTOR becomes the next covariance
              // (one-parameter chains only: rsf_mcmc_init refuses the mode for D = 3); np.cov's own arithmetic (mcmc_kernel)
              Vn[0] = A.dict_scale * rsf::np_cov_1d([&](int k) { return at(A.wbuf, k, s)[tl]; }, A.adapt_interval);
              if (rsf::chol_lower<1>(Vn, Ln)) c.V1 = Ln[0];
            } else if (rsf::window_covariance<D>(c.ws, c.wq, nn, 2.38 * 2.38 / (double)D, A.am_eps, Vn, Ln)) {
              if constexpr (D == 1) {
                c.V1 = Vn[0];
              } else {
#pragma unroll
                for (int e = 0; e < D * D; ++e) at(A.V, e, s)[tl] = Vn[e];
                store_factor(s, Ln);
              }
            }
          }
          // (am keeps its sums: the covariance of the whole history; mcmc_kernel)
        }
        if (!kWinRegs) store_window(s, tl);
      }
    }
  }

#pragma unroll
  for (int s = 0; s < NC; ++s) {
    const Chain<D> &c = ch[s];
    if (c.valid) {
#pragma unroll
      for (int p = 0; p < D; ++p) at(A.q, p, s)[t] = c.q[p];
      if constexpr (D == 1) at(A.V, 0, s)[t] = c.V1;
      at(A.ssq, 0, s)[t] = c.ssq;
      at(A.std2, 0, s)[t] = c.std2;
      if (kWinRegs && A.adapt_mode != RSF_ADAPT_NONE) store_window(s, t);
    }
  }
  // statistics: wave shuffle reduction, one atomic per wave and counter
  const unsigned long long s0 = rsf::wave_sum(n_acc), s1 = rsf::wave_sum(n_eval), s2 = rsf::wave_sum(n_nonfinite), s3 = rsf::wave_sum(n_oob);
  if ((threadIdx.x & 63) == 0) {
    if (s0) atomicAdd(&A.stats[RSF_CNT_ACCEPTED], s0);
    if (s1) atomicAdd(&A.stats[RSF_CNT_EVALUATED], s1);
    if (s2) atomicAdd(&A.stats[RSF_CNT_NONFINITE], s2);
    if (s3) atomicAdd(&A.stats[RSF_CNT_OUT_OF_BOUNDS], s3);
    if (n_solves) atomicAdd(&A.stats[RSF_CNT_WAVE_SOLVES], (unsigned long long)n_solves);
    if (trips.incr) atomicAdd(&A.stats[RSF_CNT_STEPS_TIGHT], (unsigned long long)trips.incr);
    if (trips.full) atomicAdd(&A.stats[RSF_CNT_STEPS_FULL], (unsigned long long)trips.full);
    if (trips.redone) atomicAdd(&A.stats[RSF_CNT_STEPS_REDONE], (unsigned long long)trips.redone);
  }
}

// rsf_mcmc_adapt: update_covariance_matrix (MCMC.py:200-204) of a window of samples win[n][d] with the sampler's own arithmetic
// (shifted sums about the window's first sample, rsf::window_covariance).  out[0 .. d*d) = the next "Vold" of the
// reference's loop — reference_dict: the Cholesky FACTOR (MCMC.py:203-204, 525); am: the covariance — out[d*d] = 1 if
// positive definite, else 0.
template <int D>
__device__ void adapt_window(int n, const double *win, int mode, double dict_scale, double *out) {
  double ws[D], wq[D * D], Vn[D * D], Ln[D * D];
#pragma unroll
  for (int p = 0; p < D; ++p) ws[p] = 0.0;
#pragma unroll
  for (int e = 0; e < D * D; ++e) wq[e] = 0.0;
  for (int k = 0; k < n; ++k)
#pragma unroll
    for (int p = 0; p < D; ++p) {
      ws[p] += win[k * D + p] - win[p];
#pragma unroll
      for (int r = 0; r < D; ++r) wq[p * D + r] += (win[k * D + p] - win[p]) * (win[k * D + r] - win[r]);
    }
  const bool dict = mode == RSF_ADAPT_REFERENCE_DICT;
  bool ok = n >= 2;
  if (dict) {  // one parameter: np.cov's own arithmetic, like the sampler's
    Vn[0] = ok ? dict_scale * rsf::np_cov_1d([&](int k) { return win[k * D]; }, n) : 0.0;
    ok = ok && rsf::chol_lower<1>(Vn, Ln);
  } else {
    const double no_eps[3] = {0.0, 0.0, 0.0};  // the given samples alone: what the sampler computes once its history is these
    ok = ok && rsf::window_covariance<D>(ws, wq, (double)n, 2.38 * 2.38 / (double)D, no_eps, Vn, Ln);
  }
#pragma unroll
  for (int e = 0; e < D * D; ++e) out[e] = dict ? Ln[e] : Vn[e];
  out[D * D] = ok ? 1.0 : 0.0;
}

__global__ void probe_adapt_kernel(int d, int n, const double *win, int mode, double dict_scale, double *out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (d == 1) adapt_window<1>(n, win, mode, dict_scale, out);
  else adapt_window<3>(n, win, mode, dict_scale, out);
}

// [n][d] <-> [d][n] between the C ABI's per-chain layout and the kernels' structure of arrays (d = 3 only; for one
// parameter the two coincide)
__global__ void __launch_bounds__(kMaxBlock) transpose_kernel(int64_t n, int d, const double *__restrict__ src, double *__restrict__ dst, bool to_soa) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int e = 0; e < d; ++e) {
    if (to_soa) dst[(int64_t)e * n + i] = src[i * d + e];
    else dst[i * d + e] = src[(int64_t)e * n + i];
  }
}

// ---------------------------------------------------------------------------------------------
// posterior post-processing on pooled samples (RSF.plot_dist, RSF.py:717-746)
// ---------------------------------------------------------------------------------------------
constexpr int kPoolBlocks = 1024;  // 4 workgroups per CU; partials are combined deterministically (no atomics)

struct PoolPartial {
  double cnt, sum, sumsq, mn, mx;  // sums are taken about a common shift for stability
};

__global__ void __launch_bounds__(kMaxBlock)
pool_moments_kernel(int64_t n, const double *__restrict__ x, int64_t stride, double shift, PoolPartial *__restrict__ part) {
  __shared__ PoolPartial sh[kMaxBlock / 64];
  double cnt = 0.0, sum = 0.0, sumsq = 0.0, mn = INFINITY, mx = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double v = x[i * stride], dlt = v - shift;
    cnt += 1.0; sum += dlt; sumsq = __builtin_fma(dlt, dlt, sumsq);
    mn = fmin(mn, v); mx = fmax(mx, v);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    cnt += __shfl_down(cnt, off, 64); sum += __shfl_down(sum, off, 64); sumsq += __shfl_down(sumsq, off, 64);
    mn = fmin(mn, __shfl_down(mn, off, 64)); mx = fmax(mx, __shfl_down(mx, off, 64));
  }
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = {cnt, sum, sumsq, mn, mx};
  __syncthreads();
  if (threadIdx.x == 0) {
    PoolPartial p = sh[0];
    for (unsigned w = 1; w < blockDim.x / 64; ++w) {
      p.cnt += sh[w].cnt; p.sum += sh[w].sum; p.sumsq += sh[w].sumsq; p.mn = fmin(p.mn, sh[w].mn); p.mx = fmax(p.mx, sh[w].mx);
    }
    part[blockIdx.x] = p;
  }
}

// Fixed-bin histogram (rsf_pool_histogram): HBM-bound, one pass.  Every workgroup counts into an LDS copy of the bins
// (ds_add_u32), then adds its non-empty bins to the global 64-bit counters — integer atomics, so the result does not
// depend on the order of arrival.  The bin of a sample is numpy.histogram's, edge cases included: a first guess
// floor((x - lo) * nbins/(hi - lo)), then numpy's own correction against the bin EDGES np.linspace(lo, hi, nbins + 1)
// (edge b = b * step + lo, two roundings — formed here with contraction switched off), so that a sample
// sitting exactly on an edge — a chain that rejects repeats values like q0 — lands where numpy puts it.
constexpr int kHistMaxBins = 4096;

__device__ __forceinline__ double hist_edge(int b, double lo, double hi, double step, int nbins) {
#pragma clang fp contract(off)  // numpy's edge is a product rounded, then a sum rounded: no fused multiply-add here
  const double m = (double)b * step;
  return b == nbins ? hi : m + lo;
}

__device__ __forceinline__ int hist_bin(double v, double lo, double hi, double scale, double step, int nbins) {
  if (v < lo) return 0;
  if (!(v <= hi)) return nbins + 1;                      // above hi, or NaN
  int b = (int)((v - lo) * scale);
  b = b < nbins ? b : nbins - 1;                         // v == hi (or rounding at the upper edge) -> last bin
  if (v < hist_edge(b, lo, hi, step, nbins)) --b;        // the guess is within one bin of the truth; the edges decide
  if (b != nbins - 1 && v >= hist_edge(b + 1, lo, hi, step, nbins)) ++b;
  return 1 + b;
}

__global__ void __launch_bounds__(kMaxBlock)
pool_hist_kernel(int64_t n, const double *__restrict__ x, int64_t stride, int nbins, double lo, double hi, double scale, double step,
                 unsigned long long *__restrict__ counts) {
  extern __shared__ unsigned int hbins[];
  for (int b = threadIdx.x; b < nbins + 2; b += blockDim.x) hbins[b] = 0u;
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    atomicAdd(&hbins[hist_bin(x[i * stride], lo, hi, scale, step, nbins)], 1u);
  __syncthreads();
  for (int b = threadIdx.x; b < nbins + 2; b += blockDim.x)
    if (hbins[b]) atomicAdd(&counts[b], (unsigned long long)hbins[b]);
}

__global__ void __launch_bounds__(kMaxBlock) pool_hist_finish_kernel(int nb, const unsigned long long *__restrict__ counts, double *__restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < nb) out[b] = (double)counts[b];
}

// Each workgroup owns a contiguous slice of the samples, streamed through LDS in tiles; every thread
// accumulates the kernel sum of its grid points over the slice (LDS broadcast reads).  partial[block][m].
constexpr int kKdeTile = 1024;

__global__ void __launch_bounds__(kMaxBlock)
pool_kde_kernel(int64_t n, const double *__restrict__ x, int64_t stride, int m, const double *__restrict__ grid, double inv2c,
                double *__restrict__ partial) {
  __shared__ double tile[kKdeTile];
  const int64_t per = (n + gridDim.x - 1) / gridDim.x, lo = (int64_t)blockIdx.x * per, hi = min(n, lo + per);
  for (int j0 = 0; j0 < m; j0 += blockDim.x) {
    const int j = j0 + threadIdx.x;
    const double g = j < m ? grid[j] : 0.0;
    double acc = 0.0;
    for (int64_t t0 = lo; t0 < hi; t0 += kKdeTile) {
      const int tn = (int)min((int64_t)kKdeTile, hi - t0);
      __syncthreads();
      for (int t = threadIdx.x; t < tn; t += blockDim.x) tile[t] = x[(t0 + t) * stride];
      __syncthreads();
      for (int t = 0; t < tn; ++t) {
        const double dlt = g - tile[t];
        acc += rsf::fm::exp(-dlt * dlt * inv2c);
      }
    }
    if (j < m) partial[(int64_t)blockIdx.x * m + j] = acc;
  }
}

__global__ void __launch_bounds__(kMaxBlock)
pool_kde_reduce_kernel(int nblocks, int m, const double *__restrict__ partial, double norm, double *__restrict__ density) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= m) return;
  double acc = 0.0;
  for (int b = 0; b < nblocks; ++b) acc += partial[(int64_t)b * m + j];  // fixed order: reproducible
  density[j] = acc * norm;
}

// out[0..3] = philox words (as doubles are not used here): layout documented at the call sites
__global__ void probe_philox_kernel(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                    uint32_t *out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    uint32_t w[4];
    rsf::philox4x32_10(c0, c1, c2, c3, k0, k1, w);
    for (int j = 0; j < 4; ++j) out[j] = w[j];
  }
}

// out = { z0, z1, z2, u, g }
__global__ void probe_draws_kernel(uint64_t seed, uint64_t chain, uint32_t iter, int d, double shape, double *out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    uint32_t w[4];
    double z[4] = {0, 0, 0, 0};
    rsf::draw_words(seed, chain, iter, rsf::SLOT_Z01, w);
    rsf::normal_pair(w, z[0], z[1]);
    if (d > 2) { rsf::draw_words(seed, chain, iter, rsf::SLOT_Z2, w); rsf::normal_pair(w, z[2], z[3]); }
    rsf::draw_words(seed, chain, iter, rsf::SLOT_U, w);
    out[0] = z[0]; out[1] = z[1]; out[2] = z[2];
    out[3] = rsf::u53(w[0], w[1]);
    out[4] = rsf::gamma_draw(seed, chain, iter, shape - 1.0 / 3.0, 1.0 / sqrt(9.0 * (shape - 1.0 / 3.0)));
  }
}

}  // namespace rsfk
