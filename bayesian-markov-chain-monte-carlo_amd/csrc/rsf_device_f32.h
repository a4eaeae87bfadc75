// rsf_device_f32.h — float32 forward solve (BASELINE config 5: "float32 vs float64 tolerance sweep").
//
// Selected per model with RSF_FLAG_FP32_SOLVE.  The ODE integration and the summation of squared residuals run in float32
// (the latter in a way that keeps ~48 bits, Out32); every interface array and the whole sampler logic (proposal, accept
// test, sigma^2 update, adaptation, initial covariance) stay float64.
//
// Same rescaled state as the float64 path — ms = mu/k', x = V_ref theta/Dc — and the same regrouping of the RHS around
// w = v/V_ref with dV/dt in units of vk (rsf_device.h).  The acceleration sample is formed from the step's velocity
// INCREMENT, not from the difference of two velocities near V_ref, which would lose ~4 digits in float32.
//
// What a CHAIN computes (a function of its own parameters and trajectory only — restated step for step by
// oracle/rsf_oracle.c::solve_f32; what the wave around it does is a speed decision and changes no result):
//
//   * It starts in the INCREMENTAL form (round 4; rk4_incr): the state is (w, Rh = (h/2Dc)/x), ms rides along, and every
//     stage of an RK4 step reaches its (w, 1/x) from the step's start point by short series — log1p to rho^2/2, expm1 to
//     dlt^3/6, 1/x' = (1/x)(1 - rho + rho^2) — the float64 TIGHT tier's step (rsf_device.h, rk4_tight) with one expm1 term
//     fewer and the operations ordered for a short dependency chain (incr): NO transcendental instruction.  (Until round 4
//     every stage was a full evaluation with v_log_f32 / v_exp_f32 / v_rcp_f32 — 12 per chain-step at half rate, a third
//     of the step's issue cycles.)  In float32 the series are good
//     to < 1e-9 relative for |rho| < 2^-9, |dlt| < 2^-6: one tier reaches further than the float64 path's three.
//   * A step whose END increments leave |rho| < 2^-10, |dlt| < 2^-7 (a factor of two inside the series' range, for the
//     stages of the step, whose increments are of the end increments' size) is not taken incrementally: from that step
//     on, to the end of the solve, the chain takes FULL evaluations at every stage (rk4_full: the hardware
//     transcendentals, the form every chain had until round 4), starting from (ms, x = hhd/Rh) at that step's start.
//     NaN increments pass the test, as in the float64 path: a dead trajectory ends in a non-finite sum of squares.
//   * No resync: in float32 ms ~ 6 Dc carries an absolute rounding error of 2^-12 .. 2^-11 per step, which a
//     re-evaluation of w from it would inject (1e-4 relative over a solve), whereas w carried by its own products drifts
//     by 6e-8 sqrt(steps).  (w, x) is a closed system; ms is carried for the switch to full evaluations alone.
//
// Two forms of the same arithmetic: V = float (one chain per lane: forward kernel, initial SSq) and V = a packed pair
// (TWO chains per lane: the sampler; each instruction a v_pk_*_f32 that advances both — 2x the work per issue slot, which
// is what float32 can give on this part, plain v_fma_f32 runs at the float64 rate).  Operation for operation identical
// (v_pk_fma_f32 is an IEEE fma per half, implicit contraction is off), so a chain's result does not depend on the form.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsf_device.h"
#include "rsf_f32_trip.inc"  // generated: the two-chain incremental trip as scheduled assembly (tools/gen_f32_trip.py)

namespace rsf {
namespace f32 {

typedef float float2v __attribute__((ext_vector_type(2)));

template <typename V>
struct Vt;
template <>
struct Vt<float> {
  static constexpr int N = 1;
  static __device__ __forceinline__ float splat(float s) { return s; }
  static __device__ __forceinline__ float get(float v, int) { return v; }
  static __device__ __forceinline__ void set(float &v, int, float s) { v = s; }
};
template <>
struct Vt<float2v> {
  static constexpr int N = 2;
  static __device__ __forceinline__ float2v splat(float s) { return float2v{s, s}; }
  static __device__ __forceinline__ float get(float2v v, int i) { return v[i]; }
  static __device__ __forceinline__ void set(float2v &v, int i, float s) { v[i] = s; }
};

template <typename V>
__device__ __forceinline__ V vfma(V a, V b, V c) { return __builtin_elementwise_fma(a, b, c); }
// the hardware transcendentals, once per chain
template <typename V>
__device__ __forceinline__ V vlog2(V x) {
  if constexpr (Vt<V>::N == 1) return __builtin_amdgcn_logf(x);
  else return V{__builtin_amdgcn_logf(x.x), __builtin_amdgcn_logf(x.y)};
}
template <typename V>
__device__ __forceinline__ V vexp2(V x) {
  if constexpr (Vt<V>::N == 1) return __builtin_amdgcn_exp2f(x);
  else return V{__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y)};
}
template <typename V>
__device__ __forceinline__ V vrcp(V x) {
  if constexpr (Vt<V>::N == 1) return __builtin_amdgcn_rcpf(x);
  else return V{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)};
}

// the guard of the incremental step (see the header): |rho| and |dlt| of a step's end increment
constexpr float kRhoMax = 0x1p-10f, kDltMax = 0x1p-7f;

// per-chain constants of a solve, float64 expressions rounded once
struct Lane32 {
  // full evaluation: the exponent pre-scaled to base 2 (v_exp_f32 is 2^x, v_log_f32 is log2)
  float kia2;    // (k'/a) log2(e)
  float tc2;     // -(mu_ref/a) log2(e)
  float boa;     // b/a
  float beta;    // (V_ref b/Dc)/k' = 10 b V_ref: b/theta dtheta/dt in units of k' (rsf_device.h, struct Lane)
  float c3;      // beta - V_ref
  float kvk;     // (k1/k') vk = k1 V_ref/a: radiation damping
  float cv;      // (h/6)/delta_t * vk: acceleration sample = cv * (weighted sum of w g over the interval's steps)
  float hhd, hd, h6d;   // (h/2, h, h/6) V_ref/Dc: step fractions of x
  // incremental step (natural exponent)
  float khh, kh, kh6;   // (k'/a)(h/2, h, h/6): d(mu)/a of a stage / of the step from the ms derivative
  float nhboa;          // -b/2a: (b/a) log1p(rho) = rho (b/a - (b/2a) rho)
  float bh;             // beta/hhd: (beta/x) d1 = bh (1 + q) d1' with d1' = Rh d1
  float w0;             // w(0) = exp((mu(0) - mu_ref)/a) (x(0) = 1), evaluated in float64
  float ms0;            // ms(0) = mu(0)/k'
  // the same for every chain
  float vref, hh, h, h6;
};

__device__ __forceinline__ Lane32 make_lane32(double dc, double a, double b, const Consts &K) {
  const double log2e = 1.4426950408889634074;
  const double inv_a = 1.0 / a, inv_dc = 1.0 / dc, kprime = (1e-2 * 10) / dc;
  const double vdc = K.V_ref * inv_dc;  // dx/dt = (V_ref/Dc) (1 - w x)
  const double beta = b * K.V_ref * (1.0 / (1e-2 * 10));
  Lane32 L;
  L.kia2 = (float)(kprime * inv_a * log2e);
  L.tc2 = (float)(-K.mu_ref * inv_a * log2e);
  L.boa = (float)(b * inv_a);
  L.beta = (float)beta;
  L.c3 = (float)(beta - K.V_ref);
  L.kvk = (float)(K.k1 * K.V_ref * inv_a);
  L.cv = (float)(K.cacc * (K.V_ref * inv_a * kprime));
  L.hhd = (float)(K.hh * vdc);
  L.hd = (float)(K.h * vdc);
  L.h6d = (float)(K.h6 * vdc);
  L.khh = (float)(kprime * inv_a * K.hh);
  L.kh = (float)(kprime * inv_a * K.h);
  L.kh6 = (float)(kprime * inv_a * K.h6);
  L.nhboa = (float)(-0.5 * (b * inv_a));
  L.bh = (float)(beta / (K.hh * vdc));
  L.w0 = (float)fm::exp((K.mu0 - K.mu_ref) * inv_a);
  L.ms0 = (float)(K.mu0 / kprime);
  L.vref = (float)K.V_ref;
  L.hh = (float)K.hh;
  L.h = (float)K.h;
  L.h6 = (float)K.h6;
  return L;
}

template <typename V>
struct LaneV {
  V kia2, tc2, boa, beta, c3, kvk, cv, hhd, hd, h6d, khh, kh, kh6, nhboa, bh, vref, hh, h, h6;
};

template <typename V>
__device__ __forceinline__ LaneV<V> make_lanev(const Lane32 (&S)[Vt<V>::N]) {
  typedef Vt<V> T;
  LaneV<V> L;
#define RSF_F32_FIELD(f)                                   \
  L.f = T::splat(S[0].f);                                  \
  if constexpr (T::N == 2) T::set(L.f, 1, S[1].f);
  RSF_F32_FIELD(kia2) RSF_F32_FIELD(tc2) RSF_F32_FIELD(boa) RSF_F32_FIELD(beta) RSF_F32_FIELD(c3) RSF_F32_FIELD(kvk)
  RSF_F32_FIELD(cv) RSF_F32_FIELD(hhd) RSF_F32_FIELD(hd) RSF_F32_FIELD(h6d) RSF_F32_FIELD(khh) RSF_F32_FIELD(kh)
  RSF_F32_FIELD(kh6) RSF_F32_FIELD(nhboa) RSF_F32_FIELD(bh)
#undef RSF_F32_FIELD
  // the same value in every lane: held in SGPRs (a packed instruction reads one scalar pair)
  auto uniform = [](float f) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, f))); };
  L.vref = T::splat(uniform(S[0].vref)); L.hh = T::splat(uniform(S[0].hh)); L.h = T::splat(uniform(S[0].h)); L.h6 = T::splat(uniform(S[0].h6));
  return L;
}

// ---------------------------------------------------------------------------------------------
// FULL evaluation step.  The RHS at (ms, x), RateStateModel.py:318-355 in the float64 path's regrouping: with
// w = v/V_ref = 2^(kia2 ms + tc2 - (b/a) log2 x) the bracket of dV/dt = vk w g is linear in w,
//     g = (V_l - beta/x) + (beta - V_ref) w,
// and the damping pass (RateStateModel.py:349-353) subtracts the same (kvk w) g from d(ms)/dt and from g.
// → d0 = d(ms)/dt, d1 = dtheta/dt and w g, the stage's dV/dt in units of vk.
// ---------------------------------------------------------------------------------------------
template <bool DAMP, typename V>
__device__ __forceinline__ V rhs_full(V ms, V x, float vl, const LaneV<V> &L, V &d0, V &d1) {
#pragma clang fp contract(off)  // every fused multiply-add of this path is written out: one-chain and two-chain forms round alike
  const V lg = vlog2(x), rx = vrcp(x);
  const V w = vexp2(vfma(-L.boa, lg, vfma(ms, L.kia2, L.tc2)));
  const V vl2 = Vt<V>::splat(vl), one = Vt<V>::splat(1.0f);
  const V t1 = vfma(-L.beta, rx, vl2);
  d0 = vfma(-L.vref, w, vl2);   // V_l - v
  d1 = vfma(-w, x, one);        // 1 - w x
  V g = vfma(L.c3, w, t1);
  if (DAMP) {
    const V kw = L.kvk * w;
    d0 = vfma(-kw, g, d0);
    g = vfma(-kw, g, g);
  }
  return w * g;
}

// one RK4 step; returns the weighted sum k1 + 2 k2 + 2 k3 + k4 of dV/dt in units of vk
template <bool DAMP, typename V>
__device__ __forceinline__ V rk4_full(V &ms, V &x, float vl0, float vlm, float vl1, const LaneV<V> &L) {
#pragma clang fp contract(off)
  V a0, a1, b0, b1, c0, c1, e0, e1;
  const V wa = rhs_full<DAMP>(ms, x, vl0, L, a0, a1);
  const V wb = rhs_full<DAMP>(vfma(L.hh, a0, ms), vfma(L.hhd, a1, x), vlm, L, b0, b1);
  const V wc = rhs_full<DAMP>(vfma(L.hh, b0, ms), vfma(L.hhd, b1, x), vlm, L, c0, c1);
  const V we = rhs_full<DAMP>(vfma(L.h, c0, ms), vfma(L.hd, c1, x), vl1, L, e0, e1);
  const V two = Vt<V>::splat(2.0f);
  ms = vfma(L.h6, vfma(two, b0 + c0, a0 + e0), ms);
  x = vfma(L.h6d, vfma(two, b1 + c1, a1 + e1), x);
  return vfma(two, wb + wc, wa + we);
}

// ---------------------------------------------------------------------------------------------
// INCREMENTAL step (rsf_device.h, rk4_tight, in float32).  Rh = hhd/x is the state; every theta derivative of the step
// is carried scaled by it, d1' = Rh (1 - w x) = Rh - w (Rh x_s) with Rh x_s = hhd + c_s d1'_prev — so rho = dx/x of a
// half-step stage IS the previous stage's d1', of the full-step stage twice it, of the step's end a third of the
// weighted sum — and (beta/x_s) d1 = bh (1 + q) d1'.
//   d0 = V_l - V_ref w,   g = d0 - brx d1'   (+ the damping pass),   the stage's dV/dt = vk w g.
// ---------------------------------------------------------------------------------------------
template <bool DAMP, typename V>
__device__ __forceinline__ void rhs_incr(V w, V xr, V Rh, V vl2, V brx, const LaneV<V> &L, V &d0, V &d1, V &g) {
#pragma clang fp contract(off)
  d1 = vfma(-w, xr, Rh);
  d0 = vfma(-L.vref, w, vl2);
  g = vfma(-brx, d1, d0);
  if (DAMP) {
    const V kw = L.kvk * w;
    d0 = vfma(-kw, g, d0);
    g = vfma(-kw, g, g);
  }
}

// (w', 1 + q = x/x') at the point reached from the step's start point (w0, x) by rho = dx/x and d(mu)/a = kd d0:
//   dlt = kd d0 - (b/a) log1p(rho),  w' = w0 exp(dlt),  1/x' = (1/x)(1 + q)
// written for a SHORT DEPENDENCY CHAIN at equal operation count — d0, the damped d(ms)/dt, is the last value of a stage to arrive:
// rho P is formed beside it, so dlt is ONE operation after d0; w' = w0 (1 + dlt) + dlt^2 (w0/2 + (w0/6) dlt) is two levels after
// dlt (Horner: three); w02 = w0/2, w06 = w0/6 once per step.  25 dependent levels per step instead of 33: the scheduler of the
// generated trip (tools/gen_f32_trip.py) then always has something to issue that does not read the instruction before it (a
// reader directly behind its producer costs a cycle, and in hipcc's own code a wait state: tools/microbench_issue.hip).
template <typename V>
__device__ __forceinline__ void incr(V rho, V kd, V d0, const LaneV<V> &L, V w0, V w02, V w06, V &w, V &q, V &dlt) {
#pragma clang fp contract(off)
  const V rP = rho * vfma(rho, L.nhboa, L.boa);
  dlt = vfma(kd, d0, -rP);
  const V d2 = dlt * dlt, A = vfma(dlt, w0, w0), B = vfma(dlt, w06, w02);
  w = vfma(d2, B, A);
  q = vfma(rho, rho, -rho);
}

// one RK4 step of the incremental form: (w, Rh, ms) → the step's end; rho_end / dlt_end = the end increment (the guard's input)
template <bool DAMP, typename V>
__device__ __forceinline__ V rk4_incr(V &w_io, V &Rh_io, V &ms_io, float vl0, float vlm, float vl1, const LaneV<V> &L, V &rho_end,
                                      V &dlt_end) {
#pragma clang fp contract(off)
  typedef Vt<V> T;
  const V w0 = w_io, Rh = Rh_io, vm = T::splat(vlm), two = T::splat(2.0f), third = T::splat(1.0f / 3.0f);
  const V w02 = w0 * T::splat(0.5f), w06 = w0 * T::splat(1.0f / 6.0f);
  V a0, a1, ga, b0, b1, gb, c0, c1, gc, e0, e1, ge, w, q, dlt;
  rhs_incr<DAMP>(w0, L.hhd, Rh, T::splat(vl0), L.bh, L, a0, a1, ga);
  V sv = w0 * ga;  // k1 + k4 of dV/dt (in units of vk), and k2 + k3 below
  incr(a1, L.khh, a0, L, w0, w02, w06, w, q, dlt);
  rhs_incr<DAMP>(w, vfma(L.hhd, a1, L.hhd), Rh, vm, vfma(L.bh, q, L.bh), L, b0, b1, gb);
  V sm = w * gb;
  incr(b1, L.khh, b0, L, w0, w02, w06, w, q, dlt);
  rhs_incr<DAMP>(w, vfma(L.hhd, b1, L.hhd), Rh, vm, vfma(L.bh, q, L.bh), L, c0, c1, gc);
  sm = vfma(w, gc, sm);
  const V T0 = vfma(two, b0 + c0, a0), T13 = vfma(two, b1 + c1, a1) * third;  // the weighted sums but for stage 4: ready before it
  incr(c1 + c1, L.kh, c0, L, w0, w02, w06, w, q, dlt);
  rhs_incr<DAMP>(w, vfma(L.hd, c1, L.hhd), Rh, T::splat(vl1), vfma(L.bh, q, L.bh), L, e0, e1, ge);
  sv = vfma(w, ge, sv);
  const V t0 = T0 + e0;
  rho_end = vfma(e1, third, T13);  // (h/6Dc)/x times the unscaled sum
  incr(rho_end, L.kh6, t0, L, w0, w02, w06, w, q, dlt_end);
  ms_io = vfma(L.h6, t0, ms_io);
  w_io = w;
  Rh_io = vfma(Rh, q, Rh);
  return vfma(two, sm, sv);
}

__device__ __forceinline__ bool step_fits(float rho, float dlt) {  // NaN fits (header)
  return !(__builtin_fabsf(rho) >= kRhoMax) && !(__builtin_fabsf(dlt) >= kDltMax);
}

// What a wave's solves ran, in steps (wave-uniform; the sampler adds them to rsf_mcmc_counters' steps_tight / steps_full /
// steps_redone: incremental trips, full-evaluation trips — a wave holding chains of both forms runs both —, replayed trips).
struct Trips32 {
  uint32_t incr = 0, full = 0, redone = 0;
};

// State of the chains of a lane.  full[c]: chain c takes full evaluations — (ms, x) is its state, (w, Rh) stale; otherwise
// (w, Rh, ms) is, x stale.
template <typename V>
struct State32 {
  V w, Rh, ms, x;
  bool full[Vt<V>::N];
};

// One step of every chain of the lane by the chain's own rule (header): the definition the straight-line trips below are a
// fast path of, and the step of the loops that are not unrolled (substeps > 1, the tail of a chunk, the replay of a trip in
// which a chain left the incremental form).
template <bool DAMP, typename V>
__device__ __forceinline__ V step_any(State32<V> &s, float vl0, float vlm, float vl1, const LaneV<V> &L) {
#pragma clang fp contract(off)
  typedef Vt<V> T;
  V dv = T::splat(0.0f);
  bool some_incr = false;
#pragma unroll
  for (int c = 0; c < T::N; ++c) some_incr |= !s.full[c];
  if (some_incr) {
    V w1 = s.w, Rh1 = s.Rh, ms1 = s.ms, rho, dlt;
    const V dvi = rk4_incr<DAMP>(w1, Rh1, ms1, vl0, vlm, vl1, L, rho, dlt);
#pragma unroll
    for (int c = 0; c < T::N; ++c)
      if (!s.full[c]) {
        if (step_fits(T::get(rho, c), T::get(dlt, c))) {
          T::set(s.w, c, T::get(w1, c)); T::set(s.Rh, c, T::get(Rh1, c)); T::set(s.ms, c, T::get(ms1, c));
          T::set(dv, c, T::get(dvi, c));
        } else {  // this step and every later one by full evaluations, from (ms, x = hhd/Rh) at this step's start; one rounding
          T::set(s.x, c, (float)((double)T::get(L.hhd, c) / (double)T::get(s.Rh, c)));
          s.full[c] = true;
        }
      }
  }
  bool some_full = false;
#pragma unroll
  for (int c = 0; c < T::N; ++c) some_full |= s.full[c];
  if (some_full) {
    V msf = s.ms, xf = s.x;
    const V dvf = rk4_full<DAMP>(msf, xf, vl0, vlm, vl1, L);
#pragma unroll
    for (int c = 0; c < T::N; ++c)
      if (s.full[c]) {
        T::set(s.ms, c, T::get(msf, c)); T::set(s.x, c, T::get(xf, c));
        T::set(dv, c, T::get(dvf, c));
      }
  }
  return dv;
}

// byte address in LDS of a pointer into the workgroup's shared array (for the ds_read instructions of the assembly trip)
__device__ __forceinline__ unsigned lds_addr(const float *p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) float *)p;
}

// Output of a solve: the sums of squares of the lane's chains and (one-chain form) the trajectory.
// The sum of squares of float residuals is formed WITHOUT float64 instructions inside the solve:
//   * in GROUPS of eight samples (k = 1..8, 9..16, ...; the last group may be short) the squares are summed in float32 —
//     s = fma(r, r, s), for both chains of a lane one packed instruction;
//   * the total is carried as an unevaluated sum of two floats (hi, lo), and a group's sum is added to it by the exact
//     two-sum (Knuth): s = hi + x, bb = s - hi, e = (hi - (s - bb)) + (x - bb), lo += e, hi = s — ~48 bits;
//   * the solve returns (double)hi + (double)lo.
// (Until round 4 every residual was converted and squared in float64.  A float64 instruction among packed-float32 ones costs
// the wave ~8 cycles beyond its issue slot when they come every few dozen instructions — tools/microbench_issue.hip — and
// ~130 cycles each when they come once per trip after ~670 packed instructions: two conversions and two additions per trip
// took 12 % of the sampler's time, profiles/r04/ab_f32_flush.log.)  A group's sum carries a relative rounding error of ~1e-7,
// the total far less: inside what float32 residuals themselves carry.  The rule is by sample index, so it does not depend
// on how the series is cut into chunks and trips; oracle/rsf_oracle.c restates it operation for operation.
template <bool WANT_SSQ, bool WANT_ACC, typename V>
struct Out32 {
  V hi, lo;  // the running total
  V s32;     // the running group's sum
  double *acc_out;
  int64_t stride;
  V cv;
  __device__ __forceinline__ void start(double total, bool any) {
    typedef Vt<V> T;
    const float h = any ? (float)total : 0.0f, l = any ? (float)(total - (double)h) : 0.0f;
    hi = T::splat(h); lo = T::splat(l); s32 = T::splat(0.0f);
  }
  __device__ __forceinline__ void flush() {
#pragma clang fp contract(off)
    const V x = s32, s = hi + x, bb = s - hi, t = s - bb;
    const V e1 = hi - t, e2 = x - bb;
    lo = lo + (e1 + e2);
    hi = s;
    s32 = Vt<V>::splat(0.0f);
  }
  __device__ __forceinline__ double total(int c) const { return (double)Vt<V>::get(hi, c) + (double)Vt<V>::get(lo, c); }
  // sample k of the series (index kk of the staged chunk `ld`): dv = the interval's weighted sum of dV/dt in units of vk
  __device__ __forceinline__ void emit(int k, int kk, const float *ld, V dv, bool store) {
#pragma clang fp contract(off)
    typedef Vt<V> T;
    const V ak = dv * cv;  // RateStateModel.py:388, from the interval's velocity increment
    if (WANT_ACC) {
      if (store) acc_out[(int64_t)k * stride] = (double)T::get(ak, 0);
    }
    if (WANT_SSQ) {
      const V r = ak - T::splat(ld[kk]);
      s32 = vfma(r, r, s32);
      if ((k & 7) == 0) flush();
    }
  }
};

// NU steps of every chain of the lane, one step per output sample (samples k .. k + NU - 1, chunk indices kk ..), as
// straight-line code; which code runs is decided per WAVE (all chains incremental / all full / both), what a chain
// computes is step_any's rule.  Every sample is emitted where it is computed — the sums of squares of the trip's start are
// kept, so that a chain's sum holds the samples of the form the chain is in, added in series order whatever ran.
template <bool DAMP, int NU, bool WANT_SSQ, bool WANT_ACC, typename V>
__device__ __forceinline__ void trip32(State32<V> &s, const float *v, const float *ld, int k, int kk, bool has_next, const LaneV<V> &L,
                                       Out32<WANT_SSQ, WANT_ACC, V> &out, Trips32 &tc) {
#pragma clang fp contract(off)
  typedef Vt<V> T;
  bool some_incr = false, some_full = false;
#pragma unroll
  for (int c = 0; c < T::N; ++c) { some_incr |= !s.full[c]; some_full |= s.full[c]; }
  const bool wave_incr = __any(some_incr), wave_full = __any(some_full);
  const V w_s = s.w, Rh_s = s.Rh, ms_s = s.ms;
  const V hi_s = out.hi, lo_s = out.lo;
  bool redo = false;
  if (wave_incr) {
    tc.incr += NU;
    bool left = false;  // may a chain of this lane that was incremental have met a step that does not fit?
    if constexpr (T::N == 2) {
      // The sampler's form: the NU steps, the emission of their samples and the trip's guard sums as ONE statement of
      // scheduled assembly (rsf_f32_trip.inc; why the order of issue is not left to the compiler: tools/gen_f32_trip.py).
      // It works in the solve's private register file — v[RSF_F32_TRIP_COMPILER_VGPRS .. 255], which the kernel's
      // amdgpu_num_vgpr keeps the compiler out of — where solve32v parked the chains' constants.
      // The guard: sums of squares of the steps' end increments (two packed operations per step).  Below the squared
      // bounds, every step fitted (each square is at most the sum; squaring and adding round monotonically, the bounds are
      // powers of two); otherwise — a NaN sum included — the trip is replayed step by step under step_fits itself.
      static_assert(NU == RSF_F32_TRIP_STEPS && NU == 8 && WANT_SSQ && !WANT_ACC, "rsf_f32_trip.inc: eight steps = one group of Out32, sums of squares only");
      // The trip's table values are already in the private file: the trip before it read them ahead (each load placed
      // behind the last reader of the registers it overwrites), the chunk's first trip's by solve32v's preload — so no trip
      // starts by waiting for LDS.  This one reads ahead for the next (`has_next`; else its own again: stays inside the chunk).
      V g2r, g2d, s32;  // (the trip is one group of the sum of squares, Out32: its float32 sum starts from zero)
      const unsigned next_vv = lds_addr(v + (has_next ? 2 * NU : 0)), next_ob = lds_addr(ld + kk + (has_next ? NU : 0));
      if constexpr (DAMP) RSF_F32_TRIP_DAMPED(s.w, s.Rh, s.ms, s32, g2r, g2d, next_vv, next_ob);
      else RSF_F32_TRIP_UNDAMPED(s.w, s.Rh, s.ms, s32, g2r, g2d, next_vv, next_ob);
      out.s32 = s32;
      out.flush();
#pragma unroll
      for (int c = 0; c < T::N; ++c)  // (no short circuits: straight-line code)
        left |= (int)!s.full[c] & (int)!((int)(T::get(g2r, c) < kRhoMax * kRhoMax) & (int)(T::get(g2d, c) < kDltMax * kDltMax));
    } else {
      bool fits = true;
#pragma unroll
      for (int j = 0; j < NU; ++j) {
        V rho, dlt;
        const V dv = rk4_incr<DAMP>(s.w, s.Rh, s.ms, v[2 * j], v[2 * j + 1], v[2 * j + 2], L, rho, dlt);
        out.emit(k + j, kk + j, ld, dv, !s.full[0]);
        fits = fits && step_fits(T::get(rho, 0), T::get(dlt, 0));
      }
      left = !s.full[0] && !fits;
    }
    redo = __any(left);
  }
  if (redo) {  // rare: the trip again from its start, step by step (identical results for the chains whose steps all fitted)
    tc.redone += NU;
    s.w = w_s; s.Rh = Rh_s; s.ms = ms_s;
    out.hi = hi_s; out.lo = lo_s; out.s32 = T::splat(0.0f);
#pragma unroll 1
    for (int j = 0; j < NU; ++j) out.emit(k + j, kk + j, ld, step_any<DAMP>(s, v[2 * j], v[2 * j + 1], v[2 * j + 2], L), true);
  } else if (wave_full) {
    tc.full += NU;
    const V hi_i = out.hi, lo_i = out.lo;  // the incremental chains' sums are final; the full chains' start again from the trip's start
    out.hi = hi_s; out.lo = lo_s; out.s32 = T::splat(0.0f);
    V msf = ms_s, xf = s.x;  // (the incremental trip advanced ms of every chain; the full chains' ms is the trip's start value)
#pragma unroll
    for (int j = 0; j < NU; ++j) out.emit(k + j, kk + j, ld, rk4_full<DAMP>(msf, xf, v[2 * j], v[2 * j + 1], v[2 * j + 2], L), s.full[0]);
#pragma unroll
    for (int c = 0; c < T::N; ++c) {
      if (s.full[c]) { T::set(s.ms, c, T::get(msf, c)); T::set(s.x, c, T::get(xf, c)); }
      else { T::set(out.hi, c, T::get(hi_i, c)); T::set(out.lo, c, T::get(lo_i, c)); }
    }
  }
}

// LDS layout (floats): [ vl : 2*S*kc+1 ][ data : kc ]; all threads of the workgroup must call it.
__device__ __forceinline__ int lds_data_offset32(const Consts &K) { return 2 * K.S * K.kc + 1; }

__device__ __forceinline__ void stage_chunk32(float *lds, const Consts &K, int k0, int kn) {
  const int nv = 2 * K.S * kn + 1;
  const int base = 2 * K.S * (k0 - 1);
  __syncthreads();
  for (int i = threadIdx.x; i < nv; i += blockDim.x) lds[i] = (float)K.vl[base + i];
  if (K.data) {
    float *ld = lds + lds_data_offset32(K);
    for (int i = threadIdx.x; i < kn; i += blockDim.x) ld[i] = (float)K.data[k0 + i];
  }
  __syncthreads();
}

// Forward solve of the lane's chains (which share the observation series: they belong to the workgroup's group).
// active[c] = false: the slot integrates a harmless default point and its results are not used.
// S = 1 (the BASELINE configs): eight steps per trip, the eight observations read ahead of the arithmetic, so that loop
// control, LDS addressing and the LDS latency are paid once per eight steps.
template <bool DAMP, bool WANT_SSQ, bool WANT_ACC, typename V>
__device__ __forceinline__ void solve32v(float *lds, const Consts &K, bool resident, const bool (&active)[Vt<V>::N],
                                         const double (&dc)[Vt<V>::N], const double (&a)[Vt<V>::N], const double (&b)[Vt<V>::N],
                                         double (&ssq)[Vt<V>::N], double *acc_out, int64_t stride, Trips32 &tc) {
#pragma clang fp contract(off)
  typedef Vt<V> T;
  constexpr int N = T::N;
  static_assert(!WANT_ACC || N == 1, "trajectories are written by the one-chain form");
  Lane32 S[N];
  bool any = false;
#pragma unroll
  for (int c = 0; c < N; ++c) {
    S[c] = make_lane32(active[c] ? dc[c] : 1000.0, active[c] ? a[c] : K.a_def, active[c] ? b[c] : K.b_def, K);
    any |= active[c];
  }
  const LaneV<V> L = make_lanev<V>(S);
  if constexpr (N == 2) RSF_F32_TRIP_SETUP(L);  // the constants of the assembly trip → the private register file (trip32)
  State32<V> st;
  st.x = T::splat(1.0f);  // x = V_ref theta(0)/Dc = 1 with theta(0) = Dc/V_ref
  st.Rh = L.hhd;
#pragma unroll
  for (int c = 0; c < N; ++c) {
    T::set(st.w, c, S[c].w0);
    T::set(st.ms, c, S[c].ms0);
    st.full[c] = false;
  }
  Trips32 lt;  // this solve's trips, as the lanes that take part see them
  Out32<WANT_SSQ, WANT_ACC, V> out;
  out.acc_out = acc_out; out.stride = stride; out.cv = L.cv;
  double total0 = 0.0;
  if (WANT_SSQ && any) {
    const double d0 = (double)(float)K.data[0];
    total0 = d0 * d0;
  }
  out.start(total0, any);
  if (WANT_ACC && any) acc_out[0] = 0.0;
  for (int k0 = 1; k0 < K.nout; k0 += K.kc) {
    const int kn = min(K.kc, K.nout - k0);
    if (!resident) stage_chunk32(lds, K, k0, kn);
    if (!any) continue;
    const float *ld = lds + lds_data_offset32(K);
    int kk = 0;
    if (K.S == 1 && (k0 & 7) == 1) {  // (a trip is one group of the sum of squares: chunks begin on a group boundary, rsf_set_model)
      constexpr int NU = 8;
      if constexpr (N == 2) {
        if (kn >= NU) RSF_F32_TRIP_PRELOAD(lds_addr(lds), lds_addr(ld));  // the chunk's first trip's table values (trip32)
      }
      while (kk + NU <= kn) {
        if constexpr (N == 2) {
          // All of the chunk's trips but the last in ONE statement (rsf_f32_trip.inc, RSF_F32_TRIPS_LOOP_*), while no chain of the
          // wave takes full evaluations: what compiled code does between two trips — guard test, the group's sum into the
          // total, addresses, loop control — is ~25 instructions of the statement's own instead of ~60.  It comes back early,
          // with the state of the failing trip's START, when a guard sum is not below its bound; that trip then goes through
          // trip32 like any other (which replays it step by step) — after its tables, overwritten by the read-ahead, are read again.
          int n = (kn - kk) / NU - 1;
          bool some_full = false;
#pragma unroll
          for (int c = 0; c < N; ++c) some_full |= st.full[c];
          if (n > 0 && !__any(some_full)) {
            const int n0 = n;
            unsigned vv_next = lds_addr(lds + 2 * (kk + NU)), ob_next = lds_addr(ld + kk + NU);
            unsigned long long m0, m1;
            const float t2r = kRhoMax * kRhoMax, t2d = kDltMax * kDltMax;
            if constexpr (DAMP) RSF_F32_TRIPS_LOOP_DAMPED(st.w, st.Rh, st.ms, out.hi, out.lo, vv_next, ob_next, n, m0, m1, t2r, t2d);
            else RSF_F32_TRIPS_LOOP_UNDAMPED(st.w, st.Rh, st.ms, out.hi, out.lo, vv_next, ob_next, n, m0, m1, t2r, t2d);
            kk += NU * (n0 - n);
            lt.incr += NU * (n0 - n);
            if (n > 0) RSF_F32_TRIP_PRELOAD(lds_addr(lds + 2 * kk), lds_addr(ld + kk));
          }
        }
        trip32<DAMP, NU>(st, lds + 2 * kk, ld, k0 + kk, kk, kk + 2 * NU <= kn, L, out, lt);
        kk += NU;
      }
    }
    int j = 2 * K.S * kk;
    for (; kk < kn; ++kk) {
      V dv = T::splat(0.0f);
      for (int sub = 0; sub < K.S; ++sub, j += 2) dv += step_any<DAMP>(st, lds[j], lds[j + 1], lds[j + 2], L);
      out.emit(k0 + kk, kk, ld, dv, true);
    }
  }
  if (WANT_SSQ) out.flush();  // the last, short group
#pragma unroll
  for (int c = 0; c < N; ++c) ssq[c] = out.total(c);
  const unsigned long long who = __builtin_amdgcn_ballot_w64(any);  // (lanes without an active chain ran no trip)
  if (who != 0) {
    const int src = __builtin_ctzll(who);
    tc.incr += __builtin_amdgcn_readlane(lt.incr, src);
    tc.full += __builtin_amdgcn_readlane(lt.full, src);
    tc.redone += __builtin_amdgcn_readlane(lt.redone, src);
  }
}

// one chain per lane
template <bool DAMP, bool WANT_SSQ, bool WANT_ACC>
__device__ __forceinline__ double solve32(float *lds, const Consts &K, bool resident, bool active, double dc, double a,
                                          double b, double *acc_out, int64_t stride) {
  const bool act[1] = {active};
  const double dcs[1] = {dc}, as[1] = {a}, bs[1] = {b};
  double ssq[1];
  Trips32 tc;
  solve32v<DAMP, WANT_SSQ, WANT_ACC, float>(lds, K, resident, act, dcs, as, bs, ssq, acc_out, stride, tc);
  return ssq[0];
}

// two chains per lane: sums of squares only.  ONLY for a kernel compiled with amdgpu_num_vgpr(RSF_F32_TRIP_COMPILER_VGPRS):
// the assembly trip owns the registers above (trip32).
template <bool DAMP>
__device__ __forceinline__ void solve32x2(float *lds, const Consts &K, bool resident, const bool (&active)[2], const double (&dc)[2],
                                          const double (&a)[2], const double (&b)[2], double (&ssq)[2], Trips32 &tc) {
  solve32v<DAMP, true, false, float2v>(lds, K, resident, active, dc, a, b, ssq, nullptr, 0, tc);
}

}  // namespace f32
}  // namespace rsf
