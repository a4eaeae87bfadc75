// rsf_device.h — device-side building blocks of the gfx950 kernels (included by rsf_hip.hip only).
//
//   Philox4x32-10 counter RNG + Box-Muller normals + Marsaglia-Tsang gamma   (K3)
//   rate-and-state friction RHS and the classical RK4 step, per lane          (K1 core)
//     hot path: 8-16 steps per loop trip, transcendentals carried incrementally (rk4_tight / integrate_multi; the
//               remainder in pairs, integrate_pairs)
//     cold path: full log/exp evaluation (rk4_cold), taken when an increment leaves the series' guard region
//   cooperative LDS staging of the chain-independent tables (loading V_l(t), observation)
//
// One lane integrates one chain; a wave64 is 64 independent chains.  The time recurrence is
// sequential, so the sum of squares is a per-lane running sum; cross-lane work is confined to
// the statistics counters (wave shuffle reduction) and LDS broadcast reads of the tables.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsf_math.h"

namespace rsf {

// ---------------------------------------------------------------------------------------------
// Kernel-argument blocks (wave-uniform → live in SGPRs)
// ---------------------------------------------------------------------------------------------
struct Consts {
  double mu_ref, V_ref, k1, mu0;  // RateStateModel.py:167-180
  double a_def, b_def;            // model.a / model.b where no per-lane value is given
  double h, hh, h6;               // RK4 step, h/2, h/6
  double inv_dt;                  // 1 / delta_t
  double t0, dt;                  // t_start, delta_t
  double inv_vref;                // 1/V_ref
  double cacc;                    // (h/6)/delta_t: acceleration sample = cacc * (k1 + 2 k2 + 2 k3 + k4)_V summed over the interval
  const double *vl;               // V_l at stage times t_start + j*h/2, j = 0 .. 2*S*(nout-1)
  const double *data;             // observation [nout] (of this workgroup's chain group) or nullptr
  int64_t group_chains;           // chains per observation group (0: one series for all)
  int32_t nout;                   // output samples (RateStateModel.py:358)
  int32_t S;                      // RK4 steps per output interval
  int32_t kc;                     // output intervals per LDS chunk
  int32_t nchunks;                // ceil((nout-1)/kc); 1 => tables stay resident in LDS
};

// select the observation series of this workgroup (all its chains belong to one group)
__device__ __forceinline__ void select_group(Consts &K) {
  if (K.group_chains > 0) K.data += ((int64_t)blockIdx.x * blockDim.x / K.group_chains) * K.nout;
}

// Per-lane proposal constants, hoisted out of the time loop.  The integrator works on the rescaled state
//   ms = mu / k'      (k' = 1e-2*10/Dc, RateStateModel.py:324)      x = V_ref theta / Dc
// — a linear change of variables, so RK4 produces the same trajectory to rounding — because two
// multiplications then drop out of every RHS evaluation: d(ms)/dt = V_l - v and d(theta)/dt = 1 - w x
// (w = v/V_ref, so w x = v theta/Dc, and log(V_ref theta/Dc) = log x).
struct Lane {
  double inv_dc;  // 1/Dc
  double vdc;     // V_ref/Dc      : dx/dt = vdc * d1
  double kprime;  // k'
  double kia;     // k'/a          : d(mu)/a = kia * d(ms)
  double khh, kh, kh6;  // kia*(h/2), kia*h, kia*(h/6): d(mu)/a of a stage / step straight from the ms derivative
  // dV/dt = (v/a)(dmu/dt - b/theta dtheta/dt) = vk w g with g = d0 - beta d1/x (d0 = V_l - v = d(ms)/dt): the integrator works on
  // g, i.e. on dV/dt in units of vk per unit w — vk multiplies only the finished sample (cv), not every stage
  double vk;      // (V_ref/a) k'
  double beta;    // (b/theta dtheta/dt)/(k' d(ms)/dt) coefficient: (V_ref b/Dc)/k' = 10 b V_ref (independent of Dc)
  double kvk;     // (k1/k') vk = k1 V_ref/a : radiation damping, d0 -= kvk w g
  double cv;      // cacc vk       : acceleration sample = cv * (g-weighted RK4 sum of the step)
  double h6v;     // (h/6) vk      : velocity increment of a step = h6v * that sum
  double boa;     // b/a
  double nhboa;   // -b/2a : (b/a) log1p(rho) = rho (b/a - (b/2a) rho) in the TIGHT tier, two fmas with dlt
  double tc;      // -mu_ref/a
  double hhd, hd, h6d;  // (h/2)/Dc, h/Dc, (h/6)/Dc : theta increments in x units
  double inv_hhd, bh;   // 1/hhd and beta/hhd: the TIGHT tier carries hhd/x in place of 1/x (rk4_tight)
  double c_l1p, c_l1q;  // series coefficients of the active tier kept in VGPRs (a VOP3 takes one SGPR source and the first Horner
  double c_em1;         //   term has two non-inline constants), see set_tier: b/3a and -b/4a of log1p, expm1's leading coefficient
};

template <int T>
__device__ __forceinline__ void set_tier(Lane &L);

__device__ __forceinline__ Lane make_lane(double dc, double a, double b, const Consts &K) {
  Lane L;
  // reciprocals by rsf_math.h (<= 1 ulp from the IEEE quotient at a fifth of its instruction count)
  const double inv_a = fm::rcp(a);
  L.inv_dc = fm::rcp(dc);
  L.kprime = (1e-2 * 10) * L.inv_dc;
  L.kia = L.kprime * inv_a;
  L.khh = L.kia * K.hh;
  L.kh = L.kia * K.h;
  L.kh6 = L.kia * K.h6;
  const double via = K.V_ref * inv_a;
  L.vdc = K.V_ref * L.inv_dc;
  L.vk = via * L.kprime;
  L.beta = (b * K.V_ref) * (1.0 / (1e-2 * 10));
  L.kvk = K.k1 * via;
  L.cv = K.cacc * L.vk;
  L.h6v = K.h6 * L.vk;
  L.boa = b * inv_a;
  L.tc = -K.mu_ref * inv_a;
  L.hhd = K.hh * L.vdc;
  L.hd = K.h * L.vdc;
  L.h6d = K.h6 * L.vdc;
  L.nhboa = -0.5 * L.boa;
  L.inv_hhd = fm::rcp(L.hhd);
  L.bh = L.beta * L.inv_hhd;
  set_tier<2>(L);
  return L;
}

// ---------------------------------------------------------------------------------------------
// RHS, RateStateModel.py:318-355.  y[2] (V) never feeds back, so only (mu, theta) are inputs.
// Same quantities as the reference, regrouped so that every reciprocal is hoisted into Lane and
// the slip rate appears only as w = v/V_ref = exp((mu-mu_ref)/a - (b/a) log(V_ref*theta/Dc)).
// ---------------------------------------------------------------------------------------------

// Integration state of one lane: ms = mu/k', x = V_ref theta/Dc, V, and the transcendental parts of the RHS at
// that point:  w = v/V_ref = exp(mu/a - mu_ref/a - (b/a) log x),   rx = 1/x.
// INSIDE the TIGHT tier's loops rx holds Rh = hhd/x instead and x is NOT carried (tier_enter / tier_leave): the step-end
// update 1/x' = (1/x)(1 + q) is indifferent to a constant factor, the tier's step works on theta derivatives scaled by
// Rh and never reads x itself (rk4_tight); where x is needed — a resync, a cold trip, leaving the tier — it is hhd / Rh.
struct State {
  double ms, x, V;
  double w, rx;
};

// the RHS once w and 1/x are known.  d0 = d(ms)/dt, d1 = d(theta)/dt (so dx = d1/Dc), d2 = (dV/dt)/vk.
template <bool DAMP>
__device__ __forceinline__ void rhs_tail(double w, double rx, double x, double vl, const Lane &L,
                                         const Consts &K, double &d0, double &d1, double &d2) {
  d1 = __builtin_fma(-w, x, 1.0);                    // ageing law: 1 - v*theta/Dc
  d0 = __builtin_fma(-K.V_ref, w, vl);               // spring loading / k':  V_l - v
  const double bt = (L.beta * d1) * rx;              // b/theta * dtheta/dt, in units of k'
  double g = d0 - bt;                                // dV/dt = vk w g:  v/a (dmu/dt - b/theta dtheta/dt), RateStateModel.py:346
  if (DAMP) {                                        // one fixed-point pass, RateStateModel.py:349-353: d0 -= k1/k' * dV/dt,
    d0 = __builtin_fma(-(L.kvk * w), g, d0);         //   then dV/dt again
    g = d0 - bt;
  }
  d2 = w * g;
}

// (w, 1/x) by full evaluation
__device__ __forceinline__ void eval_full(double ms, double x, const Lane &L, const Consts &K, double &w, double &rx) {
  w = fm::exp(__builtin_fma(-L.boa, fm::log(x), __builtin_fma(ms, L.kia, L.tc)));
  rx = fm::rcp(x);
}

enum Tier : int { TIGHT = 0, NARROW = 1, WIDE = 2, FULL = 3 };  // FULL: a full evaluation at every stage (struct Wave)

// the TIGHT and NARROW tiers' representation of 1/x (struct State)
template <int T>
__device__ __forceinline__ void tier_enter(State &s, const Lane &L) { s.rx *= L.hhd; }
template <int T>
__device__ __forceinline__ void tier_x(State &s, const Lane &L) { s.x = L.hhd * fm::rcp(s.rx); }  // x from Rh
template <int T>
__device__ __forceinline__ void tier_leave(State &s, const Lane &L) {
  tier_x<T>(s, L);
  s.rx *= L.inv_hhd;
}
template <int T>
__device__ __forceinline__ void eval_full_t(State &s, const Lane &L, const Consts &K) {  // from (ms, x)
  eval_full(s.ms, s.x, L, K, s.w, s.rx);
  tier_enter<T>(s, L);
}
template <int T>
__device__ __forceinline__ void resync_t(State &s, const Lane &L, const Consts &K) {  // inside tier T's loop
  tier_x<T>(s, L);
  eval_full_t<T>(s, L, K);
}

// (w', 1/x') at (ms + dms, x1 = x + dx) from (w, rx) at (ms, x).  With rho = dx/x (= dtheta/theta) and
// dlt = dk - (b/a) log1p(rho), dk = dmu/a = kia*dms:   w' = w exp(dlt),   1/x' = (1/x)/(1 + rho),
// by short series — the same function of (ms', x') to rounding inside the tier's guard region:
//   tier     |rho| <   |dlt| <   log1p to     expm1 to       1/x' = (1/x)(1 + q), q =                 truncation
//   TIGHT    2^-20     2^-9      rho^2/2      dlt^4/24       -rho + rho^2                             rho^3 < 2^-60; expm1: < 2.4e-16 at the
//            (its two half-step stages: |dlt| < 2^-10, truncation < 8e-18 relative)                     guard's edge (1 ulp), 8e-18 at |dlt| = 1e-3
//   NARROW   2^-14     2^-7      rho^3/3      dlt^6/720      -rho + rho^2 - rho^3                     rho^4 < 2^-56; expm1: < 4.5e-17
//   WIDE     2^-11     2^-5      rho^4/4      dlt^8/40320    -rho + rho^2 - rho^3 + rho^4             rho^5 < 2^-55; expm1: < 2.4e-18
// (guard_ok holds exactly these bounds; no tier takes a Newton step.)  Beyond WIDE: the FULL tier, log / exp / reciprocal at
// every stage (struct Wave).
// Guard tracks the largest |rho| / |dlt| seen since it was last reset — through the HIGH WORD of each double read as
// a float: for |x| < 2^1017 that reading is finite and monotone in |x|, the thresholds are powers of two (exact in the
// high word), and two values fold into one v_max3_f32 with |.| as source modifiers (4 instructions per step instead
// of 8 v_max_f64).  |x| >= 2^1017, Inf and NaN read as float NaNs, which max ignores: NaN passes through as it always
// did (a dead trajectory stays on the fast path and ends in a non-finite SSq), and an increment that large can only
// come out of a state that is already beyond 1e150 — whose sum of squares rejects the proposal whichever path
// integrates it.  (Testing w for Inf at the end of the pair instead put five dependent instructions between the last
// result and the loop branch: +3 % at cfg1.)

struct Guard {
  float rho, dlt;
  float dlt_h;  // TIGHT: |dlt| of the two half-step stages, held to 2^-10 (truncation of their expm1 series < 8e-18)
};

__device__ __forceinline__ float hi_as_float(double x) { return __builtin_bit_cast(float, __double2hiint(x)); }
constexpr float hi_pow2(int e) { return __builtin_bit_cast(float, (1023 + e) << 20); }  // high word of 2^e, as float


// leading series coefficients of a tier (kept in VGPRs, see Lane)
template <int T>
__device__ __forceinline__ void set_tier(Lane &L) {
  L.c_l1p = L.boa * (1.0 / 3.0);                                             // (TIGHT needs neither: Lane::nhboa)
  L.c_l1q = L.boa * (-1.0 / 4.0);                                            // (WIDE only)
  L.c_em1 = T == TIGHT ? 1.0 / 24.0 : (T == NARROW ? 1.0 / 720.0 : 1.0 / 40320.0);
  asm volatile("" : "+v"(L.c_l1p), "+v"(L.c_l1q), "+v"(L.c_em1));  // opaque: stays a register value, not re-materialised per step
}

template <int T>
__device__ __forceinline__ bool guard_ok(const Guard &g) {  // NaN passes through (see Guard)
  return (g.rho < (T == WIDE ? hi_pow2(-11) : (T == NARROW ? hi_pow2(-14) : hi_pow2(-20)))) &&
         (g.dlt < (T == TIGHT ? hi_pow2(-9) : (T == NARROW ? hi_pow2(-7) : hi_pow2(-5)))) && (T != TIGHT || g.dlt_h < hi_pow2(-10));
}

// Would the increments this trip measured also have fitted the next TIGHTER tier's guard, with a quarter to spare?  (The
// evidence a wave demotes itself on, struct Wave.)  Thresholds: 3/4 of tier T-1's bounds — 0.75 * 2^e has the high word
// of 1.5 * 2^(e-1).
constexpr float hi_pow2_34(int e) { return __builtin_bit_cast(float, ((1023 + e - 1) << 20) | 0x80000); }
template <int T>
__device__ __forceinline__ bool calm_enough(const Guard &g) {
  static_assert(T == NARROW || T == WIDE, "TIGHT has no tighter tier");
  return T == WIDE ? (g.rho < hi_pow2_34(-14) && g.dlt < hi_pow2_34(-7))
                   : (g.rho < hi_pow2_34(-20) && g.dlt < hi_pow2_34(-10));  // (TIGHT's half-step bound 2^-10 for every stage: stricter, simpler)
}

// ---------------------------------------------------------------------------------------------
// One classical RK4 step of size h; vl0/vlm/vl1 = V_l at t, t+h/2, t+h.
//
// Hot path (rk4_tight): the state carries (w, 1/th) at its own point, so stage 1 needs no
// transcendental at all; stages 2-4 and the step's end point are reached by tight_incr from the
// step's start point.  Straight-line code on purpose: with one wave per SIMD (cfg1) every
// instruction, nop and branch costs a full ~5-cycle issue slot (tools/microbench_fp64.hip).
// The largest increments of a PAIR of steps are checked once (integrate_pairs); if a lane left the tier's
// guard region (stiff small-Dc proposals) the pair is redone from the saved start point with full
// evaluations (rk4_cold).  Every kResync steps (w, 1/x) are recomputed in full so rounding in the
// incremental products cannot accumulate.
// ---------------------------------------------------------------------------------------------
constexpr int kResync = 512;  // power of two; every trip length below divides it.  (128 until round 3: 0.7 instructions per step.
                              // Four roundings per step on w make 5e-15 relative over 512 steps.)  Not at a chunk's first step: the
                              // state arrives there from a full evaluation or from the previous chunk's steps, never stale.

// Both step functions return the weighted sum of the V derivatives, k1 + 2 k2 + 2 k3 + k4, IN UNITS OF vk (rhs_tail /
// rhs_tight): the velocity increment of the step is (h/6) vk = L.h6v times it.  V itself never feeds back into the RHS, so the hot loop does not carry it: with one
// step per output sample the acceleration (V_k - V_{k-1})/delta_t (RateStateModel.py:388) IS cacc * sum.
template <bool DAMP>
__device__ __forceinline__ double rk4_cold(State &s, double vl0, double vlm, double vl1, const Lane &L,
                                           const Consts &K) {
  double k0 = 0.0, k1 = 0.0, s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma nounroll
  for (int st = 0; st < 4; ++st) {
    const double c = st == 0 ? 0.0 : (st == 3 ? K.h : K.hh);
    const double wgt = (st == 0 || st == 3) ? 1.0 : 2.0;
    const double vl = st == 0 ? vl0 : (st == 3 ? vl1 : vlm);
    const double x1 = __builtin_fma(c * L.vdc, k1, s.x);
    double w, rx, d0, d1, d2;
    eval_full(__builtin_fma(c, k0, s.ms), x1, L, K, w, rx);
    rhs_tail<DAMP>(w, rx, x1, vl, L, K, d0, d1, d2);
    s0 = __builtin_fma(wgt, d0, s0);
    s1 = __builtin_fma(wgt, d1, s1);
    s2 = __builtin_fma(wgt, d2, s2);
    k0 = d0;
    k1 = d1;
  }
  s.ms = __builtin_fma(K.h6, s0, s.ms);
  s.x = __builtin_fma(L.h6d, s1, s.x);
  return s2;
}

// The TIGHT tier's step.  s.rx holds Rh = (h/2Dc)/x (tier_enter), and every theta derivative of the step is carried
// SCALED by it, d1' = Rh (1 - w x): then rho of a half-step stage IS the previous stage's d1' (of the full-step stage
// twice it, of the step's end a third of the weighted sum), and with Rh x_0 = h/2Dc =: hhd the scaled derivative needs
// only constants and the previous d1':
//     Rh x_s = hhd + c_s d1'_prev  (c_s = hhd, hhd, hd),      d1' = Rh - w (Rh x_s),      (beta/x_s) d1 = bh (1 + q) d1',  bh = beta/hhd,
// so neither rho = R d1 (four products per step) nor Rf, R6 and beta/x_0 (three per step) are formed: 5 instructions
// fewer than the unscaled form.  x itself is not carried at all: Rh IS the state (x = hhd / Rh where a full evaluation
// needs it), updated once per step, so Rh x_0 = hhd holds by construction.
//
// The wider tiers are the same step with longer series: NARROW one more term each — log1p to rho^3/3, expm1 to dlt^6/720,
// 1/x' = (1/x)(1 - rho + rho^2 - rho^3): |rho| < 2^-14, |dlt| < 2^-7 (truncations 4.5e-18, 4.5e-17, 1.4e-17), four
// instructions more per evaluation — and WIDE one and two more again — rho^4/4, dlt^8/40320, ... + rho^4: |rho| < 2^-11,
// |dlt| < 2^-5 (7e-18, 2.4e-18, 2.8e-17).  (Until round 3 they were an unscaled form with Newton steps on 1/x: 131 and 144
// instructions per step against 102 and 118, and their a-priori bounds reached less far down in Dc.)
template <int T, bool HALF>
__device__ __forceinline__ void tight_incr(double rho, double dk, const Lane &L, double w0, double &w, double &q, Guard &g) {
  g.rho = __builtin_fmaxf(g.rho, __builtin_fabsf(hi_as_float(rho)));
  double pb = L.nhboa;  // (b/a)(1 - rho/2 [+ rho^2/3 [- rho^3/4]])
  if (T == NARROW) pb = __builtin_fma(rho, L.c_l1p, pb);
  if (T == WIDE) pb = __builtin_fma(rho, __builtin_fma(rho, L.c_l1q, L.c_l1p), pb);
  const double dlt = __builtin_fma(-rho, __builtin_fma(rho, pb, L.boa), dk);
  // TIGHT: expm1 to dlt^4/24 at every stage, half-step stages held to |dlt| < 2^-10, the others to 2^-9
  if (T == TIGHT && HALF) g.dlt_h = __builtin_fmaxf(g.dlt_h, __builtin_fabsf(hi_as_float(dlt)));
  else g.dlt = __builtin_fmaxf(g.dlt, __builtin_fabsf(hi_as_float(dlt)));
  double e = L.c_em1;
  if (T == WIDE) {
    e = fm::hfma(e, dlt, 1.0 / 5040.0);
    e = fm::hfma(e, dlt, 1.0 / 720.0);
  }
  if (T != TIGHT) {
    e = fm::hfma(e, dlt, 1.0 / 120.0);
    e = fm::hfma(e, dlt, 1.0 / 24.0);
  }
  e = fm::hfma(e, dlt, 1.0 / 6.0);
  e = __builtin_fma(e, dlt, 0.5);
  e = __builtin_fma(e, dlt, 1.0);
  w = __builtin_fma(w0 * dlt, e, w0);
  const double r1 = __builtin_fma(rho, rho, -rho);                 // -rho + rho^2
  q = r1;                                                          // 1/x' = (1/x)(1 + q)
  if (T != TIGHT) q = __builtin_fma(-rho, q, -rho);                // NARROW: -rho + rho^2 - rho^3
  if (T == WIDE) q = __builtin_fma(-rho, q, -rho);                 // WIDE: ... + rho^4
}

// d0 = V_l - V_ref w,  d1' = Rh - w xr  (xr = Rh x at the stage),  g = d0 - brx d1'  (brx = bh (1 + q)) and the damping pass.
// g is the bracket of dV/dt = vk w g, one fma on the two derivatives the stage forms anyway; working on g rather than on
// vk g makes the damping pass ONE product shared by both corrections: d0 -= (kvk w) g and g -= (kvk w) g; the caller forms
// w g (the stage's dV/dt in units of vk) inside its weighted sum.  (Rounds 1-2 wrote g linear in w so that all but one fma
// was ready before w — the end of the previous stage's dependency chain — arrived: one instruction more per stage for one
// dependent level less; measured in round 3, profiles/r03/ab_gform.log, the shorter stream wins at every shape.)
template <bool DAMP>
__device__ __forceinline__ void rhs_tight(double w, double xr, double Rh, double vl, double brx, const Lane &L, const Consts &K,
                                          double &d0, double &d1, double &g) {
  d1 = __builtin_fma(-w, xr, Rh);
  d0 = __builtin_fma(-K.V_ref, w, vl);
  g = __builtin_fma(-brx, d1, d0);
  if (DAMP) {
    const double kw = L.kvk * w;
    d0 = __builtin_fma(-kw, g, d0);
    g = __builtin_fma(-kw, g, g);
  }
}

template <bool DAMP, int T>
__device__ __forceinline__ double rk4_tight(State &s, double vl0, double vlm, double vl1, const Lane &L, const Consts &K, Guard &g) {
  double a0, a1, a2, b0, b1, b2, c0, c1, c2, e0, e1, e2, w, q;
  const double Rh = s.rx;
  rhs_tight<DAMP>(s.w, L.hhd, Rh, vl0, L.bh, L, K, a0, a1, a2);
  double sv = s.w * a2;  // k1 + k4 of dV/dt (in units of vk), and k2 + k3 below: 5 instructions for the weighted sum
  tight_incr<T, true>(a1, L.khh * a0, L, s.w, w, q, g);
  rhs_tight<DAMP>(w, __builtin_fma(L.hhd, a1, L.hhd), Rh, vlm, __builtin_fma(L.bh, q, L.bh), L, K, b0, b1, b2);
  double sm = w * b2;
  tight_incr<T, true>(b1, L.khh * b0, L, s.w, w, q, g);
  rhs_tight<DAMP>(w, __builtin_fma(L.hhd, b1, L.hhd), Rh, vlm, __builtin_fma(L.bh, q, L.bh), L, K, c0, c1, c2);
  sm = __builtin_fma(w, c2, sm);
  tight_incr<T, false>(c1 + c1, L.kh * c0, L, s.w, w, q, g);
  rhs_tight<DAMP>(w, __builtin_fma(L.hd, c1, L.hhd), Rh, vl1, __builtin_fma(L.bh, q, L.bh), L, K, e0, e1, e2);
  sv = __builtin_fma(w, e2, sv);
  const double t0 = a0 + 2.0 * b0 + 2.0 * c0 + e0;
  const double t1 = a1 + 2.0 * b1 + 2.0 * c1 + e1;
  const double rho = t1 * (1.0 / 3.0);  // (h/6Dc)/x times the unscaled sum
  tight_incr<T, false>(rho, L.kh6 * t0, L, s.w, w, q, g);
  s.ms = __builtin_fma(K.h6, t0, s.ms);
  s.w = w;
  s.rx = __builtin_fma(Rh, q, Rh);  // x' = x (1 + rho), exact to rounding for |rho| < 2^-20; like w, resynced
  return __builtin_fma(2.0, sm, sv);
}

__device__ __forceinline__ State initial_state(double dc, const Lane &L, const Consts &K) {
  State s;
  s.ms = (K.mu0 * dc) * (1.0 / (1e-2 * 10));  // mu(0)/k' = mu_t_zero/k', RateStateModel.py:367-377
  s.x = K.V_ref * ((dc * K.inv_vref) * L.inv_dc);  // V_ref theta(0)/Dc with theta(0) = Dc/V_ref: 1 to rounding
  s.V = K.V_ref;
  eval_full(s.ms, s.x, L, K, s.w, s.rx);
  return s;
}

// ---------------------------------------------------------------------------------------------
// LDS staging.  Chunk c covers output samples k0 .. k0+kn-1 (k0 = 1 + c*kc): it needs
// 2*S*kn+1 loading values and kn observations.  Layout: [ vl : 2*S*kc+1 ][ data : kc ][ data_0 ] — the last word holds the
// observation's sample 0, which belongs to no chunk (acc[0] = 0, RateStateModel.py:371: its square starts every sum of
// squares).  Until round 4 every forward solve read it from global memory and waited for it: a cache round trip per
// proposal, and — vector loads and stores retire in order on one counter — a wait for the trace row the previous iteration
// had just stored.  With it staged here the sampler's iteration loop holds no vector load at all.
// Every thread of the workgroup must call stage_chunk (it contains the barriers).
// ---------------------------------------------------------------------------------------------
constexpr int kLdsPad = 0;
__device__ __forceinline__ int lds_data_offset(const Consts &K) { return 2 * K.S * K.kc + 1 + kLdsPad; }
__device__ __forceinline__ int lds_d0_offset(const Consts &K) { return lds_data_offset(K) + K.kc; }

__device__ __forceinline__ void stage_chunk(double *lds, const Consts &K, int k0, int kn) {
  const int nv = 2 * K.S * kn + 1;
  const int base = 2 * K.S * (k0 - 1);
  __syncthreads();  // every wave is done with the previous chunk
  for (int i = threadIdx.x; i < nv; i += blockDim.x) lds[i] = K.vl[base + i];
  if (K.data) {
    double *ld = lds + lds_data_offset(K);
    for (int i = threadIdx.x; i < kn; i += blockDim.x) ld[i] = K.data[k0 + i];
    if (threadIdx.x == 0) lds[lds_d0_offset(K)] = K.data[0];
  }
  __syncthreads();
}

// Output bookkeeping of a chunk.  S1: one step per output sample (the BASELINE configs), every step of a trip emits a
// sample; otherwise a sample is emitted every K.S steps (wave-uniform phase counter).
struct Emit {            // output bookkeeping of a chunk
  int ko;                // next output sample of the chunk (0 .. kn-1)
  int phase;             // RK4 steps since the last emitted sample (S > 1 only)
  double vprev;          // V at the last emitted sample
};

// acceleration sample ko of the chunk from the velocities at its two ends (RateStateModel.py:388) + SSq term;
// `obs` is the observation at that sample (read from LDS by the caller, early, so its latency is hidden)
template <bool WANT_SSQ, bool WANT_ACC>
__device__ __forceinline__ void emit_at(double vnow, double vprev, int ko, double obs, const Consts &K, int k0,
                                        double &ssq, bool active, double *acc_out, int64_t stride) {
  const double ak = (vnow - vprev) * K.inv_dt;
  if (WANT_ACC && active) acc_out[(int64_t)(k0 + ko) * stride] = ak;
  if (WANT_SSQ) {
    const double r = ak - obs;
    ssq = __builtin_fma(r, r, ssq);
  }
}

// the same sample straight from the step's derivative sum (one step per output sample): ak = cacc vk * dvs
template <bool WANT_SSQ, bool WANT_ACC>
__device__ __forceinline__ void emit_incr(double dvs, int ko, double obs, const Lane &L, int k0, double &ssq,
                                          bool active, double *acc_out, int64_t stride) {
  if (WANT_ACC && active) acc_out[(int64_t)(k0 + ko) * stride] = dvs * L.cv;
  if (WANT_SSQ) {
    const double r = __builtin_fma(dvs, L.cv, -obs);  // one rounding; identical with and without WANT_ACC
    ssq = __builtin_fma(r, r, ssq);
  }
}

template <bool WANT_SSQ, bool WANT_ACC>
__device__ __forceinline__ void emit_sample(double vnow, Emit &em, double obs, const Consts &K, int k0, double &ssq,
                                            bool active, double *acc_out, int64_t stride) {
  emit_at<WANT_SSQ, WANT_ACC>(vnow, em.vprev, em.ko, obs, K, k0, ssq, active, acc_out, stride);
  em.vprev = vnow;
  ++em.ko;
}

// One trip: NU straight-line steps of tier T from s (L already carries the tier's coefficients, set_tier);
// dv[j] = the weighted V-derivative sum of step j.  Returns whether this lane left the tier's guard region — its
// values are then not to be used: the caller restores the state it saved and takes trip_cold.
template <bool DAMP, int T, int NU>
__device__ __forceinline__ bool trip_fast(const double *v, const Lane &L, const Consts &K, State &s, double (&dv)[NU], bool *calm = nullptr) {
  Guard g = {0, 0, 0};
#pragma unroll
  for (int j = 0; j < NU; ++j) dv[j] = rk4_tight<DAMP, T>(s, v[2 * j], v[2 * j + 1], v[2 * j + 2], L, K, g);
  if constexpr (T != TIGHT) {
    if (calm) *calm = calm_enough<T>(g);
  }
  return !guard_ok<T>(g);
}

// the same trip with a full evaluation at every stage (exact whatever the increments), (w, 1/x) re-formed at its end in
// tier T's representation
template <bool DAMP, int T, int NU>
__device__ __forceinline__ void trip_cold(const double *v, const Lane &L, const Consts &K, State &s, double (&dv)[NU]) {
  tier_x<T>(s, L);
#pragma unroll 1
  for (int j = 0; j < NU; ++j) {  // one copy of the cold step; its result goes to dv[j] by selects, so that dv stays in registers
    const double r = rk4_cold<DAMP>(s, v[2 * j], v[2 * j + 1], v[2 * j + 2], L, K);
#pragma unroll
    for (int m = 0; m < NU; ++m) dv[m] = m == j ? r : dv[m];
  }
  eval_full_t<T>(s, L, K);
}

// the same for a caller that holds the PLAIN state between trips — (ms, x) with rx = 1 / x, as the init kernels do: cold steps
// straight from (ms, x), then (w, 1/x) re-evaluated at the trip's end point
template <bool DAMP, int NU>
__device__ __forceinline__ void trip_cold_plain(const double *v, const Lane &L, const Consts &K, State &s, double (&dv)[NU]) {
#pragma unroll 1
  for (int j = 0; j < NU; ++j) {  // one copy of the cold step; its result goes to dv[j] by selects, so that dv stays in registers
    const double r = rk4_cold<DAMP>(s, v[2 * j], v[2 * j + 1], v[2 * j + 2], L, K);
#pragma unroll
    for (int m = 0; m < NU; ++m) dv[m] = m == j ? r : dv[m];
  }
  eval_full(s.ms, s.x, L, K, s.w, s.rx);
}

// ---------------------------------------------------------------------------------------------
// Wave-level control of one forward solve.  Every member is wave-uniform (ballots, step counts), i.e. lives in SGPRs.
//
// alive    lanes whose result can still matter: in bounds (the caller's activity mask) and not yet CERTAINLY REJECTED.
//          The sum of squares is a sum of non-negative terms, so its partial sums only grow, and the accept test
//          (MCMC.py:327-331) is monotone in SSq: once a lane's running sum exceeds `thr` — the caller's conservative bound
//          on the largest SSq that could still be accepted with this iteration's uniform — the proposal is rejected
//          whatever the rest of the series adds.  Such a lane stops counting: its guard trips are ignored, it no longer
//          holds its wave in a wider tier, and a wave with no lane left ends the solve.  Nothing observable changes: a
//          rejected proposal leaves the chain where it was, and its SSq is never stored.
// need[]   need[T-1] = lanes that cannot run a tier tighter than T (T = NARROW, WIDE, FULL): the a-priori bound of the lane's
//          mu increment (|V_l - v| <~ 1.2 V_ref against the tiers' |dlt| guards), escalated when the lane's guard trips.
//          The wave's tier is the widest one an ALIVE lane needs, re-decided whenever that set changes — so a stiff
//          small-Dc proposal costs its wave the wider tier only until its sum of squares has disqualified it.
// FULL     the full-evaluation tier (log / exp / reciprocal at every stage, exact whatever the increments): entered by a lane
//          whose WIDE guard trips, left again after kFullRetry steps to try WIDE once more (`stiff` lanes, whose a-priori
//          bound is beyond 2^-3, stay).  Until round 4 a tripped WIDE trip was redone cold and the next trip tried the
//          fast step again: a wave with one stiff lane paid fast + cold on every trip.
// calm     demotion on evidence.  The a-priori classes are BOUNDS (|V_l - v| <~ 1.2 V_ref), and the loading oscillation that
//          drives the increments decays like exp(-t/20): a lane that needs WIDE for its first forty steps is fine in NARROW for
//          the other four hundred.  Every trip measures its largest increments anyway (Guard); when, for kCalmSteps steps
//          in a row (two periods of the oscillation at h = 0.1), every alive lane's would also have fitted the next tighter
//          tier's guard with a quarter to spare, the wave's alive lanes move one tier tighter.  If that turns out wrong the
//          ordinary escalation takes the lane back (one cold trip) — rare, because the evidence is the increments themselves.
// ---------------------------------------------------------------------------------------------
constexpr int kFullRetry = 32;
constexpr int kCalmSteps = 16;  // steps of evidence before a wave demotes itself one tier (>= two loading periods at h = 0.1)
struct Wave {
  unsigned long long alive;
  unsigned long long need[3];
  unsigned long long stiff;
  int next_resync;   // chunk-local step at which (w, Rh) are next re-evaluated in full (kResync)
  int full_run;      // steps since the FULL tier was (re-)entered
  int calm;          // consecutive steps in which every alive lane's increments would have fitted the next tighter tier
  // statistics of this solve (rsf_mcmc_counters): wave-steps per tier, wave-steps of fast trips thrown away by a tripped
  // guard, and the sum over steps of the number of alive lanes (lane utilisation = lane_steps / (64 * all steps))
  uint32_t steps[4], redone, lane_steps;
};

__device__ __forceinline__ unsigned long long ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool lane_in(unsigned long long mask) { return (mask >> (threadIdx.x & 63)) & 1; }

// (the statistics members are the caller's to zero: they accumulate over the solves of a launch)
__device__ __forceinline__ void wave_begin(Wave &W, bool active, const Lane &L, const Consts &K) {
  const double dk = 1.2 * K.V_ref * K.h * L.kia;
  W.alive = ballot(active);
  W.need[0] = ballot(active && !(dk < 0x1.0p-9));
  W.need[1] = ballot(active && !(dk < 0x1.0p-7));
  W.need[2] = W.stiff = ballot(active && !(dk < 0x1.0p-3));
  W.next_resync = kResync;
  W.full_run = 0;
  W.calm = 0;
}

__device__ __forceinline__ int wave_tier(const Wave &W) {
  return (W.need[2] & W.alive) ? FULL : ((W.need[1] & W.alive) ? WIDE : ((W.need[0] & W.alive) ? NARROW : TIGHT));
}

// samples completed by a trip of NU steps starting at chunk step r (dv[j]: the steps' V-derivative sums).  S1: every step
// is a sample, its observation obs[j] read ahead by the caller; else one every K.S steps (wave-uniform phase counter), V
// carried, the observation read from LDS where the sample completes (at most one per K.S >= 2 steps: selecting among
// observations read ahead by a run-time count costs a scratch array once that count is known to be scalar).
template <bool WANT_SSQ, bool WANT_ACC, bool S1, int NU>
__device__ __forceinline__ void emit_trip(const double (&dv)[NU], const double (&obs)[NU], const double *ld, int r, State &s, Emit &em,
                                          const Lane &L, const Consts &K, int k0, double &ssq, bool active, double *acc_out, int64_t stride) {
  if (S1) {
#pragma unroll
    for (int j = 0; j < NU; ++j) emit_incr<WANT_SSQ, WANT_ACC>(dv[j], r + j, obs[j], L, k0, ssq, active, acc_out, stride);
  } else {
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      s.V = __builtin_fma(L.h6v, dv[j], s.V);
      if (++em.phase == K.S) {
        em.phase = 0;
        const double o = WANT_SSQ ? ld[em.ko] : 0.0;
        emit_sample<WANT_SSQ, WANT_ACC>(s.V, em, o, K, k0, ssq, active, acc_out, stride);
      }
    }
  }
}

// NU steps per trip through the loop (straight-line code): the resync test, the guard compare and the branch that waits
// for it, and the LDS addressing are paid per TRIP — and a lone wave sits out the whole latency of that
// compare-and-branch, ~150 cycles, whatever the trip holds (2 -> 4 -> 8 steps per trip: +12 %, +14 % at cfg1).
// Integrates trips of tier T from chunk step r while the wave's tier stays T; returns the first step not integrated.
// It comes back early — for integrate_tiers to re-decide the tier — when an alive lane's guard tripped (that trip is redone
// with full evaluations for the lane, which from then on needs the next wider tier), when no alive lane needs a tier this
// wide any more, or when no lane is alive.  All of that is ONE scalar branch at the end of the trip; the early-rejection
// compare uses the sum of squares as the PREVIOUS trip left it, so the branch never waits for the trip's last result.
constexpr int kTightUnroll = 8;  // steps per trip of the TIGHT loop (the one-parameter sampler runs 2 * kTightUnroll, rsf_kernels.h)
constexpr int kNarrowUnroll = 8, kWiderUnroll = 4;  // NARROW; WIDE
template <bool DAMP, bool WANT_SSQ, bool WANT_ACC, int T, bool S1, int NU>
__device__ __forceinline__ int integrate_multi(const double *lds, const double *ld, const Consts &K, Lane L, int k0, int kn, int r,
                                               int nsteps, State &s, Emit &em, double &ssq, double thr, bool active, double *acc_out,
                                               int64_t stride, Wave &W) {
  static_assert(NU >= 1 && (kResync % NU) == 0, "trip lengths divide the resync interval");
  set_tier<T>(L);
  for (; r + NU <= nsteps; r += NU) {
    const unsigned long long deadmask = WANT_SSQ ? ballot(ssq > thr) : 0ull;  // NaN compares false: such a lane runs on
    const double *v = lds + 2 * r;
    // one step per sample: the observations this trip completes, read before the arithmetic (a lone wave has nothing
    // else to hide the LDS latency behind)
    double obs[NU], dv[NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) obs[j] = (WANT_SSQ && S1) ? ld[r + j] : 0.0;
    if (r >= W.next_resync) {
      resync_t<T>(s, L, K);
      W.next_resync = (r & ~(kResync - 1)) + kResync;
    }
    const State save = s;
    bool calm = false;
    const bool bad = trip_fast<DAMP, T, NU>(v, L, K, s, dv, &calm);
    const unsigned long long badmask = ballot(bad) & W.alive;  // wave-uniform, straight from the compares
    W.steps[T] += NU;
    W.lane_steps += NU * (uint32_t)__builtin_popcountll(W.alive);
    if constexpr (T != TIGHT) {  // evidence for one tier tighter: every alive lane calm in this trip (struct Wave)
      W.calm = (ballot(!calm) & W.alive) == 0 ? W.calm + NU : 0;
    }
    if (__builtin_expect(badmask != 0, 0)) {  // scalar branch: the hot path carries no exec-mask bookkeeping
      if (lane_in(badmask)) {
        s = save;
        trip_cold<DAMP, T, NU>(v, L, K, s, dv);
      }
      W.need[T] |= badmask;
      W.redone += NU;
      W.full_run = 0;
      W.calm = 0;
    }
    emit_trip<WANT_SSQ, WANT_ACC, S1, NU>(dv, obs, ld, r, s, em, L, K, k0, ssq, active, acc_out, stride);
    W.alive &= ~deadmask;
    if constexpr (T != TIGHT) {
      if (W.calm >= kCalmSteps && badmask == 0) {  // demote: constant index (a run-time one would put the Wave in scratch memory)
        W.need[T - 1] &= ~W.alive | W.stiff;
        W.calm = 0;
      }
    }
    if (badmask != 0 || W.alive == 0 || (T != TIGHT && (W.need[T - 1] & W.alive) == 0)) return r + NU;
  }
  return r;
}

// The FULL tier: one step per trip with a full evaluation at every stage (rk4_cold), on the plain state (ms, x) — w and
// 1/x are not carried.  Runs while an alive lane needs it; every kFullRetry steps the lanes that came here from a tripped
// WIDE guard are sent back to try WIDE again.
template <bool DAMP, bool WANT_SSQ, bool WANT_ACC, bool S1>
__device__ __forceinline__ int integrate_full(const double *lds, const double *ld, const Consts &K, const Lane &L, int k0, int kn, int r,
                                              int nsteps, State &s, Emit &em, double &ssq, double thr, bool active, double *acc_out,
                                              int64_t stride, Wave &W) {
#pragma unroll 1
  for (; r < nsteps; ++r) {
    const unsigned long long deadmask = WANT_SSQ ? ballot(ssq > thr) : 0ull;
    const double *v = lds + 2 * r;
    const double obs[1] = {(WANT_SSQ && S1) ? ld[r] : 0.0};
    const double dv[1] = {rk4_cold<DAMP>(s, v[0], v[1], v[2], L, K)};
    W.steps[FULL] += 1;
    W.lane_steps += (uint32_t)__builtin_popcountll(W.alive);
    emit_trip<WANT_SSQ, WANT_ACC, S1, 1>(dv, obs, ld, r, s, em, L, K, k0, ssq, active, acc_out, stride);
    W.alive &= ~deadmask;
    if (++W.full_run >= kFullRetry) {
      W.need[2] &= W.stiff;
      W.full_run = 0;
    }
    if (W.alive == 0 || (W.need[2] & W.alive) == 0) return r + 1;
  }
  return r;
}

// Wave-uniform choice of the starting tier of the init kernel's lockstep trajectories (rsf_kernels.h) — a speed decision
// only: every tier is exact to rounding inside its guard.  TIGHT and NARROW are tried whenever the mu increment allows it
// (|V_l - v| <~ 1.2 V_ref against their |dlt| guards 2^-9 / 2^-7): theta tracks its steady state closely (|dtheta/theta| ~ 1e-7
// per stage at Dc ~ 1000, h = 0.1), which no a-priori bound captures.  (The solve below decides the same per lane: wave_begin.)
__device__ __forceinline__ int start_tier(const Lane &L, const Consts &K) {
  const double dk = 1.2 * K.V_ref * K.h * L.kia;
  return !__any(!(dk < 0x1.0p-9)) ? TIGHT : (!__any(!(dk < 0x1.0p-7)) ? NARROW : WIDE);
}

// trips of tier T from chunk step r (at least two steps are left): long trips while they fit, then pairs
template <bool DAMP, bool WANT_SSQ, bool WANT_ACC, int T, bool S1, int NU>
__device__ __forceinline__ int integrate_tier(const double *lds, const double *ld, const Consts &K, const Lane &L, int k0, int kn, int r,
                                              int nsteps, State &s, Emit &em, double &ssq, double thr, bool active, double *acc_out,
                                              int64_t stride, Wave &W) {
  if (r + NU <= nsteps) return integrate_multi<DAMP, WANT_SSQ, WANT_ACC, T, S1, NU>(lds, ld, K, L, k0, kn, r, nsteps, s, em, ssq, thr, active, acc_out, stride, W);
  return integrate_multi<DAMP, WANT_SSQ, WANT_ACC, T, S1, 2>(lds, ld, K, L, k0, kn, r, nsteps, s, em, ssq, thr, active, acc_out, stride, W);
}

template <bool DAMP, bool WANT_SSQ, bool WANT_ACC, bool S1, int NUT>
__device__ __forceinline__ void integrate_tiers(const double *lds, const double *ld, const Consts &K, const Lane &L, int k0,
                                                int kn, State &s, double &ssq, double thr, bool active, double *acc_out, int64_t stride, Wave &W) {
  const int nsteps = S1 ? kn : K.S * kn;
  Emit em = {0, 0, s.V};
  int r = 0;
  W.next_resync = kResync;  // chunk-local; not at a chunk's first step (see kResync)
  bool scaled = false;      // s.rx holds Rh = hhd / x (the incremental tiers' state) rather than 1 / x
  while (r < nsteps && W.alive != 0) {
    const int tier = wave_tier(W);
    if (tier == FULL) {
      if (scaled) { tier_leave<WIDE>(s, L); scaled = false; }
      r = integrate_full<DAMP, WANT_SSQ, WANT_ACC, S1>(lds, ld, K, L, k0, kn, r, nsteps, s, em, ssq, thr, active, acc_out, stride, W);
      eval_full(s.ms, s.x, L, K, s.w, s.rx);  // (w, 1/x) at the point the incremental tiers — or the next chunk — continue from
      continue;
    }
    if (!scaled) { tier_enter<TIGHT>(s, L); scaled = true; }
    // the chunk's odd last step (it always completes a sample): with the WIDE series whatever the tier — one instance
    if (nsteps - r == 1) r = integrate_multi<DAMP, WANT_SSQ, WANT_ACC, WIDE, S1, 1>(lds, ld, K, L, k0, kn, r, nsteps, s, em, ssq, thr, active, acc_out, stride, W);
    else if (tier == TIGHT) r = integrate_tier<DAMP, WANT_SSQ, WANT_ACC, TIGHT, S1, NUT>(lds, ld, K, L, k0, kn, r, nsteps, s, em, ssq, thr, active, acc_out, stride, W);
    else if (tier == NARROW) r = integrate_tier<DAMP, WANT_SSQ, WANT_ACC, NARROW, S1, kNarrowUnroll>(lds, ld, K, L, k0, kn, r, nsteps, s, em, ssq, thr, active, acc_out, stride, W);
    else r = integrate_tier<DAMP, WANT_SSQ, WANT_ACC, WIDE, S1, kWiderUnroll>(lds, ld, K, L, k0, kn, r, nsteps, s, em, ssq, thr, active, acc_out, stride, W);
  }
  if (scaled) tier_leave<WIDE>(s, L);
}

// Integrate kn output intervals from the staged chunk.  Accumulates the sum of squares
// (MCMC.py:387) and optionally stores acc time-major (`active` lanes only).  Called by whole waves.
template <bool DAMP, bool WANT_SSQ, bool WANT_ACC, int NUT>
__device__ __forceinline__ void integrate_chunk(const double *lds, const Consts &K, const Lane &L, int k0, int kn, State &s,
                                                double &ssq, double thr, bool active, double *acc_out, int64_t stride, Wave &W) {
  const double *ld = lds + lds_data_offset(K);
  if (K.S == 1) integrate_tiers<DAMP, WANT_SSQ, WANT_ACC, true, NUT>(lds, ld, K, L, k0, kn, s, ssq, thr, active, acc_out, stride, W);
  else integrate_tiers<DAMP, WANT_SSQ, WANT_ACC, false, NUT>(lds, ld, K, L, k0, kn, s, ssq, thr, active, acc_out, stride, W);
}

// Full forward solve for one lane.  Every thread of the workgroup must call it (chunk staging has barriers);
// `resident` (workgroup-uniform): the single chunk is already staged, nothing is re-staged.
// NUT: RK4 steps per trip of the TIGHT loop (integrate_multi); 16 where the kernel's registers allow it (one-parameter
// sampler), 8 otherwise.  thr: a sum of squares above it cannot be accepted (struct Wave; +inf where every lane's result is
// wanted) — the value returned for a lane that stopped counting is the partial sum that disqualified it.  W: the solve's
// wave-level control and statistics, for the caller's counters.
template <bool DAMP, bool WANT_SSQ, bool WANT_ACC, int NUT = kTightUnroll>
__device__ __forceinline__ double solve(double *lds, const Consts &K, bool resident, bool active, double dc, double a,
                                        double b, double thr, double *acc_out, int64_t stride, Wave &W) {
  const Lane L = make_lane(dc, a, b, K);
  State s = initial_state(dc, L, K);
  wave_begin(W, active, L, K);
  double ssq = 0.0;
  if (WANT_ACC && active) acc_out[0] = 0.0;
  for (int k0 = 1; k0 < K.nout; k0 += K.kc) {
    const int kn = min(K.kc, K.nout - k0);
    if (!resident) stage_chunk(lds, K, k0, kn);
    if (WANT_SSQ && k0 == 1) {  // sample 0 belongs to no chunk: acc[0] = 0, RateStateModel.py:371 (staged with every chunk)
      const double d0 = lds[lds_d0_offset(K)];
      ssq = d0 * d0;
    }
    // a wave-uniform branch: every lane of a wave with work goes in — the lanes that are not `active` (out of bounds, past
    // the end of the batch) ride along masked out of W.alive and of the trajectory stores.  (Under `if (active)`, a divergent
    // branch, everything the solve leaves in W would count as divergent after it and move from scalar to vector registers.)
    if (W.alive != 0) integrate_chunk<DAMP, WANT_SSQ, WANT_ACC, NUT>(lds, K, L, k0, kn, s, ssq, thr, active, acc_out, stride, W);
  }
  return ssq;
}

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. SC'11; Random123 constants) and the variates built on it.
// counter = (chain_lo, chain_hi, iteration, slot), key = seed.  Slots of one (chain, iteration):
//   0: proposal normals z0,z1   1: z2   2: accept uniform   8+2j / 9+2j: gamma attempt j
// ---------------------------------------------------------------------------------------------
enum : uint32_t { SLOT_Z01 = 0, SLOT_Z2 = 1, SLOT_U = 2, SLOT_GAMMA = 8 };

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ void draw_words(uint64_t seed, uint64_t chain, uint32_t iter, uint32_t slot,
                                           uint32_t w[4]) {
  philox4x32_10((uint32_t)chain, (uint32_t)(chain >> 32), iter, slot, (uint32_t)seed, (uint32_t)(seed >> 32), w);
}

// 53-bit uniform in (0, 1]
__device__ __forceinline__ double u53(uint32_t hi, uint32_t lo) {
  const uint64_t k = (((uint64_t)hi << 32) | lo) >> 11;
  return (double)(k + 1) * 0x1.0p-53;
}

// The variates use the kernel's own log / sincos (rsf_math.h, <= 4 ulp): the oracle's libm values differ in the last
// bits, which matters only when an accept test sits within ~1e-15 of its threshold (tests: draws to 1e-12).
__device__ __forceinline__ void normal_pair(const uint32_t w[4], double &z0, double &z1) {
  const double u1 = u53(w[0], w[1]), u2 = u53(w[2], w[3]);
  const double r = sqrt(-2.0 * fm::log(u1));
  double s, c;
  fm::sincos2pi(u2, s, c);
  z0 = r * c;
  z1 = r * s;
}

__device__ __forceinline__ double rng_log(double x) { return fm::log(x); }

// Marsaglia & Tsang (2000), shape >= 1, log acceptance test only.  d = shape - 1/3 and c = 1/sqrt(9 d) come from
// the host (chain-independent).
__device__ __forceinline__ double gamma_draw(uint64_t seed, uint64_t chain, uint32_t iter, double d, double c) {
  for (uint32_t j = 0; j < 64; ++j) {
    uint32_t w[4];
    double x, unused;
    draw_words(seed, chain, iter, SLOT_GAMMA + 2 * j, w);
    normal_pair(w, x, unused);
    double v = 1.0 + c * x;
    if (!(v > 0.0)) continue;
    v = v * v * v;
    draw_words(seed, chain, iter, SLOT_GAMMA + 2 * j + 1, w);
    const double u = u53(w[0], w[1]);
    if (rng_log(u) < 0.5 * x * x + d * (1.0 - v + rng_log(v))) return d * v;
  }
  return d;
}

// ---------------------------------------------------------------------------------------------
// small dense helpers, fully unrolled on the compile-time dimension
// ---------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ bool chol_lower(const double *V, double *L) {
  bool ok = true;
#pragma unroll
  for (int e = 0; e < D * D; ++e) L[e] = 0.0;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    double s = V[j * D + j];
#pragma unroll
    for (int k = 0; k < j; ++k) s -= L[j * D + k] * L[j * D + k];
    if (!(s > 0.0)) { ok = false; continue; }
    const double ljj = sqrt(s);
    L[j * D + j] = ljj;
#pragma unroll
    for (int i = j + 1; i < D; ++i) {
      double t = V[i * D + j];
#pragma unroll
      for (int k = 0; k < j; ++k) t -= L[i * D + k] * L[j * D + k];
      L[i * D + j] = t / ljj;
    }
  }
  return ok;
}

template <int D>
__device__ __forceinline__ void sym_inverse(const double *A, double *Ai) {
  if (D == 1) {
    Ai[0] = 1.0 / A[0];
  } else {
    const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
    const double id = 1.0 / (A[0] * c00 + A[1] * c01 + A[2] * c02);
    Ai[0] = c00 * id; Ai[1] = (A[2] * A[7] - A[1] * A[8]) * id; Ai[2] = (A[1] * A[5] - A[2] * A[4]) * id;
    Ai[3] = c01 * id; Ai[4] = (A[0] * A[8] - A[2] * A[6]) * id; Ai[5] = (A[2] * A[3] - A[0] * A[5]) * id;
    Ai[6] = c02 * id; Ai[7] = (A[1] * A[6] - A[0] * A[7]) * id; Ai[8] = (A[0] * A[4] - A[1] * A[3]) * id;
  }
}

// scale * (cov + diag(eps)) from shifted sums ws[p] = sum (q_p - ref_p), wq[p][r] = sum (q_p - ref_p)(q_r - ref_r) over nn
// samples (ddof = 1, like np.cov), and its lower Cholesky factor.  false: not positive definite (the caller keeps its
// covariance; with eps > 0 that does not happen).  This is the corrected adaptive Metropolis of `adapt_mode = am`
// (Haario, Saksman & Tamminen 2001): the sums run over the chain's WHOLE history since rsf_mcmc_init — adaptation that
// diminishes, so the chain still converges to the posterior — and eps_p = (1e-6 (hi_p - lo_p))^2 keeps the matrix positive
// definite when the history holds fewer than D + 1 distinct points.  (Until round 4 `am` used the last adapt_interval samples
// only, as the reference's broken update does: measured on the three-parameter problem, that shifts the pooled posterior —
// the mean of Dc*a by 89 standard errors, its spread by 8 % — because a proposal that forgets is a different chain, not an
// adaptive one; and a window with too few distinct points made "positive definite" a coin toss of the last bit.)
template <int D>
__device__ __forceinline__ bool window_covariance(const double *ws, const double *wq, double nn, double scale, const double *eps, double *Vn,
                                                  double *Ln) {
#pragma unroll
  for (int p = 0; p < D; ++p)
#pragma unroll
    for (int r = 0; r < D; ++r) Vn[p * D + r] = scale * ((wq[p * D + r] - ws[p] * ws[r] / nn) / (nn - 1.0) + (p == r ? eps[p] : 0.0));
  return chol_lower<D>(Vn, Ln);
}

// np.cov of ONE variable's window as NumPy forms it (numpy/lib/_function_base_impl.py::cov): the mean from np.add.reduce's
// pairwise summation — fewer than 8 elements one by one from 0; up to 128 (RSF_DICT_MAX_INTERVAL) through eight interleaved
// accumulators combined pairwise, the remainder one by one (numpy/_core/src/umath/loops_utils.h.src) — then the
// deviations' dot product times 1 / (n - 1).  The ORDER of that sum is what `reference_dict` adaptation needs: whether the
// mean of a window of identical samples is that sample (covariance exactly 0: np.linalg.cholesky raises and the reference
// keeps its proposal, MCMC.py:524-527) or one ulp off (covariance ~1e-31: the Cholesky "succeeds" and the reference's
// proposal collapses to ~1e-8) depends on the sample's low bits through exactly these additions.  a(k): sample k.
template <typename F>
__device__ __forceinline__ double np_cov_1d(F a, int n) {
  double sum;
  if (n < 8) {
    sum = 0.0;
    for (int i = 0; i < n; ++i) sum += a(i);
  } else {
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = a(j);
    int i = 8;
    for (; i < n - (n % 8); i += 8)
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] += a(i + j);
    sum = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) sum += a(i);
  }
  const double avg = sum / (double)n;
  double c = 0.0;
  for (int k = 0; k < n; ++k) {
    const double dlt = a(k) - avg;
    c = __builtin_fma(dlt, dlt, c);
  }
  return c * (1.0 / (double)(n - 1));
}

// wave64 sum via DPP-free shuffles; result valid in lane 0
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

}  // namespace rsf
