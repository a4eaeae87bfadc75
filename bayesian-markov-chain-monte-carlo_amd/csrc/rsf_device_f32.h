// rsf_device_f32.h — float32 forward solve (BASELINE config 5: "float32 vs float64 tolerance sweep").
//
// Selected per model with RSF_FLAG_FP32_SOLVE.  Only the ODE integration runs in float32; every
// interface array, the sum of squares accumulator and the whole sampler logic (proposal, accept
// test, sigma^2 update, adaptation, initial covariance) stay float64.
//
// In float32 the hardware transcendentals are one instruction each (v_log_f32 = log2, v_exp_f32 =
// 2^x, v_rcp_f32; ~1 ulp), so every RK4 stage is evaluated in full — no incremental series — and
// the constants are pre-scaled for base 2.  The acceleration sample is formed from the step's
// velocity INCREMENT, not from the difference of two velocities near V_ref, which would lose
// ~4 digits in float32.  Same rescaled state as the float64 path — ms = mu/k', x = V_ref theta/Dc — and the same
// regrouping of the RHS around w = v/V_ref with dV/dt in units of vk (rsf_device.h, rhs_tight); the two state components
// and their derivatives are carried as packed pairs (v_pk_fma_f32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsf_device.h"

namespace rsf {
namespace f32 {

// two floats in one 64-bit register pair: (d(ms)/dt, d(x)/dt) of a stage travel together, so that the stage inputs, the
// RK4 combination and the damping correction are v_pk_fma_f32 / v_pk_mul_f32 — one instruction for both components
typedef float float2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float2v pk_fma(float2v a, float2v b, float2v c) { return __builtin_elementwise_fma(a, b, c); }

struct Lane32 {
  float kia2;    // (k'/a) log2(e)
  float tc2;     // -(mu_ref/a) log2(e)
  float boa;     // b/a
  float beta;    // (V_ref b/Dc)/k' = 10 b V_ref: b/theta dtheta/dt in units of k' (rsf_device.h, struct Lane)
  float c3;      // beta - V_ref
  float kvk;     // (k1/k') vk = k1 V_ref/a: radiation damping
  float vref;
  float cv;      // (h/6)/delta_t * vk: acceleration sample = cv * (weighted sum of w g over the interval's steps)
  float2v chh, ch, ch6;  // (h/2, (h/2) V_ref/Dc), (h, h V_ref/Dc), (h/6, (h/6) V_ref/Dc): step fractions of (ms, x)
};

__device__ __forceinline__ Lane32 make_lane32(double dc, double a, double b, const Consts &K) {
  const double log2e = 1.4426950408889634074;
  const double inv_a = 1.0 / a, inv_dc = 1.0 / dc, kprime = (1e-2 * 10) / dc;
  const double vdc = K.V_ref * inv_dc;  // dx/dt = (V_ref/Dc) (1 - w x)
  Lane32 L;
  L.kia2 = (float)(kprime * inv_a * log2e);
  L.tc2 = (float)(-K.mu_ref * inv_a * log2e);
  L.boa = (float)(b * inv_a);
  L.beta = (float)(b * K.V_ref * (1.0 / (1e-2 * 10)));
  L.c3 = (float)(b * K.V_ref * (1.0 / (1e-2 * 10)) - K.V_ref);
  L.kvk = (float)(K.k1 * K.V_ref * inv_a);
  L.vref = (float)K.V_ref;
  L.cv = (float)(K.cacc * (K.V_ref * inv_a * kprime));
  L.chh = float2v{(float)K.hh, (float)(K.hh * vdc)};
  L.ch = float2v{(float)K.h, (float)(K.h * vdc)};
  L.ch6 = float2v{(float)K.h6, (float)(K.h6 * vdc)};
  return L;
}

// The RHS at (ms, x) = s, RateStateModel.py:318-355 in the float64 path's regrouping (rsf_device.h, rhs_tight): with
// w = v/V_ref = 2^(kia2 ms + tc2 - (b/a) log2 x) the bracket of dV/dt = vk w g is linear in w,
//     g = (V_l - beta/x) + (beta - V_ref) w,
// and the damping pass (RateStateModel.py:349-353) subtracts the same (kvk w) g from d(ms)/dt and from g.
// → d = (d(ms)/dt, dtheta/dt) and w g, the stage's dV/dt in units of vk.
template <bool DAMP>
__device__ __forceinline__ float rhs32(float2v s, float vl, const Lane32 &L, float2v &d) {
#pragma clang fp contract(off)  // every fused multiply-add of this path is written out: one-chain and two-chain forms round alike
  const float lg = __builtin_amdgcn_logf(s.y), rx = __builtin_amdgcn_rcpf(s.y);
  const float w = __builtin_amdgcn_exp2f(__builtin_fmaf(-L.boa, lg, __builtin_fmaf(s.x, L.kia2, L.tc2)));
  const float t1 = __builtin_fmaf(-L.beta, rx, vl);
  d.x = __builtin_fmaf(-L.vref, w, vl);   // V_l - v
  d.y = __builtin_fmaf(-w, s.y, 1.0f);    // 1 - w x   (the two halves of one register pair: no move to form the pair)
  float g = __builtin_fmaf(L.c3, w, t1);
  if (DAMP) {
    const float kw = L.kvk * w;
    d.x = __builtin_fmaf(-kw, g, d.x);
    g = __builtin_fmaf(-kw, g, g);
  }
  return w * g;
}

// one RK4 step; returns the weighted sum k1 + 2 k2 + 2 k3 + k4 of dV/dt in units of vk
template <bool DAMP>
__device__ __forceinline__ float rk4_step32(float2v &s, float vl0, float vlm, float vl1, const Lane32 &L) {
#pragma clang fp contract(off)  // every fused multiply-add of this path is written out: one-chain and two-chain forms round alike
  float2v a, b, c, e;
  const float wa = rhs32<DAMP>(s, vl0, L, a);
  const float wb = rhs32<DAMP>(pk_fma(L.chh, a, s), vlm, L, b);
  const float wc = rhs32<DAMP>(pk_fma(L.chh, b, s), vlm, L, c);
  const float we = rhs32<DAMP>(pk_fma(L.ch, c, s), vl1, L, e);
  const float2v two = {2.0f, 2.0f};
  s = pk_fma(L.ch6, pk_fma(two, b + c, a + e), s);
  return __builtin_fmaf(2.0f, wb + wc, wa + we);
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

// S1: one step per output sample (the BASELINE configs) — the sample loop is then unrolled eight-fold into straight-line
// code with the eight observations read ahead of the arithmetic, so that loop control, LDS addressing and the LDS
// latency are paid once per eight steps (the float64 path's trip structure; a wave with one or two resident peers
// cannot hide them otherwise).
template <bool DAMP, bool WANT_SSQ, bool WANT_ACC, bool S1>
__device__ __forceinline__ void integrate_chunk32(const float *lds, const Consts &K, const Lane32 &L, int k0, int kn,
                                                  float2v &st, double &ssq, double *acc_out, int64_t stride) {
#pragma clang fp contract(off)  // every fused multiply-add of this path is written out: one-chain and two-chain forms round alike
  const float *ld = lds + lds_data_offset32(K);
  auto emit = [&](int kk, float dv, float obs) {
    const float ak = dv * L.cv;  // RateStateModel.py:388, from the interval's velocity increment
    if (WANT_ACC) acc_out[(int64_t)(k0 + kk) * stride] = (double)ak;
    if (WANT_SSQ) {
      const double r = (double)(ak - obs);
      ssq = __builtin_fma(r, r, ssq);
    }
  };
  int kk = 0;
  if (S1) {
    constexpr int NU = 8;
    for (; kk + NU <= kn; kk += NU) {
      const float *v = lds + 2 * kk;
      float obs[NU], dv[NU];
#pragma unroll
      for (int j = 0; j < NU; ++j) obs[j] = WANT_SSQ ? ld[kk + j] : 0.0f;
#pragma unroll
      for (int j = 0; j < NU; ++j) dv[j] = rk4_step32<DAMP>(st, v[2 * j], v[2 * j + 1], v[2 * j + 2], L);
#pragma unroll
      for (int j = 0; j < NU; ++j) emit(kk + j, dv[j], obs[j]);
    }
  }
  int j = 2 * K.S * kk;
  for (; kk < kn; ++kk) {
    float dv = 0.0f;
    for (int sub = 0; sub < K.S; ++sub, j += 2) dv += rk4_step32<DAMP>(st, lds[j], lds[j + 1], lds[j + 2], L);
    emit(kk, dv, WANT_SSQ ? ld[kk] : 0.0f);
  }
}

template <bool DAMP, bool WANT_SSQ, bool WANT_ACC>
__device__ __forceinline__ double solve32(float *lds, const Consts &K, bool resident, bool active, double dc, double a,
                                          double b, double *acc_out, int64_t stride) {
  const Lane32 L = make_lane32(dc, a, b, K);
  float2v st = {(float)(K.mu0 / ((1e-2 * 10) / dc)), 1.0f};  // (ms, x); x = V_ref theta(0)/Dc = 1 with theta(0) = Dc/V_ref
  double ssq = 0.0;
  if (WANT_SSQ && active) {
    const double d0 = (double)(float)K.data[0];
    ssq = d0 * d0;
  }
  if (WANT_ACC && active) acc_out[0] = 0.0;
  for (int k0 = 1; k0 < K.nout; k0 += K.kc) {
    const int kn = min(K.kc, K.nout - k0);
    if (!resident) stage_chunk32(lds, K, k0, kn);
    if (active) {
      if (K.S == 1) integrate_chunk32<DAMP, WANT_SSQ, WANT_ACC, true>(lds, K, L, k0, kn, st, ssq, acc_out, stride);
      else integrate_chunk32<DAMP, WANT_SSQ, WANT_ACC, false>(lds, K, L, k0, kn, st, ssq, acc_out, stride);
    }
  }
  return ssq;
}

// ---------------------------------------------------------------------------------------------
// TWO chains per lane (the float32 sampler kernel): every quantity of the solve is a packed pair {chain A, chain B}, so
// each arithmetic instruction is a v_pk_*_f32 that advances both chains — 2x the work per issue slot, which is what
// float32 can give on this part (plain v_fma_f32 runs at the float64 rate).  Operation for operation the same
// arithmetic as the one-chain functions above (v_pk_fma_f32 is an IEEE fma per half), so a chain's result does not
// depend on which form integrated it (tested: the sampler against the forward kernel and the float32 restatement).
// The three transcendentals per stage are issued once per chain.
// ---------------------------------------------------------------------------------------------
struct Lane32x2 {
  float2v kia2, tc2, boa, beta, c3, kvk, cv, hhd, hd, h6d;  // per chain (see Lane32)
  float2v vref, hh, h, h6;                                   // the same value in both halves
};

__device__ __forceinline__ Lane32x2 make_lane32x2(const double dc[2], const double a[2], const double b[2], const Consts &K) {
  const Lane32 A = make_lane32(dc[0], a[0], b[0], K), B = make_lane32(dc[1], a[1], b[1], K);
  Lane32x2 L;
  L.kia2 = float2v{A.kia2, B.kia2}; L.tc2 = float2v{A.tc2, B.tc2}; L.boa = float2v{A.boa, B.boa};
  L.beta = float2v{A.beta, B.beta}; L.c3 = float2v{A.c3, B.c3}; L.kvk = float2v{A.kvk, B.kvk}; L.cv = float2v{A.cv, B.cv};
  L.hhd = float2v{A.chh.y, B.chh.y}; L.hd = float2v{A.ch.y, B.ch.y}; L.h6d = float2v{A.ch6.y, B.ch6.y};
  L.vref = float2v{A.vref, A.vref}; L.hh = float2v{A.chh.x, A.chh.x}; L.h = float2v{A.ch.x, A.ch.x}; L.h6 = float2v{A.ch6.x, A.ch6.x};
  return L;
}

template <bool DAMP>
__device__ __forceinline__ float2v rhs32x2(float2v ms, float2v x, float vl, const Lane32x2 &L, float2v &d0, float2v &d1) {
#pragma clang fp contract(off)  // every fused multiply-add of this path is written out: one-chain and two-chain forms round alike
  const float2v lg = {__builtin_amdgcn_logf(x.x), __builtin_amdgcn_logf(x.y)};
  const float2v rx = {__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)};
  const float2v arg = pk_fma(-L.boa, lg, pk_fma(ms, L.kia2, L.tc2));
  const float2v w = {__builtin_amdgcn_exp2f(arg.x), __builtin_amdgcn_exp2f(arg.y)};
  const float2v vl2 = {vl, vl}, one = {1.0f, 1.0f};
  const float2v t1 = pk_fma(-L.beta, rx, vl2);
  d0 = pk_fma(-L.vref, w, vl2);
  d1 = pk_fma(-w, x, one);
  float2v g = pk_fma(L.c3, w, t1);
  if (DAMP) {
    const float2v kw = L.kvk * w;
    d0 = pk_fma(-kw, g, d0);
    g = pk_fma(-kw, g, g);
  }
  return w * g;
}

template <bool DAMP>
__device__ __forceinline__ float2v rk4_step32x2(float2v &ms, float2v &x, float vl0, float vlm, float vl1, const Lane32x2 &L) {
#pragma clang fp contract(off)  // every fused multiply-add of this path is written out: one-chain and two-chain forms round alike
  float2v a0, a1, b0, b1, c0, c1, e0, e1;
  const float2v wa = rhs32x2<DAMP>(ms, x, vl0, L, a0, a1);
  const float2v wb = rhs32x2<DAMP>(pk_fma(L.hh, a0, ms), pk_fma(L.hhd, a1, x), vlm, L, b0, b1);
  const float2v wc = rhs32x2<DAMP>(pk_fma(L.hh, b0, ms), pk_fma(L.hhd, b1, x), vlm, L, c0, c1);
  const float2v we = rhs32x2<DAMP>(pk_fma(L.h, c0, ms), pk_fma(L.hd, c1, x), vl1, L, e0, e1);
  const float2v two = {2.0f, 2.0f};
  ms = pk_fma(L.h6, pk_fma(two, b0 + c0, a0 + e0), ms);
  x = pk_fma(L.h6d, pk_fma(two, b1 + c1, a1 + e1), x);
  return pk_fma(two, wb + wc, wa + we);
}

// sums of squares of two chains that share the observation series (both belong to the workgroup's group)
template <bool DAMP>
__device__ __forceinline__ void solve32x2(float *lds, const Consts &K, bool resident, const bool active[2], const double dc[2],
                                          const double a[2], const double b[2], double ssq[2]) {
#pragma clang fp contract(off)  // every fused multiply-add of this path is written out: one-chain and two-chain forms round alike
  const Lane32x2 L = make_lane32x2(dc, a, b, K);
  float2v ms = {(float)(K.mu0 / ((1e-2 * 10) / dc[0])), (float)(K.mu0 / ((1e-2 * 10) / dc[1]))}, x = {1.0f, 1.0f};
  const bool any = active[0] || active[1];
  ssq[0] = ssq[1] = 0.0;
  if (any) {
    const double d0 = (double)(float)K.data[0];
    ssq[0] = ssq[1] = d0 * d0;
  }
  auto emit = [&](float2v dv, float obs) {
    const float2v ak = dv * L.cv;  // RateStateModel.py:388, from the interval's velocity increment
    const double r0 = (double)(ak.x - obs), r1 = (double)(ak.y - obs);
    ssq[0] = __builtin_fma(r0, r0, ssq[0]);
    ssq[1] = __builtin_fma(r1, r1, ssq[1]);
  };
  for (int k0 = 1; k0 < K.nout; k0 += K.kc) {
    const int kn = min(K.kc, K.nout - k0);
    if (!resident) stage_chunk32(lds, K, k0, kn);
    if (!any) continue;
    const float *ld = lds + lds_data_offset32(K);
    int kk = 0;
    if (K.S == 1) {
      constexpr int NU = 8;
      for (; kk + NU <= kn; kk += NU) {
        const float *v = lds + 2 * kk;
        float obs[NU];
        float2v dv[NU];
#pragma unroll
        for (int j = 0; j < NU; ++j) obs[j] = ld[kk + j];
#pragma unroll
        for (int j = 0; j < NU; ++j) dv[j] = rk4_step32x2<DAMP>(ms, x, v[2 * j], v[2 * j + 1], v[2 * j + 2], L);
#pragma unroll
        for (int j = 0; j < NU; ++j) emit(dv[j], obs[j]);
      }
    }
    int j = 2 * K.S * kk;
    for (; kk < kn; ++kk) {
      float2v dv = {0.0f, 0.0f};
      for (int sub = 0; sub < K.S; ++sub, j += 2) dv += rk4_step32x2<DAMP>(ms, x, lds[j], lds[j + 1], lds[j + 2], L);
      emit(dv, ld[kk]);
    }
  }
}

}  // namespace f32
}  // namespace rsf
