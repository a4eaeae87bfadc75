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
// ~4 digits in float32.  Same rescaled state as the float64 path: ms = mu/k', x = V_ref theta/Dc.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rsf_device.h"

namespace rsf {
namespace f32 {

struct Lane32 {
  float kia2;    // (k'/a) log2(e)
  float tc2;     // -(mu_ref/a) log2(e)
  float boa;     // b/a
  float kprime, k1k, via, bdc;
  float hh, h, h6, hhd, hd, h6d, vref;
};

__device__ __forceinline__ Lane32 make_lane32(double dc, double a, double b, const Consts &K) {
  const double log2e = 1.4426950408889634074;
  const double inv_a = 1.0 / a, inv_dc = 1.0 / dc, kprime = (1e-2 * 10) / dc;
  Lane32 L;
  L.kia2 = (float)(kprime * inv_a * log2e);
  L.tc2 = (float)(-K.mu_ref * inv_a * log2e);
  L.boa = (float)(b * inv_a);
  L.kprime = (float)kprime;
  L.k1k = (float)(K.k1 / kprime);
  L.via = (float)(K.V_ref * inv_a);
  L.bdc = (float)(b * K.V_ref * inv_dc);
  L.hh = (float)K.hh; L.h = (float)K.h; L.h6 = (float)K.h6;
  const double vdc = K.V_ref * inv_dc;  // dx/dt = (V_ref/Dc) (1 - w x)
  L.hhd = (float)(K.hh * vdc); L.hd = (float)(K.h * vdc); L.h6d = (float)(K.h6 * vdc);
  L.vref = (float)K.V_ref;
  return L;
}

template <bool DAMP>
__device__ __forceinline__ void rhs32(float ms, float x, float vl, const Lane32 &L, float &d0, float &d1, float &d2) {
  const float w = __builtin_amdgcn_exp2f(__builtin_fmaf(-L.boa, __builtin_amdgcn_logf(x), __builtin_fmaf(ms, L.kia2, L.tc2)));
  const float rx = __builtin_amdgcn_rcpf(x);
  d1 = __builtin_fmaf(-w, x, 1.0f);
  d0 = __builtin_fmaf(-L.vref, w, vl);
  const float bt = (L.bdc * d1) * rx;
  const float va = w * L.via;
  d2 = va * __builtin_fmaf(L.kprime, d0, -bt);
  if (DAMP) {
    d0 = __builtin_fmaf(-L.k1k, d2, d0);
    d2 = va * __builtin_fmaf(L.kprime, d0, -bt);
  }
}

// one RK4 step; returns the velocity increment of the step
template <bool DAMP>
__device__ __forceinline__ float rk4_step32(float &ms, float &x, float vl0, float vlm, float vl1, const Lane32 &L) {
  float a0, a1, a2, b0, b1, b2, c0, c1, c2, e0, e1, e2;
  rhs32<DAMP>(ms, x, vl0, L, a0, a1, a2);
  rhs32<DAMP>(__builtin_fmaf(L.hh, a0, ms), __builtin_fmaf(L.hhd, a1, x), vlm, L, b0, b1, b2);
  rhs32<DAMP>(__builtin_fmaf(L.hh, b0, ms), __builtin_fmaf(L.hhd, b1, x), vlm, L, c0, c1, c2);
  rhs32<DAMP>(__builtin_fmaf(L.h, c0, ms), __builtin_fmaf(L.hd, c1, x), vl1, L, e0, e1, e2);
  ms = __builtin_fmaf(L.h6, a0 + 2.0f * b0 + 2.0f * c0 + e0, ms);
  x = __builtin_fmaf(L.h6d, a1 + 2.0f * b1 + 2.0f * c1 + e1, x);
  return L.h6 * (a2 + 2.0f * b2 + 2.0f * c2 + e2);
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

template <bool DAMP, bool WANT_SSQ, bool WANT_ACC>
__device__ __forceinline__ void integrate_chunk32(const float *lds, const Consts &K, const Lane32 &L, int k0, int kn,
                                                  float &ms, float &x, double &ssq, double *acc_out, int64_t stride) {
  const float *ld = lds + lds_data_offset32(K);
  const float inv_dt = (float)K.inv_dt;
  int j = 0;
  for (int kk = 0; kk < kn; ++kk) {
    float dv = 0.0f;
    for (int sub = 0; sub < K.S; ++sub, j += 2) dv += rk4_step32<DAMP>(ms, x, lds[j], lds[j + 1], lds[j + 2], L);
    const float ak = dv * inv_dt;  // RateStateModel.py:388, from the increment
    if (WANT_ACC) acc_out[(int64_t)(k0 + kk) * stride] = (double)ak;
    if (WANT_SSQ) {
      const double r = (double)(ak - ld[kk]);
      ssq = __builtin_fma(r, r, ssq);
    }
  }
}

template <bool DAMP, bool WANT_SSQ, bool WANT_ACC>
__device__ __forceinline__ double solve32(float *lds, const Consts &K, bool resident, bool active, double dc, double a,
                                          double b, double *acc_out, int64_t stride) {
  const Lane32 L = make_lane32(dc, a, b, K);
  float ms = (float)(K.mu0 / ((1e-2 * 10) / dc)), x = 1.0f;  // x = V_ref theta(0)/Dc, theta(0) = Dc/V_ref
  double ssq = 0.0;
  if (WANT_SSQ && active) {
    const double d0 = (double)(float)K.data[0];
    ssq = d0 * d0;
  }
  if (WANT_ACC && active) acc_out[0] = 0.0;
  for (int k0 = 1; k0 < K.nout; k0 += K.kc) {
    const int kn = min(K.kc, K.nout - k0);
    if (!resident) stage_chunk32(lds, K, k0, kn);
    if (active) integrate_chunk32<DAMP, WANT_SSQ, WANT_ACC>(lds, K, L, k0, kn, ms, x, ssq, acc_out, stride);
  }
  return ssq;
}

}  // namespace f32
}  // namespace rsf
