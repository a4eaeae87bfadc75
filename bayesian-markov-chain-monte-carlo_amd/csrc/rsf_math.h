// rsf_math.h — fp64 log / exp / reciprocal for the RHS inner loop on gfx950.
//
// The kernel is bound by fp64 VALU issue (one wave64 v_fma_f64 occupies a SIMD for 4 cycles), so
// the cost of an RK4 step is its fp64 instruction count.  OCML's log/exp/division are written for
// <= 1 ulp over the whole double range with double-double arithmetic and special-case handling
// (~110 fp64 instructions per RHS).  The RHS needs only ~1e-15 relative accuracy on positive,
// normal arguments, so these versions use plain Horner evaluation:
//   rcp  : v_rcp_f64 seed + Newton-Raphson                                (3-5 instructions)
//   log  : frexp split to m in [sqrt(1/2), sqrt(2)), s = (m-1)/(m+1), 2*atanh(s) series in s^2
//   exp  : k = rint(x*log2 e), r = x - k*ln2 (two-piece), Taylor in r, v_ldexp_f64
// Non-finite behaviour that the sampler relies on is kept: log(x) is NaN for x <= 0 or NaN, and
// NaN / +-inf inputs never produce a finite result (so a diverged trajectory still yields a
// non-finite sum of squares and the proposal is rejected, as in the CPU restatement).
#pragma once
#include <hip/hip_runtime.h>

namespace rsf {
namespace fm {

// Horner step p*x + c of the hot-loop series.  Plain fma: with the full log/exp out of the hot loop its
// nine coefficients stay in SGPRs and hipcc emits one 3-address v_fma_f64 v, v, v, s[..] per term.
__device__ __forceinline__ double hfma(double p, double x, double c) { return __builtin_fma(p, x, c); }

constexpr int kRcpNewtonSteps = 2;  // after the v_rcp_f64 seed (measured 4.6e-8): 2e-15, then < 1 ulp

__device__ __forceinline__ double rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);  // v_rcp_f64: hardware seed
#pragma unroll
  for (int i = 0; i < kRcpNewtonSteps; ++i) {
    const double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
  }
  return r;
}

__device__ __forceinline__ double log(double x) {
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool low = m < 0.70710678118654752440;
  m = low ? m + m : m;                         // [sqrt(1/2), sqrt(2))
  e = low ? e - 1 : e;
  const double f = m - 1.0;                    // exact
  const double s = f * rcp(m + 1.0);
  const double z = s * s;                      // <= 0.02944
  // 2*atanh(s) = 2s + s*z*(2/3 + 2/5 z + 2/7 z^2 + ... + 2/21 z^9); truncation < 2.2e-19
  double p = 2.0 / 21.0;
  p = __builtin_fma(p, z, 2.0 / 19.0);
  p = __builtin_fma(p, z, 2.0 / 17.0);
  p = __builtin_fma(p, z, 2.0 / 15.0);
  p = __builtin_fma(p, z, 2.0 / 13.0);
  p = __builtin_fma(p, z, 2.0 / 11.0);
  p = __builtin_fma(p, z, 2.0 / 9.0);
  p = __builtin_fma(p, z, 2.0 / 7.0);
  p = __builtin_fma(p, z, 2.0 / 5.0);
  p = __builtin_fma(p, z, 2.0 / 3.0);
  const double lm = __builtin_fma(s * z, p, s + s);
  const double r = __builtin_fma((double)e, 0.69314718055994530942, lm);
  return x > 0.0 ? r : __builtin_nan("");
}

__device__ __forceinline__ double exp(double x) {
  const double k = __builtin_rint(x * 1.4426950408889634074);
  double r = __builtin_fma(-k, 0x1.62e42fefa38p-1, x);      // ln2 high part: 42 significant bits, k*hi exact
  r = __builtin_fma(-k, 0x1.ef35793c7673p-45, r);           // ln2 low part
  // |r| <= 0.3466: Taylor to r^13/13!, truncation < 5e-18 relative
  double p = 1.0 / 6227020800.0;
  p = __builtin_fma(p, r, 1.0 / 479001600.0);
  p = __builtin_fma(p, r, 1.0 / 39916800.0);
  p = __builtin_fma(p, r, 1.0 / 3628800.0);
  p = __builtin_fma(p, r, 1.0 / 362880.0);
  p = __builtin_fma(p, r, 1.0 / 40320.0);
  p = __builtin_fma(p, r, 1.0 / 5040.0);
  p = __builtin_fma(p, r, 1.0 / 720.0);
  p = __builtin_fma(p, r, 1.0 / 120.0);
  p = __builtin_fma(p, r, 1.0 / 24.0);
  p = __builtin_fma(p, r, 1.0 / 6.0);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)k);  // v_cvt_i32_f64 saturates; v_ldexp_f64 over/underflows to inf/0
}

// sin and cos of 2*pi*u for u in [0, 1] (Box-Muller angle): quadrant k = rint(4u), r = u - k/4 exactly in
// [-1/8, 1/8], theta = 2*pi*r in [-pi/4, pi/4] (two-piece 2*pi), Taylor to theta^15 / theta^16 (truncation < 5e-17).
// ~35 instructions against ~150 for OCML's sincos with its full-range argument reduction.
__device__ __forceinline__ void sincos2pi(double u, double &sn, double &cs) {
  const double k = __builtin_rint(4.0 * u);
  const double r = __builtin_fma(k, -0.25, u);
  const double t = __builtin_fma(r, 0x1.1a62633145c07p-52, r * 0x1.921fb54442d18p+2);  // 2*pi = hi + lo
  const double z = t * t;
  double ps = -1.0 / 1307674368000.0;               // -1/15!
  ps = __builtin_fma(ps, z, 1.0 / 6227020800.0);    //  1/13!
  ps = __builtin_fma(ps, z, -1.0 / 39916800.0);     // -1/11!
  ps = __builtin_fma(ps, z, 1.0 / 362880.0);        //  1/9!
  ps = __builtin_fma(ps, z, -1.0 / 5040.0);         // -1/7!
  ps = __builtin_fma(ps, z, 1.0 / 120.0);           //  1/5!
  ps = __builtin_fma(ps, z, -1.0 / 6.0);            // -1/3!
  const double s0 = __builtin_fma(t * z, ps, t);
  double pc = 1.0 / 20922789888000.0;               //  1/16!
  pc = __builtin_fma(pc, z, -1.0 / 87178291200.0);  // -1/14!
  pc = __builtin_fma(pc, z, 1.0 / 479001600.0);     //  1/12!
  pc = __builtin_fma(pc, z, -1.0 / 3628800.0);      // -1/10!
  pc = __builtin_fma(pc, z, 1.0 / 40320.0);         //  1/8!
  pc = __builtin_fma(pc, z, -1.0 / 720.0);          // -1/6!
  pc = __builtin_fma(pc, z, 1.0 / 24.0);            //  1/4!
  pc = __builtin_fma(pc, z, -0.5);
  const double c0 = __builtin_fma(pc, z, 1.0);
  const int q = (int)k & 3;                          // k in 0..4
  const double a = (q & 1) ? c0 : s0, b = (q & 1) ? s0 : c0;
  sn = (q & 2) ? -a : a;                             // q: 0 (s, c)  1 (c, -s)  2 (-s, -c)  3 (-c, s)
  cs = ((q + 1) & 2) ? -b : b;
}

}  // namespace fm
}  // namespace rsf
