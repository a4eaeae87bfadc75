// rsf_device_dop853.h — the reference's own integration scheme on the GPU (RSF_FLAG_DOP853).
//
// RateStateModel.evaluate drives scipy.integrate.ode('dop853', rtol=1e-6, atol=1e-10) once per output interval
// (RateStateModel.py:374-389).  This is that algorithm per lane: Hairer's DOP853 step (12 stages, 8th order, 5th/3rd
// order error estimators), step-size control with safety 0.9 and factors 0.3 .. 6 (beta = 0), at most 500 steps per
// call, HMAX = interval length, HINIT on the first call and the predicted step size carried from call to call
// (scipy keeps it in the work array).  After the first interval every call is normally ONE accepted step of
// length delta_t, so lanes stay convergent.  The RHS is evaluated in full at the end points of every step and
// incrementally from the step's start point at its eleven inner stages (friction_incr) — this mode is for fidelity to the
// reference's numbers (agreement ~1e-12 with its trajectories), the fixed-step RK4 path is the fast one.  Tableau: include/rsf_dop853_tableau.h (generated from SciPy's table).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rsf_dop853_tableau.h"
#include "rsf_device.h"

namespace rsf {
namespace dp {

constexpr double kRtol = 1e-6, kAtol = 1e-10;

struct LaneD {
  double inv_dc, kprime, inv_a, b;
};

// loading velocity V_l(t), RateStateModel.py:327-329
__device__ __forceinline__ double loading(const Consts &K, double t) {
  return K.V_ref * (1.0 + fm::exp(t * (-1.0 / 20.0)) * ::sin(10.0 * t));
}

// slip rate and 1/theta at a point: what an incremental evaluation near that point starts from
struct Base {
  double v, rth;
};

// the derivative once v and 1/theta are known (RateStateModel.py:331-353)
template <bool DAMP>
__device__ __forceinline__ void friction_tail(const Consts &K, const LaneD &L, double vl, double v, double rth, double theta,
                                              double f[3]) {
  f[1] = 1.0 - v * theta * L.inv_dc;
  f[0] = L.kprime * (vl - v);
  const double bt = L.b * rth * f[1], va = v * L.inv_a;
  f[2] = va * (f[0] - bt);
  if (DAMP) {
    f[0] = f[0] - K.k1 * f[2];
    f[2] = va * (f[0] - bt);
  }
}

// RateStateModel.py:318-355 given the loading velocity vl at the evaluation time
template <bool DAMP>
__device__ __forceinline__ void friction(const Consts &K, const LaneD &L, double vl, const double y[3], double f[3], Base &b) {
  b.v = K.V_ref * fm::exp(L.inv_a * (y[0] - K.mu_ref - L.b * fm::log(K.V_ref * y[1] * L.inv_dc)));
  b.rth = fm::rcp(y[1]);
  friction_tail<DAMP>(K, L, vl, b.v, b.rth, y[1], f);
}

template <bool DAMP>
__device__ __forceinline__ void friction(const Consts &K, const LaneD &L, double vl, const double y[3], double f[3]) {
  Base b;
  friction<DAMP>(K, L, vl, y, f, b);
}

// The same derivative at a stage point ys = y + (dmu, dth, .) of a step whose start point has slip rate / reciprocal
// state b0: v = b0.v exp(dlt), dlt = (dmu - b log1p(rho))/a, rho = dth/theta — by the series of rsf_device.h's NARROW
// tier (log1p to rho^6/6, expm1 to dlt^7/5040, 1/theta by one Newton step; truncation < 1e-19), valid inside
// |rho| < 2^-9, |dlt| < 2^-6.  Outside, `bad` is raised and the caller redoes the step's stages with full evaluations
// — ONE test per step: a branch per stage would stall a lone wave for the latency of its compare eleven times per
// step.  A DOP853 step spans one output interval, so its stage increments are those of an RK4 step: the full log/exp
// is needed only at the step's end points.
// SHORT: the series lengths and thresholds of rsf_device.h's TIGHT tier (log1p to rho^2/2 inside |rho| < 2^-20, expm1 to
// dlt^5/120 inside |dlt| < 2^-9) — what the steady-state fast path of call() tries first.
template <bool DAMP, bool SHORT>
__device__ __forceinline__ void friction_incr(const Consts &K, const LaneD &L, double vl, const Base &b0, double dmu, double dth,
                                              const double ys[3], double f[3], bool &bad) {
  const double rho = dth * b0.rth;
  double p;
  if (SHORT) {
    p = __builtin_fma(rho, -0.5, 1.0);
  } else {
    p = -1.0 / 6.0;
    p = __builtin_fma(p, rho, 1.0 / 5.0);
    p = __builtin_fma(p, rho, -1.0 / 4.0);
    p = __builtin_fma(p, rho, 1.0 / 3.0);
    p = __builtin_fma(p, rho, -0.5);
    p = __builtin_fma(p, rho, 1.0);
  }
  const double dlt = L.inv_a * __builtin_fma(-L.b, p * rho, dmu);
  bad = bad || !(__builtin_fabs(rho) < (SHORT ? 0x1.0p-20 : 0x1.0p-9) && __builtin_fabs(dlt) < (SHORT ? 0x1.0p-9 : 0x1.0p-6));
  double e;
  if (SHORT) {
    e = 1.0 / 120.0;
  } else {
    e = 1.0 / 5040.0;
    e = __builtin_fma(e, dlt, 1.0 / 720.0);
    e = __builtin_fma(e, dlt, 1.0 / 120.0);
  }
  e = __builtin_fma(e, dlt, 1.0 / 24.0);
  e = __builtin_fma(e, dlt, 1.0 / 6.0);
  e = __builtin_fma(e, dlt, 0.5);
  e = __builtin_fma(e, dlt, 1.0);
  const double v = __builtin_fma(b0.v * dlt, e, b0.v);
  double rth = __builtin_fma(b0.rth, __builtin_fma(rho, rho, -rho), b0.rth);
  if (!SHORT) rth = __builtin_fma(rth, __builtin_fma(-ys[1], rth, 1.0), rth);  // (|rho| < 2^-20: the series is exact to rounding)
  friction_tail<DAMP>(K, L, vl, v, rth, ys[1], f);
}

template <bool DAMP>
__device__ __forceinline__ double hinit(const Consts &K, const LaneD &L, double x, const double y[3], const double f0[3],
                                        double hmax) {
  double dnf = 0.0, dny = 0.0, y1[3], f1[3], der2 = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double sk = kAtol + kRtol * fabs(y[i]);
    dnf += (f0[i] / sk) * (f0[i] / sk);
    dny += (y[i] / sk) * (y[i] / sk);
  }
  double h = (dnf <= 1e-10 || dny <= 1e-10) ? 1.0e-6 : sqrt(dny / dnf) * 0.01;
  h = fmin(h, hmax);
#pragma unroll
  for (int i = 0; i < 3; ++i) y1[i] = y[i] + h * f0[i];
  friction<DAMP>(K, L, loading(K, x + h), y1, f1);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double sk = kAtol + kRtol * fabs(y[i]);
    der2 += ((f1[i] - f0[i]) / sk) * ((f1[i] - f0[i]) / sk);
  }
  der2 = sqrt(der2) / h;
  const double der12 = fmax(fabs(der2), sqrt(dnf));
  const double h1 = der12 <= 1e-15 ? fmax(1.0e-6, fabs(h) * 1.0e-3) : pow(0.01 / der12, 1.0 / 8.0);
  return fmin(fmin(100.0 * fabs(h), h1), hmax);
}

// one dop853 call (forward in time): y from x to xend; hc = carried step size (0 => HINIT).  false on failure.
// the eleven inner stages of one step of size h from (x, y, k[0]; b0), evaluated incrementally; `bad`: some increment
// left the series' range (the values of that lane are then not to be used).  TABULATED = the fast path of call(): the
// loading table and the short series.
template <bool DAMP, bool TABULATED>
__device__ __forceinline__ void stages_incr(const Consts &K, const LaneD &L, const double *tab, bool standard, double x, double h,
                                            const double y[3], double (&k)[12][3], const Base &b0, bool &bad) {
#pragma unroll
  for (int st = 1; st < 12; ++st) {
    double inc[3], ys[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < st; ++j)
        if (RSF_DP_A[st - 1][j] != 0.0) s += RSF_DP_A[st - 1][j] * k[j][i];
      inc[i] = h * s;
      ys[i] = y[i] + inc[i];
    }
    const double vl = (TABULATED || standard) ? tab[st] : loading(K, st == 11 ? x + h : x + RSF_DP_C[st] * h);
    friction_incr<DAMP, TABULATED>(K, L, vl, b0, inc[0], inc[1], ys, k[st], bad);
  }
}

// 8th-order solution k5 and the error estimate of the step; returns err, and err ** (1/8) in fac11
__device__ __forceinline__ double solution_and_error(double h, const double y[3], const double (&k)[12][3], double k5[3],
                                                     double &fac11) {
  double err = 0.0, err2 = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    double s = 0.0, e3 = 0.0, e5 = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const double kj = k[RSF_DP_W_STAGE[j]][i];
      s += RSF_DP_B[j] * kj;
      e3 += RSF_DP_E3[j] * kj;
      e5 += RSF_DP_E5[j] * kj;
    }
    k5[i] = y[i] + h * s;
    // Step-size control: the error norm and the step-size factor only steer h (and the accept test err <= 1), so
    // they use the kernel's reciprocal / log / exp (<= 4 ulp from the divisions and pow of the Fortran code, a fifth
    // of their instructions); the solution itself (k5) is formed exactly as the reference forms it.
    const double isk = fm::rcp(kAtol + kRtol * fmax(fabs(y[i]), fabs(k5[i])));
    err2 += (e3 * isk) * (e3 * isk);
    err += (e5 * isk) * (e5 * isk);
  }
  double deno = err + 0.01 * err2;
  if (deno <= 0.0) deno = 1.0;
  err = fabs(h) * err * sqrt(fm::rcp(3.0 * deno));
  fac11 = err > 0.0 ? fm::exp(0.125 * fm::log(err)) : (err == 0.0 ? 0.0 : err);  // err ** (1/8); NaN stays NaN
  return err;
}

// `tab` (LDS, 12 values) holds V_l at the stage times of the STANDARD step of this interval — the first step
// clipped to h = xend - x, which is what every call after the first interval takes; x and xend are the
// accumulated grid times shared by all lanes, so the host can tabulate them bit-exactly (rsf_set_model).
// Any other step (HINIT's first interval, steps after a rejection) evaluates V_l(t) directly.
// `kf` carries the derivative at (x, y) from call to call: every call starts by evaluating the RHS at its start point
// (SciPy does), which is the point — and, for the tabulated standard step, bit for bit the value — at which the
// previous call's last accepted step ended (first-same-as-last); have_kf = false on the first call.
template <bool DAMP>
__device__ __forceinline__ bool call(const Consts &K, const LaneD &L, const double *tab, double &x, double xend, double y[3],
                                     double &hc, double kf[3], Base &bf, bool &have_kf) {
  constexpr double safe = 0.9, facc1 = 1.0 / 0.3, facc2 = 1.0 / 6.0, uround = 2.3e-16;
  const double hmax = fabs(xend - x);
  double k[12][3], ys[3], k5[3];
  double h = hc;
  bool last = false, reject = false;
  Base b0 = bf;  // slip rate and 1/theta at (x, y): the stages of a step are evaluated incrementally from it
  if (have_kf) {
#pragma unroll
    for (int i = 0; i < 3; ++i) k[0][i] = kf[i];
  } else {
    friction<DAMP>(K, L, tab[0], y, k[0], b0);  // V_l(x): x is the interval's start time for every lane
  }
  if (h == 0.0) h = hinit<DAMP>(K, L, x, y, k[0], hmax);
  // Fast path, the steady state of every call after the first interval: the carried step size reaches past xend for
  // EVERY lane of the wave, so all take the tabulated step h = xend - x, and all accept it.  Two wave-uniform tests
  // instead of the general loop's six per-lane branches (each of which a lone wave sits out for the latency of its
  // compare).  Anything else — a lane that wants a smaller step, leaves the series' range or rejects — falls through
  // to the general loop, which starts again from the untouched (x, y, k[0]).
  if (__all(have_kf && (x + 1.01 * h - xend > 0.0) && !(0.1 * fabs(h) <= fabs(x) * uround))) {
    const double hs = xend - x;
    bool bad = false;
    double fac11;
    stages_incr<DAMP, true>(K, L, tab, true, x, hs, y, k, b0, bad);
    const double err = solution_and_error(hs, y, k, k5, fac11);
    if (__all(!bad && err <= 1.0)) {
      friction<DAMP>(K, L, tab[11], k5, kf, bf);  // first-same-as-last, at xend: full evaluation
#pragma unroll
      for (int i = 0; i < 3; ++i) y[i] = k5[i];
      x = x + hs;
      hc = hs * fm::rcp(fmax(facc2, fmin(facc1, fac11 * (1.0 / safe))));
      return true;
    }
  }
  for (int nstep = 0;; ) {
    if (nstep > 500) return false;
    if (0.1 * fabs(h) <= fabs(x) * uround) return false;
    if (x + 1.01 * h - xend > 0.0) { h = xend - x; last = true; }
    const bool standard = last && nstep == 0;  // the tabulated step
    ++nstep;
    bool bad = false;
    stages_incr<DAMP, false>(K, L, tab, standard, x, h, y, k, b0, bad);
    if (__builtin_expect(__any(bad), 0)) {  // an increment outside the series' range: this step again, every stage in full
      if (bad) {
#pragma unroll
        for (int st = 1; st < 12; ++st) {
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < st; ++j)
              if (RSF_DP_A[st - 1][j] != 0.0) s += RSF_DP_A[st - 1][j] * k[j][i];
            ys[i] = y[i] + h * s;
          }
          const double vl = standard ? tab[st] : loading(K, st == 11 ? x + h : x + RSF_DP_C[st] * h);
          friction<DAMP>(K, L, vl, ys, k[st]);
        }
      }
    }
    double fac11;
    const double err = solution_and_error(h, y, k, k5, fac11);
    double hnew = h * fm::rcp(fmax(facc2, fmin(facc1, fac11 * (1.0 / safe))));
    if (err <= 1.0) {
      friction<DAMP>(K, L, standard ? tab[11] : loading(K, x + h), k5, k[0], b0);  // first-same-as-last, at x + h: full
#pragma unroll
      for (int i = 0; i < 3; ++i) y[i] = k5[i];
      x = x + h;
      if (last) {
        hc = hnew;
#pragma unroll
        for (int i = 0; i < 3; ++i) kf[i] = k[0][i];
        bf = b0;
        have_kf = true;
        return true;
      }
      if (fabs(hnew) > hmax) hnew = hmax;
      if (reject) hnew = fmin(fabs(hnew), fabs(h));
      reject = false;
    } else {
      hnew = h * fm::rcp(fmin(facc1, fac11 * (1.0 / safe)));
      reject = true;
      last = false;
    }
    h = hnew;
  }
}

// LDS chunk of the DOP853 mode: [ 12 loading values per interval : 12*kc ][ data : kc ]
constexpr int kTab = 12;
__device__ __forceinline__ int lds_data_offset_dp(const Consts &K) { return kTab * K.kc; }

__device__ __forceinline__ void stage_chunk_dp(double *lds, const Consts &K, int k0, int kn) {
  __syncthreads();
  for (int i = threadIdx.x; i < kTab * kn; i += blockDim.x) lds[i] = K.vl[(int64_t)kTab * (k0 - 1) + i];
  if (K.data) {
    double *ld = lds + lds_data_offset_dp(K);
    for (int i = threadIdx.x; i < kn; i += blockDim.x) ld[i] = K.data[k0 + i];
  }
  __syncthreads();
}

// Forward solve in the reference's scheme.  Same calling convention as rsf::solve (all threads call it; the
// loading table and the observation are read from the LDS chunk staged by stage_chunk_dp).
template <bool DAMP, bool WANT_SSQ, bool WANT_ACC>
__device__ __forceinline__ double solve(double *lds, const Consts &K, bool resident, bool active, double dc, double a,
                                        double b, double *acc_out, int64_t stride) {
  LaneD L;
  L.inv_dc = 1.0 / dc; L.kprime = (1e-2 * 10) / dc; L.inv_a = 1.0 / a; L.b = b;
  const double delta_t = K.dt, inv_dt = K.inv_dt;
  double y[3] = {K.mu0, dc / K.V_ref, K.V_ref};
  double x = K.t0, vprev = K.V_ref, hc = 0.0, ssq = 0.0, kf[3] = {0.0, 0.0, 0.0};
  Base bf = {0.0, 0.0};
  bool failed = false, have_kf = false;
  if (WANT_SSQ && active) { const double d0 = K.data[0]; ssq = d0 * d0; }
  if (WANT_ACC && active) acc_out[0] = 0.0;
  const double *ld = lds + lds_data_offset_dp(K);
  for (int k0 = 1; k0 < K.nout; k0 += K.kc) {
    const int kn = min(K.kc, K.nout - k0);
    if (!resident) stage_chunk_dp(lds, K, k0, kn);
    if (!active) continue;
    for (int kk = 0; kk < kn; ++kk) {
      double ak = 0.0;  // after a failed call the reference's arrays keep their zeros (RateStateModel.py:361-381)
      if (!failed) {
        failed = !call<DAMP>(K, L, lds + kTab * kk, x, x + delta_t, y, hc, kf, bf, have_kf);
        ak = (y[2] - vprev) * inv_dt;
        vprev = y[2];
      }
      if (WANT_ACC) acc_out[(int64_t)(k0 + kk) * stride] = ak;
      if (WANT_SSQ) { const double r = ak - ld[kk]; ssq = __builtin_fma(r, r, ssq); }
    }
  }
  return ssq;
}

}  // namespace dp
}  // namespace rsf
