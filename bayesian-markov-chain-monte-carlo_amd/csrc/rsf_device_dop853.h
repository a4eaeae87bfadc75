// rsf_device_dop853.h — the reference's own integration scheme on the GPU (RSF_FLAG_DOP853).
//
// RateStateModel.evaluate drives scipy.integrate.ode('dop853', rtol=1e-6, atol=1e-10) once per output interval
// (RateStateModel.py:374-389).  This is that algorithm per lane: Hairer's DOP853 step (12 stages, 8th order, 5th/3rd
// order error estimators), step-size control with safety 0.9 and factors 0.3 .. 6 (beta = 0), at most 500 steps per
// call, HMAX = interval length, HINIT on the first call and the predicted step size carried from call to call
// (scipy keeps it in the work array).  After the first interval every call is normally ONE accepted step of
// length delta_t, so lanes stay convergent.  That steady state is the fast path of call(): the RHS is evaluated in full
// at the step's end point and incrementally from its start point at the eleven inner stages (friction_incr); every other
// step goes through the general loop with full evaluations.  This mode is for fidelity to the reference's numbers
// (agreement ~1e-12 with its trajectories), the fixed-step RK4 path is the fast one.
// Tableau: include/rsf_dop853_tableau.h (generated from SciPy's table).
//
// What is NOT carried per stage: the third component.  V never feeds back into the RHS (RateStateModel.py:336-353), so
// no stage needs the V derivatives of earlier stages; they enter only the step's closing sums (8th-order weights B and
// the error estimators, stages 1 and 6..12).  Those sums are accumulated as each stage completes (VSums) instead of
// keeping twelve values live — with the mu and theta derivatives that is 24 doubles per lane instead of 36, which is
// what lets the sampler kernel hold the step in registers at two waves per SIMD.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/rsf_dop853_tableau.h"
#include "rsf_device.h"

namespace rsf {
namespace dp {

constexpr double kRtol = 1e-6, kAtol = 1e-10;
// dop853.f forms the 3rd-order estimator from the 8th-order sum: err2_i = (sum_j b_j k_j)_i - bhh1 k1 - bhh2 k9 - bhh3 k12
// (= sum_j E3_j k_j with the tableau's E3 = B - bhh on those three stages): three products per component instead of eight.
constexpr double kBhh1 = 0.244094488188976377952755905512, kBhh2 = 0.733846688281611857341361741547,
                 kBhh3 = 0.0220588235294117647058823529412;
// (tests/test_oracle_golden.py::test_dop853_bhh_constants checks them against the tableau's E3)

// The fast path's copy of the tableau, in constant memory, one packed row per stage: the non-zero A_st,j in column
// order, then (stages with weight in the closing sums) B_st and E5_st; the last row holds the closing sums' own
// {B, E5} pairs and bhh.  As literals the ~75 constants of a step were 150 s_mov_b32 per interval — and an instruction
// of any kind costs its wave one issue turn (every 4th cycle), so a wave spent 16 % of its cycles materialising
// constants.  From here a stage's row arrives in one or two s_load_dwordx8/x16 (scalar data cache), requested one
// stage ahead — between two `pin`s, which fence the instruction scheduler — and from a base the compiler cannot see through
// (`row_of`, once per step): hoisted to the kernel's entry instead, the loads held 150 SGPRs for the whole kernel
// and spilled to VGPR lanes; sunk to their uses, each was an exposed scalar-cache latency.
constexpr int kRowLen = 12, kClosingLen = 19;
struct alignas(64) StepTable {
  double a[11][kRowLen];
  double w[8][2];
  double bhh[4];
};
constexpr StepTable make_step_table() {
  StepTable t{};
  for (int i = 0; i < 11; ++i) {
    int n = 0;
    for (int j = 0; j <= i; ++j)
      if (RSF_DP_A[i][j] != 0.0) t.a[i][n++] = RSF_DP_A[i][j];
    if (i + 1 >= 5) { t.a[i][n++] = RSF_DP_B[i + 1 - 4]; t.a[i][n++] = RSF_DP_E5[i + 1 - 4]; }
  }
  for (int j = 0; j < 8; ++j) { t.w[j][0] = RSF_DP_B[j]; t.w[j][1] = RSF_DP_E5[j]; }
  t.bhh[0] = kBhh1; t.bhh[1] = kBhh2; t.bhh[2] = kBhh3;
  return t;
}
constexpr StepTable kStepTable0 = make_step_table();
__constant__ StepTable kStepTable = kStepTable0;
typedef const double __attribute__((address_space(4))) *ConstRow;
__device__ __forceinline__ ConstRow row_of(const double *p) {
  ConstRow r = (ConstRow)p;
  asm volatile("" : "+s"(r));
  return r;
}
template <int N>
__device__ __forceinline__ void pin(double (&c)[N], int n) {  // the first n values are in scalar registers from here on
#pragma unroll
  for (int i = 0; i < N; ++i)
    if (i < n) asm volatile("" : "+s"(c[i]));
}
constexpr int row_len(int st) {  // stage st (0-based, 1..11): its non-zero A entries + {B, E5} where it carries weight
  int n = st >= 5 ? 2 : 0;
  for (int j = 0; j < st; ++j) n += RSF_DP_A[st - 1][j] != 0.0;
  return n;
}

struct LaneD {
  double inv_dc, kprime, inv_a, b;
  double a, boa, nhboa;  // a, b/a and -b/2a: the fast path's folded constants (FastStep)
};

// loading velocity V_l(t) = V_ref (1 + exp(-t/20) sin(10 t)), RateStateModel.py:327-329, at a time that is not in the
// table (HINIT's first interval, steps after a rejection).  sin(10 t) = sin(2 pi u), u = frac(10 t / 2 pi): 10 t / 2 pi stays
// below ~100 over the model's time span, so the one rounding of the constant costs < 2e-14 in u and the sine agrees with
// libm's to ~1e-13 absolute — against the 1e-9 this mode is held to — at a tenth of the registers and instructions of
// the full-range argument reduction in OCML's sin (which alone cost the sampler kernel 80 B of scratch per lane).
__device__ __forceinline__ double loading(const Consts &K, double t) {
  const double w = t * 1.5915494309189533577;  // 10 / (2 pi)
  const double u = __builtin_fma(t, 1.5915494309189533577, -__builtin_floor(w));
  double sn, cs;
  fm::sincos2pi(u, sn, cs);
  return K.V_ref * (1.0 + fm::exp(t * (-1.0 / 20.0)) * sn);
}

// slip rate and 1/theta at a point: what an incremental evaluation near that point starts from
struct Base {
  double v, rth;
};

// derivative of (mu, theta, V) at one point
struct Deriv {
  double m, t, v;
};

// the derivative once v and 1/theta are known (RateStateModel.py:331-353); WANT_V = false leaves d.v at the undamped
// value where only the damping feedback needs it (stages whose V derivative enters no sum)
template <bool DAMP, bool WANT_V>
__device__ __forceinline__ Deriv friction_tail(const Consts &K, const LaneD &L, double vl, double v, double rth, double theta) {
  Deriv d;
  d.t = 1.0 - v * theta * L.inv_dc;
  d.m = L.kprime * (vl - v);
  const double bt = L.b * rth * d.t, va = v * L.inv_a;
  d.v = va * (d.m - bt);
  if (DAMP) {
    d.m = d.m - K.k1 * d.v;
    if (WANT_V) d.v = va * (d.m - bt);
  }
  return d;
}

// RateStateModel.py:318-355 given the loading velocity vl at the evaluation time
template <bool DAMP>
__device__ __forceinline__ Deriv friction(const Consts &K, const LaneD &L, double vl, double mu, double theta, Base &b) {
  b.v = K.V_ref * fm::exp(L.inv_a * (mu - K.mu_ref - L.b * fm::log(K.V_ref * theta * L.inv_dc)));
  b.rth = fm::rcp(theta);
  return friction_tail<DAMP, true>(K, L, vl, b.v, b.rth, theta);
}

// Largest |rho| / |dlt| of a step's incremental stages, tracked through the high words of the doubles read as floats
// (monotone in |x|, power-of-two thresholds exact; rsf_device.h, struct Guard): one v_max_f32 per value instead of a
// compare and a mask operation, and ONE test per step — a branch per stage would stall a lone wave for the latency of
// its compare eleven times per step.
struct GuardD {
  float rho, dlt;
};

// The derivative at a point (mu + dmu, theta + dth) near a point whose slip rate / reciprocal state b0 are known:
// v = b0.v exp(dlt), dlt = (dmu - b log1p(rho))/a, rho = dth/theta — by the series of rsf_device.h's TIGHT tier (log1p to
// rho^2/2, expm1 to dlt^5/120, 1/theta = (1/theta_0)(1 + q), q = rho^2 - rho; truncation < 1e-17 relative), valid inside
// |rho| < 2^-20, |dlt| < 2^-9.  A DOP853 step spans one output interval, so its stage increments are those of an RK4
// step: the full log/exp is needed only where rounding must not accumulate (kResyncDp).  Outside the range the step's
// guard trips and the step is taken again by the general loop of call(), every stage evaluated in full.
//
// FastStep holds what a step's twelve incremental evaluations share.  The fast path carries the stage derivatives SCALED,
//     km' = (h/a) dmu/dt,   kt' = (h/theta_0) dtheta/dt,   kv' = (h/a) dV/dt,
// so that the tableau sums ARE the series' arguments — rho = sum_j A kt'_j and (h/a) dmu = sum_j A km'_j, no product per
// stage — and every other use of a derivative absorbs its factor into a per-step constant:
//     (h/theta_0) theta_s/Dc = (h/Dc)(1 + rho),      (h/a) k'(V_l - v) = hk (V_l - v),  hk = (h/a) k',
//     (h/a)(b/theta_s) dtheta/dt = (b/a)(1 + q) kt',  q = rho^2 - rho        (b/theta_s = (b/theta_0)(1 + q)),
// the bracket of dV/dt = (v/a)(dmu/dt - (b/theta) dtheta/dt) being one fma on the two derivatives the stage forms anyway
// (rsf_device.h's rhs_tight); the damping pass (RateStateModel.py:349-353) subtracts the same k1 kv' from km' and from the
// bracket.  The closing sums undo the scaling with a, theta_0, a in place of h (closing_fast).  21 operations per stage.
struct FastStep {
  double ha, hr, hd, hk;   // h/a, h/theta_0, h/Dc, (h/a) k'
};

__device__ __forceinline__ FastStep fast_step(const LaneD &L, double h, const Base &b0) {
  FastStep F;
  F.ha = h * L.inv_a; F.hr = h * b0.rth; F.hd = h * L.inv_dc;
  F.hk = F.ha * L.kprime;
  return F;
}

// The UNSCALED derivative at the point (mu, theta) whose series arguments are rho = dtheta/theta_0 and tha = dmu/a
// (the fast path's closing sums), with (v, 1/theta) there: the step's end point, first stage of the next step.
template <bool DAMP>
__device__ __forceinline__ Deriv friction_incr(const Consts &K, const LaneD &L, double theta0, double vl, const Base &b0, double tha,
                                               double rho, GuardD &g, Base &at_point) {
  g.rho = __builtin_fmaxf(g.rho, __builtin_fabsf(hi_as_float(rho)));
  const double dlt = __builtin_fma(-rho, __builtin_fma(rho, L.nhboa, L.boa), tha);  // dmu/a - (b/a)(rho - rho^2/2)
  g.dlt = __builtin_fmaxf(g.dlt, __builtin_fabsf(hi_as_float(dlt)));
  double e = 1.0 / 120.0;
  e = __builtin_fma(e, dlt, 1.0 / 24.0);
  e = __builtin_fma(e, dlt, 1.0 / 6.0);
  e = __builtin_fma(e, dlt, 0.5);
  e = __builtin_fma(e, dlt, 1.0);
  const double v = __builtin_fma(b0.v * dlt, e, b0.v);
  const double q = __builtin_fma(rho, rho, -rho);
  at_point.v = v;
  at_point.rth = __builtin_fma(b0.rth, q, b0.rth);
  const double thd0 = theta0 * L.inv_dc;
  Deriv d;
  d.t = __builtin_fma(-v, __builtin_fma(thd0, rho, thd0), 1.0);   // 1 - v theta_s / Dc,  theta_s = theta_0 (1 + rho)
  d.m = L.kprime * (vl - v);                                       // k' (V_l - v)
  const double gg = __builtin_fma(-L.b * at_point.rth, d.t, d.m);  // k'(V_l - v) - (b/theta_s)(1 - v theta_s/Dc)
  const double va = v * L.inv_a;
  d.v = va * gg;
  if (DAMP) {
    d.m = __builtin_fma(-K.k1, d.v, d.m);
    d.v = va * __builtin_fma(-K.k1, d.v, gg);
  }
  return d;
}

__device__ __forceinline__ bool guard_tripped(const GuardD &g) {  // (NaN reads as a float NaN, which max ignores: a dead lane
  return !(g.rho < hi_pow2(-20) && g.dlt < hi_pow2(-9));          //  ends in err = NaN and fails its call like the reference's)
}

template <bool DAMP>
__device__ __forceinline__ double hinit(const Consts &K, const LaneD &L, double x, const double y[3], const Deriv &f0, double hmax) {
  const double f0a[3] = {f0.m, f0.t, f0.v};
  double dnf = 0.0, dny = 0.0, y1[3], der2 = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double sk = kAtol + kRtol * fabs(y[i]);
    dnf += (f0a[i] / sk) * (f0a[i] / sk);
    dny += (y[i] / sk) * (y[i] / sk);
  }
  double h = (dnf <= 1e-10 || dny <= 1e-10) ? 1.0e-6 : sqrt(dny / dnf) * 0.01;
  h = fmin(h, hmax);
#pragma unroll
  for (int i = 0; i < 3; ++i) y1[i] = y[i] + h * f0a[i];
  Base unused;
  const Deriv f1 = friction<DAMP>(K, L, loading(K, x + h), y1[0], y1[1], unused);
  const double f1a[3] = {f1.m, f1.t, f1.v};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double sk = kAtol + kRtol * fabs(y[i]);
    der2 += ((f1a[i] - f0a[i]) / sk) * ((f1a[i] - f0a[i]) / sk);
  }
  der2 = sqrt(der2) / h;
  const double der12 = fmax(fabs(der2), sqrt(dnf));
  const double h1 = der12 <= 1e-15 ? fmax(1.0e-6, fabs(h) * 1.0e-3) : fm::exp(0.125 * fm::log(0.01 / der12));  // ** (1/8)
  return fmin(fmin(100.0 * fabs(h), h1), hmax);
}

// the V component's share of a step's closing sums, accumulated stage by stage (see the header)
struct VSums {
  double s, e5;          // sum_j B_j k_j and sum_j E5_j k_j over the stages done so far
  double k1, k9, k12;    // the three stage values the 3rd-order estimator subtracts (kBhh)
};

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, I1)
template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (I0 < I1) {
    f(std::integral_constant<int, I0>{});
    static_for<I0 + 1, I1>(f);
  }
}
constexpr int nz_count(int st) {  // non-zero A entries of stage st (0-based, 1..11)
  int n = 0;
  for (int j = 0; j < st; ++j) n += RSF_DP_A[st - 1][j] != 0.0;
  return n;
}
constexpr int nz_col(int st, int n) {  // column of the n-th of them
  for (int j = 0; j < st; ++j)
    if (RSF_DP_A[st - 1][j] != 0.0 && n-- == 0) return j;
  return -1;
}

// The eleven inner stages of one step of size h from (x, y; first-stage derivative km[0], kt[0], vs.k1), every stage
// evaluated in full (`standard`: the loading table applies): the general loop's form — HINIT's first interval, steps
// after a rejection, waves that hold a stiff small-Dc lane.  Straight-line code (eleven unrolled stages): that keeps the
// km / kt arrays in registers.  There is deliberately no "incremental with longer series" variant for the general loop:
// the loop body would then hold two unrolled stage blocks, and the sampler kernel no longer fits two waves per SIMD
// without spilling (measured: 128-364 B of scratch per lane).
template <bool DAMP>
__device__ __forceinline__ void stages_full(const Consts &K, const LaneD &L, const double *tab, bool standard, double x, double h,
                                            const double y[3], double (&km)[12], double (&kt)[12], VSums &vs) {
  vs.s = RSF_DP_B[0] * vs.k1;
  vs.e5 = RSF_DP_E5[0] * vs.k1;
#pragma unroll
  for (int st = 1; st < 12; ++st) {
    double sm = 0.0, sth = 0.0;
#pragma unroll
    for (int j = 0; j < st; ++j)
      if (RSF_DP_A[st - 1][j] != 0.0) {
        sm += RSF_DP_A[st - 1][j] * km[j];
        sth += RSF_DP_A[st - 1][j] * kt[j];
      }
    const double vl = standard ? tab[st] : loading(K, st == 11 ? x + h : x + RSF_DP_C[st] * h);
    Base unused;
    const Deriv d = friction<DAMP>(K, L, vl, y[0] + h * sm, y[1] + h * sth, unused);
    km[st] = d.m;
    kt[st] = d.t;
    if (st >= 5) {  // stages 6..12 (1-based) carry weight in the closing sums
      vs.s = __builtin_fma(RSF_DP_B[st - 4], d.v, vs.s);
      vs.e5 = __builtin_fma(RSF_DP_E5[st - 4], d.v, vs.e5);
    }
    if (st == 8) vs.k9 = d.v;
    if (st == 11) vs.k12 = d.v;
  }
}

// The same eleven stages for the tabulated standard step, evaluated incrementally from (v, 1/theta) at the step's start
// (friction_incr's arithmetic): the steady state of call().  → whether the lane's increments left the series' range
// (its values are then not to be used).
// A stage is a serial chain — the last tableau term, rho, the log1p and expm1 series, v, the derivatives, the damping
// pass: 14 dependent fp64 operations at ~9 cycles each, of which a lone wave fills 4 — and the stages are serial among
// themselves, so written stage after stage a wave ran at half its issue rate and two waves per SIMD kept the pipe 80 %
// busy.  What is independent of a stage's evaluation is the NEXT stage's tableau sum over the derivatives already known:
// here those terms are written between the links of the evaluation's chain (P below), and the sum that waits for the
// stage is one fma per component.  The compiler keeps the order (near the register limit its scheduler falls back on
// source order).  Summation order per stage is unchanged: columns ascending.
// Constants: kStepTable, rows requested two stages ahead of their last-column use (one ahead of the partial sums).
template <bool DAMP>
__device__ __forceinline__ bool stages_fast(const Consts &K, const LaneD &L, const double *tab, double h, const double y[3],
                                            double (&km)[12], double (&kt)[12], const Base &b0, VSums &vs,
                                            double (&closing)[kClosingLen]) {
  GuardD g = {0.0f, 0.0f};
  // the step's eleven loading values in ONE batch of LDS reads, issued before the step's per-step constants are formed
  // and held in registers from there (early in a step the stage arrays are still empty: there is room) — read one by one
  // next to their uses each was a separate exposed LDS latency (five s_waitcnt per step)
  double tv[12];
#pragma unroll
  for (int i = 1; i < 12; ++i) tv[i] = tab[i];
  const ConstRow table = row_of(kStepTable.a[0]);
  double row[kRowLen], nxt[kRowLen], req[kRowLen];
#pragma unroll
  for (int i = 0; i < kRowLen; ++i) {  // the first two rows (three constants) as literals: needed before a load could arrive
    row[i] = kStepTable0.a[0][i];
    nxt[i] = kStepTable0.a[1][i];
  }
  const FastStep F = fast_step(L, h, b0);
  km[0] *= F.ha;  // the first stage's derivative arrives unscaled (Carry)
  kt[0] *= F.hr;
  vs.k1 *= F.ha;
  vs.s = RSF_DP_B[0] * vs.k1;
  vs.e5 = RSF_DP_E5[0] * vs.k1;
  double psm = 0.0, psth = 0.0;  // stage st's sums over the columns before its last
  static_for<1, 12>([&](auto st_) {
    constexpr int st = decltype(st_)::value;
    constexpr int np = nz_count(st) - 1;  // (the last column, st - 1, is non-zero in every row)
    static_assert(nz_col(st, np) == st - 1, "stage st's row ends in column st - 1");
    const double sm = np ? __builtin_fma(row[np], km[st - 1], psm) : row[np] * km[st - 1];
    const double sth = np ? __builtin_fma(row[np], kt[st - 1], psth) : row[np] * kt[st - 1];
    if constexpr (st + 2 <= 11) {
#pragma unroll
      for (int i = 0; i < row_len(st + 2 <= 11 ? st + 2 : 11); ++i) req[i] = table[kRowLen * (st + 1) + i];
    } else if constexpr (st == 11) {
#pragma unroll
      for (int i = 0; i < kClosingLen; ++i) closing[i] = table[kRowLen * 11 + i];
    }
    __builtin_amdgcn_sched_barrier(0);  // (the scheduler would sink the requests to the pin below)
    // the next stage's sums over the columns known by now, one term per call
    constexpr int npn = st < 11 ? nz_count(st < 11 ? st + 1 : 11) - 1 : 0;
    double nsm = 0.0, nsth = 0.0;
    // P(k, x): term k of those sums, placed beside the chain link that produces x — the empty asm ties x and the two
    // running sums to one program point, so the link, this term and nothing else of either chain sit between two ties
    auto P = [&](auto k_, double &link) {
      constexpr int k = decltype(k_)::value;
      if constexpr (k < npn) {
        constexpr int col = nz_col(st < 11 ? st + 1 : 11, k);
        nsm = k ? __builtin_fma(nxt[k], km[col], nsm) : nxt[k] * km[col];
        nsth = k ? __builtin_fma(nxt[k], kt[col], nsth) : nxt[k] * kt[col];
        asm volatile("" : "+v"(link), "+v"(nsm), "+v"(nsth));
      }
    };
    using std::integral_constant;
    constexpr bool want_v = st >= 5;  // stages 6..12 (1-based) carry weight in the closing sums
    // the stage's evaluation (FastStep), its chain written out link by link; the links that have no independent work
    // of their own get the terms
    const double rho = sth;
    g.rho = __builtin_fmaxf(g.rho, __builtin_fabsf(hi_as_float(rho)));
    const double thds = __builtin_fma(F.hd, rho, F.hd);
    const double pb = __builtin_fma(rho, L.nhboa, L.boa);
    const double q = __builtin_fma(rho, rho, -rho);
    const double bq = __builtin_fma(L.boa, q, L.boa);
    double dlt = __builtin_fma(-rho, pb, sm);  // (h/a) dmu - (b/a)(rho - rho^2/2)
    const double kvl = F.hk * tv[st];
    P(integral_constant<int, 0>{}, dlt);
    g.dlt = __builtin_fmaxf(g.dlt, __builtin_fabsf(hi_as_float(dlt)));
    double e = __builtin_fma(1.0 / 120.0, dlt, 1.0 / 24.0);
    const double bd = b0.v * dlt;
    e = __builtin_fma(e, dlt, 1.0 / 6.0);
    P(integral_constant<int, 1>{}, e);
    e = __builtin_fma(e, dlt, 0.5);
    P(integral_constant<int, 2>{}, e);
    e = __builtin_fma(e, dlt, 1.0);
    P(integral_constant<int, 3>{}, e);
    double v = __builtin_fma(bd, e, b0.v);
    P(integral_constant<int, 4>{}, v);
    Deriv d;
    d.t = __builtin_fma(-v, thds, F.hr);   // (h/theta_0)(1 - v theta_s / Dc)
    d.m = __builtin_fma(-F.hk, v, kvl);    // (h/a) k' (V_l - v)
    const double va = v * L.inv_a;
    double gg = __builtin_fma(-bq, d.t, d.m);  // (h/a)(k'(V_l - v) - (b/theta_s)(1 - v theta_s/Dc))
    P(integral_constant<int, 5>{}, gg);
    d.v = va * gg;
    P(integral_constant<int, 6>{}, d.v);
    if (DAMP) {
      d.m = __builtin_fma(-K.k1, d.v, d.m);
      if (want_v) d.v = va * __builtin_fma(-K.k1, d.v, gg);
    }
    P(integral_constant<int, 7>{}, d.m);
    static_assert(npn <= 8, "P is called eight times");
    km[st] = d.m;
    kt[st] = d.t;
    if constexpr (st + 2 <= 11) pin(req, row_len(st + 2 <= 11 ? st + 2 : 11));
    else if constexpr (st == 11) pin(closing, kClosingLen);
    if (want_v) {
      vs.s = __builtin_fma(row[np + 1], d.v, vs.s);
      vs.e5 = __builtin_fma(row[np + 2], d.v, vs.e5);
    }
    if (st == 8) vs.k9 = d.v;
    if (st == 11) vs.k12 = d.v;
#pragma unroll
    for (int i = 0; i < kRowLen; ++i) { row[i] = nxt[i]; nxt[i] = req[i]; }
    psm = nsm;
    psth = nsth;
  });
  return guard_tripped(g);
}

// 8th-order solution k5 and the error estimate of a step of the general loop; returns err, and err ** (1/8) in fac11.
// The solution is formed as the reference forms it; reciprocals, square root and err ** (1/8) to ~1 ulp (rsf_math.h):
// where the controller really chooses step sizes (HINIT's first interval, stiff small-Dc lanes: thousands of steps whose
// sizes feed back into the solution at the level of the tolerance), the step sequence has to be the reference's.
__device__ __forceinline__ double solution_and_error(double h, const double y[3], const double (&km)[12], const double (&kt)[12],
                                                     const VSums &vs, double k5[3], double &fac11) {
  double e5[3] = {0.0, 0.0, vs.e5}, e3[3], s[3] = {0.0, 0.0, vs.s};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s[0] += RSF_DP_B[j] * km[RSF_DP_W_STAGE[j]];
    s[1] += RSF_DP_B[j] * kt[RSF_DP_W_STAGE[j]];
    e5[0] += RSF_DP_E5[j] * km[RSF_DP_W_STAGE[j]];
    e5[1] += RSF_DP_E5[j] * kt[RSF_DP_W_STAGE[j]];
  }
  e3[0] = s[0] - kBhh1 * km[0] - kBhh2 * km[8] - kBhh3 * km[11];
  e3[1] = s[1] - kBhh1 * kt[0] - kBhh2 * kt[8] - kBhh3 * kt[11];
  e3[2] = s[2] - kBhh1 * vs.k1 - kBhh2 * vs.k9 - kBhh3 * vs.k12;
  double err = 0.0, err2 = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    k5[i] = y[i] + h * s[i];
    const double sk = kAtol + kRtol * fmax(fabs(y[i]), fabs(k5[i]));
    const double isk = fm::rcp(sk);
    err2 += (e3[i] * isk) * (e3[i] * isk);
    err += (e5[i] * isk) * (e5[i] * isk);
  }
  double deno = err + 0.01 * err2;
  if (deno <= 0.0) deno = 1.0;
  err = fabs(h) * err * sqrt(fm::rcp(3.0 * deno));
  fac11 = err > 0.0 ? fm::exp(0.125 * fm::log(err)) : (err == 0.0 ? 0.0 : err);  // err ** (1/8); NaN stays NaN
  return err;
}

// The same for the fast path, from the SCALED stage derivatives (FastStep): the sums s carry the factors h/a, h/theta_0,
// h/a, so h s_i is a s_0, theta_0 s_1, a s_2, and with w_i = {a, theta_0, a} / sk_i the norm's |h| cancels:
//     err = |h| sqrt(sum (e5_i/sk_i)^2 / 3) * [sum (e5/sk)^2 / (sum (e5/sk)^2 + 0.01 sum (e3/sk)^2)]^(1/2)   (dop853.f)
//         = E5 / sqrt(3 (E5 + 0.01 E3)),   E5 = sum (e5'_i w_i)^2,  E3 = sum (e3'_i w_i)^2.
// Here the step IS the output interval whatever the controller says, and the carried prediction is only compared with it
// (x + 1.01 h > xend), so the hardware's approximate reciprocal / reciprocal square root / square root (v_rcp_f64,
// v_rsq_f64, v_sqrt_f64: ~1e-7 relative, one instruction each) stand in for dop853.f's divisions, sqrt and pow
// (predicted_step: err ** (1/8) is three square roots).  A prediction that ends the steady state enters the general loop
// as a step size 1e-7 off — a perturbation of the solution ~1e-7 times a local error.  Constants: the closing row of kStepTable, loaded
// during the last stage.  s[0], s[1] are left for the end point's evaluation (they are its series arguments).
__device__ __forceinline__ double closing_fast(const LaneD &L, const double y[3], const double (&km)[12], const double (&kt)[12],
                                               const VSums &vs, const double (&w)[kClosingLen], double k5[3], double (&s)[3]) {
  double e5[3] = {0.0, 0.0, vs.e5}, e3[3];
  s[0] = 0.0; s[1] = 0.0; s[2] = vs.s;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s[0] += w[2 * j] * km[RSF_DP_W_STAGE[j]];
    s[1] += w[2 * j] * kt[RSF_DP_W_STAGE[j]];
    e5[0] += w[2 * j + 1] * km[RSF_DP_W_STAGE[j]];
    e5[1] += w[2 * j + 1] * kt[RSF_DP_W_STAGE[j]];
  }
  e3[0] = s[0] - w[16] * km[0] - w[17] * km[8] - w[18] * km[11];
  e3[1] = s[1] - w[16] * kt[0] - w[17] * kt[8] - w[18] * kt[11];
  e3[2] = s[2] - w[16] * vs.k1 - w[17] * vs.k9 - w[18] * vs.k12;
  const double unscale[3] = {L.a, y[1], L.a};
  double err = 0.0, err2 = 0.0;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    k5[i] = __builtin_fma(unscale[i], s[i], y[i]);
    const double sk = kAtol + kRtol * fmax(fabs(y[i]), fabs(k5[i]));
    const double wi = unscale[i] * __builtin_amdgcn_rcp(sk);
    err2 += (e3[i] * wi) * (e3[i] * wi);
    err += (e5[i] * wi) * (e5[i] * wi);
  }
  double deno = err + 0.01 * err2;
  if (deno <= 0.0) deno = 1.0;
  return err * __builtin_amdgcn_rsq(3.0 * deno);
}

// What a call hands to the next one: the carried step size (0 => HINIT), and the derivative and (v, 1/theta) at (x, y)
// — every call starts by evaluating the RHS at its start point (SciPy does), which is the point and, for the tabulated
// standard step, bit for bit the value at which the previous call's last accepted step ended (first-same-as-last).
struct Carry {
  double hc;
  Deriv kf;
  Base bf;
  bool have_kf;
};

// In the steady state even the derivative at the step's END point — the next step's first stage — is reached
// incrementally from the step's start point (the same series as the inner stages: the whole step's increments are inside
// their range, as in the RK4 path's TIGHT tier).  (v, 1/theta) then pass from step to step by multiplicative updates;
// every kResyncDp-th interval of a chunk the end point is evaluated in full, so rounding cannot accumulate.
constexpr int kResyncDp = 64;

// Whether a lane's carried step size reaches past xend, i.e. its next step is the tabulated standard step h = xend - x
// (dop853.f's `last` on the first pass of its loop, after the too-small-step test).
__device__ __forceinline__ bool takes_standard_step(const Carry &c, double x, double xend) {
  constexpr double uround = 2.3e-16;
  return c.have_kf && (x + 1.01 * c.hc - xend > 0.0) && !(0.1 * fabs(c.hc) <= fabs(x) * uround);
}

// The steady state of every call after the first interval: EVERY lane of the wave takes the tabulated step (the caller has
// tested takes_standard_step for all of them) and all accept it.  Two wave-uniform tests instead of the general loop's six
// per-lane branches (each of which a lone wave sits out for the latency of its compare).  → true: the state is at xend;
// false (wave-uniform): a lane left the series' range or rejected — nothing was touched, the interval goes to call_general.
// `tab` (LDS, 12 values) holds V_l at the stage times of the standard step of this interval; x and xend are the accumulated
// grid times shared by all lanes, so the host can tabulate them bit-exactly (rsf_set_model).
// The step size dop853.f would carry out of an accepted step of size hs with error norm err: hs / max(facc2, min(facc1,
// err^(1/8) / safe)).  The steady state never forms it: all it would be used for there is takes_standard_step of the next
// interval, 1.01 hs / fac > hs, i.e. fac < 1.01, i.e. err < (1.01 safe)^8 — one compare on the error norm instead of
// three v_sqrt_f64, a v_rcp_f64 and four more operations per interval.  It is formed where the steady state ends.
constexpr double kSafe = 0.9, kFacc1 = 1.0 / 0.3, kFacc2 = 1.0 / 6.0;
constexpr double kErrStandard = (1.01 * kSafe) * (1.01 * kSafe) * (1.01 * kSafe) * (1.01 * kSafe) * (1.01 * kSafe) * (1.01 * kSafe) *
                                (1.01 * kSafe) * (1.01 * kSafe);
__device__ __forceinline__ double predicted_step(double hs, double err) {  // (a prediction: the approximate forms of closing_fast)
  const double fac11 = __builtin_amdgcn_sqrt(__builtin_amdgcn_sqrt(__builtin_amdgcn_sqrt(err)));  // 0 stays 0, NaN stays NaN
  return hs * __builtin_amdgcn_rcp(fmax(kFacc2, fmin(kFacc1, fac11 * (1.0 / kSafe))));
}

// On acceptance err / hs return the step's error norm and size (for predicted_step); c.hc is NOT updated.
template <bool DAMP>
__device__ __forceinline__ bool fast_interval(const Consts &K, const LaneD &L, const double *tab, double &x, double xend, double y[3],
                                              Carry &c, bool resync, double &err, double &hs) {  // resync: wave-uniform
  double km[12], kt[12], k5[3], ssum[3], closing[kClosingLen];
  VSums vs;
  const Base b0 = c.bf;  // slip rate and 1/theta at (x, y): the stages of the step are evaluated incrementally from it
  const double h1 = xend - x;
  km[0] = c.kf.m; kt[0] = c.kf.t; vs.k1 = c.kf.v;
  const bool bad = stages_fast<DAMP>(K, L, tab, h1, y, km, kt, b0, vs, closing);
  const double e1 = closing_fast(L, y, km, kt, vs, closing, k5, ssum);
  if (!__all(!bad && e1 <= 1.0)) return false;
  err = e1;
  hs = h1;
  // first-same-as-last, at xend
  bool full = resync;
  if (!full) {
    GuardD g = {0.0f, 0.0f};
    c.kf = friction_incr<DAMP>(K, L, y[1], tab[11], b0, ssum[0], ssum[1], g, c.bf);
    full = __any(guard_tripped(g));
  }
  if (full) c.kf = friction<DAMP>(K, L, tab[11], k5[0], k5[1], c.bf);
#pragma unroll
  for (int i = 0; i < 3; ++i) y[i] = k5[i];
  x = x + h1;
  return true;
}

// one dop853 call (forward in time) through the general loop: y from x to xend.  false on failure.
// HINIT's first interval, steps after a rejection, stiff small-Dc lanes.  The first step of a call whose carried step size
// reaches past xend is the tabulated one (`tab`); any other evaluates V_l(t) directly.
template <bool DAMP>
__device__ __forceinline__ bool call_general(const Consts &K, const LaneD &L, const double *tab, double &x, double xend, double y[3],
                                             Carry &c) {
  constexpr double safe = kSafe, facc1 = kFacc1, facc2 = kFacc2, uround = 2.3e-16;
  const double hmax = fabs(xend - x);
  double km[12], kt[12], k5[3];
  VSums vs;
  double h = c.hc;
  bool last = false, reject = false;
  Base b0 = c.bf;
  Deriv k1 = c.kf;
  if (!c.have_kf) k1 = friction<DAMP>(K, L, tab[0], y[0], y[1], b0);  // V_l(x): x is the interval's start time for every lane
  if (h == 0.0) h = hinit<DAMP>(K, L, x, y, k1, hmax);
  for (int nstep = 0;;) {
    if (nstep > 500) return false;
    if (0.1 * fabs(h) <= fabs(x) * uround) return false;
    if (x + 1.01 * h - xend > 0.0) { h = xend - x; last = true; }
    const bool standard = last && nstep == 0;  // the tabulated step
    ++nstep;
    km[0] = k1.m; kt[0] = k1.t; vs.k1 = k1.v;
    stages_full<DAMP>(K, L, tab, standard, x, h, y, km, kt, vs);
    double fac11;
    const double err = solution_and_error(h, y, km, kt, vs, k5, fac11);
    double hnew = h * fm::rcp(fmax(facc2, fmin(facc1, fac11 * (1.0 / safe))));
    if (err <= 1.0) {
      k1 = friction<DAMP>(K, L, standard ? tab[11] : loading(K, x + h), k5[0], k5[1], b0);  // first-same-as-last, at x + h: full
#pragma unroll
      for (int i = 0; i < 3; ++i) y[i] = k5[i];
      x = x + h;
      if (last) {
        c.hc = hnew;
        c.kf = k1;
        c.bf = b0;
        c.have_kf = true;
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

// one dop853 call: the steady-state step if every lane of the wave takes it, else the general loop
template <bool DAMP>
__device__ __forceinline__ bool call(const Consts &K, const LaneD &L, const double *tab, double &x, double xend, double y[3], Carry &c,
                                     bool resync) {
  double err, hs;
  if (__all(takes_standard_step(c, x, xend)) && fast_interval<DAMP>(K, L, tab, x, xend, y, c, resync, err, hs)) {
    c.hc = predicted_step(hs, err);
    return true;
  }
  return call_general<DAMP>(K, L, tab, x, xend, y, c);
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

__device__ __forceinline__ LaneD make_lane_dp(double dc, double a, double b) {
  LaneD L;
  L.inv_dc = 1.0 / dc; L.kprime = (1e-2 * 10) / dc; L.inv_a = 1.0 / a; L.b = b;
  L.a = a;
  L.boa = b * L.inv_a;
  L.nhboa = -0.5 * L.boa;
  return L;
}

__device__ __forceinline__ Carry fresh_carry() {
  Carry c;
  c.hc = 0.0; c.kf = {0.0, 0.0, 0.0}; c.bf = {0.0, 0.0}; c.have_kf = false;
  return c;
}

// Forward solve in the reference's scheme.  Same calling convention as rsf::solve (all threads call it; the
// loading table and the observation are read from the LDS chunk staged by stage_chunk_dp).
template <bool DAMP, bool WANT_SSQ, bool WANT_ACC>
__device__ __forceinline__ double solve(double *lds, const Consts &K, bool resident, bool active, double dc, double a,
                                        double b, double *acc_out, int64_t stride) {
  const LaneD L = make_lane_dp(dc, a, b);
  const double delta_t = K.dt, inv_dt = K.inv_dt;
  double y[3] = {K.mu0, dc / K.V_ref, K.V_ref};
  double x = K.t0, vprev = K.V_ref, ssq = 0.0;
  Carry c = fresh_carry();
  bool failed = false;
  if (WANT_SSQ && active) { const double d0 = K.data[0]; ssq = d0 * d0; }
  if (WANT_ACC && active) acc_out[0] = 0.0;
  const double *ld = lds + lds_data_offset_dp(K);
  for (int k0 = 1; k0 < K.nout; k0 += K.kc) {
    const int kn = min(K.kc, K.nout - k0);
    if (!resident) stage_chunk_dp(lds, K, k0, kn);
    if (!active) continue;
    auto sample = [&](int kk, double ak, double obs) {
      if (WANT_ACC) acc_out[(int64_t)(k0 + kk) * stride] = ak;
      if (WANT_SSQ) {
        const double r = ak - obs;
        ssq = __builtin_fma(r, r, ssq);
      }
    };
    for (int kk = 0; kk < kn;) {
      // the steady state: a loop of its own that holds nothing but the tabulated step, so that its registers and exec
      // masks are not merged with the general loop's at every interval (that bookkeeping was ~50 instructions per interval).
      // Entered on the carried step size; continued on the error norm alone (kErrStandard); the carried step size is
      // formed when it ends.
      if (__all(!failed && takes_standard_step(c, x, x + delta_t))) {
        double err = -1.0, hs = 0.0;
        bool took;
        do {
          double obs = WANT_SSQ ? ld[kk] : 0.0;  // read ahead of the step: its LDS latency passes under it
          took = fast_interval<DAMP>(K, L, lds + kTab * kk, x, x + delta_t, y, c, (kk & (kResyncDp - 1)) == kResyncDp - 1, err, hs);
          if (!took) break;
          const double ak = (y[2] - vprev) * inv_dt;
          vprev = y[2];
          asm volatile("" : "+v"(obs));
          sample(kk, ak, obs);
          ++kk;
        } while (kk < kn && __all(err < kErrStandard));
        if (err >= 0.0) c.hc = predicted_step(hs, err);
        if (took) continue;  // (the next interval decides again, on the step size just formed)
      }
      if (kk >= kn) break;
      double ak = 0.0;  // after a failed call the reference's arrays keep their zeros (RateStateModel.py:361-381)
      if (!failed) {
        failed = !call_general<DAMP>(K, L, lds + kTab * kk, x, x + delta_t, y, c);
        ak = (y[2] - vprev) * inv_dt;
        vprev = y[2];
      }
      sample(kk, ak, WANT_SSQ ? ld[kk] : 0.0);
      ++kk;
    }
  }
  return ssq;
}

}  // namespace dp
}  // namespace rsf
