// proto_f32x2.hip — PROTOTYPE / measurement only (not product code, not linked into librsf_hip.so).
//
// Question for the next round: what does the float32 solve gain from (a) the incremental stage evaluation of the
// float64 hot loop instead of three quarter-rate transcendentals per stage and (b) packed math (v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32 process two floats per lane-instruction) with TWO chains per lane?
//
// The kernel integrates 2 chains per lane with classical RK4 (radiation damping on, loading table and observation in
// LDS as floats), float2 state, incremental (w, 1/x) from the step's start point with short series
// (log1p to rho^2/2, expm1 to dlt^3/6 — ample for float32), a full re-evaluation every 16 steps with the hardware
// exp2/log2/rcp, and the running sum of squares in float64.  It prints RK4 steps x chains / s and the SSq of a few
// chains next to a plain float64 RK4 on the host.
//
//   hipcc -O3 --offload-arch=gfx950 -o build/proto_f32x2 tools/proto_f32x2.hip && ./build/proto_f32x2 [chains] [nsteps] [reps]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float x) { return (f2){x, x}; }

struct LaneC {  // per-lane constants of two chains
  f2 kia2, tc2, boa, vk, vb, c3, k1k, khh, kh, kh6, hhd, hd, h6d;
};

__device__ __forceinline__ void eval_full(f2 ms, f2 x, const LaneC &L, f2 &w, f2 &rx) {
  f2 lg, ex;
  lg.x = __builtin_amdgcn_logf(x.x); lg.y = __builtin_amdgcn_logf(x.y);
  const f2 arg = fma2(-L.boa, lg, fma2(ms, L.kia2, L.tc2));
  ex.x = __builtin_amdgcn_exp2f(arg.x); ex.y = __builtin_amdgcn_exp2f(arg.y);
  w = ex;
  rx.x = __builtin_amdgcn_rcpf(x.x); rx.y = __builtin_amdgcn_rcpf(x.y);
}

// RHS with w arriving last (rsf_device.h rhs_fast): t1 = vk*vl - vb/x formed by the caller
__device__ __forceinline__ void rhs(f2 w, f2 x, float vl, f2 t1, const LaneC &L, float vref, f2 &d0, f2 &d1, f2 &d2) {
  d1 = fma2(-w, x, splat(1.0f));
  d0 = fma2(splat(-vref), w, splat(vl));
  f2 in = fma2(L.c3, w, t1);
  const f2 kw = L.k1k * w;
  d0 = fma2(-kw, in, d0);
  in = fma2(-(L.vk * kw), in, in);
  d2 = w * in;
}

// (w', q) at a stage from the step's start values; q = rho^2 - rho stands in for 1/x' (stage_t1 of rsf_device.h)
__device__ __forceinline__ void incr(f2 kf, f2 d0, f2 R, f2 d1, const LaneC &L, f2 w0, f2 &w, f2 &q) {
  const f2 rho = d1 * R;
  const f2 p = fma2(rho, splat(-0.5f), splat(1.0f));
  const f2 dlt = fma2(-(L.boa * rho), p, kf * d0);
  f2 e = fma2(splat(1.0f / 6.0f), dlt, splat(0.5f));
  e = fma2(e, dlt, splat(1.0f));
  w = fma2(w0 * dlt, e, w0);
  q = fma2(rho, rho, -rho);
}

__global__ void __launch_bounds__(256) solve(int nlanes, int nout, const float *vl_g, const float *obs_g, const double *dc_g,
                                             double a, double b, double h, double vref, double k1, double mu_ref, double mu0,
                                             double *ssq_out) {
  extern __shared__ float lds[];
  float *vl = lds, *obs = lds + (2 * (nout - 1) + 1);
  for (int i = threadIdx.x; i < 2 * (nout - 1) + 1; i += blockDim.x) vl[i] = vl_g[i];
  for (int i = threadIdx.x; i < nout; i += blockDim.x) obs[i] = obs_g[i];
  __syncthreads();
  const int lane = blockIdx.x * blockDim.x + threadIdx.x;
  if (lane >= nlanes) return;
  const double dc0 = dc_g[2 * lane], dc1 = dc_g[2 * lane + 1];
  const double log2e = 1.4426950408889634;
  LaneC L;
  auto mk = [&](double dc, int c, f2 LaneC::*m, double v) { (void)dc; (L.*m)[c] = (float)v; };
  double ms0[2];
  for (int c = 0; c < 2; ++c) {
    const double dc = c ? dc1 : dc0, kp = 0.1 / dc, kia = kp / a, via = vref / a, vdc = vref / dc, bdc = b * vdc;
    mk(dc, c, &LaneC::kia2, kia * log2e); mk(dc, c, &LaneC::tc2, -mu_ref / a * log2e); mk(dc, c, &LaneC::boa, b / a);
    mk(dc, c, &LaneC::vk, via * kp); mk(dc, c, &LaneC::vb, via * bdc); mk(dc, c, &LaneC::c3, via * bdc - via * kp * vref);
    mk(dc, c, &LaneC::k1k, k1 / kp); mk(dc, c, &LaneC::khh, kia * 0.5 * h); mk(dc, c, &LaneC::kh, kia * h);
    mk(dc, c, &LaneC::kh6, kia * h / 6.0); mk(dc, c, &LaneC::hhd, 0.5 * h * vdc); mk(dc, c, &LaneC::hd, h * vdc);
    mk(dc, c, &LaneC::h6d, h / 6.0 * vdc);
    ms0[c] = mu0 / kp;
  }
  f2 ms = {(float)ms0[0], (float)ms0[1]}, x = splat(1.0f), w, rx;
  eval_full(ms, x, L, w, rx);
  const float vr = (float)vref, h6 = (float)(h / 6.0), cacc = (float)(1.0 / 6.0);
  double ssq0 = (double)obs[0] * obs[0], ssq1 = ssq0;
  for (int k = 0; k < nout - 1; ++k) {
    if ((k & 15) == 0) eval_full(ms, x, L, w, rx);
    const float v0 = vl[2 * k], vm = vl[2 * k + 1], v1 = vl[2 * k + 2];
    const f2 Rh = L.hhd * rx, Rf = Rh + Rh, R6 = L.h6d * rx, vbr0 = L.vb * rx;
    f2 a0, a1, a2, b0, b1, b2, c0, c1, c2, e0, e1, e2, ws, q, xs;
    rhs(w, x, v0, fma2(L.vk, splat(v0), -vbr0), L, vr, a0, a1, a2);
    xs = fma2(L.hhd, a1, x);
    incr(L.khh, a0, Rh, a1, L, w, ws, q);
    rhs(ws, xs, vm, fma2(-vbr0, q, fma2(L.vk, splat(vm), -vbr0)), L, vr, b0, b1, b2);
    xs = fma2(L.hhd, b1, x);
    incr(L.khh, b0, Rh, b1, L, w, ws, q);
    rhs(ws, xs, vm, fma2(-vbr0, q, fma2(L.vk, splat(vm), -vbr0)), L, vr, c0, c1, c2);
    xs = fma2(L.hd, c1, x);
    incr(L.kh, c0, Rf, c1, L, w, ws, q);
    rhs(ws, xs, v1, fma2(-vbr0, q, fma2(L.vk, splat(v1), -vbr0)), L, vr, e0, e1, e2);
    const f2 t0 = a0 + splat(2.0f) * b0 + splat(2.0f) * c0 + e0, t1 = a1 + splat(2.0f) * b1 + splat(2.0f) * c1 + e1;
    const f2 dv = a2 + splat(2.0f) * b2 + splat(2.0f) * c2 + e2;
    incr(L.kh6, t0, R6, t1, L, w, ws, q);
    ms = fma2(splat(h6), t0, ms);
    x = fma2(L.h6d, t1, x);
    rx = fma2(rx, q, rx);
    w = ws;
    const f2 r = fma2(dv, splat(cacc), splat(-obs[k + 1]));
    ssq0 = __builtin_fma((double)r.x, (double)r.x, ssq0);
    ssq1 = __builtin_fma((double)r.y, (double)r.y, ssq1);
  }
  ssq_out[2 * lane] = ssq0;
  ssq_out[2 * lane + 1] = ssq1;
}

// plain float64 RK4 on the host (the literal RHS), for the sanity column
static double host_ssq(double dc, double a, double b, int nout, double h, const std::vector<double> &vl, const std::vector<double> &obs) {
  const double vref = 1.0, mu_ref = 0.6, k1 = 1e-7, kp = 0.1 / dc;
  double y[3] = {mu_ref, dc / vref, vref}, ssq = obs[0] * obs[0];
  auto f = [&](double vlt, const double *yy, double *d) {
    const double v = vref * std::exp((yy[0] - mu_ref - b * std::log(vref * yy[1] / dc)) / a);
    d[1] = 1.0 - v * yy[1] / dc; d[0] = kp * vlt - kp * v; d[2] = v / a * (d[0] - b / yy[1] * d[1]);
    d[0] -= k1 * d[2]; d[2] = v / a * (d[0] - b / yy[1] * d[1]);
  };
  for (int k = 0; k < nout - 1; ++k) {
    double k1_[3], k2[3], k3[3], k4[3], t[3];
    f(vl[2 * k], y, k1_);
    for (int i = 0; i < 3; ++i) t[i] = y[i] + 0.5 * h * k1_[i];
    f(vl[2 * k + 1], t, k2);
    for (int i = 0; i < 3; ++i) t[i] = y[i] + 0.5 * h * k2[i];
    f(vl[2 * k + 1], t, k3);
    for (int i = 0; i < 3; ++i) t[i] = y[i] + h * k3[i];
    f(vl[2 * k + 2], t, k4);
    const double vprev = y[2];
    for (int i = 0; i < 3; ++i) y[i] += h / 6.0 * (k1_[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
    const double r = (y[2] - vprev) / h - obs[k + 1];
    ssq += r * r;
  }
  return ssq;
}

int main(int argc, char **argv) {
  const int chains = argc > 1 ? atoi(argv[1]) : 131072, nsteps = argc > 2 ? atoi(argv[2]) : 500, reps = argc > 3 ? atoi(argv[3]) : 20;
  const int nout = nsteps, nlanes = chains / 2;
  const double h = 50.0 / nsteps, a = 0.011, b = 0.014;
  std::vector<double> vl(2 * (nout - 1) + 1), obs(nout, 0.0), dc(chains);
  for (size_t j = 0; j < vl.size(); ++j) { const double t = 0.5 * h * j; vl[j] = 1.0 + std::exp(-t / 20) * std::sin(10 * t); }
  for (int i = 0; i < chains; ++i) dc[i] = 900.0 + 200.0 * (i % 1024) / 1024.0;
  {  // observation: the host solve at Dc = 1000 (clean)
    std::vector<double> zero(nout, 0.0);
    const double kp = 0.1 / 1000.0; (void)kp;
    double y[3] = {0.6, 1000.0, 1.0};
    for (int k = 0; k < nout - 1; ++k) {
      auto f = [&](double vlt, const double *yy, double *d) {
        const double v = std::exp((yy[0] - 0.6 - b * std::log(yy[1] / 1000.0)) / a);
        d[1] = 1.0 - v * yy[1] / 1000.0; d[0] = 1e-4 * (vlt - v); d[2] = v / a * (d[0] - b / yy[1] * d[1]);
        d[0] -= 1e-7 * d[2]; d[2] = v / a * (d[0] - b / yy[1] * d[1]);
      };
      double k1_[3], k2[3], k3[3], k4[3], t[3];
      f(vl[2 * k], y, k1_);
      for (int i = 0; i < 3; ++i) t[i] = y[i] + 0.5 * h * k1_[i];
      f(vl[2 * k + 1], t, k2);
      for (int i = 0; i < 3; ++i) t[i] = y[i] + 0.5 * h * k2[i];
      f(vl[2 * k + 1], t, k3);
      for (int i = 0; i < 3; ++i) t[i] = y[i] + h * k3[i];
      f(vl[2 * k + 2], t, k4);
      const double vprev = y[2];
      for (int i = 0; i < 3; ++i) y[i] += h / 6.0 * (k1_[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
      obs[k + 1] = (y[2] - vprev) / h * 1.05;
    }
  }
  std::vector<float> vlf(vl.begin(), vl.end()), obsf(obs.begin(), obs.end());
  float *dvl, *dobs;
  double *ddc, *dssq;
  CHECK(hipMalloc(&dvl, vlf.size() * 4)); CHECK(hipMalloc(&dobs, obsf.size() * 4));
  CHECK(hipMalloc(&ddc, chains * 8)); CHECK(hipMalloc(&dssq, chains * 8));
  CHECK(hipMemcpy(dvl, vlf.data(), vlf.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dobs, obsf.data(), obsf.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(ddc, dc.data(), chains * 8, hipMemcpyHostToDevice));
  const size_t lds = (vlf.size() + obsf.size()) * 4;
  const dim3 grid((nlanes + 255) / 256), block(256);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w)
    hipLaunchKernelGGL(solve, grid, block, lds, nullptr, nlanes, nout, dvl, dobs, ddc, a, b, h, 1.0, 1e-7, 0.6, 0.6, dssq);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(solve, grid, block, lds, nullptr, nlanes, nout, dvl, dobs, ddc, a, b, h, 1.0, 1e-7, 0.6, 0.6, dssq);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<double> ssq(chains);
  CHECK(hipMemcpy(ssq.data(), dssq, chains * 8, hipMemcpyDeviceToHost));
  printf("chains %d (2 per lane, %d waves) nsteps %d: %.3f ms per solve of all chains, %.4e RK4 steps*chains/s\n", chains, (nlanes + 63) / 64,
         nsteps, ms / reps, (double)chains * (nout - 1) / (ms / reps * 1e-3));
  for (int i : {0, 1, 511, 1023}) {
    const double ref = host_ssq(dc[i], a, b, nout, h, vl, obs);
    printf("  chain %4d Dc %.2f: SSq f32x2 %.9e   float64 host %.9e   rel diff %.2e\n", i, dc[i], ssq[i], ref, std::fabs(ssq[i] - ref) / ref);
  }
  return 0;
}
