// peak_fp64.hip — what the fp64 vector pipe of this MI355X SUSTAINS (diagnostic tool, not product code).
//
// The roofline in bench.py divides by the nominal peak 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz = 78.6 TFLOP/s.
// Under back-to-back v_fma_f64 the power manager does not hold 2.4 GHz; this program measures the rate that a pure,
// perfectly parallel FMA stream reaches (8 independent chains per lane, W waves per SIMD, seconds-long launches) and the
// shader clock it ran at, so that DESIGN.md can quote the kernel against the peak the part actually delivers.
//   hipcc -O3 --offload-arch=gfx950 -o build/peak_fp64 tools/peak_fp64.hip && build/peak_fp64
#include <hip/hip_runtime.h>

#include <cstdio>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ILP = 8, UNROLL = 8;

__global__ void __launch_bounds__(256) fma_stream(double *out, unsigned long long *cyc, int iters, double seed) {
  double r[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) r[i] = seed + i * 1e-3 + threadIdx.x * 1e-6;
  const double a = 1.0000001, b = 1e-9;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u)
#pragma unroll
      for (int i = 0; i < ILP; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  hipDeviceProp_t p;
  CHECK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  printf("%s: %d CUs, nominal clock %.2f GHz\n", p.name, cus, p.clockRate * 1e-6);
  for (int waves_per_simd : {1, 2, 4, 8}) {
    const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD of a CU
    double *out;
    unsigned long long *cyc;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(double)));
    CHECK(hipMalloc(&cyc, (size_t)blocks * sizeof(unsigned long long)));
    const int iters = 8000000 / waves_per_simd;  // ~1 s per launch: long enough for the power manager to settle
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fma_stream, dim3(blocks), dim3(256), 0, 0, out, cyc, iters / 20, 1.0);  // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(fma_stream, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double fmas = (double)blocks * 256 * (double)iters * UNROLL * ILP;
    const double tflops = 2.0 * fmas / (ms * 1e-3) / 1e12;
    // each wave issues iters*UNROLL*ILP instructions; with W waves per SIMD sharing the pipe at 4 cycles per instruction:
    const double cycles_needed = (double)iters * UNROLL * ILP * 4.0 * waves_per_simd;
    printf("waves/SIMD %d: %8.1f ms  %6.2f TFLOP/s fp64 (%.3f of 78.6)  implied shader clock %.3f GHz if the pipe never idles\n",
           waves_per_simd, ms, tflops, tflops / 78.6, cycles_needed / (ms * 1e-3) / 1e9);
    CHECK(hipFree(out));
    CHECK(hipFree(cyc));
  }
  return 0;
}
