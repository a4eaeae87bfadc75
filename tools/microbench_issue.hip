// microbench_issue.hip — issue cost and dependent latency of the instructions of the float32 trip, one wave per SIMD.
// Each variant is an asm block of 8 instructions, repeated; "independent" = eight separate accumulators (issue cost),
// "dependent" = a chain through one (issue to issue).  Cycles are counted by the shader clock itself (clock64 = s_memtime
// around the loop of one wave), so the figures do not depend on what the power manager does to the clock.
//   hipcc -O3 --offload-arch=gfx950 -o build/microbench_issue tools/microbench_issue.hip && build/microbench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float float2v __attribute__((ext_vector_type(2)));

#define R8(s) s s s s s s s s
#define R64(s) R8(R8(s))   // long straight-line bodies: the loop branch (tens of cycles) must not show in a per-instruction figure
template <int VAR>
__global__ void __launch_bounds__(256) k(float2v *out, const float2v *in, int iters, long long *cycles) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float2v a = in[i], b = in[i + 1], c = {0.5f, 0.25f};
  double d0 = a.x, d1 = a.y, d2 = b.x, d3 = b.y, d4 = 1.0, d5 = 2.0, d6 = 3.0, d7 = 4.0, m = 0.999;
  float2v p0 = a, p1 = b, p2 = a * 2, p3 = b * 2, p4 = a * 3, p5 = b * 3, p6 = a * 4, p7 = b * 4;
  float f0 = a.x, f1 = a.y;
  const long long c0 = clock64();
  for (int n = 0; n < iters; ++n) {
    if (VAR == 0)  // independent packed fma
      asm volatile(R8(R8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                   "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"))
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c), "v"(b));
    if (VAR == 1)  // dependent packed fma
      asm volatile(R8(R64("v_pk_fma_f32 %0, %0, %1, %2\n")) : "+v"(p0) : "v"(c), "v"(b));
    if (VAR == 2)  // independent v_fma_f64
      asm volatile(R8(R8("v_fma_f64 %0, %0, %8, %0\n v_fma_f64 %1, %1, %8, %1\n v_fma_f64 %2, %2, %8, %2\n v_fma_f64 %3, %3, %8, %3\n"
                   "v_fma_f64 %4, %4, %8, %4\n v_fma_f64 %5, %5, %8, %5\n v_fma_f64 %6, %6, %8, %6\n v_fma_f64 %7, %7, %8, %7\n"))
                   : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(m));
    if (VAR == 3)  // dependent v_fma_f64
      asm volatile(R8(R64("v_fma_f64 %0, %0, %1, %0\n")) : "+v"(d0) : "v"(m));
    if (VAR == 4)  // independent v_cvt_f64_f32
      asm volatile(R8(R8("v_cvt_f64_f32 %0, %8\n v_cvt_f64_f32 %1, %8\n v_cvt_f64_f32 %2, %8\n v_cvt_f64_f32 %3, %8\n"
                   "v_cvt_f64_f32 %4, %8\n v_cvt_f64_f32 %5, %8\n v_cvt_f64_f32 %6, %8\n v_cvt_f64_f32 %7, %8\n"))
                   : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(f0));
    if (VAR == 5)  // the emission chain: sub → cvt → fma64, twice, then back to float (4 x (sub, cvt) = 8 instructions + 0)
      asm volatile(R8(R8("v_sub_f32 %0, %0, %2\n v_cvt_f64_f32 %1, %0\n v_cvt_f32_f64 %0, %1\n")) : "+v"(f0), "+v"(d0) : "v"(f1));
    if (VAR == 6)  // independent v_sub_f32
      asm volatile(R8(R64("v_sub_f32 %0, %1, %2\n")) : "=v"(f0) : "v"(f1), "v"(a.x));
    if (VAR == 7)  // packed result consumed by a scalar float op and back: pk_mul → v_sub_f32 on its low half → pk_mul ...
      asm volatile(R8(R8("v_pk_mul_f32 %0, %0, %1\n v_sub_f32 %2, %2, %2\n")) : "+v"(p0), "+v"(c), "+v"(f0));
    if (VAR == 8)  // dependent v_fma_f32 (plain)
      asm volatile(R8(R64("v_fma_f32 %0, %0, %1, %2\n")) : "+v"(f0) : "v"(f1), "v"(a.x));
    if (VAR == 9)  // dependent packed fma with the wait state hipcc puts between such a pair
      asm volatile(R8(R64("v_pk_fma_f32 %0, %0, %1, %2\n s_nop 0\n")) : "+v"(p0) : "v"(c), "v"(b));
    if (VAR == 10)  // two interleaved dependent packed chains
      asm volatile(R8(R64("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n")) : "+v"(p0), "+v"(p1) : "v"(c), "v"(b));
    if (VAR == 11)  // three interleaved dependent packed chains
      asm volatile(R8(R64("v_pk_fma_f32 %0, %0, %3, %4\n v_pk_fma_f32 %1, %1, %3, %4\n v_pk_fma_f32 %2, %2, %3, %4\n")) : "+v"(p0), "+v"(p1), "+v"(p2) : "v"(c), "v"(b));
  }
  const long long c1 = clock64();
  if (i == 0) *cycles = c1 - c0;
  out[i] = p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7 + float2v{(float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7), f0};
}

template <int VAR>
double run(float2v *out, float2v *in, int n, int iters, long long *dcyc) {
  long long h = 0;
  for (int rep = 0; rep < 2; ++rep) {
    k<VAR><<<n / 256, 256>>>(out, in, iters, dcyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(&h, dcyc, sizeof(h), hipMemcpyDeviceToHost));
  }
  return (double)h;
}

int main() {
  for (int wps : {1, 2}) {
    const int n = 256 * 4 * 64 * wps;
    float2v *in, *out;
    long long *dcyc;
    CHECK(hipMalloc(&in, (n + 1) * sizeof(float2v))); CHECK(hipMalloc(&out, n * sizeof(float2v))); CHECK(hipMalloc(&dcyc, 8));
    CHECK(hipMemset(in, 0, (n + 1) * sizeof(float2v)));
    const int iters = 2000;
    printf("%d wave(s) per SIMD: shader-clock cycles per instruction of ONE wave\n", wps);
    auto rep = [&](const char *name, double cyc, int per_iter) { printf("  %-62s %6.2f\n", name, cyc / (double)per_iter / iters); };
    rep("v_pk_fma_f32, independent", run<0>(out, in, n, iters, dcyc), 512);
    rep("v_pk_fma_f32, dependent chain", run<1>(out, in, n, iters, dcyc), 512);
    rep("v_fma_f64, independent", run<2>(out, in, n, iters, dcyc), 512);
    rep("v_fma_f64, dependent chain", run<3>(out, in, n, iters, dcyc), 512);
    rep("v_cvt_f64_f32, independent", run<4>(out, in, n, iters, dcyc), 512);
    rep("v_sub_f32 -> v_cvt_f64_f32 -> v_cvt_f32_f64 chain", run<5>(out, in, n, iters, dcyc), 192);
    rep("v_sub_f32, independent", run<6>(out, in, n, iters, dcyc), 512);
    rep("v_pk_mul_f32 chain with an independent v_sub_f32 between each", run<7>(out, in, n, iters, dcyc), 128);
    rep("v_fma_f32, dependent chain", run<8>(out, in, n, iters, dcyc), 512);
    rep("v_pk_fma_f32, dependent chain, s_nop 0 after each (per pair)", run<9>(out, in, n, iters, dcyc), 512);
    rep("v_pk_fma_f32, two interleaved dependent chains", run<10>(out, in, n, iters, dcyc), 1024);
    rep("v_pk_fma_f32, three interleaved dependent chains", run<11>(out, in, n, iters, dcyc), 1536);
    CHECK(hipFree(in)); CHECK(hipFree(out)); CHECK(hipFree(dcyc));
  }
  return 0;
}
