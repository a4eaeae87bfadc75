// microbench_pk_f32.hip — does a DEPENDENT chain of v_pk_fma_f32 need the wait state hipcc puts between a packed-f32
// result and the next instruction that reads it?
//
// hipcc (ROCm 7.2) emits `s_nop 0` after a v_pk_*_f32 whose result the very next instruction consumes (its hazard
// recogniser treats op_sel_hi of source 0 — set by default on packed f32 — like a destination op_sel).  The RK4 step of
// the float32 solve is one long dependent chain, so that is one issue slot in five.  Kernel 0 is the compiler's form
// (with the nops; one asm statement per instruction, kernel 1, gets them too: the recogniser treats an asm result alike),
// kernel 2 the same chain as ONE asm block with nothing between the dependent instructions.  If the results are bit-identical over many
// chains and lengths, the hardware needs no wait state there; the DIFFERENCE of the times gives what the nops cost (4 cycles each).
// (The absolute "cycles per packed op" this program prints include the loop branch of an 8-instruction body — they are not
// instruction costs; tools/microbench_issue.hip measures those with long straight-line bodies and the shader clock.)
//
//   hipcc -O3 --offload-arch=gfx950 -o build/microbench_pk_f32 tools/microbench_pk_f32.hip && build/microbench_pk_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ float2v fma_c(float2v a, float2v b, float2v c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float2v fma_a(float2v a, float2v b, float2v c) {
  float2v d;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ float2v mul_a(float2v a, float2v b) {
  float2v d;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

// the whole chain of one iteration as ONE asm block: nothing between the dependent instructions
__device__ __forceinline__ float2v chain_block(float2v x, float2v r, float2v one, float2v h, float2v m) {
  float2v t, u;
  asm("v_pk_fma_f32 %1, %6, %0, %4\n\t"   // t = m x + 1
      "v_pk_mul_f32 %1, %1, %0\n\t"       // t = t x
      "v_pk_mul_f32 %1, %1, %3\n\t"       // t = t r
      "v_pk_fma_f32 %2, %1, %5, %5\n\t"   // u = t h + h
      "v_pk_fma_f32 %2, %2, %1, %1\n\t"   // u = u t + t
      "v_pk_mul_f32 %2, %2, %5\n\t"       // u = u h
      "v_pk_fma_f32 %2, %2, %6, %4\n\t"   // u = u m + 1
      "v_pk_fma_f32 %0, %2, %5, %5"         // x = u h + h
      : "+v"(x), "=&v"(t), "=&v"(u)
      : "v"(r), "v"(one), "v"(h), "v"(m));
  return x;
}

// x <- a chain of 8 dependent packed operations per iteration (a logistic-like map, bounded, sensitive to every bit)
template <int ASM>
__global__ void __launch_bounds__(256) chain(float2v *out, const float2v *in, int iters) {
#pragma clang fp contract(off)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float2v x = in[i];
  const float2v r = {3.7f, 3.9f}, one = {1.0f, 1.0f}, h = {0.5f, 0.25f}, m = {-1.0f, -1.0f};
  for (int n = 0; n < iters; ++n) {
    if (ASM == 2) {
      x = chain_block(x, r, one, h, m);
    } else if (ASM == 1) {
      float2v t = fma_a(m, x, one);      // 1 - x
      t = mul_a(t, x);                   // x (1 - x)
      t = mul_a(t, r);                   // r x (1 - x)
      float2v u = fma_a(t, h, h);        // mix
      u = fma_a(u, t, t);
      u = mul_a(u, h);
      u = fma_a(u, m, one);
      x = fma_a(u, h, h);
    } else {
      float2v t = fma_c(m, x, one);
      t = t * x;
      t = t * r;
      float2v u = fma_c(t, h, h);
      u = fma_c(u, t, t);
      u = u * h;
      u = fma_c(u, m, one);
      x = fma_c(u, h, h);
    }
  }
  out[i] = x;
}

int main() {
  const int waves_per_simd[] = {1, 2, 4};
  for (int wps : waves_per_simd) {
    const int n = 256 * 4 * 64 * wps;  // 256 CUs x 4 SIMDs x 64 lanes
    std::vector<float2v> h(n);
    srand(1);
    for (auto &v : h) { v.x = 0.1f + 0.8f * (rand() / (float)RAND_MAX); v.y = 0.1f + 0.8f * (rand() / (float)RAND_MAX); }
    float2v *in, *oa, *ob;
    CHECK(hipMalloc(&in, n * sizeof(float2v))); CHECK(hipMalloc(&oa, n * sizeof(float2v))); CHECK(hipMalloc(&ob, n * sizeof(float2v)));
    CHECK(hipMemcpy(in, h.data(), n * sizeof(float2v), hipMemcpyHostToDevice));
    long mism = 0;
    for (int iters : {1, 7, 100, 20000}) {
      chain<0><<<n / 256, 256>>>(oa, in, iters);
      chain<2><<<n / 256, 256>>>(ob, in, iters);
      CHECK(hipDeviceSynchronize());
      std::vector<float2v> a(n), b(n);
      CHECK(hipMemcpy(a.data(), oa, n * sizeof(float2v), hipMemcpyDeviceToHost));
      CHECK(hipMemcpy(b.data(), ob, n * sizeof(float2v), hipMemcpyDeviceToHost));
      mism += memcmp(a.data(), b.data(), n * sizeof(float2v)) != 0;
    }
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 200000;
    float ms[2];
    for (int k = 0; k < 2; ++k) {
      for (int rep = 0; rep < 2; ++rep) {  // second repetition is the one kept
        CHECK(hipEventRecord(e0));
        if (k == 0) chain<0><<<n / 256, 256>>>(oa, in, iters); else chain<2><<<n / 256, 256>>>(ob, in, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms[k], e0, e1));
      }
    }
    printf("%d wave(s)/SIMD: results %s; compiler form (wait states) %.2f ms, one asm block (none) %.2f ms (x%.3f); %.2f / %.2f cycles per packed op at 2.4 GHz\n", wps,
           mism ? "DIFFER" : "bit-identical", ms[0], ms[1], ms[0] / ms[1], ms[0] * 1e-3 * 2.4e9 / (8.0 * iters) / wps, ms[1] * 1e-3 * 2.4e9 / (8.0 * iters) / wps);
    CHECK(hipFree(in)); CHECK(hipFree(oa)); CHECK(hipFree(ob));
  }
  return 0;
}
