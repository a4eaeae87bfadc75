// microbench_fp64.hip — fp64 VALU issue / latency on gfx950 (diagnostic tool, not product code).
// Each test runs ITERS x UNROLL instructions per wave in ILP independent dependency chains, with
// W waves per SIMD (grid = 1024 SIMDs * W waves), and reports shader cycles per wave-instruction
// (s_memtime) and the wall-clock equivalent.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 60000;

template <int OP, int ILP, int ACTIVE = 64>
__global__ void __launch_bounds__(256) bench(double *out, unsigned long long *cyc, unsigned long long *rt, double seed) {
  if ((threadIdx.x & 63) >= ACTIVE) return;  // leave only the first ACTIVE lanes of each wave64 running
  double r[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) r[i] = seed + i * 1e-3 + threadIdx.x * 1e-6;
  const double a = 1.0000001, b = 1e-9, c3 = 0.5 + seed;
  float gf = 0.0f;
  double r2[ILP];
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0)::"memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int u = 0; u < 16 / ILP; ++u) {
#pragma unroll
      for (int i = 0; i < ILP; ++i) {
        if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
        if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r[i]) : "v"(a));
        if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(r[i]) : "v"(b));
        if (OP == 3) asm volatile("v_rcp_f64 %0, %0" : "+v"(r[i]));
        if (OP == 4) asm volatile("v_ldexp_f64 %0, %0, 0" : "+v"(r[i]));
        if (OP == 5) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(r[i]));
        if (OP == 6) asm volatile("v_rndne_f64 %0, %0" : "+v"(r[i]));
        if (OP == 7) asm volatile("v_mov_b64 %0, %0" : "+v"(r[i]));
        if (OP == 8) { float f = (float)r[i]; asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f)); r[i] = f; }
        if (OP == 9) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
        if (OP == 10) asm volatile("s_nop 0\n\tv_fma_f64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
        if (OP == 12) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "s"(b));        // one SGPR source
        if (OP == 13) asm volatile("v_fma_f64 %0, %0, %1, 1.0" : "+v"(r[i]) : "v"(a));               // inline constant
        if (OP == 14) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(r[i]) : "v"(a), "v"(b), "v"(c3)); // no dependence at all
        if (OP == 15) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(r[i]) : "v"(a), "v"(b));
        if (OP == 16) { float f = __builtin_bit_cast(float, __double2hiint(r[i])); asm volatile("v_max3_f32 %0, %0, |%1|, |%1|" : "+v"(gf) : "v"(f)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b)); }  // fma + max3 on its high word: 2 instrs
        if (OP == 17) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b)); asm volatile("v_mul_f64 %0, %1, %1" : "=v"(r2[i]) : "v"(r[i])); }  // fma then a directly dependent mul: 2 instrs
        if (OP == 18) asm volatile("s_mov_b32 s20, 0x9999999a\n\ts_mov_b32 s21, 0x3fb99999\n\tv_fma_f64 %0, %0, %1, s[20:21]" : "+v"(r[i]) : "v"(a) : "s20", "s21");  // literal pair + fma: 3 instrs, 1 VALU
        if (OP == 19) asm volatile("s_mov_b32 s20, 0x9999999a\n\tv_fma_f64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b) : "s20");  // 1 SALU + fma
        if (OP == 11) { int lo = __double2loint(r[i]); asm volatile("v_cndmask_b32 %0, %0, %0, vcc" : "+v"(lo)); r[i] = __hiloint2double(__double2hiint(r[i]), lo); }
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1)::"memory");
  double s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += r[i] + ((OP == 17) ? r2[i] : 0.0);
  s += gf;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; rt[blockIdx.x] = r1 - r0; }
}

template <int OP, int ILP, int ACTIVE = 64>
int run(const char *name, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd;  // 256 CUs, 256-thread blocks = one wave per SIMD per block
  double *out;
  unsigned long long *cyc, *rt;
  CHECK(hipMalloc(&out, sizeof(double) * blocks * 256));
  CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * blocks));
  CHECK(hipMalloc(&rt, sizeof(unsigned long long) * blocks));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((bench<OP, ILP, ACTIVE>), dim3(blocks), dim3(256), 0, nullptr, out, cyc, rt, 1.0);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((bench<OP, ILP, ACTIVE>), dim3(blocks), dim3(256), 0, nullptr, out, cyc, rt, 1.0);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(blocks);
  CHECK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
  std::vector<unsigned long long> hr(blocks);
  CHECK(hipMemcpy(hr.data(), rt, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
  double mean = 0, mrt = 0;
  for (auto v : h) mean += v;
  for (auto v : hr) mrt += v;
  mean /= blocks;
  mrt /= blocks;
  const double n = (double)ITERS * 16;
  printf("%-14s act=%d ILP=%d waves/SIMD=%d : memtime %6.2f ticks/instr/wave (%5.2f per SIMD) ; in-kernel %.3f ms (realtime@100MHz) wall %.3f ms ; memtime rate %.3f GHz ; %.3f ns per instr per SIMD\n", name, ACTIVE, ILP,
         waves_per_simd, mean / n, mean / n / waves_per_simd, mrt / 1e5, ms, mean / (mrt * 10.0), mrt * 10.0 / n / waves_per_simd);
  (void)hipFree(out);
  (void)hipFree(cyc);
  (void)hipFree(rt);
  return 0;
}

int main(int argc, char **argv) {
  if (argc > 1 && argv[1][0] == '2') {  // two waves per SIMD: what dependence and scalar instructions cost the shared pipe (per-SIMD column; 4.0 = pipe full)
    run<0, 1, 64>("fma dep", 2); run<0, 2, 64>("fma ilp2", 2); run<0, 4, 64>("fma ilp4", 2);
    run<12, 1, 64>("fma dep sgpr", 2); run<12, 2, 64>("fma ilp2 sgpr", 2);
    run<19, 1, 64>("smov+fma dep", 2); run<19, 2, 64>("smov+fma ilp2", 2); run<19, 4, 64>("smov+fma ilp4", 2);
    run<18, 1, 64>("2smov+fma dep", 2); run<18, 2, 64>("2smov+fma ilp2", 2); run<18, 4, 64>("2smov+fma ilp4", 2);
    run<16, 1, 64>("fma+max3hi dep", 2); run<17, 1, 64>("fma+depmul", 2); run<17, 2, 64>("fma+depmul ilp2", 2);
    run<0, 1, 64>("fma dep", 1); run<18, 1, 64>("2smov+fma dep", 1); run<19, 1, 64>("smov+fma dep", 1);
    return 0;
  }
  if (argc > 1) {  // lone-wave issue cost by instruction form (cycles per wave-instruction; OP 16/17 count 2 per slot)
    run<0, 4, 64>("fma v,v,v,v", 1); run<0, 8, 64>("fma v,v,v,v", 1); run<0, 16, 64>("fma v,v,v,v", 1);
    run<12, 4, 64>("fma v,v,v,s", 1); run<12, 8, 64>("fma v,v,v,s", 1);
    run<13, 4, 64>("fma v,v,v,1.0", 1); run<13, 8, 64>("fma v,v,v,1.0", 1);
    run<14, 8, 64>("fma indep", 1); run<15, 8, 64>("mul indep", 1);
    run<1, 4, 64>("mul v,v,v", 1); run<1, 8, 64>("mul v,v,v", 1);
    run<2, 8, 64>("add v,v,v", 1); run<9, 8, 64>("fmac", 1);
    run<16, 4, 64>("fma+max3hi", 1); run<16, 8, 64>("fma+max3hi", 1);
    run<17, 4, 64>("fma+depmul", 1); run<17, 8, 64>("fma+depmul", 1);
    run<0, 8, 64>("fma v,v,v,v", 2); run<14, 8, 64>("fma indep", 2);
    return 0;
  }
  // does the SIMD skip inactive 16/32-lane groups of a wave64 for fp64 ops?
  run<0, 1, 64>("v_fma_f64", 1); run<0, 4, 64>("v_fma_f64", 1);
  run<0, 1, 32>("v_fma_f64", 1); run<0, 4, 32>("v_fma_f64", 1);
  run<0, 1, 16>("v_fma_f64", 1); run<0, 4, 16>("v_fma_f64", 1);
  run<0, 1, 32>("v_fma_f64", 2); run<0, 4, 32>("v_fma_f64", 2);
  run<0, 1, 16>("v_fma_f64", 4); run<0, 4, 16>("v_fma_f64", 4);
  run<0, 4, 64>("v_fma_f64", 2);
  return 0;
}
