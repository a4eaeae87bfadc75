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
  const long long r0 = wall_clock64();
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
    // register banks (bank = index mod 4; a pair at an even index lies in banks {0,1} or {2,3}): independent packed fmas whose
    // three sources and destination are placed by hand
    if (VAR == 12)  // everything in banks {0,1}
      asm volatile(R8(R8("v_pk_fma_f32 v[100:101], v[104:105], v[108:109], v[112:113]\n v_pk_fma_f32 v[116:117], v[120:121], v[124:125], v[128:129]\n"
                         "v_pk_fma_f32 v[132:133], v[104:105], v[124:125], v[112:113]\n v_pk_fma_f32 v[136:137], v[120:121], v[108:109], v[128:129]\n"))
                   ::: "v100", "v101", "v116", "v117", "v132", "v133", "v136", "v137");
    if (VAR == 13)  // sources in {0,1}, {2,3}, {0,1}; destination {2,3}
      asm volatile(R8(R8("v_pk_fma_f32 v[102:103], v[104:105], v[110:111], v[112:113]\n v_pk_fma_f32 v[118:119], v[120:121], v[126:127], v[128:129]\n"
                         "v_pk_fma_f32 v[134:135], v[104:105], v[126:127], v[112:113]\n v_pk_fma_f32 v[138:139], v[120:121], v[110:111], v[128:129]\n"))
                   ::: "v102", "v103", "v118", "v119", "v134", "v135", "v138", "v139");
    if (VAR == 14)  // two sources the SAME pair (x*x+c), third elsewhere
      asm volatile(R8(R8("v_pk_fma_f32 v[100:101], v[104:105], v[104:105], v[110:111]\n v_pk_fma_f32 v[116:117], v[120:121], v[120:121], v[126:127]\n"
                         "v_pk_fma_f32 v[132:133], v[108:109], v[108:109], v[114:115]\n v_pk_fma_f32 v[136:137], v[124:125], v[124:125], v[130:131]\n"))
                   ::: "v100", "v101", "v116", "v117", "v132", "v133", "v136", "v137");
    if (VAR == 15)  // packed mul, both sources in {0,1}
      asm volatile(R8(R8("v_pk_mul_f32 v[100:101], v[104:105], v[108:109]\n v_pk_mul_f32 v[116:117], v[120:121], v[124:125]\n"
                         "v_pk_mul_f32 v[132:133], v[104:105], v[124:125]\n v_pk_mul_f32 v[136:137], v[120:121], v[108:109]\n"))
                   ::: "v100", "v101", "v116", "v117", "v132", "v133", "v136", "v137");
    if (VAR == 16)  // packed fma with one source read by op_sel broadcast and an inline constant
      asm volatile(R8(R8("v_pk_fma_f32 v[100:101], v[104:105], v[110:111], v[112:113] op_sel_hi:[1,1,0]\n v_pk_fma_f32 v[116:117], 2.0, v[126:127], v[128:129] op_sel_hi:[0,1,1]\n"
                         "v_pk_fma_f32 v[132:133], v[104:105], v[126:127], v[112:113] op_sel:[0,0,1]\n v_pk_fma_f32 v[136:137], v[120:121], v[110:111], v[128:129] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"))
                   ::: "v100", "v101", "v116", "v117", "v132", "v133", "v136", "v137");
    // which operand carries the dependency, and how many operands are still in flight
    if (VAR == 17)  // dependent chain through src2 (the addend)
      asm volatile(R8(R64("v_pk_fma_f32 %0, %1, %2, %0\n")) : "+v"(p0) : "v"(c), "v"(b));
    if (VAR == 18)  // dependent chain through src1
      asm volatile(R8(R64("v_pk_fma_f32 %0, %1, %0, %2\n")) : "+v"(p0) : "v"(c), "v"(b));
    if (VAR == 19)  // three values, each instruction reads the results of the two instructions before it (distance 1 and 2)
      asm volatile(R8(R64("v_pk_fma_f32 %0, %1, %2, %3\n v_pk_fma_f32 %1, %2, %0, %3\n v_pk_fma_f32 %2, %0, %1, %3\n")) : "+v"(p0), "+v"(p1), "+v"(p2) : "v"(c));
    if (VAR == 20)  // four values, each instruction reads the results at distance 2 and 3 (none adjacent, two in flight)
      asm volatile(R8(R64("v_pk_fma_f32 %0, %1, %2, %4\n v_pk_fma_f32 %1, %2, %3, %4\n v_pk_fma_f32 %2, %3, %0, %4\n v_pk_fma_f32 %3, %0, %1, %4\n"))
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c));
    if (VAR == 21)  // five values, distance 3 and 4
      asm volatile(R8(R64("v_pk_fma_f32 %0, %1, %2, %5\n v_pk_fma_f32 %1, %2, %3, %5\n v_pk_fma_f32 %2, %3, %4, %5\n v_pk_fma_f32 %3, %4, %0, %5\n v_pk_fma_f32 %4, %0, %1, %5\n"))
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4) : "v"(c));
    if (VAR == 22)  // seven independent packed fmas and one independent v_fma_f64 per eight
      asm volatile(R8(R8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_fma_f64 %10, %10, %11, %10\n"
                   "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"))
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c), "v"(b), "v"(d0), "v"(m));
    if (VAR == 23)  // one value read by the NEXT instruction only as src0 of a packed mul, then unrelated work (distance 1 once per 4)
      asm volatile(R8(R64("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_mul_f32 %1, %0, %4\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"))
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c), "v"(b));
    // where the registers are and how many the wave holds
    if (VAR == 24)  // independent packed fmas on HIGH registers (v200..), 256 registers allocated
      asm volatile(R8(R8("v_pk_fma_f32 v[200:201], v[200:201], v[216:217], v[218:219]\n v_pk_fma_f32 v[202:203], v[202:203], v[216:217], v[218:219]\n"
                         "v_pk_fma_f32 v[204:205], v[204:205], v[216:217], v[218:219]\n v_pk_fma_f32 v[206:207], v[206:207], v[216:217], v[218:219]\n"
                         "v_pk_fma_f32 v[208:209], v[208:209], v[216:217], v[218:219]\n v_pk_fma_f32 v[210:211], v[210:211], v[216:217], v[218:219]\n"
                         "v_pk_fma_f32 v[212:213], v[212:213], v[216:217], v[218:219]\n v_pk_fma_f32 v[214:215], v[214:215], v[216:217], v[218:219]\n"))
                   ::: "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v255");
    if (VAR == 25)  // the same on LOW registers (v100..), 256 registers allocated all the same
      asm volatile(R8(R8("v_pk_fma_f32 v[100:101], v[100:101], v[116:117], v[118:119]\n v_pk_fma_f32 v[102:103], v[102:103], v[116:117], v[118:119]\n"
                         "v_pk_fma_f32 v[104:105], v[104:105], v[116:117], v[118:119]\n v_pk_fma_f32 v[106:107], v[106:107], v[116:117], v[118:119]\n"
                         "v_pk_fma_f32 v[108:109], v[108:109], v[116:117], v[118:119]\n v_pk_fma_f32 v[110:111], v[110:111], v[116:117], v[118:119]\n"
                         "v_pk_fma_f32 v[112:113], v[112:113], v[116:117], v[118:119]\n v_pk_fma_f32 v[114:115], v[114:115], v[116:117], v[118:119]\n"))
                   ::: "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v255");
    if (VAR == 26)  // the same on low registers, 128 registers allocated
      asm volatile(R8(R8("v_pk_fma_f32 v[100:101], v[100:101], v[116:117], v[118:119]\n v_pk_fma_f32 v[102:103], v[102:103], v[116:117], v[118:119]\n"
                         "v_pk_fma_f32 v[104:105], v[104:105], v[116:117], v[118:119]\n v_pk_fma_f32 v[106:107], v[106:107], v[116:117], v[118:119]\n"
                         "v_pk_fma_f32 v[108:109], v[108:109], v[116:117], v[118:119]\n v_pk_fma_f32 v[110:111], v[110:111], v[116:117], v[118:119]\n"
                         "v_pk_fma_f32 v[112:113], v[112:113], v[116:117], v[118:119]\n v_pk_fma_f32 v[114:115], v[114:115], v[116:117], v[118:119]\n"))
                   ::: "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v127");
    if (VAR == 27)  // 1024 independent packed fmas per loop iteration (8 KB of code)
      asm volatile(R8(R8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                   "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                   "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                   "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"))
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c), "v"(b));
    // opcode mix (all independent)
    if (VAR == 28)  // v_pk_fma_f32 and v_pk_mul_f32 alternating
      asm volatile(R8(R8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_mul_f32 %1, %1, %8\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_mul_f32 %3, %3, %8\n"
                   "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_mul_f32 %5, %5, %8\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_mul_f32 %7, %7, %8\n"))
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c), "v"(b));
    if (VAR == 29)  // v_pk_fma_f32 and v_pk_add_f32 alternating
      asm volatile(R8(R8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_add_f32 %1, %1, %8\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_add_f32 %3, %3, %8\n"
                   "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_add_f32 %5, %5, %8\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_add_f32 %7, %7, %8\n"))
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c), "v"(b));
    if (VAR == 30)  // three fmas, one mul
      asm volatile(R8(R8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_mul_f32 %3, %3, %8\n"
                   "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_mul_f32 %7, %7, %8\n"))
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c), "v"(b));
    if (VAR == 31)  // v_pk_mul_f32 only
      asm volatile(R8(R8("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                   "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"))
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c), "v"(b));
    if (VAR == 32)  // packed multiply written as an fma with a zero addend (one opcode for everything), alternating with fmas
      asm volatile(R8(R8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, 0 op_sel_hi:[1,1,0]\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, 0 op_sel_hi:[1,1,0]\n"
                   "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, 1.0, %8 op_sel_hi:[1,0,1]\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, 1.0, %8 op_sel_hi:[1,0,1]\n"))
                   : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(c), "v"(b));
    if (VAR == 10)  // two interleaved dependent packed chains
      asm volatile(R8(R64("v_pk_fma_f32 %0, %0, %2, %3\n v_pk_fma_f32 %1, %1, %2, %3\n")) : "+v"(p0), "+v"(p1) : "v"(c), "v"(b));
    if (VAR == 11)  // three interleaved dependent packed chains
      asm volatile(R8(R64("v_pk_fma_f32 %0, %0, %3, %4\n v_pk_fma_f32 %1, %1, %3, %4\n v_pk_fma_f32 %2, %2, %3, %4\n")) : "+v"(p0), "+v"(p1), "+v"(p2) : "v"(c), "v"(b));
  }
  const long long c1 = clock64();
  const long long r1 = wall_clock64();
  if (i == 0) { cycles[0] = c1 - c0; cycles[1] = r1 - r0; }
  out[i] = p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7 + float2v{(float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7), f0};
}

static double g_ratio = 0;  // clock64 ticks per wall_clock64 tick (100 MHz) of the last run
template <int VAR>
double run(float2v *out, float2v *in, int n, int iters, long long *dcyc) {
  long long h[2] = {0, 0};
  for (int rep = 0; rep < 2; ++rep) {
    k<VAR><<<n / 256, 256>>>(out, in, iters, dcyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(h, dcyc, sizeof(h), hipMemcpyDeviceToHost));
  }
  g_ratio = (double)h[0] / (double)h[1];
  return (double)h[0];
}

int main() {
  for (int wps : {1, 2}) {
    const int n = 256 * 4 * 64 * wps;
    float2v *in, *out;
    long long *dcyc;
    CHECK(hipMalloc(&in, (n + 1) * sizeof(float2v))); CHECK(hipMalloc(&out, n * sizeof(float2v))); CHECK(hipMalloc(&dcyc, 16));
    CHECK(hipMemset(in, 0, (n + 1) * sizeof(float2v)));
    const int iters = 2000;
    printf("%d wave(s) per SIMD: shader-clock cycles per instruction of ONE wave\n", wps);
    auto rep = [&](const char *name, double cyc, int per_iter) { printf("  %-62s %6.2f   (clock64 at %.0f MHz)\n", name, cyc / (double)per_iter / iters, 100.0 * g_ratio); };
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
    rep("v_pk_fma_f32, dependent chain through src2", run<17>(out, in, n, iters, dcyc), 512);
    rep("v_pk_fma_f32, dependent chain through src1", run<18>(out, in, n, iters, dcyc), 512);
    rep("v_pk_fma_f32, reads results at distance 1 and 2", run<19>(out, in, n, iters, dcyc), 1536);
    rep("v_pk_fma_f32, reads results at distance 2 and 3", run<20>(out, in, n, iters, dcyc), 2048);
    rep("v_pk_fma_f32, reads results at distance 3 and 4", run<21>(out, in, n, iters, dcyc), 2560);
    rep("7 independent v_pk_fma_f32 + 1 independent v_fma_f64", run<22>(out, in, n, iters, dcyc), 512);
    rep("one adjacent dependent pair per four instructions", run<23>(out, in, n, iters, dcyc), 2048);
    rep("independent v_pk_fma_f32 on v200.., 256 registers allocated", run<24>(out, in, n, iters, dcyc), 512);
    rep("independent v_pk_fma_f32 on v100.., 256 registers allocated", run<25>(out, in, n, iters, dcyc), 512);
    rep("independent v_pk_fma_f32 on v100.., 128 registers allocated", run<26>(out, in, n, iters, dcyc), 512);
    rep("independent v_pk_fma_f32, 1024 per loop iteration (8 KB)", run<27>(out, in, n, iters, dcyc), 1024);
    rep("v_pk_fma_f32 / v_pk_mul_f32 alternating, independent", run<28>(out, in, n, iters, dcyc), 512);
    rep("v_pk_fma_f32 / v_pk_add_f32 alternating, independent", run<29>(out, in, n, iters, dcyc), 512);
    rep("three v_pk_fma_f32, one v_pk_mul_f32, independent", run<30>(out, in, n, iters, dcyc), 512);
    rep("v_pk_mul_f32 only, independent", run<31>(out, in, n, iters, dcyc), 512);
    rep("fma / fma-with-constant-operand (mul, add as fma) alternating", run<32>(out, in, n, iters, dcyc), 512);
    rep("v_pk_fma_f32, independent, all operands in banks {0,1}", run<12>(out, in, n, iters, dcyc), 256);
    rep("v_pk_fma_f32, independent, sources {0,1} {2,3} {0,1}", run<13>(out, in, n, iters, dcyc), 256);
    rep("v_pk_fma_f32, independent, x*x+c (one pair read twice)", run<14>(out, in, n, iters, dcyc), 256);
    rep("v_pk_mul_f32, independent, both sources in banks {0,1}", run<15>(out, in, n, iters, dcyc), 256);
    rep("v_pk_fma_f32, independent, op_sel / inline constant / neg forms", run<16>(out, in, n, iters, dcyc), 256);
    CHECK(hipFree(in)); CHECK(hipFree(out)); CHECK(hipFree(dcyc));
  }
  return 0;
}
