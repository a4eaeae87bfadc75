// microbench_trip.hip — the generated float32 trip (csrc/rsf_f32_trip.inc) alone: shader-clock cycles per trip of one wave,
// nothing around it but a loop.  Separates the cost of the scheduled assembly from the cost of the C++ between trips.
//   hipcc -O3 --offload-arch=gfx950 -I bayesian-markov-chain-monte-carlo_amd/csrc -o build/microbench_trip tools/microbench_trip.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "rsf_f32_trip.inc"
#include "f32_trip_bench.inc"  // python tools/gen_f32_trip.py --bench build/f32_trip_bench.inc   (bodies begin on an 8-byte boundary)
// the same bodies begun on a 4-byte boundary: RSF_TRIP_ALIGN=$'.p2align 3\\n\\ts_nop 0' python tools/gen_f32_trip.py --bench build/f32_trip_bench_mis.inc
#include "f32_trip_bench_mis.inc"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef float float2v __attribute__((ext_vector_type(2)));
struct L32 { float2v hhd, hd, khh, kh, kh6, boa, nhboa, kvk, bh, cv, vref, h6; };

template <int VAR>
__global__ void __launch_bounds__(256, 2) __attribute__((amdgpu_num_vgpr(RSF_F32_TRIP_COMPILER_VGPRS / 2))) kv(double *out, int trips, long long *cycles) {
  // (the private file holds whatever it holds: only the time is looked at)
  asm volatile("" ::: "v255");
  const long long r0 = wall_clock64();
  const long long c0 = clock64();
  for (int n = 0; n < trips; ++n) {
    if (VAR == 0) RSF_F32_TRIP_BENCH_ORIGINAL();
    if (VAR == 1) RSF_F32_TRIP_BENCH_FIXED_SOURCES();
    if (VAR == 2) RSF_F32_TRIP_BENCH_ROTATING_DESTINATIONS();
    if (VAR == 3) RSF_F32_TRIP_BENCH_SHUFFLED();
    if (VAR == 4) RSF_F32_TRIP_BENCH_ORIGINAL_MIS();
    if (VAR == 5) RSF_F32_TRIP_BENCH_SHUFFLED_MIS();
  }
  const long long c1 = clock64();
  const long long r1 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) { cycles[0] = c1 - c0; cycles[1] = r1 - r0; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (double)(c1 - c0);
}

template <bool DAMP, bool PREFETCH_ONLY>
__global__ void __launch_bounds__(256, 2) __attribute__((amdgpu_num_vgpr(RSF_F32_TRIP_COMPILER_VGPRS / 2))) k(double *out, int trips, long long *cycles) {
  __shared__ __attribute__((aligned(16))) float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 1.0f + 1e-3f * (i % 17);
  __syncthreads();
  L32 L;
  const float dc = 1000.0f + threadIdx.x, a = 0.011f, b = 0.014f, h = 0.1f, vdc = 1.0f / dc;
  L.hhd = float2v{0.5f * h * vdc, 0.5f * h * vdc}; L.hd = L.hhd * 2.0f; L.khh = float2v{0.1f / dc / a * 0.5f * h, 0.1f / dc / a * 0.5f * h}; L.kh = L.khh * 2.0f;
  L.kh6 = L.khh * (1.0f / 3.0f); L.boa = float2v{b / a, b / a}; L.nhboa = L.boa * -0.5f; L.kvk = float2v{0.0f, 0.0f}; L.bh = float2v{10 * b, 10 * b} / L.hhd;
  L.cv = float2v{1.0f, 1.0f}; L.vref = float2v{1.0f, 1.0f}; L.h6 = float2v{h / 6, h / 6};
  RSF_F32_TRIP_SETUP(L);
  float2v w = {1.0f, 1.0f}, Rh = L.hhd, ms = {6000.0f, 6000.0f}, g2r, g2d, s32;
  double q0 = 0.0, q1 = 0.0;
  const unsigned va = (unsigned)(size_t)(const __attribute__((address_space(3))) float *)lds, oa = va + 4 * 2049;
  RSF_F32_TRIP_PRELOAD(va, oa);
  const long long c0 = clock64();
  for (int n = 0; n < trips; ++n) {
    const unsigned nv = va + 64 * (n & 15), no = oa + 32 * (n & 15);
    if (DAMP) RSF_F32_TRIP_DAMPED(w, Rh, ms, s32, g2r, g2d, nv, no);
    else RSF_F32_TRIP_UNDAMPED(w, Rh, ms, s32, g2r, g2d, nv, no);
    q0 += (double)s32.x; q1 += (double)s32.y;  // the group's sum into the float64 totals, as Out32::flush does
  }
  const long long c1 = clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) *cycles = c1 - c0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = q0 + q1 + w.x + Rh.y + ms.x + g2r.x + g2d.y;
}

int main() {
  double *out; long long *dc;
  const int blocks = 256;
  CHECK(hipMalloc(&out, blocks * 256 * sizeof(double))); CHECK(hipMalloc(&dc, 16));
  for (int rep = 0; rep < 2; ++rep)
    for (int damp = 1; damp >= 0; --damp) {
      const int trips = 20000;
      if (damp) k<true, false><<<blocks, 256>>>(out, trips, dc); else k<false, false><<<blocks, 256>>>(out, trips, dc);
      CHECK(hipDeviceSynchronize());
      long long h; CHECK(hipMemcpy(&h, dc, 8, hipMemcpyDeviceToHost));
      if (rep) printf("%s trip alone, one wave per SIMD: %.0f cycles per trip = %.1f per step\n", damp ? "damped" : "undamped", (double)h / trips, (double)h / trips / RSF_F32_TRIP_STEPS);
    }
  const char *names[6] = {"the trip's vector instructions as scheduled", "every source a fixed register", "destinations redirected to 8 rotating pairs",
                          "the same instructions, shuffled", "as scheduled, begun on a 4-byte boundary", "shuffled, begun on a 4-byte boundary"};
  for (int v = 0; v < 6; ++v) {
    const int trips = 20000;
    for (int rep = 0; rep < 2; ++rep) {
      if (v == 0) kv<0><<<blocks, 256>>>(out, trips, dc);
      if (v == 1) kv<1><<<blocks, 256>>>(out, trips, dc);
      if (v == 2) kv<2><<<blocks, 256>>>(out, trips, dc);
      if (v == 3) kv<3><<<blocks, 256>>>(out, trips, dc);
      if (v == 4) kv<4><<<blocks, 256>>>(out, trips, dc);
      if (v == 5) kv<5><<<blocks, 256>>>(out, trips, dc);
      CHECK(hipDeviceSynchronize());
    }
    long long h[2]; CHECK(hipMemcpy(h, dc, 16, hipMemcpyDeviceToHost));
    printf("  %-52s %.0f clock64 ticks per trip; clock64 / wall_clock64 (100 MHz) = %.2f -> %.0f ns per trip\n", names[v], (double)h[0] / trips,
           (double)h[0] / (double)h[1], 10.0 * (double)h[1] / trips);
  }
  return 0;
}
