// probe_math.hip — test-only harness: evaluates the kernel's fp64 math helpers (csrc/rsf_math.h)
// elementwise on the GPU so tests/test_gpu_math.py can compare them with NumPy.
//   usage: probe_math <log|exp|rcp|rcp_seed|sin2pi|cos2pi> <in.f64> <out.f64>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

#include "../../bayesian-markov-chain-monte-carlo_amd/csrc/rsf_math.h"

__global__ void probe(int kind, int n, const double *in, double *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = in[i];
  double y;
  switch (kind) {
    case 0: y = rsf::fm::log(x); break;
    case 1: y = rsf::fm::exp(x); break;
    case 2: y = rsf::fm::rcp(x); break;
    case 3: y = __builtin_amdgcn_rcp(x); break;
    default: { double sn, cs; rsf::fm::sincos2pi(x, sn, cs); y = kind == 4 ? sn : cs; } break;
  }
  out[i] = y;
}

int main(int argc, char **argv) {
  if (argc != 4) return 2;
  const char *kinds[] = {"log", "exp", "rcp", "rcp_seed", "sin2pi", "cos2pi"};
  int kind = -1;
  for (int k = 0; k < 6; ++k) if (!strcmp(argv[1], kinds[k])) kind = k;
  if (kind < 0) return 2;
  FILE *f = fopen(argv[2], "rb");
  if (!f) return 3;
  fseek(f, 0, SEEK_END);
  const long bytes = ftell(f);
  fseek(f, 0, SEEK_SET);
  const int n = (int)(bytes / sizeof(double));
  std::vector<double> h(n), o(n);
  if (fread(h.data(), sizeof(double), n, f) != (size_t)n) return 3;
  fclose(f);
  double *din, *dout;
  if (hipMalloc(&din, bytes) != hipSuccess || hipMalloc(&dout, bytes) != hipSuccess) return 4;
  (void)hipMemcpy(din, h.data(), bytes, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3((n + 255) / 256), dim3(256), 0, nullptr, kind, n, din, dout);
  if (hipMemcpy(o.data(), dout, bytes, hipMemcpyDeviceToHost) != hipSuccess) return 4;
  f = fopen(argv[3], "wb");
  fwrite(o.data(), sizeof(double), n, f);
  fclose(f);
  return 0;
}
