#!/bin/bash
# Compile ONE instantiation of a kernel template for gfx950 (seconds instead of the full library's minute) and print its
# registers / scratch / occupancy; the ISA is left in /tmp/one_kernel-hip-amdgcn-amd-amdhsa-gfx950.s for tools/isa_loops.py.
#   tools/one_kernel.sh 'mcmc_kernel<1, true, false, DOP853>(rsf::Consts, McmcArgs)' [extra hipcc flags...]
INST=$1; shift
CSRC="$(cd "$(dirname "$0")/../bayesian-markov-chain-monte-carlo_amd/csrc" && pwd)"
cat > /tmp/one_kernel.hip <<SRC
#include "$CSRC/rsf_kernels.h"
namespace rsfk { template __global__ void $INST; }
SRC
cd /tmp && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -no-hip-rt "$@" -Rpass-analysis=kernel-resource-usage -save-temps -c -o /tmp/one_kernel.o /tmp/one_kernel.hip 2>&1 \
  | grep -E "error|VGPRs:|AGPRs|ScratchSize|Occupancy|SGPRs:|LDS Size" | sed -e 's/.*remark: //' -e 's/^.*:[0-9]*:0: *//' -e 's/ \[-Rpass.*//' | tail -7 | tr '\n' ';'
echo
