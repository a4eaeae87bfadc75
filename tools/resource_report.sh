#!/bin/bash
# Per-kernel VGPR / AGPR / scratch / occupancy of the product build (compiler remarks; no GPU needed).
#   tools/resource_report.sh [extra hipcc flags...]
cd "$(dirname "$0")/../bayesian-markov-chain-monte-carlo_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -no-hip-rt "$@" -Rpass-analysis=kernel-resource-usage -c -o /tmp/rsf_hip_report.o rsf_hip.hip 2> /tmp/rsf_report.txt
python3 - <<'PY'
import re
txt = open('/tmp/rsf_report.txt').read()
KEYS = (("VGPR", r"VGPRs"), ("AGPR", r"AGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"), ("occ", r"Occupancy \[waves/SIMD\]"), ("SGPR", r"TotalSGPRs"))
for b in re.split(r'remark: Function Name: ', txt)[1:]:
    name = b.split()[0]
    vals = " ".join("%s %4s" % (k, (re.search(pat + r": (\d+)", b) or [None, "?"])[1]) for k, pat in KEYS)
    name = re.sub(r"^_ZN4rsfk\d+", "", name)  # kernels live in namespace rsfk
    print("%-62s %s" % (name[:62], vals))
PY
