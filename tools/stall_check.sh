#!/bin/bash
# Where a lone wave per SIMD spends its cycles (cfg1): SQ wait / active / instruction-fetch counters of the MCMC kernel.
#   tools/stall_check.sh [bench args]    (through gpurun; two PMC passes, no tracing domains besides the kernel trace)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/stall_a gpurun_out/stall_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d gpurun_out/stall_a -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 "$@" > gpurun_out/stall_a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INST_CYCLES_SALU SQ_INSTS_VALU --kernel-trace --output-format csv -d gpurun_out/stall_b -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 "$@" > gpurun_out/stall_b.log 2>&1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
tot = defaultdict(list)
for d in ("a", "b"):
    per = defaultdict(float)
    for f in glob.glob(f"gpurun_out/stall_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "mcmc_kernel" in r["Kernel_Name"]:
                per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (_, n), v in per.items():
        tot[n].append(v)
avg = {k: sum(v) / len(v) for k, v in tot.items()}
wc = avg.get("SQ_WAVE_CYCLES", 1.0)
for k in sorted(avg):
    print(f"{k:24s} {avg[k]:16.0f}   / SQ_WAVE_CYCLES = {avg[k] / wc:.3f}")
PY
