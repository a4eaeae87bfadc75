#!/bin/bash
# Instruction counts of the sampler kernel on the in-tree build and on other builds, same launches, one process:
#   tools/pmc_ab.sh <tag> [ab_bench args ...] name=path.so ...
# rocprofv3 --pmc in its own run (no other trace domains); prints per-dispatch counters grouped by launch order.
set -e
TAG=$1; shift
OUT=gpurun_out/pmc_ab_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc -- python3 tools/ab_bench.py "$@" > $OUT/ab.log 2>&1
tail -4 $OUT/ab.log
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
f = sorted(glob.glob(os.path.join(out, "pmc", "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f)) if "mcmc_kernel" in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = list(by)
print(len(ids), "sampler dispatches")
for i, d in enumerate(ids):
    c = by[d]
    print(i, " ".join(f"{k}={c.get(k, 0):.4g}" for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVES")))
PY
