#!/bin/bash
# Where the sampler kernel's wave cycles go, build against build, same launches in one process (two PMC passes):
#   tools/stall_ab.sh <tag> [ab_bench args ...] name=path.so ...
# Sampler dispatches arrive in library order (default, then the named ones), so dispatch i belongs to library i mod N.
set -e
TAG=$1; shift
N=1; for a in "$@"; do case "$a" in *=*) N=$((N+1));; esac; done
OUT=gpurun_out/stall_ab_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/a -- python3 tools/ab_bench.py "$@" > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INST_CYCLES_SALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/b -- python3 tools/ab_bench.py "$@" > $OUT/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH --kernel-trace --output-format csv -d $OUT/c -- python3 tools/ab_bench.py "$@" > $OUT/c.log 2>&1 || true
python3 - "$OUT" "$N" <<'PY'
import csv, glob, sys, collections
out, n = sys.argv[1], int(sys.argv[2])
res = [collections.defaultdict(list) for _ in range(n)]
for d in ("a", "b", "c"):
    per = collections.OrderedDict()
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "mcmc_" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"] and "propose" not in r["Kernel_Name"]:  # the sampler kernels (float64, float32 two-chain)
                per.setdefault(int(r["Dispatch_Id"]), collections.defaultdict(float))[r["Counter_Name"]] += float(r["Counter_Value"])
    for i, disp in enumerate(sorted(per)):
        if i < n:
            continue  # the warm-up launches
        for k, v in per[disp].items():
            res[i % n][k].append(v)
names = sorted({k for r in res for k in r})
print("%-22s" % "counter" + "".join("%16s" % f"lib{i}" for i in range(n)) + "   (per launch; ratio to SQ_WAVE_CYCLES of that lib)")
for k in names:
    vals = [sum(r[k]) / max(1, len(r[k])) for r in res]
    wcs = [sum(r["SQ_WAVE_CYCLES"]) / max(1, len(r["SQ_WAVE_CYCLES"])) for r in res]
    print("%-22s" % k + "".join("%16.4g" % v for v in vals) + "   " + " ".join("%.4f" % (v / w if w else 0) for v, w in zip(vals, wcs)))
PY
