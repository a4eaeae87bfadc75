#!/bin/bash
# Average shader clock of the MCMC kernel for short and long launches on the SAME box:
#   GRBM_GUI_ACTIVE (cycles, summed over the 8 XCDs) / 8 / kernel duration.   tools/clock_check.sh (through gpurun)
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ips in 10 100; do
  OUT=gpurun_out/clock_$ips
  rm -rf $OUT
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT -- python3 bench.py --workload cfg1 --no-also --no-cpu-baseline --iters-per-step $ips --steps 10 --warmup 2 > $OUT.log 2>&1
done
python3 - <<'PY'
import csv, glob
for ips in (10, 100):
    cc = glob.glob(f"gpurun_out/clock_{ips}/**/*counter_collection.csv", recursive=True)[0]
    kt = glob.glob(f"gpurun_out/clock_{ips}/**/*kernel_trace.csv", recursive=True)[0]
    cyc = {}
    for r in csv.DictReader(open(cc)):
        if "mcmc_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cyc[r["Dispatch_Id"]] = cyc.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    dur = {r["Dispatch_Id"]: float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(kt)) if "mcmc_kernel" in r["Kernel_Name"]}
    ids = sorted(cyc, key=int)[2:]
    c = sum(cyc[i] for i in ids) / len(ids) / 8
    d = sum(dur[i] for i in ids) / len(ids)
    print(f"ips {ips}: {c / ips / 1e3:.1f} Kcycles per proposal-iteration, kernel {d / 1e6:.3f} ms, clock {c / d:.3f} GHz")
PY
