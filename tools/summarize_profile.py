#!/usr/bin/env python3
"""Summarise the rocprofv3 passes written by tools/profile_gpu.sh into profiles/<round>/ (tracked).

  python tools/summarize_profile.py gpurun_out/prof_cfg1 cfg1 profiles/r01 [build-tag]

Writes <tag>_<build>_kernel_stats.csv (copy of rocprofv3 --stats), <tag>_<build>_pmc.json (per-launch
averages of the dominant kernel's counters, HBM traffic with the gfx950 FETCH_SIZE correction from
MI355X_MICROARCH.md §HBM) and updates profiles/pmc_traffic.json, which bench.py reports as
roofline.traffic."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def rows(pattern):
    """Rows of the NEWEST file matching the pattern (gpurun merges successive runs into one directory)."""
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return list(csv.DictReader(open(files[-1]))) if files else []


def main():
    src, tag, dst = sys.argv[1:4]
    build = sys.argv[4] if len(sys.argv) > 4 else "latest"
    os.makedirs(dst, exist_ok=True)
    stats = sorted(glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1]
    shutil.copy(stats, os.path.join(dst, f"{tag}_{build}_kernel_stats.csv"))
    top = max(csv.DictReader(open(stats)), key=lambda r: float(r["TotalDurationNs"]))
    kern = top["Name"]
    summary = {"kernel": kern, "calls": int(top["Calls"]), "avg_ms": float(top["AverageNs"]) / 1e6,
               "share_of_gpu_time_pct": float(top["Percentage"])}
    trace = [r for r in rows(os.path.join(src, "trace", "**", "*kernel_trace.csv")) if r["Kernel_Name"] == kern]
    if trace:
        t = trace[-1]
        summary.update(vgpr=int(t["VGPR_Count"]), sgpr=int(t["SGPR_Count"]), lds_bytes=int(t["LDS_Block_Size"]),
                       scratch=int(t["Scratch_Size"]), workgroup=int(t["Workgroup_Size_X"]), grid=int(t["Grid_Size_X"]))
    counters = defaultdict(list)
    for p in ("pmc_sq", "pmc_fetch", "pmc_write"):
        per_dispatch = defaultdict(float)
        for r in rows(os.path.join(src, p, "**", "*counter_collection.csv")):
            if r["Kernel_Name"] == kern:
                per_dispatch[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (_, name), v in per_dispatch.items():
            counters[name].append(v)
    # per-launch figure = the MEDIAN over the launches of the pass (one launch of a pass occasionally carries another
    # kernel's write-back: e.g. WRITE_SIZE 131328 / 131328 / 296535 KiB for three identical DOP853 launches)
    pmc = {k: sorted(v)[len(v) // 2] if len(v) % 2 else 0.5 * (sorted(v)[len(v) // 2 - 1] + sorted(v)[len(v) // 2]) for k, v in counters.items()}
    summary["pmc_launches_per_counter"] = {k: len(v) for k, v in counters.items()}
    summary["pmc_per_launch"] = pmc
    if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
        # rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B read request for wide coalesced
        # streams (MI355X_MICROARCH.md §HBM): doubled bytes are the upper estimate, raw the lower.
        fetch_lo, write = pmc["FETCH_SIZE"] * 1024, pmc["WRITE_SIZE"] * 1024
        summary["hbm_bytes_per_launch"] = {"fetch_raw": fetch_lo, "fetch_corrected_x2": 2 * fetch_lo, "write": write,
                                           "total_corrected": 2 * fetch_lo + write}
    if "SQ_INSTS_VALU" in pmc and "SQ_WAVES" in pmc:
        summary["valu_insts_per_wave"] = pmc["SQ_INSTS_VALU"] / pmc["SQ_WAVES"]
    if "SQ_ACTIVE_INST_VALU" in pmc and "SQ_BUSY_CYCLES" in pmc:
        summary["note_units"] = "SQ_* cycle counters are in quad-cycles summed over SEs/XCDs (MI355X_MICROARCH.md)"
    # the bench line printed under the kernel-trace pass: kept beside the summaries; its proposals per launch say
    # which launch shape the per-launch counters belong to
    ips = bench_roof = None
    log = os.path.join(src, "trace.log")
    if os.path.exists(log):
        lines = [l for l in open(log) if l.startswith('{"metric"')]
        if lines:
            bench = json.loads(lines[-1])
            ips = bench["config"].get("proposals_per_chain_per_step")
            summary["build_id"] = bench.get("build_id")  # rsf_build_id() of the library that ran under the profiler
            summary["bench_kernel_ms_hip_events"] = bench["roofline"].get("kernel_ms")
            bench_roof = bench["roofline"] if "rk4_steps_per_s" in bench["roofline"] else bench.get("roofline_valu")
            if bench_roof is not None:
                bench_roof = dict(bench_roof, kernel_ms=bench["roofline"].get("kernel_ms"))
            json.dump(bench, open(os.path.join(dst, f"{tag}_{build}_bench.json"), "w"))
    summary["proposals_per_chain_per_launch"] = ips
    if bench_roof and "SQ_INSTS_VALU" in pmc and "GRBM_GUI_ACTIVE" in pmc:
        # SQ_INSTS_VALU counts wave-instructions; each occupies a SIMD's fp64 pipe for 4 cycles; 1024 SIMDs; GRBM_GUI_ACTIVE is
        # summed over the 8 XCDs.  RK4 steps per launch come from the bench line of the kernel-trace pass (same launch shape).
        wave_steps = bench_roof["rk4_steps_per_s"] * bench_roof["kernel_ms"] * 1e-3 / 64.0
        summary["valu_insts_per_rk4_step"] = pmc["SQ_INSTS_VALU"] / wave_steps
        summary["salu_insts_per_rk4_step"] = pmc.get("SQ_INSTS_SALU", 0.0) / wave_steps
        summary["pipe_busy"] = 4.0 * pmc["SQ_INSTS_VALU"] / 1024.0 / (pmc["GRBM_GUI_ACTIVE"] / 8.0)
        summary["clock_ghz"] = pmc["GRBM_GUI_ACTIVE"] / 8.0 / (summary["avg_ms"] * 1e-3) / 1e9
    json.dump(summary, open(os.path.join(dst, f"{tag}_{build}_pmc.json"), "w"), indent=1)
    tfile = os.path.join(os.path.dirname(dst.rstrip("/")), "pmc_traffic.json")
    traffic = json.load(open(tfile)) if os.path.exists(tfile) else {}
    if "hbm_bytes_per_launch" in summary:
        traffic[tag] = summary["hbm_bytes_per_launch"]["total_corrected"]
        traffic[tag + "_source"] = f"{dst}/{tag}_{build}_pmc.json"
        traffic[tag + "_iters_per_step"] = ips
        traffic[tag + "_build_id"] = summary.get("build_id")
        traffic[tag + "_valu_per_rk4_step"] = summary.get("valu_insts_per_rk4_step")
        traffic[tag + "_pipe_busy"] = summary.get("pipe_busy")
        json.dump(traffic, open(tfile, "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
