#!/usr/bin/env python3
"""
bench.py — headline metric of the MCMC-over-ODE hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg1]

One "step" = one launch of the fused MCMC kernel: `iters_per_step` Metropolis proposals for every
chain of this GPU, each proposal costing one rate-and-state RK4 forward solve of `nsteps` output
intervals.  Inputs (observation, loading table, chain state) are resident in HBM before the timed
region.  metric = chains × proposals × nsteps / wall seconds, whole job (all ranks).

Default workload: BASELINE.json configs[2] (262 144 chains x nsteps 2000, fp64) — the largest configuration that
fits one GPU; the configs[1] shape, the reference's own main.py sweep and one GPU's share of configs[4] (float64 and float32 solve)
are measured in the same process afterwards and carried under "also".

N > 1: one rank per GPU under torch.distributed.run.  `python bench.py --gpus N` without that launcher starts it
itself: the parent touches no GPU API, spawns `python -m torch.distributed.run --nproc-per-node N bench.py ...`
as a fresh child, relays rank 0's JSON line and exits with the child's status.  Chains shard by global id with no
data-path collective (weak scaling: per-GPU chains fixed); after the timed region the post-burn sample block is
pooled with ONE RCCL all-gather, timed separately.

Also printed in the same JSON line:
  roofline      the bound that binds this kernel: fp64 VALU issue (152 nominal flops per RK4 step against the
                78.6 TFLOP/s fp64-vector peak), with the instruction count and pipe occupancy the PMC profile of
                the same launch shape gave; roofline_hbm is the HBM view the metric names (algorithmic 16 B per
                chain-proposal), O(1e-3) of peak by construction;
  cpu_baseline  the CPU restatement (oracle/, OpenMP over chains) timed on this host's cores on a
                bounded sample of the same workload (rank 0, N = 1 only);
  reference_scheme  the same workload integrated with the reference's own DOP853 scheme (short side run, N = 1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json configs[1] / configs[2]
    "cfg1": dict(chains=65536, nsteps=500, desc="configs[1]: 65536 chains x nsteps 500, fp64"),
    "cfg2": dict(chains=262144, nsteps=2000, desc="configs[2]: 262144 chains x nsteps 2000, fp64"),
    # one GPU's share of configs[4]: joint (Dc, a, b), 1 048 576 / 8 chains, nsteps 4000 (profiling / tolerance-sweep shape)
    "cfg5": dict(chains=131072, nsteps=4000, n_params=3, desc="configs[4] per-GPU shard: 131072 chains x nsteps 4000, joint (Dc, a, b)"),
}
START = [1000.0, 0.011, 0.014]                 # chain start and prior box of the 3-parameter problem
BOX_LO, BOX_HI = [0.0, 0.005, 0.005], [1.0e4, 0.02, 0.03]
FLOPS_PER_RK4_STEP = 152.0       # 4 RHS x 27 + RK4 combine 39 + observation/SSq 5
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PEAK_FP64_VALU_TFLOPS = 78.6     # 256 CU x 4 SIMD x 16 fp64 FMA lanes x 2 flop x 2.4 GHz
SUSTAINED_FMA_TFLOPS = 70.0      # what a pure, fully occupied v_fma_f64 stream holds on this part for ~1 s (the power manager settles at
                                 # ~2.14 GHz under it): tools/peak_fp64.hip -> profiles/r02/peak_fp64.log (61.3 at two waves per SIMD)


def synthetic_problem(nsteps):
    """Observation = own forward solve at Dc_true = 1000 + |acc| N(0,1); built on the GPU."""
    import bayesian_markov_chain_monte_carlo_amd as pkg

    model = pkg.RateStateModel(number_time_steps=nsteps)
    with pkg.Engine(mem="host") as e:
        e.set_model(model, 1)
        _, acc = e.forward([1000.0])
    acc = acc[:, 0]
    data = acc + np.abs(acc) * np.random.default_rng(2025).standard_normal(acc.shape[0])
    return model, data


def effective_cpus():
    """Host threads this process may really use: the affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(model, data, target_s=12.0):
    """Time the CPU restatement on a bounded sample of the same workload (all host cores)."""
    import bayesian_markov_chain_monte_carlo_amd as pkg

    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import rsf_oracle

    lib = pkg._abi.bind(ctypes.CDLL(rsf_oracle.lib_path()))
    cores = effective_cpus()
    chains = 256 * cores
    # checker=True: this leg times the CPU restatement on purpose (reported baseline, not the product)
    with pkg.Engine(lib=lib, cpu_threads=cores, checker=True) as e:
        nout = e.set_model(model, 1)
        e.mcmc_init(np.full((chains, 1), 1000.0), data, [0.0], [1.0e4], seed=2025, prior_len=3)
        e.mcmc_run(1, traces=False)  # thread-pool warm-up
        t0 = time.perf_counter()
        e.mcmc_run(3, traces=False)
        per_iter = (time.perf_counter() - t0) / 3
        iters = int(max(4, min(2000, target_s / max(per_iter, 1e-6))))
        t0 = time.perf_counter()
        e.mcmc_run(iters, traces=("q", "std2"))
        wall = time.perf_counter() - t0
    nsteps = model.num_tsteps
    with pkg.Engine(lib=lib, cpu_threads=1, checker=True) as e:  # per-core figure: a short single-thread run
        e.set_model(model, 1)
        e.mcmc_init(np.full((8, 1), 1000.0), data, [0.0], [1.0e4], seed=2025, prior_len=3)
        e.mcmc_run(1, traces=False)
        t0 = time.perf_counter()
        e.mcmc_run(40, traces=False)
        one_core = 8 * 40 * nsteps / (time.perf_counter() - t0)
    return dict(value=chains * iters * nsteps / wall, unit="ODE-steps*chains/s", cores=cores, kind="port", single_thread_value=one_core,
                cpu_model=cpu_model_name(),
                reference_python={"value": 2.7e3, "unit": "ODE-steps*chains/s", "cores": 1,
                                  "where": "the reference's own Python/SciPy path, measured in the survey container (BASELINE.md §3); "
                                           "it cannot travel to this box, so this figure is quoted, not re-measured here"},
                sample=f"{chains} chains x {iters} proposals x nsteps {nsteps} ({nout - 1} RK4 steps each), "
                       f"oracle/librsf_oracle.so with OpenMP over chains, {wall:.1f} s")


def reference_scheme_rate(model, data, C, nsteps, proposals=5):
    """The same workload with the reference's OWN integration scheme (RateStateModel.integrator = "dop853": Hairer's
    DOP853 driven like scipy.integrate.ode does, the mode whose numbers are identical to the reference's) — a short
    side measurement next to the headline RK4 figure, outside its timed region."""
    import torch

    import bayesian_markov_chain_monte_carlo_amd as pkg

    model.integrator = "dop853"
    try:
        with pkg.Engine(mem="device") as e:
            e.set_model(model, 1)
            q0 = torch.full((C, 1), 1000.0, dtype=torch.float64, device="cuda")
            e.mcmc_init(q0, data, [0.0], [1.0e4], seed=2025, prior_len=3, adapt_mode="none")
            e.mcmc_run(proposals, traces=False)
            e.sync()
            t0 = time.perf_counter()
            e.mcmc_run(proposals, traces=False)
            e.sync()
            dt = time.perf_counter() - t0
    finally:
        model.integrator = "rk4"
    return {"integrator": "dop853 (rtol 1e-6, atol 1e-10, as the reference)", "value": C * proposals * nsteps / dt,
            "unit": "ODE-steps*chains/s", "sample": f"{C} chains x {proposals} proposals x nsteps {nsteps}"}


SWEEP = dict(dc_list=[100.0, 1325.0, 2550.0, 3775.0, 5000.0], qstart=1000.0, lo=0.0, hi=1.0e4, nsteps=500,
             chains_per_group=65536, burn=300, ips=100, launches=3)


def main_py_sweep(pkg, lib=None, chains_per_group=None, burn=None, ips=None, launches=None, headline_rate=None):
    """The reference's OWN problem shape (main.py:50-56): dc_list = linspace(100, 5000, 5), every chain started at
    QSTART = 1000, the LIST prior ["Uniform", 0, 10000] — with which the reference never adapts its proposal (the
    AttributeError of MCMC.py:200 is swallowed at MCMC.py:524-527), so the proposal keeps the width of Vstart — nsteps 500.
    One observation series per true Dc, one group of chains per series (RSF.inference_batched's shape).  This is the
    data-dependent BAD case of the sampler kernel, next to the headline's good one: at Dc_true = 100 about four proposals
    in ten fall outside the box (idle lanes, MCMC.py:318-322), the rest spread over every integration tier down to stiff
    small-Dc ones, and almost all of them are rejected.  Timed: the whole sweep as ONE launch per step, and every group
    alone (same chains: the RNG is keyed by the global chain id), each after `burn` untimed proposals; the counters
    (rsf_mcmc_counters) say where the wave-steps went.  `per_evaluated_step_vs_headline`: the group's rate per RK4 step of an
    IN-BOUNDS proposal over the headline's (all in bounds, all TIGHT)."""
    import torch

    cg = chains_per_group or SWEEP["chains_per_group"]
    burn = SWEEP["burn"] if burn is None else burn
    ips, launches = ips or SWEEP["ips"], launches or SWEEP["launches"]
    nsteps, dcs = SWEEP["nsteps"], SWEEP["dc_list"]
    G = len(dcs)
    model = pkg.RateStateModel(number_time_steps=nsteps)
    mk = (lambda **kw: pkg.Engine(lib=lib, **kw)) if lib is not None else (lambda **kw: pkg.Engine(**kw))
    with mk(mem="host") as e:
        e.set_model(model, 1)
        _, acc = e.forward(dcs)  # (nout, G): the clean series of every true Dc (RSF.generate_time_series, RSF.py:355-371)
    rng = np.random.default_rng(2025)
    data = np.stack([acc[:, g] + np.abs(acc[:, g]) * rng.standard_normal(acc.shape[0]) for g in range(G)])

    def run(first_group, n_groups):
        C = cg * n_groups
        eng = mk(mem="device")
        eng.set_model(model, 1)
        q0 = torch.full((C, 1), SWEEP["qstart"], dtype=torch.float64, device="cuda")
        d = torch.as_tensor(data[first_group:first_group + n_groups] if n_groups > 1 else data[first_group], device="cuda")
        eng.mcmc_init(q0, d, [SWEEP["lo"]], [SWEEP["hi"]], seed=2025, chain_offset=first_group * cg, prior_len=3, adapt_mode="none")
        tr = (torch.empty((ips, C, 1), dtype=torch.float64, device="cuda"), torch.empty((ips, C), dtype=torch.float64, device="cuda"), None)
        done = 0
        while done < burn:  # untimed: lets the chains leave the common start (the trace rows are overwritten)
            n = min(ips, burn - done)
            eng.mcmc_run(n, out=tuple(t[:n] if t is not None else None for t in tr))
            done += n
        torch.cuda.synchronize()
        has_counters = hasattr(eng.lib, "rsf_mcmc_counters")
        c0, s0 = (eng.counters() if has_counters else None), eng.stats()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
        for a, b in ev:
            a.record()
            eng.mcmc_run(ips, out=tr)
            b.record()
        torch.cuda.synchronize()
        ms = float(np.median([a.elapsed_time(b) for a, b in ev]))
        c1, s1 = (eng.counters() if has_counters else None), eng.stats()
        props = C * ips * launches
        evaluated = (s1["evaluated"] - s0["evaluated"]) / props
        out = {"chains": C, "ms_per_launch": ms, "value": C * ips * nsteps / (ms * 1e-3), "evaluated_fraction": evaluated,
               "acceptance": (s1["accepted"] - s0["accepted"]) / props,
               "posterior_mean": float(tr[0][:, :, 0].mean()), "proposal_std": float(eng.get_state()[3].flatten().sqrt().mean())}
        out["per_evaluated_step"] = out["value"] * evaluated
        if headline_rate:
            out["per_evaluated_step_vs_headline"] = out["per_evaluated_step"] / headline_rate
        if has_counters:
            dlt = {k: c1[k] - c0[k] for k in c0 if k != "lane_utilisation"}
            steps = dlt["steps_tight"] + dlt["steps_narrow"] + dlt["steps_wide"] + dlt["steps_full"]
            waves = dlt["wave_solves"] + dlt["wave_skips"]
            out["counters"] = {
                "out_of_bounds_fraction": dlt["out_of_bounds"] / props, "early_rejected_of_evaluated": dlt["early_rejected"] / max(1, dlt["evaluated"]),
                "wave_skip_fraction": dlt["wave_skips"] / max(1, waves),
                "wave_steps_vs_full_series": steps / max(1, dlt["wave_solves"] * (nsteps - 1)),
                "tier_share": {t: dlt["steps_" + t] / max(1, steps) for t in ("tight", "narrow", "wide", "full")},
                "redone_share": dlt["steps_redone"] / max(1, steps), "lane_utilisation": dlt["lane_steps"] / max(1, 64 * steps)}
        eng.close()
        del tr
        torch.cuda.empty_cache()
        return out

    res = {"workload": f"main.py sweep: dc_list {dcs}, qstart {SWEEP['qstart']}, list prior ({SWEEP['lo']}, {SWEEP['hi']}) (no adaptation), "
                       f"nsteps {nsteps}, {cg} chains per group, {burn} untimed + {launches} x {ips} timed proposals per chain",
           "unit": "ODE-steps*chains/s (nominal: chains x proposals x nsteps / s)", "headline_rate_used": headline_rate,
           "all_groups_one_launch": run(0, G), "groups": {}}
    for g, dc in enumerate(dcs):
        res["groups"][f"dc_true_{dc:g}"] = run(g, 1)
    return res


def abi_pool_allgather(eng, local, expected_pool, rdist, timeout_s=90.0):
    """The same posterior-pool exchange through the C ABI (rsf_comm_init + rsf_pool_allgather: the library's own
    RCCL communicator on the engine's stream), checked against the torch.distributed pool.  Outside the timed
    region, and under a watchdog so that it can never cost the result line."""
    import threading

    import torch

    res = {"status": "timeout"}
    dev = torch.cuda.current_device()

    def work():
        try:
            torch.cuda.set_device(dev)
            rdist.comm_init_from_process_group(eng)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = eng.pool_allgather(local)
            eng.sync()
            ms = (time.perf_counter() - t0) * 1e3
            same = bool(torch.equal(rdist.pool_to_chain_major(out), expected_pool))
            eng.comm_destroy()
            res.update(status="ok", ms=ms, equals_torch_pool=same)
        except Exception as exc:  # reported, never fatal: the torch.distributed pool above is the one that counts
            res.update(status="error", error=str(exc)[:200])

    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(timeout_s)
    return dict(res)


def previous_round_line(workload_desc, value):
    """The driver's record of the previous round's default run (BENCH_rNN.json at the repo root), so that a reader
    diffing two rounds sees at the top level whether the headline workloads are the same and what the like-for-like
    change is — round 2 had switched the default from configs[1] to configs[2] without saying so in the line."""
    import glob
    import re

    files = sorted(glob.glob(os.path.join(ROOT, "BENCH_r*.json")), key=lambda f: int(re.search(r"_r(\d+)", f).group(1)))
    if not files:
        return None
    try:
        prev = json.load(open(files[-1])).get("parsed") or {}
        pw, pv = prev["config"]["workload"], float(prev["value"])
    except (KeyError, TypeError, ValueError, OSError):
        return None
    same = pw == workload_desc
    return {"file": os.path.basename(files[-1]), "workload": pw, "value": pv, "comparable": same,
            "ratio": value / pv if same else None}


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks under torch.distributed.run as a CHILD process
    (this parent has touched no GPU API and never execs), relay rank 0's JSON line, exit with the child's status."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = [l for l in proc.stdout.splitlines() if l.startswith('{"metric"')]
    if lines:
        print(lines[-1], flush=True)
    if proc.returncode != 0 or not lines:
        sys.stderr.write(f"bench.py: child ranks exited with status {proc.returncode}"
                         f"{'' if lines else ' and printed no result line'}\n")
        sys.exit(proc.returncode or 1)
    sys.exit(0)


def time_sampler(pkg, model, data, C, ips, steps, warmup, rank, barrier, d=1):
    """W untimed + K timed launches of the fused sampler (ips proposals per chain each) on `C` chains of this rank.
    → (wall seconds of the K launches, mean kernel ms from HIP events on the launch stream, stats, engine, traces)."""
    import torch

    eng = pkg.Engine(mem="device")
    nout = eng.set_model(model, 1)
    q0 = torch.tensor(START[:d], dtype=torch.float64, device="cuda").repeat(C, 1)
    # one parameter: the reference's own mode (list prior: no adaptation); joint (Dc, a, b): the init kernel's prior-regularised
    # proposal covariance and corrected adaptive Metropolis — the documented three-parameter workflow, nothing hand-set
    eng.mcmc_init(q0, data, BOX_LO[:d], BOX_HI[:d], seed=2025, chain_offset=rank * C, prior_len=3, adapt_mode="none" if d == 1 else "am",
                  adapt_interval=10 if d == 1 else 20, fd_rel_step=1e-6 if d == 1 else 1e-4)
    traces = (torch.empty((ips, C, d), dtype=torch.float64, device="cuda"),
              torch.empty((ips, C), dtype=torch.float64, device="cuda"), None)
    for _ in range(warmup):
        eng.mcmc_run(ips, out=traces)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    barrier()
    t0 = time.perf_counter()
    for a, b in ev:          # the engine launches on torch's current stream, so these events bracket the kernel
        a.record()
        eng.mcmc_run(ips, out=traces)
        b.record()
    barrier()
    wall = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    return wall, kernel_ms, dict(eng.stats(), counters=eng.counters()), eng, traces, nout


def roofline_views(workload, custom, C, ips, nout, stats, kernel_ms, d=1, mode="RK4", build_id=None):
    """The two roofline views of one launch of the sampler kernel.  PMC figures stored in
    profiles/pmc_traffic.json are attached only if they were collected on THIS build of the kernels (rsf_build_id(): the
    hash of the kernel sources the loaded library was compiled from) and this launch shape; otherwise they are null and
    the view says pmc_stale."""
    evaluated = stats["evaluated"] / max(1, stats["iters_done"] * C)
    per_launch_props = C * ips
    rk4_steps_per_launch = per_launch_props * (nout - 1) * evaluated
    bytes_per_proposal = 8.0 * d + 8.0   # sample (d doubles) + sigma^2 written per chain-proposal
    hbm_gbs = bytes_per_proposal * per_launch_props / (kernel_ms * 1e-3) / 1e9
    tflops = FLOPS_PER_RK4_STEP * rk4_steps_per_launch / (kernel_ms * 1e-3) / 1e12
    pmc = {}
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(path) and not custom:
        with open(path) as f:
            t = json.load(f)
        if t.get(workload + "_iters_per_step") == ips and workload in t:   # PMC figures belong to this launch shape ...
            if build_id is not None and t.get(workload + "_build_id") == build_id:  # ... and to the code that is running
                pmc = {"traffic": t.get(workload), "valu_insts_per_rk4_step": t.get(workload + "_valu_per_rk4_step"),
                       "pipe_busy": t.get(workload + "_pipe_busy"),
                       "pmc_source": f"rocprofv3 PMC passes of this launch shape on this build ({build_id}), {t.get(workload + '_source')} — "
                                     "stored, not measured in this run"}
            else:
                pmc = {"pmc_stale": True, "pmc_source": f"stored counters ({t.get(workload + '_source')}) are from build "
                                                        f"{t.get(workload + '_build_id')}, the loaded library is {build_id}: not reported"}
    valu = {"bound": "valu_fp64", "achieved": tflops, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
            "frac": tflops / PEAK_FP64_VALU_TFLOPS, "traffic": pmc.get("traffic"),
            "kernel": f"mcmc_kernel<{d},damp,philox,{mode}>", "kernel_ms": kernel_ms,
            "flops_per_rk4_step": FLOPS_PER_RK4_STEP, "rk4_steps_per_s": rk4_steps_per_launch / (kernel_ms * 1e-3),
            "valu_insts_per_rk4_step": pmc.get("valu_insts_per_rk4_step"), "pipe_busy": pmc.get("pipe_busy"),
            "pmc_source": pmc.get("pmc_source"), "pmc_stale": bool(pmc.get("pmc_stale")),
            "peak_sustained": SUSTAINED_FMA_TFLOPS, "frac_of_sustained": tflops / SUSTAINED_FMA_TFLOPS,
            "issue_rate_vs_pure_fma_stream": (pmc["valu_insts_per_rk4_step"] * rk4_steps_per_launch / (kernel_ms * 1e-3) * 2.0 / 1e12
                                              / SUSTAINED_FMA_TFLOPS) if pmc.get("valu_insts_per_rk4_step") else None,
            "peak_sustained_source": "measured: pure v_fma_f64 stream, 8 waves/SIMD, 1 s (tools/peak_fp64.hip, profiles/r02/peak_fp64.log); "
                                     "issue_rate_vs_pure_fma_stream = this kernel's per-lane VALU instructions/s (PMC count x steps/s; x 2 flop) over that stream's",
            "note": "fp64 VALU issue binds (no MFMA: elementwise ODE recurrence); peak = 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz; "
                    "pipe_busy = 4 cycles x SQ_INSTS_VALU / SIMDs / GRBM_GUI_ACTIVE is the utilisation figure, frac the nominal-flop one"}
    hbm = {"bound": "hbm", "achieved": hbm_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm_gbs / PEAK_HBM_GBS,
           "traffic": pmc.get("traffic"), "algorithmic_bytes_per_launch": bytes_per_proposal * per_launch_props,
           "note": f"HBM view the metric names: {bytes_per_proposal:.0f} B written per chain-proposal; arithmetic intensity ~4750 flop/B, not the binding bound"}
    return valu, hbm, evaluated


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--chains", type=int, default=0, help="chains per GPU (default: the workload's)")
    ap.add_argument("--nsteps", type=int, default=0)
    ap.add_argument("--iters-per-step", type=int, default=100, help="proposals per chain per launch (SURVEY §8d: 100)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the secondary configs[1] measurement")
    ap.add_argument("--no-sweep", action="store_true", help="skip also.main_py_sweep (the reference's own dc_list / list-prior problem)")
    ap.add_argument("--integrator", default="rk4", choices=["rk4", "dop853"],
                    help="dop853 = the reference's own adaptive scheme (side measurement / profiling; the metric is quoted on rk4)")
    ap.add_argument("--precision", default="float64", choices=["float64", "float32"], help="float32 = config-5 tolerance-sweep solve")
    ap.add_argument("--abi-pool", action="store_true",
                    help="N > 1: repeat the pool exchange through the C ABI's own RCCL communicator after the timed region (rsf_comm_init + "
                         "rsf_pool_allgather, under a watchdog; a hang there ends the run with status 3) — opt-in, so that the driver's "
                         "scaling run depends only on the torch.distributed exchange")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 path on one GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)  # does not return

    # stdout carries exactly ONE line, the result: everything any library prints meanwhile (RCCL's version banner at the
    # first communicator, for one) goes to stderr — file descriptor 1 is pointed at stderr until the line is written
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    import bayesian_markov_chain_monte_carlo_amd as pkg
    from bayesian_markov_chain_monte_carlo_amd import dist as rdist

    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    if world != max(1, args.gpus):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    if world > 1:
        rank, world = rdist.init_process_group(args.backend)
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count())
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"  # gloo rehearsal moves the collectives' tensors to the host

    wl = WORKLOADS[args.workload]
    custom = bool(args.chains or args.nsteps)
    d = wl.get("n_params", 1)
    variant = ("_dop853" if args.integrator == "dop853" else "") + ("_f32" if args.precision == "float32" else "")
    mode = "DOP853" if args.integrator == "dop853" else ("RK4/f32" if args.precision == "float32" else "RK4")
    C = args.chains or wl["chains"]          # per GPU: weak scaling
    nsteps = args.nsteps or wl["nsteps"]
    ips = args.iters_per_step

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    model, data = synthetic_problem(nsteps)
    model.integrator, model.precision = args.integrator, args.precision
    wall, kernel_ms, stats, eng, traces, nout = time_sampler(pkg, model, data, C, ips, args.steps, args.warmup, rank, barrier, d)
    model.integrator, model.precision = "rk4", "float64"

    if world > 1:
        t = torch.tensor([wall], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
        # the path's only collective: pool the last block of samples from every GPU (RCCL all-gather)
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        pool = rdist.pool_to_chain_major(rdist.allgather_pool(traces[0].to(coll_dev)))
        torch.cuda.synchronize()
        allgather_ms = (time.perf_counter() - g0) * 1e3
        assert pool.shape == (ips, world * C, d)
        abi_pool = abi_pool_allgather(eng, traces[0], pool, rdist) if (args.abi_pool and args.backend == "nccl") else None
    else:
        allgather_ms = abi_pool = None

    if rank == 0:
        value = world * C * ips * args.steps * nsteps / wall
        build_id = pkg._abi.load().rsf_build_id().decode()
        valu, hbm, evaluated = roofline_views(args.workload + variant, custom, C, ips, nout, stats, kernel_ms, d, mode, build_id)
        if mode != "RK4":
            valu["note"] = ("side measurement, not the headline arithmetic: " + ("the reference's adaptive DOP853 scheme — the unit of "
                            "rk4_steps_per_s is one OUTPUT INTERVAL (>= 1 step of 12 stages), flops are not counted, frac is nominal only"
                            if mode == "DOP853" else "float32 ODE solve (hardware v_exp/v_log/v_rcp_f32); frac still normalised by the fp64 peak"))
        out = {
            "metric": "ode_steps_x_chains_per_sec", "value": value, "unit": "ODE-steps*chains/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64" if args.precision == "float64" else "f32 solve, f64 sampler",
            "data": "synthetic", "build_id": build_id,
            "config": {"workload": (wl["desc"] if not custom else f"custom: {C} chains x nsteps {nsteps}, fp64") + (f" [{mode}]" if mode != "RK4" else ""),
                       "chains_per_gpu": C, "nsteps": nsteps, "rk4_substeps": 1, "proposals_per_chain_per_step": ips,
                       "n_params": d, "adapt_mode": "none" if d == 1 else "am", "integrator": args.integrator, "parallelism": f"chains sharded over {world} GPU(s), no data-path collective",
                       "evaluated_fraction": evaluated,
                       "acceptance": stats["accepted"] / max(1, stats["iters_done"] * C),
                       # where the wave-steps went (rsf_mcmc_counters, float64 RK4 sampler): in-bounds proposals whose solve was cut
                       # short because their running sum of squares already ruled acceptance out (exact: partial sums of squares
                       # only grow); share of lane-steps on lanes still integrating; wave-steps per tier.  In this workload every
                       # wave keeps an undecided lane to the end of the series: no wave-step is skipped, value counts executed work
                       "early_rejected_of_evaluated": stats["counters"]["early_rejected"] / max(1, stats["counters"]["evaluated"]),
                       "lane_utilisation": stats["counters"]["lane_utilisation"],
                       "wave_steps_vs_full_series": (sum(stats["counters"]["steps_" + t] for t in ("tight", "narrow", "wide", "full"))
                                                     / max(1, stats["counters"]["wave_solves"] * (nout - 1))) if mode in ("RK4", "RK4/f32") else None,
                       # (float32 solve: tight = incremental trips, full = full-evaluation trips — a wave holding chains of both forms
                       #  runs both —, redone = trips replayed step by step because a chain left the incremental form in them)
                       "tier_wave_steps": {t: stats["counters"]["steps_" + t] for t in ("tight", "narrow", "wide", "full", "redone")}},
            "roofline": valu, "roofline_hbm": hbm,
        }
        out["previous_round"] = previous_round_line(out["config"]["workload"], value)
        if allgather_ms is not None:
            out["pool_allgather_ms"] = allgather_ms
        if abi_pool is not None:
            out["pool_allgather_c_abi"] = abi_pool
    if world == 1:
        eng.close()
        del traces
        torch.cuda.empty_cache()
        if not args.no_also and not custom and d == 1 and mode == "RK4":
            # the other single-GPU BASELINE configuration, same process, same method (secondary figure)
            other = "cfg1" if args.workload == "cfg2" else "cfg2"
            wo = WORKLOADS[other]
            k = max(3, min(args.steps, 10))
            model_o, data_o = synthetic_problem(wo["nsteps"])
            w_o, kms_o, st_o, eng_o, tr_o, nout_o = time_sampler(pkg, model_o, data_o, wo["chains"], ips, k, 2, 0, barrier)
            valu_o, _, _ = roofline_views(other, False, wo["chains"], ips, nout_o, st_o, kms_o, build_id=out["build_id"])
            out["also"] = {other: {"workload": wo["desc"], "value": wo["chains"] * ips * k * wo["nsteps"] / w_o,
                                   "unit": "ODE-steps*chains/s", "steps": k, "ms_per_step": w_o / k * 1e3,
                                   "roofline": {kk: valu_o[kk] for kk in ("bound", "achieved", "peak", "unit", "frac", "kernel_ms",
                                                                          "valu_insts_per_rk4_step", "pipe_busy", "traffic", "pmc_stale")}}}
            eng_o.close()
            del tr_o
            torch.cuda.empty_cache()
            if not args.no_sweep:
                # the data-dependent bad case next to the headline's good one: the reference's own main.py problem
                cfg1_rate = out["also"]["cfg1"]["value"] if "cfg1" in out["also"] else value
                out["also"]["main_py_sweep"] = main_py_sweep(pkg, headline_rate=cfg1_rate)
                # BASELINE configs[4] ("joint (a, b, d_c), nsteps=4000, float32 vs float64 tolerance sweep"): one GPU's share of it, both
                # precisions, three launches each — the init kernel's own proposal covariance and adaptive Metropolis, nothing hand-set
                out["also"]["cfg5"] = {}
                w5 = WORKLOADS["cfg5"]
                model5, data5 = synthetic_problem(w5["nsteps"])
                for prec in ("float64", "float32"):
                    model5.precision = prec
                    w_5, kms_5, st_5, eng_5, tr_5, _ = time_sampler(pkg, model5, data5, w5["chains"], ips, 3, 1, 0, barrier, w5["n_params"])
                    out["also"]["cfg5"][prec] = {
                        "workload": w5["desc"] + (" [float32 solve, float64 sampler]" if prec == "float32" else ""),
                        "value": w5["chains"] * ips * 3 * w5["nsteps"] / w_5, "unit": "ODE-steps*chains/s", "steps": 3, "ms_per_step": w_5 / 3 * 1e3,
                        "kernel_ms": kms_5, "evaluated_fraction": st_5["evaluated"] / max(1, st_5["iters_done"] * w5["chains"]),
                        "tier_wave_steps": {t: st_5["counters"]["steps_" + t] for t in ("tight", "narrow", "wide", "full", "redone")}}
                    eng_5.close()
                    del tr_5
                    torch.cuda.empty_cache()
                model5.precision = "float64"
        if not args.no_cpu_baseline and d == 1 and mode == "RK4":
            out["reference_scheme"] = reference_scheme_rate(model, data, C, nsteps)
            out["cpu_baseline"] = cpu_baseline(model, data)
    if rank == 0:
        sys.stdout.flush()
        os.dup2(result_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if abi_pool is not None and abi_pool.get("status") == "timeout":
        # a rank stuck inside a collective cannot be torn down cleanly; the result line is out, but a hung collective is
        # a failure of the run, not a success: non-zero status
        os._exit(3)
    if world > 1:
        eng.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
