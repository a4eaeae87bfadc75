#!/usr/bin/env python3
"""
bench.py — headline metric of the MCMC-over-ODE hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg1|cfg2]

One "step" = one launch of the fused MCMC kernel: `iters_per_step` Metropolis proposals for every
chain of this GPU, each proposal costing one rate-and-state RK4 forward solve of `nsteps` output
intervals.  Inputs (observation, loading table, chain state) are resident in HBM before the timed
region.  metric = chains × proposals × nsteps / wall seconds, whole job (all ranks).

N > 1: launched by torch.distributed.run, one rank per GPU; chains shard by global id with no
data-path collective (weak scaling: per-GPU chains fixed); after the timed region the post-burn
sample block is pooled with ONE RCCL all-gather, timed separately.

Also printed in the same JSON line:
  roofline      HBM view the metric names (algorithmic 16 B per chain-proposal) + the fp64-VALU
                view that actually bounds this kernel (152 nominal flops per RK4 step);
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
}
FLOPS_PER_RK4_STEP = 152.0       # 4 RHS x 27 + RK4 combine 39 + observation/SSq 5
BYTES_PER_PROPOSAL = 16.0        # 8 B sample + 8 B sigma^2 written per chain-proposal (d = 1)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PEAK_FP64_VALU_TFLOPS = 78.6     # 256 CU x 4 SIMD x 16 fp64 FMA lanes x 2 flop x 2.4 GHz


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
    with pkg.Engine(lib=lib, cpu_threads=cores) as e:
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
    with pkg.Engine(lib=lib, cpu_threads=1) as e:  # per-core figure: a short single-thread run
        e.set_model(model, 1)
        e.mcmc_init(np.full((8, 1), 1000.0), data, [0.0], [1.0e4], seed=2025, prior_len=3)
        e.mcmc_run(1, traces=False)
        t0 = time.perf_counter()
        e.mcmc_run(40, traces=False)
        one_core = 8 * 40 * nsteps / (time.perf_counter() - t0)
    return dict(value=chains * iters * nsteps / wall, unit="ODE-steps*chains/s", cores=cores, kind="port", single_thread_value=one_core,
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg1", choices=sorted(WORKLOADS))
    ap.add_argument("--chains", type=int, default=0, help="chains per GPU (default: the workload's)")
    ap.add_argument("--nsteps", type=int, default=0)
    ap.add_argument("--iters-per-step", type=int, default=100, help="proposals per chain per launch (SURVEY §8d: 100)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 path on one GPU)")
    args = ap.parse_args()

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
    if world > 1:
        rank, world = rdist.init_process_group(args.backend)
    elif args.gpus > 1:
        raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count())
    coll_dev = "cuda" if args.backend == "nccl" else "cpu"  # gloo rehearsal moves the collectives' tensors to the host

    wl = WORKLOADS[args.workload]
    C = args.chains or wl["chains"]          # per GPU: weak scaling
    nsteps = args.nsteps or wl["nsteps"]
    ips = args.iters_per_step

    model, data = synthetic_problem(nsteps)
    eng = pkg.Engine(mem="device")
    nout = eng.set_model(model, 1)
    q0 = torch.full((C, 1), 1000.0, dtype=torch.float64, device="cuda")
    eng.mcmc_init(q0, data, [0.0], [1.0e4], seed=2025, chain_offset=rank * C, prior_len=3, adapt_mode="none")
    traces = (torch.empty((ips, C, 1), dtype=torch.float64, device="cuda"),
              torch.empty((ips, C), dtype=torch.float64, device="cuda"), None)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        eng.mcmc_run(ips, out=traces)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for a, b in ev:          # the engine launches on torch's current stream, so these events bracket the kernel
        a.record()
        eng.mcmc_run(ips, out=traces)
        b.record()
    barrier()
    wall = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    stats = eng.stats()

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
        assert pool.shape == (ips, world * C, 1)
        abi_pool = abi_pool_allgather(eng, traces[0], pool, rdist) if args.backend == "nccl" else None
    else:
        allgather_ms = abi_pool = None

    if rank == 0:
        proposals = world * C * ips * args.steps
        value = proposals * nsteps / wall
        per_launch_props = C * ips
        rk4_steps_per_launch = per_launch_props * (nout - 1) * stats["evaluated"] / max(1, stats["iters_done"] * C)
        hbm_gbs = BYTES_PER_PROPOSAL * per_launch_props / (kernel_ms * 1e-3) / 1e9
        tflops = FLOPS_PER_RK4_STEP * rk4_steps_per_launch / (kernel_ms * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            with open(pmc) as f:
                t = json.load(f)
            if t.get(args.workload + "_iters_per_step") == ips and not (args.chains or args.nsteps):
                traffic = t.get(args.workload)   # PMC bytes per launch, measured for this launch shape
        out = {
            "metric": "ode_steps_x_chains_per_sec", "value": value, "unit": "ODE-steps*chains/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl["desc"] if not (args.chains or args.nsteps) else f"custom: {C} chains x nsteps {nsteps}, fp64",
                       "chains_per_gpu": C, "nsteps": nsteps, "rk4_substeps": 1, "proposals_per_chain_per_step": ips,
                       "n_params": 1, "adapt_mode": "none", "parallelism": f"chains sharded over {world} GPU(s), no data-path collective",
                       "evaluated_fraction": stats["evaluated"] / max(1, stats["iters_done"] * C),
                       "acceptance": stats["accepted"] / max(1, stats["iters_done"] * C)},
            "roofline": {"bound": "hbm", "achieved": hbm_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm_gbs / PEAK_HBM_GBS,
                         "traffic": traffic, "kernel": "mcmc_kernel<1,damp,philox>", "kernel_ms": kernel_ms,
                         "note": "HBM view as the metric asks; this kernel is fp64-VALU bound (see roofline_valu)"},
            "roofline_valu": {"bound": "valu_fp64", "achieved": tflops, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
                              "frac": tflops / PEAK_FP64_VALU_TFLOPS, "flops_per_rk4_step": FLOPS_PER_RK4_STEP,
                              "rk4_steps_per_s": rk4_steps_per_launch / (kernel_ms * 1e-3)},
        }
        if allgather_ms is not None:
            out["pool_allgather_ms"] = allgather_ms
        if abi_pool is not None:
            out["pool_allgather_c_abi"] = abi_pool
        if world == 1 and not args.no_cpu_baseline:
            out["reference_scheme"] = reference_scheme_rate(model, data, C, nsteps)
            out["cpu_baseline"] = cpu_baseline(model, data)
        sys.stdout.flush()
        os.dup2(result_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if abi_pool is not None and abi_pool.get("status") == "timeout":
        os._exit(0)  # a rank stuck inside a collective cannot be torn down cleanly; the result line is out
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
