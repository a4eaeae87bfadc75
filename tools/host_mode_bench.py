#!/usr/bin/env python3
"""
PCIe-inclusive rate of the hot path for RSF_MEM_HOST callers (plain C / ctypes users handing over NumPy arrays):
wall time of rsf_mcmc_run including the copy of the trace to host memory, with the drain pipeline (default) and
without it (RSF_DRAIN_BYTES so large that the whole run is one launch followed by one copy), next to the
device-resident rate that bench.py reports.

    python tools/host_mode_bench.py [--chains 65536] [--nsteps 500] [--iters 300] [--rounds 3]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=65536)
    ap.add_argument("--nsteps", type=int, default=500)
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()

    import torch

    import bayesian_markov_chain_monte_carlo_amd as rsf

    model = rsf.RateStateModel(number_time_steps=args.nsteps)
    C, n = args.chains, args.iters
    with rsf.Engine(mem="host") as e:
        e.set_model(model, 1)
        _, acc = e.forward([1000.0])
    rng = np.random.default_rng(2025)
    data = acc[:, 0] + np.abs(acc[:, 0]) * rng.standard_normal(acc.shape[0])
    q0 = np.full((C, 1), 1000.0)
    steps = float(C) * n * args.nsteps
    out = {"chains": C, "nsteps": args.nsteps, "iters": n, "trace_bytes": C * n * 17}

    def host_run(drain_bytes):
        if drain_bytes is None:
            os.environ.pop("RSF_DRAIN_BYTES", None)
        else:
            os.environ["RSF_DRAIN_BYTES"] = str(drain_bytes)
        best = None
        with rsf.Engine(mem="host") as e:
            e.set_model(model, 1)
            e.mcmc_init(q0, data, [0.0], [1e4], seed=2025, prior_len=3)
            bufs = e._traces(n, True)
            for b in bufs:
                b.fill(0)  # touch the pages once: first-touch faults are the caller's cost, not the library's
            e.mcmc_run(n, out=bufs)
            for _ in range(args.rounds):
                t0 = time.perf_counter()
                e.mcmc_run(n, out=bufs)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
        return best, bufs[0][-1, :4, 0].tolist()

    t_pipe, tail_a = host_run(None)
    t_one, tail_b = host_run(1 << 40)
    with rsf.Engine(mem="device") as e:
        e.set_model(model, 1)
        e.mcmc_init(torch.from_numpy(q0).cuda(), torch.from_numpy(data).cuda(), [0.0], [1e4], seed=2025, prior_len=3)
        bufs = e._traces(n, True)
        e.mcmc_run(n, out=bufs)
        e.sync()
        t_dev = None
        for _ in range(args.rounds):
            t0 = time.perf_counter()
            e.mcmc_run(n, out=bufs)
            e.sync()
            dt = time.perf_counter() - t0
            t_dev = dt if t_dev is None else min(t_dev, dt)
    out.update({
        "device_resident": {"s": t_dev, "steps_per_s": steps / t_dev},
        "host_pipelined": {"s": t_pipe, "steps_per_s": steps / t_pipe, "trace_GBps": out["trace_bytes"] / t_pipe / 1e9},
        "host_one_launch_one_copy": {"s": t_one, "steps_per_s": steps / t_one, "trace_GBps": out["trace_bytes"] / t_one / 1e9},
    })
    print(json.dumps(out))


if __name__ == "__main__":
    main()
