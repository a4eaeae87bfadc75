#!/usr/bin/env python3
"""
Posterior post-processing on the pooled draws (RSF.plot_dist's gaussian_kde on 1000 grid points, RSF.py:717-746):
device KDE + moments of a cfg1-sized pool (65 536 chains x 501 kept draws = 32.8 M samples) against
scipy.stats.gaussian_kde on a subsample (its cost is linear in the sample count).

    python tools/kde_bench.py [--chains 65536] [--iters 1000] [--grid 1000]
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
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--grid", type=int, default=1000)
    ap.add_argument("--scipy-samples", type=int, default=200000)
    args = ap.parse_args()

    import torch
    from scipy.stats import gaussian_kde

    import bayesian_markov_chain_monte_carlo_amd as rsf
    from bench import synthetic_problem

    model, data = synthetic_problem(500)
    C, n = args.chains, args.iters
    with rsf.Engine(mem="device") as e:
        e.set_model(model, 1)
        e.mcmc_init(torch.full((C, 1), 1000.0, dtype=torch.float64, device="cuda"), data, [0.0], [1e4], seed=2025, prior_len=3)
        t0 = time.perf_counter()
        tq, _, _ = e.mcmc_run(n, traces=("q",))
        e.sync()
        t_sample = time.perf_counter() - t0
        pool = tq[n // 2 - 1:].contiguous()
        nsamp = pool.numel()
        s = e.pool_summary(pool)
        grid = torch.linspace(s["min"], s["max"], args.grid, dtype=torch.float64, device="cuda")
        e.pool_kde(pool, grid)
        e.sync()
        t0 = time.perf_counter()
        dens = e.pool_kde(pool, grid)
        e.sync()
        t_kde = time.perf_counter() - t0
        t0 = time.perf_counter()
        e.pool_summary(pool)
        t_sum = time.perf_counter() - t0
        sub = pool.reshape(-1)[:: max(1, nsamp // args.scipy_samples)].cpu().numpy()
    t0 = time.perf_counter()
    ref = gaussian_kde(sub)(grid.cpu().numpy())
    t_scipy = time.perf_counter() - t0
    dens = dens.cpu().numpy()
    out = {
        "pool_samples": nsamp, "grid": args.grid, "sampling_s": t_sample,
        "device_kde_s": t_kde, "device_kde_pairs_per_s": nsamp * args.grid / t_kde,
        "device_summary_s": t_sum, "summary_GBps": nsamp * 8 / t_sum / 1e9,
        "scipy_kde_s": t_scipy, "scipy_samples": int(sub.size), "scipy_pairs_per_s": sub.size * args.grid / t_scipy,
        "scipy_extrapolated_to_pool_s": t_scipy * nsamp / sub.size,
        "max_abs_density_diff_vs_subsample_kde": float(np.abs(dens - ref).max()), "density_max": float(dens.max()),
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
