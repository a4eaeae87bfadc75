#!/usr/bin/env python3
"""Tier 3 at scale: the pooled posterior of a long GPU run against the pooled posterior of independent chains of
the CPU oracle on the same observation — same sampler, different RNG stream (other seed), so agreement is statistical: the two
pooled means must agree within a few Monte-Carlo standard errors and the spreads / acceptance rates closely.  Also a soak run:
65 536 chains x 20 000 proposals (6.6e11 RK4 steps) without a non-finite value.

  python tools/soak_tier3.py [out.json]      (GPU box; ~40 s of the host's cores for the oracle)"""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bayesian_markov_chain_monte_carlo_amd as pkg  # noqa: E402
import rsf_oracle  # noqa: E402  (the checker: a measurement tool, not the product path)


def batch_means_se(per_chain_means):
    """Standard error of the pooled mean from the spread of independent chains' means."""
    return float(np.std(per_chain_means, ddof=1) / np.sqrt(per_chain_means.size))


def main():
    import torch

    lib = pkg._abi.bind(ctypes.CDLL(rsf_oracle.lib_path()))
    model = pkg.RateStateModel(500)
    with pkg.Engine(mem="host") as e:
        e.set_model(model, 1)
        _, acc = e.forward([1000.0])
    acc = acc[:, 0]
    data = acc + np.abs(acc) * np.random.default_rng(2025).standard_normal(acc.shape[0])
    out = {}
    # --- GPU: 65 536 chains, burn 1000, then 19 blocks of 1000 proposals; running moments per block (no 10 GB trace)
    C, block, nblocks = 65536, 1000, 20
    with pkg.Engine(mem="device") as e:
        e.set_model(model, 1)
        q0 = torch.full((C, 1), 1000.0, dtype=torch.float64, device="cuda")
        e.mcmc_init(q0, data, [0.0], [1.0e4], seed=2025, prior_len=3)
        tq = torch.empty((block, C, 1), dtype=torch.float64, device="cuda")
        t0 = time.perf_counter()
        s1 = torch.zeros(C, dtype=torch.float64, device="cuda")
        s2 = torch.zeros(C, dtype=torch.float64, device="cuda")
        kept = 0
        for b in range(nblocks):
            e.mcmc_run(block, out=(tq, None, None))
            if b >= 1:  # first block = burn-in
                x = tq[:, :, 0]
                s1 += x.sum(dim=0)
                s2 += (x * x).sum(dim=0)
                kept += block
        e.sync()
        dt = time.perf_counter() - t0
        st = e.stats()
    chain_mean = (s1 / kept).cpu().numpy()
    mean = float(chain_mean.mean())
    var = float((s2.sum() / (kept * C) - (s1.sum() / (kept * C)) ** 2).cpu())
    out["gpu"] = dict(chains=C, proposals_per_chain=block * nblocks, kept_per_chain=kept, mean=mean, std=var ** 0.5,
                      se_mean=batch_means_se(chain_mean), accept=st["accepted"] / (st["iters_done"] * C), nonfinite=st["nonfinite"],
                      seconds=dt, ode_steps_x_chains_per_s=C * block * nblocks * 500 / dt)
    # --- oracle: 4096 chains x 1500 proposals (other seed), burn 500
    Co, n, burn = 4096, 1500, 500
    with pkg.Engine(lib=lib, checker=True) as e:
        e.set_model(model, 1)
        e.mcmc_init(np.full((Co, 1), 1000.0), data, [0.0], [1.0e4], seed=777, prior_len=3)
        t0 = time.perf_counter()
        tq, _, _ = e.mcmc_run(n, traces=("q",))
        dto = time.perf_counter() - t0
        sto = e.stats()
    x = tq[burn:, :, 0]
    out["oracle"] = dict(chains=Co, proposals_per_chain=n, kept_per_chain=n - burn, mean=float(x.mean()), std=float(x.std()),
                         se_mean=batch_means_se(x.mean(axis=0)), accept=sto["accepted"] / (sto["iters_done"] * Co), seconds=dto)
    g, o = out["gpu"], out["oracle"]
    out["agreement"] = dict(mean_diff=g["mean"] - o["mean"], mean_diff_in_combined_se=(g["mean"] - o["mean"]) / (g["se_mean"] ** 2 + o["se_mean"] ** 2) ** 0.5,
                            std_ratio=g["std"] / o["std"], accept_diff=g["accept"] - o["accept"])
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt)


if __name__ == "__main__":
    main()
