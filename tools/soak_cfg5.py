#!/usr/bin/env python3
"""Statistical validation of the joint (Dc, a, b) inference (BASELINE config 5) — the three-parameter analogue of
tools/soak_tier3.py.  Chains start AWAY from the truth, with the proposal covariance the init kernel itself provides
(prior-regularised Gauss-Newton, csrc/rsf_kernels.h::initial_covariance) — nothing hand-set.

  1. GPU pool vs independent chains of the CPU oracle (other seed, same lengths), for the fixed initial covariance ("none":
     plain Metropolis) and for "am": pooled means of Dc, a, b and of the one combination the data identify, Dc*a, must agree
     within 3 combined Monte-Carlo standard errors (from the spread of the chains' own means), spreads and acceptance closely.
  2. What adaptation buys: integrated autocorrelation time (batch means) -> effective samples per proposal, "am" against
     "none", same chains otherwise.
  3. Whether adaptation leaves the target alone: the non-adaptive chains are plain Metropolis, their pool IS the posterior;
     the difference of the "am" pool from it is reported in standard errors.  (Round 4 found the window-only adaptation "am"
     first had — the covariance of the LAST adapt_interval samples — 89 standard errors off in the mean of Dc*a and 8 % narrow;
     "am" is the whole-history scheme of Haario et al. since.)

  python tools/soak_cfg5.py [out.json]      (GPU box; ~60 s of the host's cores for the oracle)"""
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

TRUTH = [1000.0, 0.011, 0.014]
START = [1600.0, 0.008, 0.022]
LO, HI = [0.0, 0.005, 0.005], [1.0e4, 0.02, 0.03]
NAMES = ("Dc", "a", "b", "Dc*a")


def quantities(tq):
    """(n, C, 3) trace -> (n, C, 4): Dc, a, b and the identified combination Dc*a."""
    return np.concatenate([tq, (tq[:, :, 0] * tq[:, :, 1])[:, :, None]], axis=2)


def pool(x):
    """x (n, C, k) -> per quantity: pooled mean, pooled sd, standard error of the mean from the spread of the chains' means."""
    cm = x.mean(axis=0)
    return dict(mean=cm.mean(axis=0).tolist(), sd=x.reshape(-1, x.shape[2]).std(axis=0).tolist(),
                se=(cm.std(axis=0, ddof=1) / np.sqrt(cm.shape[0])).tolist())


def ess_per_proposal(x, batch=200):
    """Batch-means integrated autocorrelation time per chain, median over chains -> effective samples per proposal."""
    n = (x.shape[0] // batch) * batch
    xb = x[:n].reshape(n // batch, batch, *x.shape[1:])
    tau = batch * xb.mean(axis=1).var(axis=0, ddof=1) / np.maximum(x[:n].var(axis=0, ddof=1), 1e-300)
    return (1.0 / np.median(np.maximum(tau, 1.0), axis=0)).tolist()


def run(engine_kw, data, model, C, n, burn, seed, adapt, as_numpy):
    with pkg.Engine(**engine_kw) as e:
        e.set_model(model, 1)
        e.mcmc_init(np.tile(START, (C, 1)), data, LO, HI, seed=seed, prior_len=3, adapt_mode=adapt, adapt_interval=20, fd_rel_step=1e-4)
        V0 = np.asarray(as_numpy(e.get_state()[3]))[0]
        t0 = time.perf_counter()
        tq, _, ta = e.mcmc_run(n, traces=("q", "accept"))
        e.sync()
        dt = time.perf_counter() - t0
        tq, ta, cnt = as_numpy(tq), as_numpy(ta), e.counters()
    x = quantities(tq[burn:])
    return dict(chains=C, proposals=n, burn=burn, adapt=adapt, seconds=dt, acceptance=float(ta[burn:].mean()),
                out_of_bounds=cnt["out_of_bounds"] / (C * n), nonfinite=cnt["nonfinite"], pool=pool(x), ess_per_proposal=ess_per_proposal(x),
                V0_sd=np.sqrt(np.diag(V0)).tolist(), V0_corr_Dc_a=float(V0[0, 1] / np.sqrt(V0[0, 0] * V0[1, 1]))), x


def diff_in_se(a, b):
    return [(ma - mb) / (sa ** 2 + sb ** 2) ** 0.5 for ma, mb, sa, sb in zip(a["mean"], b["mean"], a["se"], b["se"])]


def main():
    lib = pkg._abi.bind(ctypes.CDLL(rsf_oracle.lib_path()))
    model = pkg.RateStateModel(500)
    with pkg.Engine(mem="host") as e:
        e.set_model(model, 1)
        _, acc = e.forward([TRUTH[0]], a=[TRUTH[1]], b=[TRUTH[2]])
    acc = acc[:, 0]
    data = acc + np.abs(acc) * np.random.default_rng(2025).standard_normal(acc.shape[0])
    to_np = lambda t: t.cpu().numpy() if hasattr(t, "cpu") else np.asarray(t)  # noqa: E731
    out = {"truth": TRUTH, "start": START, "box": [LO, HI], "quantities": NAMES, "nsteps": 500}
    n, burn = 6000, 2000
    out["gpu_am"], _ = run(dict(mem="device"), data, model, 16384, n, burn, 2025, "am", to_np)
    out["gpu_fixed"], _ = run(dict(mem="device"), data, model, 16384, n, burn, 2025, "none", to_np)
    out["oracle_am"], _ = run(dict(lib=lib, checker=True), data, model, 768, n, burn, 777, "am", to_np)
    out["oracle_fixed"], _ = run(dict(lib=lib, checker=True), data, model, 768, n, burn, 777, "none", to_np)
    g, f = out["gpu_am"]["pool"], out["gpu_fixed"]["pool"]
    for mode, gp in (("am", g), ("fixed", f)):
        o = out["oracle_" + mode]["pool"]
        out["agreement_gpu_vs_oracle_" + mode] = dict(mean_diff_in_combined_se=diff_in_se(gp, o), sd_ratio=[a / b for a, b in zip(gp["sd"], o["sd"])],
                                                      acceptance_diff=out["gpu_" + mode]["acceptance"] - out["oracle_" + mode]["acceptance"])
    out["am_vs_fixed_proposal"] = dict(mean_diff_in_combined_se=diff_in_se(g, f), sd_ratio=[a / b for a, b in zip(g["sd"], f["sd"])],
                                       ess_gain=[a / b for a, b in zip(out["gpu_am"]["ess_per_proposal"], out["gpu_fixed"]["ess_per_proposal"])])
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt)


if __name__ == "__main__":
    main()
