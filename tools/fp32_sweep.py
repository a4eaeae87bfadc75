#!/usr/bin/env python3
"""float32 vs float64 tolerance sweep (BASELINE config 5) on the GPU.

  python tools/fp32_sweep.py [out.json]

For nsteps in {500, 2000, 4000}: max / median relative |SSq32 - SSq64| and trajectory error over a grid of
(Dc, a, b); then posterior-moment drift of 3-parameter chains run with identical seeds in both precisions,
and the throughput ratio of the two solves."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bayesian_markov_chain_monte_carlo_amd as pkg  # noqa: E402


def main():
    out = {}
    rng = np.random.default_rng(5)
    for n in (500, 2000, 4000):
        m64, m32 = pkg.RateStateModel(n), pkg.RateStateModel(n)
        m32.precision = "float32"
        C = 4096
        dc = rng.uniform(100.0, 9000.0, C)
        a = rng.uniform(0.008, 0.016, C)
        b = a + rng.uniform(0.0, 0.008, C)
        with pkg.Engine(mem="host") as e64, pkg.Engine(mem="host") as e32:
            e64.set_model(m64, 1)
            e32.set_model(m32, 1)
            _, ref = e64.forward([1000.0])
            ref = ref[:, 0]
            data = ref + np.abs(ref) * np.random.default_rng(2025).standard_normal(ref.shape[0])
            s64, a64 = e64.forward(dc, a=a, b=b, data=data, want_ssq=True, want_acc=True)
            s32, a32 = e32.forward(dc, a=a, b=b, data=data, want_ssq=True, want_acc=True)
            rel = np.abs(s32 - s64) / s64
            terr = np.abs(a32 - a64).max(axis=0) / np.abs(a64).max(axis=0)
            t = {}
            for name, e in (("f64", e64), ("f32", e32)):
                big = np.full(65536, 1000.0)
                e.forward(big, data=data, want_ssq=True, want_acc=False)
                t0 = time.perf_counter()
                e.forward(big, data=data, want_ssq=True, want_acc=False)
                t[name] = time.perf_counter() - t0
            out[f"nsteps_{n}"] = dict(ssq_rel_max=float(rel.max()), ssq_rel_median=float(np.median(rel)),
                                      traj_rel_max=float(terr.max()), traj_rel_median=float(np.median(terr)),
                                      forward_65536_lanes_s=t, speedup_f32=t["f64"] / t["f32"])
    # posterior-moment drift, 3 parameters (Dc, a, b), same seeds
    n = 500
    res = {}
    for prec in ("float64", "float32"):
        m = pkg.RateStateModel(n)
        m.precision = prec
        with pkg.Engine(mem="host") as e:
            e.set_model(m, 1)
            if prec == "float64":
                _, ref = e.forward([1000.0])
                ref = ref[:, 0]
                data = ref + np.abs(ref) * np.random.default_rng(2025).standard_normal(ref.shape[0])
            C = 8192
            q0 = np.tile([1000.0, 0.011, 0.014], (C, 1))
            e.mcmc_init(q0, data, [0.0, 0.005, 0.005], [1e4, 0.02, 0.03], seed=9, adapt_mode="am", adapt_interval=10)
            # (X^T X)^-1 is near-singular for (Dc, a, b): start from an explicit proposal covariance instead
            e.set_state(V=np.tile(np.diag([20.0 ** 2, 1e-4 ** 2, 1e-4 ** 2]), (C, 1, 1)))
            tq, _, ta = e.mcmc_run(400, traces=("q", "accept"))
            kept = tq[200:].reshape(-1, 3)
            res[prec] = dict(mean=kept.mean(axis=0).tolist(), std=kept.std(axis=0).tolist(), accept=float(ta.mean()))
    drift = {k: (np.abs(np.array(res["float32"][k]) - np.array(res["float64"][k])) / np.array(res["float64"]["std"])).tolist()
             for k in ("mean", "std")}
    out["posterior_3param"] = dict(res, drift_in_units_of_posterior_std=drift)
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt)


if __name__ == "__main__":
    main()
