#!/usr/bin/env python3
"""float32-vs-float64 tolerance sweep of BASELINE config 5 AT ITS OWN per-GPU SHAPE: joint (Dc, a, b), 131 072 chains
(1 048 576 / 8 GPUs), nsteps 4000 — on one MI355X, device-resident buffers.

  python tools/fp32_sweep_cfg5.py [out.json] [--iters 400]

1. sum of squares: 131 072 lanes with (Dc, a, b) drawn inside the prior box, float32 solve vs float64 solve of the same
   lanes → max / median relative |SSq32 - SSq64|;
2. sampler: 131 072 three-parameter chains x `iters` proposals in both precisions with identical Philox seeds (adaptive
   Metropolis, the init kernel's own start covariance) → posterior mean / std drift in units of the float64 posterior std,
   acceptance rates, and the throughput of both runs (ODE-steps x chains / s) — once with every chain started at the truth,
   once with the chains started away from it, dispersed over the prior box.
tests/test_gpu_parity.py::test_float32_tolerance_at_config5_shape asserts the bands on a shorter run of the same code."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHAINS, NSTEPS = 131072, 4000
LO, HI = [0.0, 0.005, 0.005], [1.0e4, 0.02, 0.03]   # prior box; truth (1000, 0.011, 0.014)


def sweep(iters=400, chains=CHAINS, nsteps=NSTEPS, seed=9):
    import torch

    import bayesian_markov_chain_monte_carlo_amd as pkg

    out = {"shape": {"chains": chains, "nsteps": nsteps, "n_params": 3, "iters": iters}}
    m = {p: pkg.RateStateModel(nsteps) for p in ("float64", "float32")}
    m["float32"].precision = "float32"
    with pkg.Engine(mem="host") as e:
        e.set_model(m["float64"], 1)
        _, ref = e.forward([1000.0])
    ref = ref[:, 0]
    data = ref + np.abs(ref) * np.random.default_rng(2025).standard_normal(ref.shape[0])
    rng = np.random.default_rng(5)
    dc = torch.as_tensor(rng.uniform(100.0, 9000.0, chains)).cuda()
    a = torch.as_tensor(rng.uniform(0.008, 0.016, chains))
    b = (a + torch.as_tensor(rng.uniform(0.0, 0.008, chains))).cuda()   # b - a < 0.1: stable sliding
    a = a.cuda()
    d_dev = torch.as_tensor(data).cuda()
    ssq = {}
    for p in ("float64", "float32"):
        with pkg.Engine(mem="device") as e:
            e.set_model(m[p], 1)
            s, _ = e.forward(dc, a=a, b=b, data=d_dev, want_ssq=True, want_acc=False)
            e.sync()
            ssq[p] = s.cpu().numpy()
    fin = np.isfinite(ssq["float64"])
    rel = np.abs(ssq["float32"][fin] - ssq["float64"][fin]) / ssq["float64"][fin]
    out["ssq"] = {"lanes": int(fin.sum()), "nonfinite_f64": int((~fin).sum()), "nonfinite_f32": int((~np.isfinite(ssq["float32"])).sum()),
                  "rel_max": float(rel.max()), "rel_p999": float(np.quantile(rel, 0.999)), "rel_median": float(np.median(rel))}
    out["posterior"] = sampler_leg(pkg, torch, m, d_dev, chains, nsteps, iters, seed,
                                   torch.tensor([1000.0, 0.011, 0.014], dtype=torch.float64, device="cuda").repeat(chains, 1))
    # the same with chains started AWAY from the truth, dispersed over the box (the same points in both precisions): the two
    # precisions must also agree while the chains travel and the adaptive proposal forms
    r2 = np.random.default_rng(77)
    a0 = r2.uniform(0.008, 0.016, chains)
    q0d = np.column_stack([r2.uniform(300.0, 3000.0, chains), a0, a0 + r2.uniform(0.0, 0.008, chains)])
    out["posterior_dispersed_starts"] = sampler_leg(pkg, torch, m, d_dev, chains, nsteps, iters, seed, torch.as_tensor(q0d).cuda())
    return out


def sampler_leg(pkg, torch, m, d_dev, chains, nsteps, iters, seed, q0):
    res = {}
    for p in ("float64", "float32"):
        with pkg.Engine(mem="device") as e:
            e.set_model(m[p], 1)
            # the init kernel's own proposal covariance (prior-regularised, float64 sensitivities in both precisions): nothing hand-set
            e.mcmc_init(q0, d_dev, LO, HI, seed=seed, prior_len=3, adapt_mode="am", adapt_interval=20, fd_rel_step=1e-4)
            e.mcmc_run(2, traces=False)
            e.sync()
            t0 = time.perf_counter()
            tq, _, ta = e.mcmc_run(iters, traces=("q", "accept"))
            e.sync()
            dt = time.perf_counter() - t0
            kept = tq[iters // 2:].reshape(-1, 3)
            st = e.stats()
            res[p] = dict(mean=kept.mean(dim=0).tolist(), std=kept.std(dim=0).tolist(), accept=float(ta.float().mean()),
                          nonfinite=int(st["nonfinite"]), ode_steps_x_chains_per_s=chains * iters * nsteps / dt, seconds=dt)
            del tq, ta, kept
            torch.cuda.empty_cache()
    drift = {k: (np.abs(np.array(res["float32"][k]) - np.array(res["float64"][k])) / np.array(res["float64"]["std"])).tolist()
             for k in ("mean", "std")}
    return dict(res, drift_in_units_of_f64_posterior_std=drift,
                accept_diff=abs(res["float32"]["accept"] - res["float64"]["accept"]),
                speedup_f32=res["float32"]["ode_steps_x_chains_per_s"] / res["float64"]["ode_steps_x_chains_per_s"])


if __name__ == "__main__":
    args = [x for x in sys.argv[1:] if not x.startswith("--")]
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 400
    if "--iters" in sys.argv:
        args = [x for x in args if x != str(iters)]
    txt = json.dumps(sweep(iters), indent=1)
    print(txt)
    if args:
        open(args[0], "w").write(txt)
