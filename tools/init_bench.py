#!/usr/bin/env python3
"""
Time of rsf_mcmc_init (the initial solve, sigma^2_0 and the sensitivity-based proposal covariance of MCMC.py:229-262 for
every chain) per build, interleaved in one process:

  python tools/init_bench.py [--chains C] [--nsteps N] [--params 1|3] [--integrator rk4|dop853] [name=path.so ...]

"default" is the in-tree library.  Prints milliseconds per call (median of --rounds) and trajectory-steps/s
(chains x (1 + params) trajectories x nsteps).
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=131072)
    ap.add_argument("--nsteps", type=int, default=4000)
    ap.add_argument("--params", type=int, default=3, choices=[1, 3])
    ap.add_argument("--integrator", default="rk4", choices=["rk4", "dop853"])
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("variants", nargs="*")
    args = ap.parse_args()

    import torch

    import bayesian_markov_chain_monte_carlo_amd as pkg
    from bench import synthetic_problem

    libs = {"default": pkg._abi.load()}
    for v in args.variants:
        name, path = v.split("=", 1)
        lib = ctypes.CDLL(os.path.abspath(path))
        for sym, (restype, argtypes) in pkg._abi.PROTOTYPES.items():
            if hasattr(lib, sym):
                getattr(lib, sym).restype, getattr(lib, sym).argtypes = restype, argtypes
        libs[name] = lib
    model, data = synthetic_problem(args.nsteps)
    model.integrator = args.integrator
    C, d = args.chains, args.params
    rng = np.random.default_rng(3)
    q0 = torch.tensor(np.column_stack([rng.uniform(400.0, 2500.0, C), rng.uniform(0.009, 0.014, C), rng.uniform(0.012, 0.018, C)])[:, :d],
                      dtype=torch.float64, device="cuda")
    lo, hi = [0.0, 0.005, 0.005][:d], [1.0e4, 0.02, 0.03][:d]
    engines = {}
    for name, lib in libs.items():
        e = pkg.Engine(lib=lib, mem="device")
        e.set_model(model, 1)
        engines[name] = e
    times = {n: [] for n in engines}
    for r in range(args.rounds + 1):
        for name, e in engines.items():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e.mcmc_init(q0, data, lo, hi, seed=1, prior_len=3 if d == 1 else 0, fd_rel_step=1e-6 if d == 1 else 1e-4)
            torch.cuda.synchronize()
            if r:
                times[name].append(time.perf_counter() - t0)
    for name, ts in times.items():
        med = float(np.median(ts))
        print(f"{name:10s} {1e3 * med:9.2f} ms  (min {1e3 * min(ts):.2f})   {C * (1 + d) * args.nsteps / med:.3e} trajectory-steps/s", flush=True)


if __name__ == "__main__":
    main()
