#!/usr/bin/env python3
"""
A/B timing of kernel builds in ONE process with interleaved rounds (same device, same clocks).

  python tools/ab_bench.py [--chains C] [--nsteps N] [--iters I] [--rounds R] name=path.so ...

Each variant is a build of an edited copy of csrc/ (the kernels' tunables are constexpr values in rsf_device.h /
rsf_kernels.h: copy the directory, change one, `make`, pass the .so here); "default" is the in-tree librsf_hip.so.  Prints median / min milliseconds per launch and ODE-steps*chains/s.
"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=65536)
    ap.add_argument("--nsteps", type=int, default=500)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--params", type=int, default=1, choices=[1, 3], help="1: Dc;  3: (Dc, a, b)")
    ap.add_argument("--substeps", type=int, default=1, help="RK4 steps per output interval")
    ap.add_argument("--integrator", default="rk4", choices=["rk4", "dop853"])
    ap.add_argument("--precision", default="float64", choices=["float64", "float32"])
    ap.add_argument("variants", nargs="*")
    args = ap.parse_args()

    import torch

    import bayesian_markov_chain_monte_carlo_amd as pkg
    from bench import synthetic_problem

    libs = {"default": pkg._abi.load()}
    for v in args.variants:
        name, path = v.split("=", 1)
        lib = ctypes.CDLL(os.path.abspath(path))
        for sym, (restype, argtypes) in pkg._abi.PROTOTYPES.items():  # an older build is bound with the symbols it has
            if hasattr(lib, sym):
                getattr(lib, sym).restype, getattr(lib, sym).argtypes = restype, argtypes
        assert lib.rsf_backend() == b"hip-gfx950", path
        libs[name] = lib
    model, data = synthetic_problem(args.nsteps)
    model.precision = args.precision
    model.integrator = args.integrator
    C, ips, d = args.chains, args.iters, args.params
    engines, traces = {}, (torch.empty((ips, C, d), dtype=torch.float64, device="cuda"),
                           torch.empty((ips, C), dtype=torch.float64, device="cuda"), None)
    q0 = torch.tensor([1000.0, 0.011, 0.014][:d], dtype=torch.float64, device="cuda").repeat(C, 1)
    V0 = torch.diag(torch.tensor([20.0 ** 2, 1e-4 ** 2, 1e-4 ** 2][:d], dtype=torch.float64, device="cuda")).repeat(C, 1, 1)
    for name, lib in libs.items():
        e = pkg.Engine(lib=lib, mem="device")
        e.set_model(model, args.substeps)
        e.mcmc_init(q0, data, [0.0, 0.005, 0.005][:d], [1.0e4, 0.02, 0.03][:d], seed=2025, prior_len=3 if d == 1 else 0)
        if d == 3:
            e.set_state(V=V0)
        e.mcmc_run(ips, out=traces)  # warm-up
        engines[name] = e
    torch.cuda.synchronize()
    times = {n: [] for n in engines}
    for _ in range(args.rounds):
        for name, e in engines.items():
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            e.mcmc_run(ips, out=traces)
            b.record()
            torch.cuda.synchronize()
            times[name].append(a.elapsed_time(b))
    work = C * ips * args.nsteps * args.substeps  # RK4 steps
    print(f"chains={C} nsteps={args.nsteps} substeps={args.substeps} iters/launch={ips} rounds={args.rounds} params={d} {args.precision}")
    for name, t in times.items():
        med, mn = float(np.median(t)), float(np.min(t))
        line = f"  {name:16s} median {med:9.3f} ms  min {mn:9.3f} ms   {work / (med * 1e-3):.4e} steps*chains/s"
        try:   # where the wave-steps went (builds that have rsf_mcmc_counters)
            c = engines[name].counters()
            line += "   wave-steps " + " ".join(f"{k[6:]}={c[k]:.3g}" for k in ("steps_tight", "steps_narrow", "steps_wide", "steps_full", "steps_redone") if c.get(k))
        except Exception:
            pass
        print(line)


if __name__ == "__main__":
    main()
