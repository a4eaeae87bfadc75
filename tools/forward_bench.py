#!/usr/bin/env python3
"""
The one HBM-significant variant of the path: rsf_forward_batch with the trajectory stored
(acc[nout][C], 8 B per RK4 step and lane, time-major so that a wave stores 512 contiguous bytes per step), next to
the SSq-only solve of the same lanes.

    python tools/forward_bench.py [--lanes 262144] [--nsteps 2000] [--rounds 5]
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
    ap.add_argument("--lanes", type=int, default=262144)
    ap.add_argument("--nsteps", type=int, default=2000)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--dc", type=float, nargs=2, default=[800.0, 1200.0],
                    help="Dc range of the lanes: below ~560 (nsteps 500) / ~140 (nsteps 2000) the NARROW tier runs, below a quarter of that WIDE")
    args = ap.parse_args()

    import torch

    import bayesian_markov_chain_monte_carlo_amd as rsf
    from bench import synthetic_problem

    model, data = synthetic_problem(args.nsteps)
    C = args.lanes
    dc = torch.linspace(args.dc[0], args.dc[1], C, dtype=torch.float64, device="cuda")
    out = {"lanes": C, "nsteps": args.nsteps, "dc": args.dc}
    with rsf.Engine(mem="device") as e:
        nout = e.set_model(model, 1)
        for name, kw in (("ssq_only", dict(data=data, want_ssq=True, want_acc=False)), ("trajectory", dict(want_acc=True))):
            e.forward(dc, **kw)
            e.sync()
            best = None
            for _ in range(args.rounds):
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                r = e.forward(dc, **kw)
                ev1.record()
                torch.cuda.synchronize()
                ms = ev0.elapsed_time(ev1)
                best = ms if best is None else min(best, ms)
                del r
            steps = C * (nout - 1)
            out[name] = {"ms": best, "rk4_steps_per_s": steps / (best * 1e-3)}
            if name == "trajectory":
                out[name]["write_GBps"] = C * nout * 8 / (best * 1e-3) / 1e9
    print(json.dumps(out))


if __name__ == "__main__":
    main()
