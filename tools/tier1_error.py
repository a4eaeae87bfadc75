#!/usr/bin/env python3
"""Measured Tier-1 agreement: HIP forward solve vs the CPU oracle (same RK4, same substeps) — largest relative trajectory and
sum-of-squares differences over a spread of Dc (and (a, b)) for the BASELINE series lengths.  Run on the GPU box:

  python tools/tier1_error.py [out.json]

The tests assert rtol 1e-9; this prints what the agreement actually is (DESIGN.md quotes it)."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import bayesian_markov_chain_monte_carlo_amd as pkg  # noqa: E402
import rsf_oracle  # noqa: E402  (the checker: this is a measurement tool, not the product path)


def main():
    lib = pkg._abi.bind(ctypes.CDLL(rsf_oracle.lib_path()))
    rng = np.random.default_rng(11)
    out = {}
    for n in (500, 2000, 4000):
        m = pkg.RateStateModel(n)
        C = 4096
        dc = np.exp(rng.uniform(np.log(150.0), np.log(9000.0), C))
        a = rng.uniform(0.008, 0.016, C)
        b = a + rng.uniform(0.0, 0.008, C)
        with pkg.Engine(mem="host") as g, pkg.Engine(lib=lib, checker=True) as c:
            g.set_model(m, 1)
            c.set_model(m, 1)
            _, ref = c.forward([1000.0])
            data = ref[:, 0] + np.abs(ref[:, 0]) * np.random.default_rng(2025).standard_normal(ref.shape[0])
            # "tight": every wave on the TIGHT tier's scaled step (all Dc > 512, rsf_device.h start_tier) — the sampler's
            # steady state; the random spread above puts a small-Dc lane into every wave, i.e. measures the wider tiers
            dct = np.sort(np.exp(rng.uniform(np.log(600.0), np.log(3000.0), C)))
            h = 50.0 / n
            edge = 1.2 * h * 0.1 / 0.011 * 512.0  # Dc at the TIGHT tier's a-priori bound (a = 0.011); NARROW's is a quarter of it
            dcn = np.sort(np.exp(rng.uniform(np.log(0.3 * edge), np.log(0.9 * edge), C)))  # "narrow": every wave on the NARROW tier
            dcw = np.sort(np.exp(rng.uniform(np.log(0.07 * edge), np.log(0.22 * edge), C)))  # "wide": below NARROW's bound (edge / 4)
            for tag, kw, dcs in (("dc_only", {}, dc), ("dc_a_b", dict(a=a, b=b), dc), ("tight_dc_only", {}, dct),
                                 ("tight_dc_a_b", dict(a=a, b=b), dct), ("narrow_dc_only", {}, dcn), ("wide_dc_only", {}, dcw)):
                sg, ag = g.forward(dcs, data=data, want_ssq=True, want_acc=True, **kw)
                sc, ac = c.forward(dcs, data=data, want_ssq=True, want_acc=True, **kw)
                traj = np.abs(ag - ac).max(axis=0) / np.abs(ac).max(axis=0)
                ssq = np.abs(sg - sc) / sc
                out[f"nsteps_{n}_{tag}"] = dict(traj_rel_max=float(traj.max()), traj_rel_median=float(np.median(traj)),
                                                ssq_rel_max=float(ssq.max()), ssq_rel_median=float(np.median(ssq)), lanes=C)
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(txt)


if __name__ == "__main__":
    main()
