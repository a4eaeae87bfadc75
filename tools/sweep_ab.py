#!/usr/bin/env python3
"""
The reference's main.py sweep (bench.main_py_sweep) on the in-tree library and on other builds of it, one process.

  python tools/sweep_ab.py [--chains-per-group C] [--burn B] [--ips I] [--launches L] [name=path.so ...] > out.json

A build that predates rsf_mcmc_counters (round 3's, kept as build/base_96c1.so) is bound with the symbols it has: its rows
carry times and the old statistics only.  Prints one JSON object {name: sweep}.
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def bind_available(pkg, path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, (restype, argtypes) in pkg._abi.PROTOTYPES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = restype, argtypes
    assert lib.rsf_backend() == b"hip-gfx950", path
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains-per-group", type=int, default=0)
    ap.add_argument("--burn", type=int, default=None)
    ap.add_argument("--ips", type=int, default=0)
    ap.add_argument("--launches", type=int, default=0)
    ap.add_argument("variants", nargs="*")
    args = ap.parse_args()

    import bayesian_markov_chain_monte_carlo_amd as pkg
    import bench

    libs = {"in_tree": None}
    pkg._abi.load()  # binds the process's one HIP runtime before any other build is opened
    for v in args.variants:
        name, path = v.split("=", 1)
        libs[name] = bind_available(pkg, path)
    out = {}
    for name, lib in libs.items():
        out[name] = bench.main_py_sweep(pkg, lib=lib, chains_per_group=args.chains_per_group or None, burn=args.burn,
                                        ips=args.ips or None, launches=args.launches or None)
        out[name]["build_id"] = (lib or pkg._abi.load()).rsf_build_id().decode()
        g = out[name]["groups"]
        sys.stderr.write(f"{name}: all groups {out[name]['all_groups_one_launch']['value']:.3e}; " +
                         ", ".join(f"{k} {v['value']:.3e} (eval {v['evaluated_fraction']:.2f}, acc {v['acceptance']:.3f})" for k, v in g.items()) + "\n")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
