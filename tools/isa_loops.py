#!/usr/bin/env python3
"""List the loops of one kernel in a device ISA listing with their instruction mix (no GPU needed).

  hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -o /tmp/rsf.s csrc/rsf_hip.hip
  python tools/isa_loops.py /tmp/rsf.s mcmc_kernelILi1ELb1ELb0ELi0E [min_valu] [max_valu]

A loop = a backward branch to a label; reported: VALU / SALU / LDS instruction counts and the commonest opcodes.  The TIGHT
hot loop of the one-parameter sampler holds 16 RK4 steps per trip, so VALU / 16 is the per-step figure DESIGN.md quotes."""
import collections
import re
import sys


def main():
    path, kern = sys.argv[1], sys.argv[2]
    lo = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    hi = int(sys.argv[4]) if len(sys.argv) > 4 else 4000
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if kern in l and l.rstrip().endswith(":") or (kern in l and l.startswith("_Z") and ":" in l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    seen = set()
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if not (m and m.group(1) in labels and labels[m.group(1)] < i):
            continue
        a = labels[m.group(1)]
        if (a, i) in seen:
            continue
        seen.add((a, i))
        ins = [x.split()[0] for x in body[a:i + 1] if x.startswith("\t") and not x.strip().startswith((";", "."))]
        c = collections.Counter(ins)
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        if lo <= valu <= hi:
            salu = sum(v for k, v in c.items() if k.startswith("s_"))
            ds = sum(v for k, v in c.items() if k.startswith("ds_"))
            print(f"{m.group(1)} @line {start + a}: VALU {valu} SALU {salu} LDS {ds} | " + ", ".join(f"{k} {v}" for k, v in c.most_common(12)))


if __name__ == "__main__":
    main()
