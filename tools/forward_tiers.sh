#!/bin/bash
# Forward solve (SSq only, and with the trajectory stored) across Dc ranges, i.e. across integration tiers, 262 144 lanes x nsteps 500:
#   tools/forward_tiers.sh > profiles/rNN/forward_tiers.log        (GPU box)
for r in "800 1200" "300 500" "150 250" "60 120" "20 40" "3 12"; do
  timeout -k 10 120 python tools/forward_bench.py --nsteps 500 --dc $r 2>/dev/null | python3 -c "
import json,sys
o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(o['dc'], '%.3e ssq-only  %.3e traj'%(o['ssq_only']['rk4_steps_per_s'],o['trajectory']['rk4_steps_per_s']))"
done
