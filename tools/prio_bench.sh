#!/usr/bin/env bash
# Developer sweep: headline generation (bench.py, 2 timed steps, no extra legs) per static trunk priority mode on ONE box.
# trunk_prio bits: 1 = waves 4-7 up, 2 = second-resident workgroups up, 4 = first-resident up, 8 = the two workgroups of a CU alternate per phase
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for p in -1 1 2 8 9 3; do
  if [ "$p" = "-1" ]; then unset AQG_TRUNK_PRIO; else export AQG_TRUNK_PRIO=$p; fi
  timeout -k 10 200 python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('trunk_prio $p rep $rep: %.1f games/s, trunk launch %.2f us' % (d['value'], d['roofline']['avg_launch_us']))"
done
done
