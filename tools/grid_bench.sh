#!/usr/bin/env bash
# Developer sweep: headline generation per cap of the trunk's persistent grid (bench.py --trunk-grid), one box.
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
for g in 0 448 384 320 256; do
  timeout -k 10 200 python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --trunk-grid $g 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('trunk_grid $g: %.1f games/s, trunk launch %.2f us' % (d['value'], d['roofline']['avg_launch_us']))"
done
