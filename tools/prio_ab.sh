#!/usr/bin/env bash
# A/B on the bench: static priority of the trunk's waves against the other sets' step / heads waves (AQG_TRUNK_PRIO bits: 1 = waves
# 4-7, 2 = second-resident workgroups, 4 = first-resident; 6 = every trunk wave at priority 1; -1 = by launch size).
for p in -1 8 -1 8; do
  AQG_TRUNK_PRIO=$p python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('trunk_prio $p:', round(d['value'],1), 'games/s', round(d['leaf_evals_per_s']/1e6,2), 'M evals/s')"
done
