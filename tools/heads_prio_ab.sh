#!/usr/bin/env bash
# A/B on the bench: wave priority 1 for the heads kernel (step kernel at its default priority 1).
for p in 0 1 0 1; do
  AQG_HEADS_PRIO=$p python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('heads_prio $p:', round(d['value'],1), 'games/s', round(d['leaf_evals_per_s']/1e6,2), 'M evals/s')"
done
