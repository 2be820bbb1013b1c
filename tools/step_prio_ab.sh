#!/usr/bin/env bash
# A/B on the bench: wave priority of the MCTS step kernel (0 = default arbitration, 1..3 = above the trunk's waves).
for p in 0 1 3 0 1 3; do
  AQG_STEP_PRIO=$p python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step_prio $p:', round(d['value'],1), 'games/s', round(d['leaf_evals_per_s']/1e6,2), 'M evals/s')"
done
