#!/usr/bin/env bash
# Experiment: games per workgroup of the fast MCTS step kernel (1, 2, 4 or 8 wavefronts) -> bench games/s, same box, one call.
for w in 4 8 4 8 2; do
  AQG_STEP_WAVES=$w python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step_waves $w:', round(d['value'],1), 'games/s', round(d['leaf_evals_per_s']/1e6,2), 'M evals/s')"
done
