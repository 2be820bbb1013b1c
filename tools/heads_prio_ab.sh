#!/usr/bin/env bash
# A/B on the bench: wave priorities of the step kernel and the heads kernel ("<step>,<heads>" pairs, 0..3 each; defaults 1,1).
for sh in ${PAIRS:-1,0 1,1 1,2 2,2 1,3 2,3 1,0 1,1}; do
  AQG_STEP_PRIO=${sh%,*} AQG_HEADS_PRIO=${sh#*,} python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --large-games 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step_prio,heads_prio $sh:', round(d['value'],1), 'games/s', round(d['leaf_evals_per_s']/1e6,2), 'M evals/s')" || exit 1
done
