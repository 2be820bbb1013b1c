#!/usr/bin/env bash
# A/B on the bench: alternating trunk priorities at the MCTS's small launches together with a step kernel above them.
run() { AQG_TRUNK_PRIO=$1 AQG_STEP_PRIO=$2 AQG_HEADS_PRIO=$3 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('trunk_prio $1 step_prio $2 heads_prio $3:', round(d['value'],1), 'games/s')"; }
run -1 1 0; run 8 2 0; run 8 2 1; run 8 3 1; run -1 1 0; run 8 2 1
