#!/usr/bin/env bash
# Experiment: 16,384 concurrent games as 2 / 3 / 4 game sets and 4 / 8 games per step workgroup -> games/s of one generation (one box).
for cfg in "4 4" "2 4" "3 4" "4 8" "2 8"; do
  set -- $cfg
  AQG_STEP_WAVES=$2 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs --games 16384 --sets $1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('16384 games, sets $1, step_waves $2:', round(d['value'],1), 'games/s')"
done
