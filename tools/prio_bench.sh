#!/usr/bin/env bash
# Headline generation (bench.py, no extra legs) per static trunk priority mode (option trunk_prio; -1 = by launch size), on one box.
for p in ${MODES:--1 0 1 2 4 8 -1}; do
  AQG_TRUNK_PRIO=$p python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --large-games 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('trunk_prio $p:', round(d['value'],1), 'games/s, trunk launch', round(d['roofline']['avg_launch_us'],2), 'us, gnn_forward', round(d['gnn_forward']['boards_per_s']/1e6,2), 'M boards/s')" || exit 1
done
