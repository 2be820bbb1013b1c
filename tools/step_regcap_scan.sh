#!/usr/bin/env bash
# Experiment (round 4): step kernel built at 80 registers x 12 games per workgroup / 64 registers x 16 games per workgroup (libraries
# from `AQG_EXTRA_FLAGS=-DAQG_STEP_WPB_MAX=12|16 OUT=tools/ubench/bin/libaqgnn_wpbNN.so bash csrc/build.sh`) against the default 96 x 8:
# fewer step workgroups take the register-file slot of a trunk workgroup for the duration of a step launch.
run() {  # label, lib (or empty), step_waves
  AQG_LIB_PATH=${2:-$(pwd)/alphaquoridorgnn_amd/libaqgnn_hip.so} AQG_STEP_WAVES=$3 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-legs --large-games ${LARGE:-0} 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1:', round(d['value'],1), 'games/s', round(d['leaf_evals_per_s']/1e6,2), 'M evals/s', (d.get('large_batch') or {}).get('games_per_s'))"
}
B=$(pwd)/tools/ubench/bin
run "96 regs x  8" "" 8
run "80 regs x 12" $B/libaqgnn_wpb12.so 12
run "64 regs x 16" $B/libaqgnn_wpb16.so 16
run "96 regs x  8" "" 8
run "80 regs x 12" $B/libaqgnn_wpb12.so 12
run "80 regs x  8" $B/libaqgnn_wpb12.so 8
