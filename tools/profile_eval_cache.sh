#!/usr/bin/env bash
# Round 4: kernel statistics of a generation with the evaluation cache on (rocprofv3 --kernel-trace; rocprofv3 serialises the dispatches
# it intercepts: per-kernel durations are those of the kernel alone).  CONFIGS as tools/eval_cache_bench.py.
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r04c
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in ${CFGS:-2048:8192 16384:2048}; do
  tag=${cfg/:/_}
  rm -rf /tmp/pc
  CONFIGS=$cfg timeout -k 10 500 rocprofv3 --kernel-trace -d /tmp/pc -o b -- python3 $R/tools/eval_cache_bench.py > $OUT/run_$tag.log 2>&1
  python3 $R/tools/rocpd_stats.py /tmp/pc/b_results.db $OUT/kernel_stats_$tag.csv | cut -c1-110
done
