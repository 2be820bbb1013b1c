#!/usr/bin/env bash
# Kernel statistics of the bench command (rocprofv3 --kernel-trace; tools/rocpd_stats.py prints the --stats table from the rocpd
# SQLite file).  Run on the GPU box from the repo root; writes gpurun_out/prof_r02/bench_kernel_stats.csv.
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p1
timeout -k 10 400 rocprofv3 --kernel-trace -d /tmp/p1 -o b -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs > $OUT/bench_under_rocprof.log 2>&1
python3 $R/tools/rocpd_stats.py /tmp/p1/b_results.db $OUT/bench_kernel_stats.csv > /dev/null && head -5 $OUT/bench_kernel_stats.csv | cut -c1-60,150-330
