#!/usr/bin/env bash
# HBM-side PMC passes (two TCC counters per pass: more exceed what one pass can collect) for the trunk and the step kernel,
# and the kernel trace of the bench command.  Run on the GPU box from the repo root; appends to gpurun_out/prof_r02/.
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run_pmc() {  # tag, kernel pattern, script, counters
  local tag=$1 pat=$2 script=$3 ctr=$4
  rm -rf /tmp/pm
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pm -o c -- python3 $R/tools/$script > $OUT/pmc_hbm_$tag.log 2>&1
  python3 $R/tools/pmc_summary.py "$pat" /tmp/pm | sed "s/^/$tag,/" >> $OUT/pmc_hbm_summary.csv
}
: > $OUT/pmc_hbm_summary.csv
for B in 512 65536; do
  export AQG_B=$B AQG_VARIANT=3 AQG_ITERS=5
  run_pmc trunk_B$B gcn_trunk_boards_mm_kernel prof_trunk.py "FETCH_SIZE WRITE_SIZE"
  run_pmc trunk_B$B gcn_trunk_boards_mm_kernel prof_trunk.py "TCC_HIT_sum TCC_MISS_sum"
done
export AQG_G=512 AQG_MOVES=6
run_pmc step_G512 engine_step_fast_kernel prof_step.py "FETCH_SIZE WRITE_SIZE"
run_pmc step_G512 engine_step_fast_kernel prof_step.py "TCC_HIT_sum TCC_MISS_sum"
cat $OUT/pmc_hbm_summary.csv
echo "[bench kernel trace]"; rm -rf /tmp/p1
timeout -k 10 400 rocprofv3 --kernel-trace -d /tmp/p1 -o b -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs > $OUT/bench_under_rocprof.log 2>&1
python3 $R/tools/rocpd_stats.py /tmp/p1/b_results.db $OUT/bench_kernel_stats.csv > /dev/null && head -6 $OUT/bench_kernel_stats.csv | cut -c1-60,150-330
