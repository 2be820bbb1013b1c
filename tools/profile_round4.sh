#!/usr/bin/env bash
# Round-4 profile set (run on the GPU box from the repo root).  Writes gpurun_out/prof_r04/ (copy the summaries into profiles/).
#   pmc_summary.csv : per-launch means, one rocprofv3 --pmc pass per counter group (no tracing besides --kernel-trace):
#       trunk at 480 and 65,536 boards per launch (instruction mix, matrix-pipe busy, LDS, waits, HBM-side counters),
#       legal_actions_kernel at 4,096 and 65,536 states (instruction mix, VALU utilisation), MCTS step kernel at 512 games
#   bench_kernel_stats.csv : kernel statistics of the bench command (tools/rocpd_stats.py on rocprofv3's rocpd file)
# STAGE=pmc|bench|all selects what runs (a whole set is ~12 min of box time).
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r04
STAGE=${STAGE:-all}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run_pmc() {  # tag, kernel pattern, script, counters
  local tag=$1 pat=$2 script=$3 ctr=$4
  rm -rf /tmp/pm
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pm -o c -- python3 $R/tools/$script > $OUT/pmc_$tag.log 2>&1
  python3 $R/tools/pmc_summary.py "$pat" /tmp/pm | sed "s/^/$tag,/" >> $OUT/pmc_summary.csv
  echo "  pass $tag [$ctr] done"
}
if [ "$STAGE" = all ] || [ "$STAGE" = pmc ]; then
  : > $OUT/pmc_summary.csv
  for B in 480 65536; do
    export AQG_B=$B AQG_VARIANT=3 AQG_ITERS=5
    run_pmc trunk_B$B "gcn_trunk_boards_mm_kernel<0, false>" prof_trunk.py "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU"
    run_pmc trunk_B$B "gcn_trunk_boards_mm_kernel<0, false>" prof_trunk.py "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
    run_pmc trunk_B$B "gcn_trunk_boards_mm_kernel<0, false>" prof_trunk.py "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"
    run_pmc trunk_B$B "gcn_trunk_boards_mm_kernel<0, false>" prof_trunk.py "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"
    run_pmc trunk_B$B "gcn_trunk_boards_mm_kernel<0, false>" prof_trunk.py "TCC_HIT_sum TCC_MISS_sum"
    run_pmc trunk_B$B "gcn_trunk_boards_mm_kernel<0, false>" prof_trunk.py "FETCH_SIZE"
    run_pmc trunk_B$B "gcn_trunk_boards_mm_kernel<0, false>" prof_trunk.py "WRITE_SIZE"
    run_pmc trunk_B$B "gcn_trunk_boards_mm_kernel<0, false>" prof_trunk.py "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
  done
  for B in 4096 65536; do
    export AQG_B=$B AQG_ITERS=5
    run_pmc legal_B$B legal_actions_kernel prof_legal.py "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM"
    run_pmc legal_B$B legal_actions_kernel prof_legal.py "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
    run_pmc legal_B$B legal_actions_kernel prof_legal.py "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
    run_pmc legal_B$B legal_actions_kernel prof_legal.py "SQ_INSTS_SMEM SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH"
  done
  export AQG_G=512 AQG_MOVES=6
  run_pmc step_G512 engine_step_fast_kernel prof_step.py "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM"
  run_pmc step_G512 engine_step_fast_kernel prof_step.py "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
  cat $OUT/pmc_summary.csv
fi
if [ "$STAGE" = all ] || [ "$STAGE" = bench ]; then
  echo "[bench kernel trace]"; rm -rf /tmp/p1
  timeout -k 10 500 rocprofv3 --kernel-trace -d /tmp/p1 -o b -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs > $OUT/bench_under_rocprof.log 2>&1
  python3 $R/tools/rocpd_stats.py /tmp/p1/b_results.db $OUT/bench_kernel_stats.csv > /dev/null && head -8 $OUT/bench_kernel_stats.csv | cut -c1-170
fi
