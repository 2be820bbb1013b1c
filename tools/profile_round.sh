#!/usr/bin/env bash
# Round profile set (run on the GPU box from the repo root): kernel statistics of the bench command and of the training
# epoch, PMC passes for the trunk and the MCTS step kernel.  Writes summaries to gpurun_out/prof_r02/ (copy into profiles/).
# rocprofv3 here = ROCm 7.2: --kernel-trace writes a rocpd SQLite file; tools/rocpd_stats.py turns it into the --stats table
# (the built-in --stats post-processing hung on this image).  One PMC pass per counter group, no other tracing.
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "[1] bench kernel trace"; rm -rf /tmp/p1
timeout -k 10 400 rocprofv3 --kernel-trace -d /tmp/p1 -o b -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs > $OUT/bench_under_rocprof.log 2>&1
python3 $R/tools/rocpd_stats.py /tmp/p1/b_results.db $OUT/bench_kernel_stats.csv > /dev/null && head -8 $OUT/bench_kernel_stats.csv | cut -c1-150
echo "[2] train kernel trace"; rm -rf /tmp/p2
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/p2 -o t -- python3 $R/tools/train_bench.py > $OUT/train_under_rocprof.log 2>&1
python3 $R/tools/rocpd_stats.py /tmp/p2/t_results.db $OUT/train_kernel_stats.csv > /dev/null && head -9 $OUT/train_kernel_stats.csv | cut -c1-150
run_pmc() {  # tag, kernel pattern, script, counters
  local tag=$1 pat=$2 script=$3 ctr=$4
  rm -rf /tmp/pm
  timeout -k 10 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pm -o c -- python3 $R/tools/$script > $OUT/pmc_$tag.log 2>&1
  python3 $R/tools/pmc_summary.py "$pat" /tmp/pm | sed "s/^/$tag,/" >> $OUT/pmc_summary.csv
}
: > $OUT/pmc_summary.csv
for B in 512 65536; do
  export AQG_B=$B AQG_VARIANT=3 AQG_ITERS=5
  run_pmc trunk_B$B gcn_trunk_boards_mm_kernel prof_trunk.py "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU"
  run_pmc trunk_B$B gcn_trunk_boards_mm_kernel prof_trunk.py "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
  run_pmc trunk_B$B gcn_trunk_boards_mm_kernel prof_trunk.py "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"
  run_pmc trunk_B$B gcn_trunk_boards_mm_kernel prof_trunk.py "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"
  run_pmc trunk_B$B gcn_trunk_boards_mm_kernel prof_trunk.py "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
done
export AQG_G=512 AQG_MOVES=6
run_pmc step_G512 engine_step_fast_kernel prof_step.py "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM"
run_pmc step_G512 engine_step_fast_kernel prof_step.py "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
run_pmc step_G512 engine_step_fast_kernel prof_step.py "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
run_pmc step_G512 engine_step_fast_kernel prof_step.py "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
cat $OUT/pmc_summary.csv
