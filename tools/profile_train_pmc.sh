#!/usr/bin/env bash
# PMC passes over the training step's kernels (run on the GPU box from the repo root): per-launch means into gpurun_out/pmc_train.csv
# (one rocprofv3 --pmc pass per counter group, --kernel-trace only beside it).
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/pmc_train.csv
run_pmc() {  # tag, kernel pattern, counters
  rm -rf /tmp/pmt
  timeout -k 10 300 rocprofv3 --pmc $3 --kernel-trace --output-format csv -d /tmp/pmt -o c -- python3 $R/tools/prof_train.py > $OUT/pmc_train_$1.log 2>&1
  python3 $R/tools/pmc_summary.py "$2" /tmp/pmt | sed "s/^/$1,/" >> $OUT/pmc_train.csv
}
for k in train_board_split_kernel train_final_kernel; do
  run_pmc $k $k "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU"
  run_pmc $k $k "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"
  run_pmc $k $k "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"
  run_pmc $k $k "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"
done
cat $OUT/pmc_train.csv
