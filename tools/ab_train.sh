#!/usr/bin/env bash
# Developer A/B of training-step builds on one GPU box: every argument is a set of -D flags for gcn_train.hip ("-" = none); each is
# built into its own library and timed with tools/train_bench.py (run_epoch ms/step per form).   bash tools/ab_train.sh "-" "-DAQG_HEADS_INLINE"
set -euo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SRC=$ROOT/alphaquoridorgnn_amd/csrc
i=0
for fl in "$@"; do
  [ "$fl" = "-" ] && flags="" || flags="$fl"
  so=/tmp/libaqgnn_abtrain_$i.so
  (cd $SRC && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared $flags legal_mask.hip gcn_forward.hip gcn_train.hip mcts.hip capi.hip host_agents.cpp -o $so 2>/dev/null)
  echo "[$i] $fl"
  AQG_LIB_PATH=$so timeout -k 10 200 python3 $ROOT/tools/train_bench.py 2>/dev/null | grep -E "train_fused=2|fallbacks"
  i=$((i+1))
done
