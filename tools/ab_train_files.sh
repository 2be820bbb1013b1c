#!/usr/bin/env bash
# A/B of two versions of gcn_train.hip on one box: bash tools/ab_train_files.sh <old file> (the in-tree file is the new one)
set -euo pipefail
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SRC=$ROOT/alphaquoridorgnn_amd/csrc
for v in old new old new; do
  f=gcn_train.hip; [ $v = old ] && f=$1
  so=/tmp/libaqgnn_abtf_$v.so
  (cd $SRC && /opt/rocm/bin/hipcc -I$SRC -I$ROOT/include -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared legal_mask.hip gcn_forward.hip $f mcts.hip capi.hip host_agents.cpp -o $so 2>/dev/null)
  echo "[$v]"; AQG_LIB_PATH=$so timeout -k 10 200 python3 $ROOT/tools/train_bench.py 2>/dev/null | grep -E "train_fused=2"
done
