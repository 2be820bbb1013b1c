#!/usr/bin/env bash
# Negative control for the padding-row fix of the fused training kernel: rebuild with the OLD clearing loop (rows 81..95 only)
# and run the small-board gradient test, which poisons the LDS first -- it must FAIL on 3x3 / 5x5.
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p /tmp/negctl && cp -r $R/alphaquoridorgnn_amd/csrc /tmp/negctl/ && mkdir -p /tmp/negctl/include_root && cp -r $R/include /tmp/negctl/
cd /tmp/negctl/csrc
sed -i 's|for (int i = t; i < (96 - V) \* 32; i += 512) st4(Hs + (V + (i >> 5)) \* SA|for (int i = t; i < 15 * 32; i += 512) st4(Hs + (81 + (i >> 5)) * SA|' gcn_train.hip
grep -c "81 + (i >> 5)" gcn_train.hip
sed -i 's|#include "../../include/aqgnn.h"|#include "/tmp/negctl/include/aqgnn.h"|' *.hip *.cpp 2>/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared legal_mask.hip gcn_forward.hip gcn_train.hip mcts.hip capi.hip host_agents.cpp -o /tmp/negctl/lib_old.so 2>/dev/null
cd $R
AQG_LIB_PATH=/tmp/negctl/lib_old.so python -m pytest tests/test_gpu_parity.py -q -k "small_boards" 2>&1 | tail -4 | cut -c1-160
