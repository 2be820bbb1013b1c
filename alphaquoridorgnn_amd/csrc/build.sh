#!/usr/bin/env bash
# Build libaqgnn_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libaqgnn_hip.so
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function"
OBJDIR=${OBJDIR:-$(mktemp -d)}
objs=()
for f in legal_mask gcn_forward gcn_train mcts capi; do
  $HIPCC $FLAGS -c $f.hip -o ${OBJDIR}/aqg_$f.o &
  objs+=(${OBJDIR}/aqg_$f.o)
done
# host-only code (CPU baseline agents over the same rule header): plain C++, no device pass
${CXX:-g++} -O2 -std=c++17 -fPIC -Wall -Wno-unknown-pragmas -c host_agents.cpp -o ${OBJDIR}/aqg_host_agents.o &
objs+=(${OBJDIR}/aqg_host_agents.o)
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT "${objs[@]}"
echo "built $(realpath $OUT)"
