#!/usr/bin/env bash
# Build libaqgnn_hip.so for gfx950 (MI355X) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=${OUT:-../libaqgnn_hip.so}
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wall -Wno-unused-function ${AQG_EXTRA_FLAGS:-}"
# objects go into a directory of this build's own: a caller-supplied OBJDIR must be empty (a stale object from another build
# would otherwise be linked silently when its compile fails)
if [ -n "${OBJDIR:-}" ]; then
  mkdir -p "$OBJDIR"
  if [ -n "$(ls -A "$OBJDIR")" ]; then echo "build.sh: OBJDIR=$OBJDIR is not empty" >&2; exit 2; fi
else
  OBJDIR=$(mktemp -d)
  trap 'rm -rf "$OBJDIR"' EXIT
fi
objs=()
pids=()
for f in legal_mask gcn_forward gcn_train mcts capi; do
  $HIPCC $FLAGS -c $f.hip -o "${OBJDIR}/aqg_$f.o" &
  pids+=($!)
  objs+=("${OBJDIR}/aqg_$f.o")
done
# host-only code (CPU baseline agents over the same rule header): plain C++, no device pass
${CXX:-g++} -O2 -std=c++17 -fPIC -Wall -Wno-unknown-pragmas -c host_agents.cpp -o "${OBJDIR}/aqg_host_agents.o" &
pids+=($!)
objs+=("${OBJDIR}/aqg_host_agents.o")
# every compile job is waited for by PID: a bare `wait` returns 0 whatever the jobs did
failed=0
for pid in "${pids[@]}"; do
  wait "$pid" || failed=1
done
if [ "$failed" -ne 0 ]; then echo "build.sh: a compile job failed" >&2; exit 1; fi
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${objs[@]}"
echo "built $(realpath "$OUT")"
