cd $GRAFT_REPO_ROOT/alphaquoridorgnn_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DAQG_NO_PRIO legal_mask.hip gcn_forward.hip gcn_train.hip mcts.hip capi.hip host_agents.cpp -o /tmp/lib_noprio.so 2>/dev/null
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  echo "--- with prologue"; python tools/trunk_scan.py 6 480,512 2>/dev/null | grep "B="
  echo "--- without"; AQG_LIB_PATH=/tmp/lib_noprio.so python tools/trunk_scan.py 6 480,512 2>/dev/null | grep "B="
done
