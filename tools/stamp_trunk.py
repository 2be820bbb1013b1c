#!/usr/bin/env python3
"""Diagnostic: build the trunk with -DAQG_STAMP and print the share of cycles per phase (workgroup 0, wave 0).
Never quote this build's run time -- read its SHARES (guide: in-kernel stamps)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = "/tmp/libaqgnn_hip_stamp.so"
src = os.path.join(ROOT, "alphaquoridorgnn_amd", "csrc")
subprocess.check_call(f"cd {src} && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DAQG_STAMP "
                      f"legal_mask.hip gcn_forward.hip gcn_train.hip mcts.hip capi.hip host_agents.cpp -o {so}", shell=True)
os.environ["AQG_LIB_PATH"] = so
import torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
from tools.microbench import synth_states
dev = _lib.require_gpu("cuda:0"); lib = _lib.load()
model = GNNNetwork().to(dev).eval(); pk = model.packed_weights(dev)
B = int(os.environ.get("AQG_B", "65536"))
st = synth_states(B)
word = torch.zeros((1,), dtype=torch.int32, device=dev)
names = ["setup", "L1a gather6", "L1b 6->128", "L2 mfma", "L2 stripe gather", "L3 mfma", "L3 gather+pool", "loop top", "wait vmcnt before L1b", "wait vmcnt before L3", "-", "-", "-", "-", "-", "-"]
names5 = ["G' rows of layer 1 (+ fragments if built here), next-board index, barrier", "barrier after layer 1", "layer-2 linear (MFMA + split)", "barrier after layer-2 aggregation", "layer-3 linear (MFMA + split)", "layer-3 aggregation + pool + store", "record wait, decode, bias offsets, bias request", "loop top (w1f request)", "layer 1 (6 MFMAs on the bias rows, relu, plane stores)", "W2 fragment request (+ fragments if built here)", "next board's inputs if prefetched", "adjacency fragments if built after the layer-2 linear map", "W3 fragment request + next record", "barrier: planes read by everybody", "layer-2 aggregation + plane stores", "-"]
for v, grid in ((6, 512),):
    _lib.set_option("trunk_variant", v); _lib.set_option("trunk_grid", grid)
    pooled = torch.zeros((B + 1, 128), device=dev)
    for _ in range(3):
        _lib.check(lib.aqg_gcn_forward_boards_guarded(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(pooled), None, None, None, None,
                                                      int(os.environ.get("AQG_GNN_FLAGS", "2")), _lib.ptr(word), _lib.stream_ptr(dev)), "t")   # 2 = the build without per-value tracking
    torch.cuda.synchronize()
    raw = pooled[B].view(torch.int64)[:17].cpu().tolist()
    n = raw[16]; tot = sum(raw[:16])
    print(f"variant {v} grid {grid}: boards by WG0 = {n}, cycles/board (s_memtime @100MHz units?) = {tot / max(n,1):.1f}")
    for nm, c in zip(names5 if v >= 3 else names, raw[:16]):
        print(f"   {nm:62s} {c / max(n,1):10.1f}  {100.0 * c / tot:5.1f}%")
