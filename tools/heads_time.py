#!/usr/bin/env python3
"""Developer timing: trunk-only vs trunk+heads launch pairs at MCTS-sized batches (HIP events, back-to-back)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
from tools.microbench import synth_states
dev = _lib.require_gpu("cuda:0"); lib = _lib.load()
model = GNNNetwork().to(dev).eval(); pk = model.packed_weights(dev)
for B in (256, 512, 1024, 2048, 8192):
    st = synth_states(B); pooled = torch.empty((B, 128), device=dev); pol = torch.empty((B, 209), device=dev); val = torch.empty((B,), device=dev)
    def trunk():
        _lib.check(lib.aqg_gcn_forward_boards(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(pooled), None, None, None, None, 0, _lib.stream_ptr(dev)), "t")
    def full():
        _lib.check(lib.aqg_gcn_forward_boards(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(pooled), None, _lib.ptr(pol), None, _lib.ptr(val), 0, _lib.stream_ptr(dev)), "t")
    res = []
    for f in (trunk, full):
        for _ in range(10): f()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 200 * 1e3)
    print(f"B={B}: trunk {res[0]:.1f} us, trunk+heads {res[1]:.1f} us, heads (difference) {res[1]-res[0]:.1f} us", flush=True)
