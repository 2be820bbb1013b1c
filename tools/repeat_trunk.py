#!/usr/bin/env python3
"""Diagnostic: run each trunk variant repeatedly on the same boards and report run-to-run differences
(a deterministic kernel must be bitwise repeatable) and the distance to the exact-f32 variant."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
from tools.microbench import synth_states
dev = _lib.require_gpu("cuda:0"); lib = _lib.load()
torch.manual_seed(0)
model = GNNNetwork().to(dev).eval(); pk = model.packed_weights(dev)
for B in (300, 4096):
    st = synth_states(B, seed=3)
    ref = None
    for v in (1, 3, 4):
        _lib.set_option("trunk_variant", v)
        outs = []
        for rep in range(30):
            pooled = torch.full((B, 128), float("nan"), device=dev)
            _lib.check(lib.aqg_gcn_forward_boards(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(pooled), None, None, None, None, _lib.stream_ptr(dev)), "t")
            outs.append(pooled.clone())
            # disturb LDS contents between runs with the other variants
            _lib.set_option("trunk_variant", 1 if rep % 2 else 4)
            tmp = torch.empty((B, 128), device=dev)
            _lib.check(lib.aqg_gcn_forward_boards(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(tmp), None, None, None, None, _lib.stream_ptr(dev)), "t")
            _lib.set_option("trunk_variant", v)
        torch.cuda.synchronize()
        base = outs[0]
        nd = [int(((o != base).any(1)).sum()) for o in outs]
        nan = int(torch.isnan(base).any(1).sum())
        if v == 1:
            ref = base
        err = float((base - ref).abs().max())
        worst = float(max((o - ref).abs().max() for o in outs))
        print(f"B={B} variant {v}: boards differing from run 0 per run: {nd}  nan boards: {nan}  max|pooled - f32 exact| run0 {err:.3e} worst {worst:.3e}")
