#!/usr/bin/env python3
"""Diagnostic: on a failing first launch after a fresh model, which pooled channels / boards differ from a clean re-run?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
from oracle import gnn as og
from tests import _util as U
dev = _lib.require_gpu("cuda:0"); lib = _lib.load()
g = U.golden("walk_9x9.npz")
params = og.init_params(0)
_lib.set_option("trunk_variant", 3)
for B in (300, 200, 600):
    sel = np.linspace(0, g["states"].shape[0] - 1, B).astype(int)
    recs = g["states"][sel]
    nbad = 0
    for rep in range(60):
        m = GNNNetwork(); m.load_state_dict({k: torch.from_numpy(x.copy()) for k, x in params.items()}); m = m.to("cuda").eval()
        st = torch.from_numpy(recs).to(dev)
        pooled = torch.full((B, 128), float("nan"), device=dev)
        policy = torch.empty((B, 209), device=dev); value = torch.empty((B,), device=dev)
        pk = m.packed_weights(dev)
        _lib.check(lib.aqg_gcn_forward_boards(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(pooled), None, _lib.ptr(policy), None, _lib.ptr(value), _lib.stream_ptr(dev)), "f")
        p1 = pooled.clone()
        torch.cuda.synchronize()
        pooled2 = torch.full((B, 128), float("nan"), device=dev)
        _lib.check(lib.aqg_gcn_forward_boards(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(pooled2), None, _lib.ptr(policy), None, _lib.ptr(value), _lib.stream_ptr(dev)), "f")
        d = (p1 != pooled2)
        if d.any():
            nbad += 1
            rows = torch.nonzero(d.any(1)).flatten().tolist()
            r0 = rows[0]
            ch = torch.nonzero(d[r0]).flatten().tolist()
            print(f"B={B} rep {rep}: boards {rows[:10]}{'...' if len(rows) > 10 else ''} ({len(rows)}); board {r0}: {len(ch)} channels differ, first {ch[:8]} last {ch[-4:]}, "
                  f"max diff {float((p1 - pooled2).abs().max()):.3e}, nan in first run: {bool(torch.isnan(p1).any())}")
    print(f"B={B}: {nbad}/60 first launches differ from the re-run")
