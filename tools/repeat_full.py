#!/usr/bin/env python3
"""Diagnostic: bitwise repeatability of trunk (pooled) and heads (logits) over many full forwards."""
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
sel = np.linspace(0, g["states"].shape[0] - 1, 300).astype(int)
recs = torch.from_numpy(g["states"][sel]).to(dev)
params = og.init_params(0)
m = GNNNetwork(); m.load_state_dict({k: torch.from_numpy(x.copy()) for k, x in params.items()}); m = m.to("cuda").eval()
pk = m.packed_weights(dev)
B = 300
for v in (3, 1):
    _lib.set_option("trunk_variant", v)
    base_p = base_l = None
    nbad_p = nbad_l = 0
    for rep in range(400):
        pooled = torch.full((B, 128), float("nan"), device=dev)
        logits = torch.full((B, 209), float("nan"), device=dev)
        policy = torch.empty((B, 209), device=dev); value = torch.empty((B,), device=dev); vpre = torch.empty((B,), device=dev)
        _lib.check(lib.aqg_gcn_forward_boards(9, _lib.ptr(recs), 0, B, _lib.ptr(pk), _lib.ptr(pooled), _lib.ptr(logits), _lib.ptr(policy),
                                              _lib.ptr(vpre), _lib.ptr(value), _lib.stream_ptr(dev)), "f")
        if rep % 3 == 0:   # perturb: allocate / free / other kernels in between
            junk = torch.randn(1 << 20, device=dev); junk = junk * 2
        if base_p is None:
            base_p, base_l = pooled.clone(), logits.clone()
        else:
            dp = (pooled != base_p).any(1); dl = (logits != base_l).any(1)
            if dp.any() or dl.any():
                nbad_p += int(dp.any()); nbad_l += int(dl.any())
                print(f"variant {v} rep {rep}: pooled differs on boards {torch.nonzero(dp).flatten().tolist()[:12]} (max {float((pooled-base_p).abs().max()):.3e}); "
                      f"logits differ on boards {torch.nonzero(dl).flatten().tolist()[:12]} (max {float((logits-base_l).abs().max()):.3e})")
    print(f"variant {v}: runs with pooled diffs {nbad_p}, runs with logits diffs {nbad_l} of 399")
