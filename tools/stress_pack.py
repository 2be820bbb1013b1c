#!/usr/bin/env python3
"""Diagnostic: is the packed-weight upload (host pack -> .to(device)) always bit-identical on the device?"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
from oracle import gnn as og
dev = _lib.require_gpu("cuda:0"); lib = _lib.load()
params = og.init_params(0)
ref = None
bad = 0
for rep in range(400):
    m = GNNNetwork(); m.load_state_dict({k: torch.from_numpy(x.copy()) for k, x in params.items()}); m = m.to("cuda").eval()
    pk = m.packed_weights(dev)
    junk = [np.random.rand(50000) for _ in range(4)]          # churn the host allocator right after
    t = torch.randn(1 << 18, device=dev) * 2                   # and the device
    got = pk.cpu().numpy().view(np.uint32)
    if ref is None:
        ref = got.copy()
    nd = int((got != ref).sum())
    if nd:
        bad += 1
        idx = np.nonzero(got != ref)[0]
        print(f"rep {rep}: {nd} differing dwords, first {idx[:5]} last {idx[-5:]} of {len(ref)}")
print("bad uploads:", bad, "of 400; total dwords", len(ref))
