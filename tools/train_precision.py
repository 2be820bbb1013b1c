#!/usr/bin/env python3
"""Diagnostic: gradient error of the training-step forms against fp64 autograd (oracle/train.py) -- per tensor max |err| / max |g|,
and the worst relative error over the elements whose gradient is at least 1e-3 of the tensor's largest (what Adam amplifies)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
from alphaquoridorgnn_amd.train_network import GNNTrainer
from oracle import gnn as og, train as ot
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_parity import _train_batch, _model
dev = _lib.require_gpu("cuda:0")
model, params = _model(6)
recs, pi, z = _train_batch(32, 11, params=params if os.environ.get("AQG_KINK_FILTER", "1") == "1" else None)
ref = ot.train_steps(params, [(recs, pi.astype(np.float64), z.astype(np.float64))])[0]
forms = [(1, None), (2, None), (0, None)]
for fused, sc in forms:
    _lib.set_option("train_fused", fused)
    tr = GNNTrainer(model, max_batch=32)
    tr.step(torch.from_numpy(recs), torch.from_numpy(pi), torch.from_numpy(z), update=False)
    print(f"train_fused={fused} fallbacks={_lib.load().aqg_gcn_train_fallbacks(1)}")
    for k, gt in zip(og.KEYS, tr.grads):
        r = ref["grads"][k]
        g = gt.cpu().numpy().astype(np.float64)
        well = np.abs(r) >= 1e-3 * np.abs(r).max()
        print(f"   {k:28s} max|err|/max|g| {np.abs(g - r).max() / np.abs(r).max():.2e}   worst rel err on well elements {(np.abs(g - r)[well] / np.abs(r)[well]).max():.2e}")
