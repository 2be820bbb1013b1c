#!/usr/bin/env python3
"""Developer diagnostic: pooled trunk features of every trunk form against the fp64 oracle on a few boards."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
from oracle import gnn as og, quoridor as oq
dev = _lib.require_gpu("cuda:0"); lib = _lib.load()
params = og.init_params(0)
rng = np.random.RandomState(1)
for k in ("gcn_layers.0.bias", "gcn_layers.1.bias", "gcn_layers.2.bias"):
    if os.environ.get("AQG_BIAS", "1") == "1":
        params[k] = (0.3 * rng.randn(128)).astype(np.float32)
model = GNNNetwork(); model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()}); model = model.to(dev).eval()
pk = model._packed = None
g = np.load(os.path.join(ROOT, "tests", "golden", "walk_9x9.npz"))
recs = np.concatenate([oq.init_record(9)[None], g["states"][[10, 500, 3000, 9000, 12000, 13500]]])
ref = og.forward_states(params, recs)
st = torch.from_numpy(recs).to(dev)
B = recs.shape[0]
for v in (1, 6, 5, 4):
    _lib.set_option("trunk_variant", v)
    pooled = torch.zeros((B, 128), device=dev); logits = torch.zeros((B, 209), device=dev)
    _lib.check(lib.aqg_gcn_forward_boards(9, _lib.ptr(st), 0, B, _lib.ptr(model.packed_weights(dev)), _lib.ptr(pooled), _lib.ptr(logits), None, None, None, 0, _lib.stream_ptr(dev)), "f")
    p = pooled.cpu().numpy().astype(np.float64); l = logits.cpu().numpy().astype(np.float64)
    rp = ref["pooled"]
    err = np.abs(p - rp)
    ratio = p[np.abs(rp) > 1e-3] / rp[np.abs(rp) > 1e-3]
    print(f"variant {v}: pooled max|err| {err.max():.3e} per board {err.max(1)}  ratio mean {ratio.mean():.6f} std {ratio.std():.3e}  logits max|err| {np.abs(l - ref['logits']).max():.3e}")
_lib.set_option("trunk_variant", 3)
