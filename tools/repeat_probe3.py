#!/usr/bin/env python3
"""Diagnostic experiments E1-E3 on the transient bf16x6 mismatch (see tools/repeat_probe2.py)."""
import os, sys, time
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
which = sys.argv[1]
_lib.set_option("trunk_variant", 4 if which.startswith("X3") else 3)
B = 300
sel = np.linspace(0, g["states"].shape[0] - 1, B).astype(int)
recs = g["states"][sel]
m = GNNNetwork(); m.load_state_dict({k: torch.from_numpy(x.copy()) for k, x in params.items()}); m = m.to("cuda").eval()
pk = m.packed_weights(dev)
st = torch.from_numpy(recs).to(dev)
policy = torch.empty((B, 209), device=dev); value = torch.empty((B,), device=dev)
def run():
    pooled = torch.full((B, 128), float("nan"), device=dev)
    _lib.check(lib.aqg_gcn_forward_boards(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(pooled), None, _lib.ptr(policy), None, _lib.ptr(value), _lib.stream_ptr(dev)), "f")
    return pooled
torch.cuda.synchronize()
if which == "E1":
    _lib.set_option("trunk_grid", 256)
if which == "E2":
    st2 = torch.from_numpy(np.concatenate([recs, recs])).to(dev); p2 = torch.empty((600, 128), device=dev)
    for _ in range(3):
        _lib.check(lib.aqg_gcn_forward_boards(9, _lib.ptr(st2), 0, 600, _lib.ptr(pk), _lib.ptr(p2), None, None, None, None, _lib.stream_ptr(dev)), "warm")
    torch.cuda.synchronize()
outs = []
for rep in range(40):
    if which == "E3":
        torch.cuda.synchronize(); time.sleep(0.3)
    outs.append(run())
torch.cuda.synchronize()
ref = outs[-1]
bad = [(i, torch.nonzero((o != ref).any(1)).flatten().tolist()[:6]) for i, o in enumerate(outs) if (o != ref).any()]
print(which, "launches differing from the last one:", bad[:8], f"({len(bad)}/40)")
