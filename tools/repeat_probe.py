#!/usr/bin/env python3
"""Diagnostic probe: which ingredient makes the first bf16x6 launch after a fresh model differ?"""
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
recs = g["states"][sel]
params = og.init_params(0)
ref = og.forward_states(params, recs)["logits"]

def new_model():
    m = GNNNetwork(); m.load_state_dict({k: torch.from_numpy(x.copy()) for k, x in params.items()})
    return m.to("cuda").eval()

def bad_boards(m):
    _, _, lg, _ = m.forward_states(torch.from_numpy(recs).to(dev), want_logits=True)
    lg = lg.cpu().numpy().astype(np.float64)
    bad = ~np.isclose(lg, ref, atol=1e-5, rtol=1e-4)
    return np.nonzero(bad.any(1))[0].tolist()

def scenario(name, fn, reps=25):
    n = 0
    first = None
    for r in range(reps):
        b = fn(r)
        if b:
            n += 1
            first = first or b[:6]
    print(f"{name}: {n}/{reps} runs with bad boards; e.g. {first}")

# A: fresh model each time, only variant 3
def A(r):
    _lib.set_option("trunk_variant", 3); return bad_boards(new_model())
# B: fresh model + full device sync before forward
def Bf(r):
    _lib.set_option("trunk_variant", 3); m = new_model(); m.packed_weights(dev); torch.cuda.synchronize(); return bad_boards(m)
# C: same model, but run variant 1 in between (LDS contents from the f32 kernel)
mC = new_model()
def C(r):
    _lib.set_option("trunk_variant", 1); bad_boards(mC); _lib.set_option("trunk_variant", 3); return bad_boards(mC)
# D: same model, poison LDS with finite garbage pattern via variant 1 then heads... plus NaN poison
def D(r):
    _lib.poison_lds(dev); _lib.set_option("trunk_variant", 3); return bad_boards(mC)
# E: fresh model, variant 1 run in between with the NEW model first
def E(r):
    m = new_model(); _lib.set_option("trunk_variant", 1); bad_boards(m); _lib.set_option("trunk_variant", 3); return bad_boards(m)
for name, fn in (("A fresh model, v3 only", A), ("B fresh model + sync", Bf), ("C same model, v1 then v3", C), ("D same model, NaN poison then v3", D), ("E fresh model, v1 then v3", E)):
    scenario(name, fn)
