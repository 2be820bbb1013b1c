#!/usr/bin/env python3
"""Round 4: the evaluation cache must not change a single byte of a generation.  Plays the same seeded generation with the cache off
and on (several table sizes, incl. one far too small: every window full, entries replaced all the time) and compares the history rows,
the per-game results and the logical evaluation counts; prints the hit rate and the wall time of each."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.engine import MultiSetSelfPlay
from alphaquoridorgnn_amd.pv_network_gnn import GraphPolicyValueNetwork
G = int(os.environ.get("GAMES", "256")); SIMS = int(os.environ.get("SIMS", "50"))
dev = _lib.require_gpu("cuda:0")
torch.manual_seed(0)
model = GraphPolicyValueNetwork().to(dev)
ref = None
for slots in (0, 8192, 1024, 64, 0):
    eng = MultiSetSelfPlay(model, num_games=G, sims=SIMS, seed=3, eval_cache_slots=slots)
    eng.sync(); t0 = time.perf_counter()
    c = eng.play_generation()
    eng.sync(); dt = time.perf_counter() - t0
    st, vis, z = (x.cpu() for x in eng.history_tensors())
    if ref is None:
        ref = (st, vis, z, c)
    same = torch.equal(st, ref[0]) and torch.equal(vis, ref[1]) and torch.equal(z, ref[2]) and c["leaf_evals"] == ref[3]["leaf_evals"] and c["terminal_sims"] == ref[3]["terminal_sims"]
    print(f"slots {slots:5d}: {G / dt:8.1f} games/s  rows {st.shape[0]}  leaf_evals {c['leaf_evals']}  hits {c['cache_hits']} ({c['cache_hits'] / max(c['leaf_evals'], 1):.3f})  identical to cache-off: {same}", flush=True)
    assert same
    del eng
print("OK")
