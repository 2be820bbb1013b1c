#!/usr/bin/env python3
"""Round 4: self-play generations with the evaluation cache off / on at the bench's configurations (one warm-up generation, one timed).
CONFIGS="games:slots,..." (default: the headline's 2,048 games and the 16,384-game configuration)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.engine import MultiSetSelfPlay
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
SIMS = int(os.environ.get("SIMS", "200"))
cfgs = [tuple(int(x) for x in c.split(":")) for c in os.environ.get("CONFIGS", "2048:0,2048:8192,2048:1024,16384:0,16384:2048").split(",")]
SETS = int(os.environ.get("SETS", "4"))
dev = _lib.require_gpu("cuda:0")
torch.manual_seed(0)
model = GNNNetwork().to(dev).eval()
for games, slots in cfgs:
    eng = MultiSetSelfPlay(model, num_games=games, sims=SIMS, num_sets=SETS, seed=1000, eval_cache_slots=slots)
    for rep in range(2):
        eng.reset()
        eng.sync(); t0 = time.perf_counter()
        c = eng.play_generation()
        eng.sync(); dt = time.perf_counter() - t0
    print(f"sets {SETS} games {games:6d} slots {slots:5d}: {c['finished'] / dt:8.1f} games/s  {dt:6.2f} s  leaf_evals {c['leaf_evals']}  "
          f"hits {c['cache_hits'] / max(c['leaf_evals'], 1):.3f}  evals/s {c['leaf_evals'] / dt / 1e6:6.1f} M  mem {torch.cuda.max_memory_allocated() / 2**30:5.1f} GiB", flush=True)
    del eng
    torch.cuda.empty_cache()
