#!/usr/bin/env python3
"""Wall time of each move of a REFILLED generation (6144 games on 2048 slots x 200 sims, 4 sets)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.engine import MultiSetSelfPlay
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
dev = _lib.require_gpu("cuda:0")
torch.manual_seed(0)
model = GNNNetwork().to(dev).eval()
eng = MultiSetSelfPlay(model, num_games=2048, sims=200, num_sets=4, seed=4242, quota=int(os.environ.get("AQG_QUOTA", "6144")))
eng.move(); eng.sync(); eng.reset(); eng.sync()
prev = 0
for ply in range(400):
    t0 = time.perf_counter(); eng.move(); t1 = time.perf_counter(); eng.sync(); dt = time.perf_counter() - t0
    c = eng.counters()
    if ply % 8 == 0 or c["active"] == 0:
        print(f"ply {ply:3d}: {dt * 1e3:7.2f} ms (host enqueue {1e3 * (t1 - t0):5.2f})  active {c['active']:5d} started {c['started']:5d} finished {c['finished']:5d}  evals {c['leaf_evals'] - prev:8d}")
    prev = c["leaf_evals"]
    if c["active"] == 0:
        break
