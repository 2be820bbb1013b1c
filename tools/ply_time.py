#!/usr/bin/env python3
"""Wall time of each move of a lock-step generation (2048 games x 200 sims, 4 sets): where in a game the time goes."""
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
eng = MultiSetSelfPlay(model, num_games=2048, sims=200, num_sets=4, seed=1000)
eng.move(); eng.sync(); eng.reset(); eng.sync()
rows = []
for ply in range(117):
    t0 = time.perf_counter(); eng.move(); eng.sync(); dt = time.perf_counter() - t0
    c = eng.counters()
    rows.append((ply, dt * 1e3, c["active"], c["leaf_evals"]))
    if c["active"] == 0:
        break
prev = 0
for ply, ms, act, ev in rows:
    if ply % 4 == 0 or act < 2048:
        print(f"ply {ply:3d}: {ms:7.2f} ms  active after {act:5d}  leaf evals this move {ev - prev:8d}  ({(ev - prev) / ms / 1e3:6.2f} M evals/s)")
    prev = ev
print("total", sum(r[1] for r in rows), "ms")
