#!/usr/bin/env python3
"""Run the fused MCTS step kernel for rocprofv3 (kernel-trace or --pmc passes): AQG_G games (default 512), fake evaluator
(uniform priors: no trunk / heads launches in between), plain launches, AQG_MOVES moves of 200 simulations."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.engine import BatchedSelfPlay
dev = _lib.require_gpu("cuda:0")
_lib.set_option("use_graph", 0)
G = int(os.environ.get("AQG_G", "512"))
eng = BatchedSelfPlay(None, num_games=G, sims=200, evaluator="fake", fake_bias=0, record_history=False)
for _ in range(int(os.environ.get("AQG_MOVES", "12"))):
    eng.move()
torch.cuda.synchronize()
print("done", G, eng.counters())
