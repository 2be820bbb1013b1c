#!/usr/bin/env python3
"""Developer probe: host time of one move() call (graph replay vs plain launches), i.e. the enqueue cost per move."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.engine import BatchedSelfPlay
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
dev = _lib.require_gpu("cuda:0")
model = GNNNetwork().to(dev).eval()
for use_graph in (1, 0):
    _lib.set_option("use_graph", use_graph)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        eng = BatchedSelfPlay(model, num_games=512, sims=200, seed=1)
        eng.move(); eng.move()
        st.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            eng.move()
        t1 = time.perf_counter()
        st.synchronize()
        t2 = time.perf_counter()
    print(f"use_graph={use_graph}: host enqueue {1e3*(t1-t0)/5:.2f} ms per move; GPU drain after {1e3*(t2-t1):.1f} ms", flush=True)
