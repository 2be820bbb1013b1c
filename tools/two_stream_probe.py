#!/usr/bin/env python3
"""Developer probe: K independent game sets on K streams vs one set (leaf evaluations per second per move)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.engine import BatchedSelfPlay
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
dev = _lib.require_gpu("cuda:0"); lib = _lib.load()
model = GNNNetwork().to(dev).eval()
sims = 200
for variant, tgrid in ((3, 0), (3, 256), (3, 384), (3, 128)):
    _lib.set_option("trunk_variant", variant)
    _lib.set_option("trunk_grid", tgrid)
    for total, K in ((2048, 1), (2048, 2), (2048, 4), (8192, 2), (8192, 4)):
        G = total // K
        streams = [torch.cuda.Stream() for _ in range(K)]
        engs = []
        for k in range(K):
            with torch.cuda.stream(streams[k]):
                engs.append(BatchedSelfPlay(model, num_games=G, sims=sims, seed=k, record_history=True))
        torch.cuda.synchronize()
        for k in range(K):
            with torch.cuda.stream(streams[k]):
                engs[k].move()
        torch.cuda.synchronize()
        nm = 3
        t0 = time.time()
        for _ in range(nm):
            for k in range(K):
                with torch.cuda.stream(streams[k]):
                    engs[k].move()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / nm
        print(f"variant {variant} trunk_grid {tgrid} total games {total} as {K} set(s) x {G}: {dt*1e3:7.2f} ms per move, {total*sims/dt/1e6:6.2f} M leaf evals/s", flush=True)
        del engs
        torch.cuda.empty_cache()
