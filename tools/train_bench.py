#!/usr/bin/env python3
"""Time the training step: per-step calls (GNNTrainer.step) and the one-call epoch (run_epoch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
from alphaquoridorgnn_amd.train_network import GNNTrainer, BATCH_SIZE
from tools.microbench import synth_states
dev = _lib.require_gpu("cuda:0")
model = GNNNetwork().to(dev)
tr = GNNTrainer(model, max_batch=BATCH_SIZE)
n = BATCH_SIZE * 400
st = synth_states(n)
A = model.policy_output_size
pi = torch.rand((n, A), device=dev); pi = pi / pi.sum(1, keepdim=True)
z = torch.randint(-1, 2, (n,), device=dev).float()
b = slice(0, BATCH_SIZE)
for fused in (0, 1, 2):
    _lib.set_option("train_fused", fused)
    order = torch.randperm(n, device=dev)
    tr.run_epoch(st, pi, z, order[:BATCH_SIZE * 10]); torch.cuda.synchronize()
    t0 = time.perf_counter(); s = tr.run_epoch(st, pi, z, order); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"train_fused={fused}: run_epoch {(t1 - t0) / 400 * 1e3:.4f} ms/step  ({n / (t1 - t0):.0f} positions/s)  loss sums {s.tolist()}")
print("fallbacks:", _lib.load().aqg_gcn_train_fallbacks(0))
for _ in range(5): tr.step(st[b], pi[b], z[b])
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): tr.step(st[b], pi[b], z[b])
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"GNNTrainer.step      : {(t1 - t0) / 200 * 1e3:.4f} ms/step")
order = torch.randperm(n, device=dev)
tr.run_epoch(st, pi, z, order[:BATCH_SIZE * 10]); torch.cuda.synchronize()
t0 = time.perf_counter(); s = tr.run_epoch(st, pi, z, order); torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"GNNTrainer.run_epoch : {(t1 - t0) / 400 * 1e3:.4f} ms/step  ({n / (t1 - t0):.0f} positions/s)  loss sums {s.tolist()}")
