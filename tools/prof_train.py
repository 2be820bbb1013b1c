#!/usr/bin/env python3
"""Run a few training steps (batch 128) for rocprofv3 (kernel-trace or --pmc passes).  AQG_TRAIN_FUSED selects the form (default 2)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
from alphaquoridorgnn_amd.train_network import GNNTrainer, BATCH_SIZE
from tools.microbench import synth_states
dev = _lib.require_gpu("cuda:0")
_lib.set_option("train_fused", int(os.environ.get("AQG_TRAIN_FUSED", "2")))
steps = int(os.environ.get("AQG_ITERS", "12"))
model = GNNNetwork().to(dev)
tr = GNNTrainer(model, max_batch=BATCH_SIZE)
n = BATCH_SIZE * steps
st = synth_states(n)
pi = torch.softmax(torch.randn((n, 209), device=dev), dim=1)
z = torch.randint(-1, 2, (n,), device=dev).float()
tr.run_epoch(st, pi, z, torch.randperm(n, device=dev))
torch.cuda.synchronize()
print("done", steps)
