#!/usr/bin/env python3
"""Developer timing of the training step (batch 128) for rocprofv3 --kernel-trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
from alphaquoridorgnn_amd.train_network import GNNTrainer
from tools.microbench import synth_states, time_ms
dev = _lib.require_gpu("cuda:0")
model = GNNNetwork().to(dev)
tr = GNNTrainer(model, max_batch=128)
st = synth_states(128); pi = torch.softmax(torch.randn((128, 209), device=dev), 1); z = torch.randint(-1, 2, (128,), device=dev).float()
print("ms per step", time_ms(lambda: tr.step(st, pi, z), 20, warmup=3))
