#!/usr/bin/env python3
"""Run the legal-actions kernel a few times for rocprofv3 (kernel-trace or --pmc passes): AQG_B states, AQG_ITERS launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alphaquoridorgnn_amd import _lib
from tools.microbench import synth_states

B = int(os.environ.get("AQG_B", "65536"))
iters = int(os.environ.get("AQG_ITERS", "5"))
dev = _lib.require_gpu("cuda:0")
lib = _lib.load()
st = synth_states(B, seed=1)
mask = torch.empty((B, 209), dtype=torch.uint8, device=dev)
order = torch.empty((B, 136), dtype=torch.uint8, device=dev)
count = torch.empty((B,), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for _ in range(iters):
    _lib.check(lib.aqg_legal_actions(9, _lib.ptr(st), B, _lib.ptr(mask), _lib.ptr(order), _lib.ptr(count), _lib.stream_ptr(dev)), "legal")
torch.cuda.synchronize()
print("done", B, float(count.float().mean()))
