#!/usr/bin/env python3
"""Developer timing: legal-actions kernel latency at small batches (one wavefront per state)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alphaquoridorgnn_amd import _lib
from tools.microbench import synth_states
dev = _lib.require_gpu("cuda:0"); lib = _lib.load()
for B in (64, 256, 512, 2048, 8192):
    st = synth_states(B, seed=3)
    order = torch.empty((B, 136), dtype=torch.uint8, device=dev); count = torch.empty((B,), dtype=torch.int32, device=dev)
    def f():
        _lib.check(lib.aqg_legal_actions(9, _lib.ptr(st), B, None, _lib.ptr(order), _lib.ptr(count), _lib.stream_ptr(dev)), "legal")
    for _ in range(10): f()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): f()
    e1.record(); torch.cuda.synchronize()
    walls = (st[:, 4:68] != 0).sum(1).float().mean().item()
    print(f"B={B}: {e0.elapsed_time(e1)/200*1e3:.1f} us per launch, mean legal {count.float().mean().item():.1f}, mean walls on board {walls:.1f}", flush=True)
