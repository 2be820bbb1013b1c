#!/usr/bin/env python3
"""Experiment: static wave priorities in the trunk (trunk_prio bit 0: waves 4-7, bit 1: second-resident workgroups, bit 2: first)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
from tools.microbench import synth_states, time_ms
dev = _lib.require_gpu("cuda:0"); lib = _lib.load()
model = GNNNetwork().to(dev).eval(); pk = model.packed_weights(dev)
for B in (1024, 2048, 16384, 65536):
    st = synth_states(B); pooled = torch.empty((B, 128), device=dev)
    def fwd():
        _lib.check(lib.aqg_gcn_forward_boards(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(pooled), None, None, None, None, 0, _lib.stream_ptr(dev)), "fwd")
    for rep in range(2):
        for prio in (0, 3, 8, 11):
            _lib.set_option("trunk_prio", prio)
            ms = min(time_ms(fwd, 200 if B < 60000 else 30, warmup=20) for _ in range(3))
            print(f"B={B:6d} prio {prio}: {ms * 1e3:8.2f} us  {B / ms / 1e3:6.2f} M boards/s")
