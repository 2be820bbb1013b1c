#!/usr/bin/env python3
"""Developer scan: trunk + heads time at the MCTS's launch sizes vs the start offset of the second-resident workgroups."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
from tools.microbench import synth_states
dev = _lib.require_gpu("cuda:0"); lib = _lib.load()
model = GNNNetwork().to(dev).eval(); pk = model.packed_weights(dev)
_lib.set_option("trunk_variant", 6); _lib.set_option("trunk_delay_min_boards", 0)
for B in (480, 512, 1024, 4096):
    st = synth_states(B); pooled = torch.empty((B, 128), device=dev)
    def trunk():
        _lib.check(lib.aqg_gcn_forward_boards(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(pooled), None, None, None, None, 0, _lib.stream_ptr(dev)), "t")
    res = {}
    for rnd in range(3):
        for d in (0, 8, 16, 24, 32, 48, 64, 100):
            _lib.set_option("trunk_phase_delay", d)
            for _ in range(5): trunk()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200): trunk()
            e1.record(); torch.cuda.synchronize()
            res.setdefault(d, []).append(e0.elapsed_time(e1) / 200)
    print(f"B={B}: " + "  ".join(f"delay {d}: {min(v)*1e3:.1f} us" for d, v in res.items()), flush=True)
