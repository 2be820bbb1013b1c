#!/usr/bin/env python3
"""Developer scan: trunk throughput vs the start delay of the second-resident workgroups (option trunk_phase_delay)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
from tools.microbench import synth_states
dev = _lib.require_gpu("cuda:0"); lib = _lib.load()
model = GNNNetwork().to(dev).eval(); pk = model.packed_weights(dev)
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 3
_lib.set_option("trunk_variant", variant)
_lib.set_option("trunk_delay_min_boards", 0)
for B in (256, 512, 1024, 2048, 4096, 16384, 65536):
    st = synth_states(B); pooled = torch.empty((B, 128), device=dev)
    def run():
        _lib.check(lib.aqg_gcn_forward_boards(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(pooled), None, None, None, None, 0, _lib.stream_ptr(dev)), "t")
    for delay in (0, 100):
        _lib.set_option("trunk_phase_delay", delay)
        for _ in range(5): run()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        n = 50 if B <= 16384 else 20
        e0.record()
        for _ in range(n): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(f"B={B} delay={delay:4d} x64cyc  {ms*1e3:8.1f} us  {B/ms/1e3:7.2f} M boards/s", flush=True)
