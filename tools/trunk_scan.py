#!/usr/bin/env python3
"""Developer scan: GCN trunk time per launch for each form (trunk_variant 4 / 5 / 6 = 4-wave x 3 per CU, 4-wave x 2, 8-wave x 2)
over launch sizes from the MCTS's ~480 boards to 65,536, variants interleaved in one process (guide rule 24)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
from tools.microbench import synth_states
dev = _lib.require_gpu("cuda:0"); lib = _lib.load()
model = GNNNetwork().to(dev).eval(); pk = model.packed_weights(dev)
flags = int(os.environ.get("AQG_GNN_FLAGS", model.gnn_flags(dev))); word = model.saturation_word(dev)
variants = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else "6,5,4".split(","))]
sizes = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else "480,512,1024,2048,4096,16384,65536".split(","))]
for B in sizes:
    st = synth_states(B); pooled = torch.empty((B, 128), device=dev)
    policy = torch.empty((B, 209), device=dev); value = torch.empty((B,), device=dev)
    # the module's own call: guarded entry with this weight set's flags (range guard: proven bound or tracking build)
    def trunk():
        _lib.check(lib.aqg_gcn_forward_boards_guarded(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(pooled), None, None, None, None, flags, _lib.ptr(word), _lib.stream_ptr(dev)), "t")
    def full():
        _lib.check(lib.aqg_gcn_forward_boards_guarded(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(pooled), None, _lib.ptr(policy), None, _lib.ptr(value), flags, _lib.ptr(word), _lib.stream_ptr(dev)), "t")
    res = {}
    for rnd in range(3):
        for v in variants:
            _lib.set_option("trunk_variant", v)
            for fn, nm in ((trunk, "trunk"), (full, "full")):
                for _ in range(5): fn()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                n = 200 if B <= 4096 else (50 if B <= 16384 else 20)
                e0.record()
                for _ in range(n): fn()
                e1.record(); torch.cuda.synchronize()
                res.setdefault((v, nm), []).append(e0.elapsed_time(e1) / n)
    for v in variants:
        t = min(res[(v, "trunk")]); f = min(res[(v, "full")])
        print(f"B={B:6d} variant {v}: trunk {t*1e3:8.1f} us ({B/t/1e3:6.2f} M boards/s)   trunk+heads {f*1e3:8.1f} us ({B/f/1e3:6.2f} M boards/s)", flush=True)
_lib.set_option("trunk_variant", 3)
