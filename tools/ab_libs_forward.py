#!/usr/bin/env python3
"""A/B of library builds on the GNN forward: python tools/ab_libs_forward.py libA.so libB.so ...  Every library in its own process
(AQG_LIB_PATH), twice, interleaved; trunk alone and trunk + heads at 480 / 4,096 / 65,536 boards (boards/s, best of 3 x N launches).
Bit identity of the builds: tools/compare_libs.py."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_WORKER = r'''
import sys, os, time
sys.path.insert(0, sys.argv[1])
import torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
from tools.microbench import synth_states, time_ms
dev = torch.device("cuda", 0); lib = _lib.load()
torch.manual_seed(0)
model = GNNNetwork().to(dev).eval(); pk = model.packed_weights(dev); flags = int(model.gnn_flags(dev)); word = model.saturation_word(dev)
res = []
for B, n in ((480, 400), (4096, 200), (65536, 30)):
    st = synth_states(B, seed=1, dev=dev)
    pooled = torch.empty((B, 128), device=dev); policy = torch.empty((B, 209), device=dev); value = torch.empty((B,), device=dev)
    def trunk():
        _lib.check(lib.aqg_gcn_forward_boards_guarded(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(pooled), None, None, None, None, flags, _lib.ptr(word), _lib.stream_ptr(dev)), "t")
    def full():
        _lib.check(lib.aqg_gcn_forward_boards_guarded(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(pooled), None, _lib.ptr(policy), None, _lib.ptr(value), flags, _lib.ptr(word), _lib.stream_ptr(dev)), "f")
    t = min(time_ms(trunk, n, warmup=10) for _ in range(3)); f = min(time_ms(full, n, warmup=10) for _ in range(3))
    res.append(f"B={B}: trunk {t * 1e3:8.1f} us ({B / t / 1e3:6.2f} M/s)  trunk+heads {f * 1e3:8.1f} us ({B / f / 1e3:6.2f} M/s)")
print(" | ".join(res))
'''
libs = sys.argv[1:]
for rep in range(2):
    for lib in libs:
        env = dict(os.environ, AQG_LIB_PATH=os.path.abspath(lib), GPU_MAX_HW_QUEUES="8")
        out = subprocess.run([sys.executable, "-c", _WORKER, ROOT], env=env, cwd=ROOT, capture_output=True, text=True)
        print(f"{os.path.basename(lib):28s} {out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:]}", flush=True)
