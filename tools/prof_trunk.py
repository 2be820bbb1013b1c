#!/usr/bin/env python3
"""Run the GNN forward a few times for rocprofv3 (kernel-trace or --pmc passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
from tools.microbench import synth_states

B = int(os.environ.get("AQG_B", "65536"))
variant = int(os.environ.get("AQG_VARIANT", "3"))
iters = int(os.environ.get("AQG_ITERS", "5"))
dev = _lib.require_gpu("cuda:0")
lib = _lib.load()
_lib.set_option("trunk_variant", variant)
model = GNNNetwork().to(dev).eval()
pk = model.packed_weights(dev)
st = synth_states(B)
pooled = torch.empty((B, 128), device=dev)
policy = torch.empty((B, 209), device=dev)
value = torch.empty((B,), device=dev)
flags = model.gnn_flags(dev)             # the module's own call: guarded entry, this weight set's flags (range guard: proven bound or tracking)
word = model.saturation_word(dev)
for _ in range(iters):
    _lib.check(lib.aqg_gcn_forward_boards_guarded(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(pooled), None, _lib.ptr(policy), None,
                                                  _lib.ptr(value), flags, _lib.ptr(word), _lib.stream_ptr(dev)), "fwd")
torch.cuda.synchronize()
print("done", B, variant)
