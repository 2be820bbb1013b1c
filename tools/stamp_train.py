#!/usr/bin/env python3
"""Diagnostic: build with -DAQG_STAMP and print where workgroup 0 of each training kernel spends its cycles
(batch 128).  Read the SHARES; never quote this build's run time."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = "/tmp/libaqgnn_hip_stamp.so"
src = os.path.join(ROOT, "alphaquoridorgnn_amd", "csrc")
subprocess.check_call(f"cd {src} && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DAQG_STAMP "
                      f"legal_mask.hip gcn_forward.hip gcn_train.hip mcts.hip capi.hip host_agents.cpp -o {so} 2>/dev/null", shell=True)
os.environ["AQG_LIB_PATH"] = so
import torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
from alphaquoridorgnn_amd.train_network import GNNTrainer, BATCH_SIZE
from tools.microbench import synth_states
dev = _lib.require_gpu("cuda:0"); lib = _lib.load()
_lib.set_option("train_fused", int(os.environ.get("AQG_TRAIN_FUSED", "2")))
model = GNNNetwork().to(dev); tr = GNNTrainer(model, max_batch=BATCH_SIZE)
n = BATCH_SIZE * 50
st = synth_states(n); A = model.policy_output_size
pi = torch.rand((n, A), device=dev); pi = pi / pi.sum(1, keepdim=True)
z = torch.randint(-1, 2, (n,), device=dev).float()
order = torch.randperm(n, device=dev)
buf = (ctypes.c_ulonglong * 160)()
fn = lib.aqg_debug_train_stamps; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
tr.run_epoch(st, pi, z, order[:BATCH_SIZE * 5]); torch.cuda.synchronize(); fn(buf, 1)
tr.run_epoch(st, pi, z, order); torch.cuda.synchronize(); fn(buf, 0)
names = {0: ("fwd12", ["loads + graph", "layer-1 mfma", "aggregate 1 (128 cols)", "layer-2 mfma", "acc -> LDS + barrier", "aggregate 2 + stores"]),
         1: ("fwd3", ["loads + graph", "mfma", "acc -> LDS + barrier", "aggregate + stores", "pool"]),
         2: ("heads", ["g load", "hidden layers", "logits", "softmax/loss reductions", "dhp partials", "dhs", "dg partials"]),
         3: ("bwd<3>", ["-", "-", "loads + graph + mask", "acc -> LDS + barrier", "aggregate + stores", "weight gradient"]),
         4: ("bwd<2>", ["loads + graph", "dgrad mfma", "mask + barrier", "acc -> LDS + barrier", "aggregate + stores", "weight gradient"]),
         5: ("bwd<1>", ["loads + graph", "dgrad mfma", "mask + barrier", "acc -> LDS + barrier", "aggregate", "weight gradient"]),
         6: ("final", ["index", "sums", ]),}
names[7] = ("heads inside the fused kernel", ["wait for pooled g", "hidden layers", "logits", "softmax/loss reductions", "dhp partials", "dhs", "dg partials"])
names[8] = ("train_board (fused)", ["loads + graph", "layer 1 + aggregate", "layer-2 mfma", "acc->LDS, aggregate 2", "layer-3 mfma", "aggregate 3 + pool", "heads (all)",
                                    "bwd3: mask, dP, aggregate", "bwd3: weight gradient", "bwd2: dgrad mfma", "bwd2: mask, dP, aggregate", "bwd2: weight gradient",
                                    "bwd1: dgrad mfma", "bwd1: mask, dP, aggregate", "bwd1: weight gradient"])
names[9] = ("train_board_split (fused, fp16 split MFMA)", ["loads + board tables", "A_hat fragments + layer-1 linear", "layer 1: aggregation T+R, stores", "layer-2 linear",
                                                            "layer 2: aggregation T+R, stores", "layer-3 linear + aggregation R + pool", "heads (all)",
                                                            "bwd3: dP, aggregation T+R, stores", "bwd3: weight gradient", "bwd2: data gradient + mask", "bwd2: aggregation T+R, stores",
                                                            "bwd2: weight gradient", "bwd1: data gradient + mask", "bwd1: aggregation R + weight gradient",
                                                            "(wave 0) layer 1: aggregation + epilogues", "(wave 0) layer 1: park, W2 split"])
fused = os.environ.get("AQG_TRAIN_FUSED", "2")
if fused == "2":
    names = {k: v for k, v in names.items() if k in (6, 7, 9)}
elif fused == "1":
    names = {k: v for k, v in names.items() if k in (6, 7, 8)}
else:
    names = {k: v for k, v in names.items() if k < 7}
for k, (nm, ph) in names.items():
    row = [buf[k * 16 + i] / 50 for i in range(16)]
    print(f"{nm}: total {sum(row):.0f} cycles per launch (workgroup 0)")
    for i, p in enumerate(ph):
        print(f"    {p:28s} {row[i]:9.0f}")
