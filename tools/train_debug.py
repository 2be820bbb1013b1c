#!/usr/bin/env python3
"""Developer diagnostic (-DAQG_TRAIN_DEBUG build): dense dumps of the layer-2 backward intermediates (dZ3, dH2 before / after the ReLU
mask) of the split-precision training step against the f32 step's, position by position."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
so = "/tmp/libaqgnn_hip_dbg.so"
src = os.path.join(ROOT, "alphaquoridorgnn_amd", "csrc")
subprocess.check_call(f"cd {src} && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DAQG_TRAIN_DEBUG "
                      f"legal_mask.hip gcn_forward.hip gcn_train.hip mcts.hip capi.hip host_agents.cpp -o {so} 2>/dev/null", shell=True)
os.environ["AQG_LIB_PATH"] = so
import numpy as np, torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.train_network import GNNTrainer
from test_gpu_parity import _train_batch, _model
dev = _lib.require_gpu("cuda:0"); lib = _lib.load()
model, params = _model(6)
B = 32
recs, pi, z = _train_batch(B, 11)
buf = torch.zeros((3, B, 96, 128), device=dev)
fn = lib.aqg_debug_train_buf; fn.restype = ctypes.c_int; fn.argtypes = [ctypes.c_void_p]
assert fn(buf.data_ptr()) == 0
out = {}
for fused in (1, 2):
    _lib.set_option("train_fused", fused)
    buf.zero_()
    tr = GNNTrainer(model, max_batch=B)
    tr.step(torch.from_numpy(recs), torch.from_numpy(pi), torch.from_numpy(z), update=False)
    torch.cuda.synchronize()
    out[fused] = buf[:, :, :81].cpu().numpy().astype(np.float64)
for slot, name in ((2, "dZ3"), (1, "dH2 before the mask"), (0, "dP2 = dH2 after the mask")):
    a, b = out[1][slot], out[2][slot]
    d = np.abs(a - b)
    print(f"{name:28s} max |f32| {np.abs(a).max():.3e}  max |diff| {d.max():.3e}  rel to max {d.max() / np.abs(a).max():.2e}   mean |diff| / mean |f32| {d.mean() / np.abs(a).mean():.2e}")
    idx = np.unravel_index(np.argmax(d), d.shape)
    print(f"     worst at (b, n, col) = {idx}: f32 {a[idx]:.6e} split {b[idx]:.6e}")
    if slot == 0:
        mism = (a == 0) != (b == 0)
        print(f"     mask mismatches: {int(mism.sum())} of {mism.size}; by node: {np.nonzero(mism.any(axis=(0, 2)))[0][:20]}; by col: {np.nonzero(mism.any(axis=(0, 1)))[0][:20]}")
