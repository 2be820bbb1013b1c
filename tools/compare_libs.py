"""Bit-identity of two builds of libaqgnn_hip.so on the same inputs (cleanup / refactoring check).

    python tools/compare_libs.py tools/ubench/bin/libaqgnn_r3.so alphaquoridorgnn_amd/libaqgnn_hip.so

Each library runs in its own process (AQG_LIB_PATH): the GNN forward at 300 / 4,096 / 1,001 boards in both guard modes, a masked engine
launch inside a 64-game GNN-evaluated generation (history rows), and three training steps.  Prints one line per item; exit code 1
on any difference."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_WORKER = r'''
import sys, os
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.engine import BatchedSelfPlay
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
from alphaquoridorgnn_amd.train_network import GNNTrainer
from tools.microbench import synth_states
from oracle import gnn as og
dev = torch.device("cuda", 0)
lib = _lib.load()
out = {}
torch.manual_seed(0)
model = GNNNetwork().to(dev).eval()
pk = model.packed_weights(dev)
for B in (300, 4096, 1001):
    st = synth_states(B, seed=B, dev=dev)
    for flags in (0, 2):
        pooled = torch.zeros((B, 128), device=dev); policy = torch.zeros((B, 209), device=dev); value = torch.zeros((B,), device=dev)
        word = torch.zeros((1,), dtype=torch.int32, device=dev)
        _lib.check(lib.aqg_gcn_forward_boards_guarded(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(pooled), None, _lib.ptr(policy), None,
                                                      _lib.ptr(value), flags, _lib.ptr(word), _lib.stream_ptr(dev)), "fwd")
        out[f"pooled_{B}_{flags}"] = pooled.cpu().numpy(); out[f"policy_{B}_{flags}"] = policy.cpu().numpy(); out[f"value_{B}_{flags}"] = value.cpu().numpy()
eng = BatchedSelfPlay(model, num_games=64, sims=24, seed=11)
eng.play_generation()
s, v, z = eng.history_tensors()
out["hist_s"], out["hist_v"], out["hist_z"] = s.cpu().numpy(), v.cpu().numpy(), z.cpu().numpy()
tm = GNNNetwork()
tm.load_state_dict({k: torch.from_numpy(x.copy()) for k, x in og.init_params(3).items()})
tm = tm.to(dev)
tr = GNNTrainer(tm, max_batch=128)
rng = np.random.RandomState(0)
for i in range(3):
    st = synth_states(128 if i < 2 else 37, seed=50 + i, dev=dev)
    n = st.shape[0]
    pi = torch.softmax(torch.from_numpy(rng.randn(n, 209).astype(np.float32)), 1).to(dev)
    zz = torch.from_numpy(rng.choice([-1.0, 0.0, 1.0], n).astype(np.float32)).to(dev)
    tr.step(st, pi, zz)
for k, t in tm.state_dict().items():
    out["param_" + k] = t.cpu().numpy()
np.savez(sys.argv[2], **out)
'''


def run(libpath, outpath):
    env = dict(os.environ, AQG_LIB_PATH=os.path.abspath(libpath), GPU_MAX_HW_QUEUES="8")
    subprocess.run([sys.executable, "-c", _WORKER, ROOT, outpath], check=True, env=env, cwd=ROOT)
    return np.load(outpath)


def main():
    a, b = sys.argv[1:3]
    tmp = os.environ.get("TMPDIR", "/tmp")
    ra, rb = run(a, os.path.join(tmp, "cmp_a.npz")), run(b, os.path.join(tmp, "cmp_b.npz"))
    bad = 0
    for k in ra.files:
        same = np.array_equal(ra[k], rb[k], equal_nan=True)
        d = 0.0 if same else float(np.abs(ra[k].astype(np.float64) - rb[k].astype(np.float64)).max())
        print(f"{k:32s} {'identical' if same else 'DIFFERENT  max |d| = %g' % d}")
        bad += 0 if same else 1
    print("all identical" if not bad else f"{bad} items differ")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
