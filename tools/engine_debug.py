#!/usr/bin/env python3
"""Developer diagnostic: root priors stored by the engine vs the stand-alone forward vs the fp64 oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.engine import BatchedSelfPlay
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
from oracle import gnn as og, quoridor as oq
dev = _lib.require_gpu("cuda:0"); lib = _lib.load()
params = og.init_params(4)
model = GNNNetwork(); model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()}); model = model.to(dev).eval()
g = np.load(os.path.join(ROOT, "tests", "golden", "walk_9x9.npz"))
recs = g["states"][[0, 5, 40, 333, 1200, 2600, 5000]]
oracle = og.OracleModel(params)
pol, val = model.forward_states(torch.from_numpy(recs).to(dev))
pol = pol.cpu().numpy()
for sv in (0, 1):
    _lib.set_option("step_variant", sv)
    eng = BatchedSelfPlay(model, num_games=recs.shape[0], sims=10, record_history=False)
    eng.search(recs); torch.cuda.synchronize()
    G, cap = eng.G, eng.node_cap
    raw = eng.t["node_rec"].view(torch.uint8).view(G, cap, 32).cpu().numpy()
    epol = eng.t["policy"].cpu().numpy()
    for i, rec in enumerate(recs):
        st = oq.State(rec); legal = st.legal_actions()
        kids = int(raw[i, 0, 20:24].view(np.uint32)[0]); first, cnt = kids & 0xFFFFFF, kids >> 24
        pri = raw[i, first:first + cnt, 8:12].copy().view(np.float32)[:, 0]
        want, _ = oracle.predict(st)
        fwd = pol[i][legal]; fwd = fwd / fwd.sum()
        print(f"step_variant {sv} state {i}: n_legal {len(legal)} cnt {cnt}  |engine-oracle| {np.abs(pri - want).max():.2e}  |fwd-oracle| {np.abs(fwd - want).max():.2e}")
_lib.set_option("step_variant", 1)

# ---- stand-alone forward on 24-byte packed records (state_fmt 1) vs state72 (state_fmt 0)
def qstate(rec):
    hw = sum(1 << i for i in range(64) if rec[4 + i] == 1); vw = sum(1 << i for i in range(64) if rec[4 + i] == 2)
    m = int(rec[0]) | int(rec[1]) << 8 | int(rec[2]) << 16 | int(rec[3]) << 24 | (int(rec[68]) | int(rec[69]) << 8) << 32
    return np.array([hw, vw, m], dtype=np.uint64).view(np.uint8)
q = torch.from_numpy(np.stack([qstate(r) for r in recs])).to(dev)
B = recs.shape[0]
for fmt, st in ((0, torch.from_numpy(recs).to(dev)), (1, q)):
    for v in (1, 6):
        _lib.set_option("trunk_variant", v)
        pooled = torch.zeros((B, 128), device=dev); logits = torch.zeros((B, 209), device=dev)
        _lib.check(lib.aqg_gcn_forward_boards(9, _lib.ptr(st), fmt, B, _lib.ptr(model.packed_weights(dev)), _lib.ptr(pooled), _lib.ptr(logits), None, None, None, 0, _lib.stream_ptr(dev)), "f")
        ref = og.forward_states(params, recs)
        print(f"fmt {fmt} variant {v}: |logits - oracle| per board {np.abs(logits.cpu().numpy() - ref['logits']).max(1)}")
_lib.set_option("trunk_variant", 3)
print("leaf/root states of the engine (24 bytes) vs expected:")
eng = BatchedSelfPlay(model, num_games=recs.shape[0], sims=1, record_history=False)
eng.search(recs); torch.cuda.synchronize()
ls = eng.t["leaf_state"].cpu().numpy()
for i in range(B):
    print(i, np.array_equal(ls[i], qstate(recs[i])), ls[i].view(np.uint64), qstate(recs[i]).view(np.uint64))
