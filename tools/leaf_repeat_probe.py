#!/usr/bin/env python3
"""Round-4 measurement: how often does a self-play game ask the network for a position it has ALREADY asked for?

The reference builds a new tree for every move (pv_mcts.py:84: `root_node = Node(state, 0)`), so the sub-tree under the move that is
played is evaluated again by the next search, and transpositions inside one search are separate nodes with separate predict() calls.
The network's output is a pure function of (walls, pawns, walls in hand) -- not of the ply counter -- and the fused trunk computes every
board independently of its launch, so an evaluation cache keyed by the exact record would return bit-identical priors and values.
This tool plays G games with the GNN evaluator one simulation at a time (plain launches: step -> trunk + heads), copies every leaf
record, and reports per game phase which fraction of the evaluations repeats (a) a leaf of the same search, (b) a leaf of any earlier
search of the same game, (c) a leaf of the previous search only.  It changes nothing in the engine."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.engine import BatchedSelfPlay
from alphaquoridorgnn_amd.pv_network_gnn import GraphPolicyValueNetwork

G = int(os.environ.get("GAMES", "128"))
SIMS = int(os.environ.get("SIMS", "200"))
dev = _lib.require_gpu("cuda:0")
lib = _lib.load()
torch.manual_seed(0)
model = GraphPolicyValueNetwork().to(dev)
eng = BatchedSelfPlay(model, num_games=G, sims=SIMS, seed=1, record_history=False)
e, st = ctypes.byref(eng.e), eng._stream()
t = eng.t
sat = model.saturation_word(dev)
flags = int(model.gnn_flags(dev))
max_moves = eng.max_plies
log_state = torch.zeros((max_moves, SIMS, G, 24), dtype=torch.uint8, device=dev)
log_flag = torch.zeros((max_moves, SIMS, G), dtype=torch.uint8, device=dev)
moves = 0
while moves < max_moves:
    u = torch.rand((G,), dtype=torch.float64, device=dev, generator=eng.gen)
    _lib.check(lib.aqg_engine_begin_move(e, st), "begin")
    for sim in range(SIMS):
        _lib.check(lib.aqg_engine_step(e, 1 if sim else 0, 1, st), "step")
        log_state[moves, sim].copy_(t["leaf_state"])
        log_flag[moves, sim].copy_(t["leaf_flag"])
        _lib.check(lib.aqg_gcn_forward_boards_guarded(9, _lib.ptr(t["leaf_state"]), 1, G, _lib.ptr(t["packed_weights"]), _lib.ptr(t["pooled"]), None,
                                                      _lib.ptr(t["policy"]), None, _lib.ptr(t["value"]), flags, _lib.ptr(sat), st), "fwd")
    _lib.check(lib.aqg_engine_step(e, 1, 0, st), "step")
    _lib.check(lib.aqg_engine_finish_move(e, _lib.ptr(u), st), "finish")
    moves += 1
    if moves % 8 == 0 and eng.counters()["active"] == 0:
        break
torch.cuda.synchronize()
print(f"{G} games x {SIMS} sims, {moves} moves played, {eng.counters()}", flush=True)
ls = log_state[:moves].cpu().numpy()
lf = log_flag[:moves].cpu().numpy()
# key = the 24-byte record without its ply counter (bytes 20..21; 22..23 pad): hw 0..7, vw 8..15, ppos, pwl, epos, ewl 16..19
keys = np.ascontiguousarray(ls[..., :20])
tot = np.zeros(moves, dtype=np.int64); same = np.zeros(moves, dtype=np.int64); earlier = np.zeros(moves, dtype=np.int64); prev = np.zeros(moves, dtype=np.int64)
for g in range(G):
    seen_all = set()
    seen_prev = set()
    for m in range(moves):
        cur = set()
        for s in range(SIMS):
            if lf[m, s, g] != 1:
                continue
            k = keys[m, s, g].tobytes()
            tot[m] += 1
            if k in cur:
                same[m] += 1
            elif k in seen_all:
                earlier[m] += 1
                if k in seen_prev:
                    prev[m] += 1
            cur.add(k)
        seen_all |= cur
        seen_prev = cur
print("moves      evals   same-search  earlier-search (of which previous search)   any repeat")
for lo in range(0, moves, 10):
    hi = min(lo + 10, moves)
    T = tot[lo:hi].sum()
    if T == 0:
        continue
    print(f"{lo:3d}-{hi - 1:3d} {T:10d}   {same[lo:hi].sum() / T:8.3f}   {earlier[lo:hi].sum() / T:8.3f}        ({prev[lo:hi].sum() / T:6.3f})                {(same[lo:hi].sum() + earlier[lo:hi].sum()) / T:8.3f}")
T = tot.sum()
print(f"all     {T:10d}   {same.sum() / T:8.3f}   {earlier.sum() / T:8.3f}        ({prev.sum() / T:6.3f})                {(same.sum() + earlier.sum()) / T:8.3f}")
