#!/usr/bin/env python3
"""Developer microbenchmarks on one MI355X: per-kernel timings with HIP events on torch's current stream
(the stream every C-ABI call is enqueued on).  Variants are interleaved in one process (guide rule 24)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from alphaquoridorgnn_amd import _lib, game_logic  # noqa: E402
from alphaquoridorgnn_amd.engine import BatchedSelfPlay  # noqa: E402
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork  # noqa: E402


def synth_states(B, seed=0, p_wall=0.3, dev="cuda"):
    """Synthetic boards: random legal play from the start position on the GPU engine's own rules
    (uniform over legal pawn moves / wall placements, P(wall)=p_wall), sampled at random plies."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    n_games = min(B, 4096)
    rec = torch.zeros((n_games, 72), dtype=torch.uint8)
    rec[:, 0] = 76; rec[:, 1] = 10; rec[:, 2] = 76; rec[:, 3] = 10; rec[:, 70] = 9
    rec = rec.to(dev)
    pool = [rec.clone()]
    for ply in range(40):
        mask, order, count = game_logic.legal_actions_batch(rec, 9)
        m = mask.float()
        pawn = m[:, :81]
        wall = m[:, 81:]
        has_wall = wall.sum(1) > 0
        use_wall = (torch.rand(n_games, generator=g).to(dev) < p_wall) & has_wall
        w = torch.where(use_wall.unsqueeze(1), torch.cat([torch.zeros_like(pawn), wall], 1), torch.cat([pawn, torch.zeros_like(wall)], 1))
        w = w + 1e-9 * m
        act = torch.multinomial(w.cpu(), 1, generator=g).squeeze(1).to(dev)
        nxt = game_logic.next_batch(rec, act, 9)
        done = game_logic.status_batch(nxt, 9, 116) != 0
        rec = torch.where(done.unsqueeze(1), rec, nxt)
        pool.append(rec.clone())
    pool = torch.cat(pool, 0)
    idx = torch.randint(0, pool.shape[0], (B,), generator=g).to(dev)
    return pool[idx].contiguous()


def board_graph_batch(states72):
    """The boards of `states72` (uint8 [B,72], 9x9, device) as the arguments of the reference's literal operator
    forward(x, edge_index, batch) (pv_network_gnn.py:53): node features pv_network_cnn.py:88-114 as x [B*81, 6], the wall-cut
    4-neighbour grid (game_logic.py:145-167) as a directed edge list in both directions, batch [B*81].  Torch ops on the device."""
    dev = states72.device
    B = states72.shape[0]
    N, S, V = 9, 8, 81
    r = states72.long()
    W = r[:, 4:68].view(B, S, S)
    Hh = torch.zeros((B, S, N + 1), dtype=torch.bool, device=dev); Hh[:, :, 1:N] = W == 1
    Vv = torch.zeros((B, N + 1, S), dtype=torch.bool, device=dev); Vv[:, 1:N, :] = W == 2
    down = ~(Hh[:, :, 1:] | Hh[:, :, :N])                        # [B, S, N]: open between rows x and x + 1 at column y
    right = ~(Vv[:, 1:, :] | Vv[:, :N, :])                       # [B, N, S]: open between columns y and y + 1 at row x
    base = (torch.arange(B, device=dev) * V).view(B, 1, 1)
    td = (torch.arange(S, device=dev).view(S, 1) * N + torch.arange(N, device=dev).view(1, N)).view(1, S, N) + base
    tr = (torch.arange(N, device=dev).view(N, 1) * N + torch.arange(S, device=dev).view(1, S)).view(1, N, S) + base
    a, b = td[down], tr[right]
    src = torch.cat([a, a + N, b, b + 1]); dst = torch.cat([a + N, a, b + 1, b])
    x = torch.zeros((B, V, 6), dtype=torch.float32, device=dev)
    rows = torch.arange(B, device=dev)
    x[rows, r[:, 0], 0] = 1.0
    x[:, :, 1] = r[:, 1].float().unsqueeze(1)
    x[rows, r[:, 2], 2] = 1.0
    x[:, :, 3] = r[:, 3].float().unsqueeze(1)
    tw = N * (torch.arange(S * S, device=dev) // S) + torch.arange(S * S, device=dev) % S
    x[:, tw, 4] = (r[:, 4:68] == 1).float()
    x[:, tw, 5] = (r[:, 4:68] == 2).float()
    batch = torch.arange(B, device=dev).repeat_interleave(V)
    return x.view(B * V, 6), torch.stack([src, dst]), batch


def time_ms(fn, iters, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="gnn,legal,mcts")
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    dev = _lib.require_gpu("cuda:0")
    lib = _lib.load()
    out = {}
    model = GNNNetwork().to(dev).eval()
    pk = model.packed_weights(dev)
    if "gnn" in args.what:
        for B in (4096, 16384, 65536):
            st = synth_states(B)
            pooled = torch.empty((B, 128), device=dev)
            policy = torch.empty((B, 209), device=dev)
            value = torch.empty((B,), device=dev)

            def trunk():
                _lib.check(lib.aqg_gcn_forward_boards(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(pooled), None, None, None, None,
                                                      0, _lib.stream_ptr(dev)), "trunk")

            def full():
                _lib.check(lib.aqg_gcn_forward_boards(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(pooled), None, _lib.ptr(policy),
                                                      None, _lib.ptr(value), 0, _lib.stream_ptr(dev)), "full")
            for rnd in range(2):
                for v in (1, 6, 4):
                    _lib.set_option("trunk_variant", v)
                    t = time_ms(trunk, args.iters)
                    f = time_ms(full, args.iters)
                    out[f"gnn_B{B}_v{v}_r{rnd}"] = dict(trunk_ms=round(t, 4), full_ms=round(f, 4), boards_per_s=round(B / f * 1e3),
                                                        mfma_frac=round(B / t * 1e3 * 5432832 / 157.3e12, 4))
            _lib.set_option("trunk_variant", 3)
    if "legal" in args.what:
        for B in (4096, 65536):
            st = synth_states(B, seed=1)
            order = torch.empty((B, 136), dtype=torch.uint8, device=dev)
            count = torch.empty((B,), dtype=torch.int32, device=dev)

            def legal():
                _lib.check(lib.aqg_legal_actions(9, _lib.ptr(st), B, None, _lib.ptr(order), _lib.ptr(count), _lib.stream_ptr(dev)), "legal")
            t = time_ms(legal, args.iters)
            out[f"legal_B{B}"] = dict(ms=round(t, 4), states_per_s=round(B / t * 1e3), mean_legal=float(count.float().mean()))
    if "mcts" in args.what:
        for G, sims in ((2048, 200), (8192, 200)):
            eng = BatchedSelfPlay(model, num_games=G, sims=sims, seed=0, record_history=True)
            eng.move(); torch.cuda.synchronize()
            t0 = time.time()
            nm = 3
            for _ in range(nm):
                eng.move()
            torch.cuda.synchronize()
            dt = (time.time() - t0) / nm
            c = eng.counters()
            out[f"mcts_G{G}_s{sims}"] = dict(s_per_move=round(dt, 4), us_per_sim_step=round(dt / sims * 1e6, 1),
                                             leaf_evals_per_s=round(G * sims / dt), counters=c)
            del eng
            torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
