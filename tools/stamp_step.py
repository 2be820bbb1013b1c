#!/usr/bin/env python3
"""Diagnostic: build with -DAQG_STAMP and print where the cycles of the fused MCTS step go (per game and simulation,
averaged over 512 games x a few moves; fake evaluator).  Read the SHARES, not the absolute run time of this build.
AQG_LEVELS=1 adds -DAQG_STAMP_LEVELS: four stamps inside every tree level below the root (each is a read-modify-write of global memory
in the loop: the level then takes ~50 % longer -- read their RATIOS only)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = "/tmp/libaqgnn_hip_stamp.so"
src = os.path.join(ROOT, "alphaquoridorgnn_amd", "csrc")
subprocess.check_call(f"cd {src} && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DAQG_STAMP {'-DAQG_STAMP_LEVELS' if os.environ.get('AQG_LEVELS') else ''} "
                      f"legal_mask.hip gcn_forward.hip gcn_train.hip mcts.hip capi.hip host_agents.cpp -o {so} 2>/dev/null", shell=True)
os.environ["AQG_LIB_PATH"] = so
import torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.engine import BatchedSelfPlay
dev = _lib.require_gpu("cuda:0")
_lib.set_option("use_graph", 0)
for warm_moves, moves in ((0, 4), (20, 4), (60, 4)):
    eng = BatchedSelfPlay(None, num_games=512, sims=200, evaluator="fake", fake_bias=0, record_history=False)
    for _ in range(warm_moves):
        eng.move()
    eng.t["pooled"].zero_()
    for _ in range(moves):
        eng.move()
    torch.cuda.synchronize()
    full = eng.t["pooled"].view(torch.int64).view(512, 64).double().cpu()
    raw = full[:, :12]
    steps = raw[:, 7].sum().item()
    names = ["round-1 loads landed", "expand + old backup (issue)", "descent (all levels)", "legal actions of the leaf", "tail stores landed"]
    tot = raw[:, :5].sum().item()
    print(f"after {warm_moves} moves: {steps:.0f} game-steps, {tot / steps:.0f} cycles per step, {raw[:, 6].sum().item() / steps:.2f} levels per descent")
    for i, n in enumerate(names):
        print(f"   {n:30s} {raw[:, i].sum().item() / steps:9.0f} cycles  {100 * raw[:, i].sum().item() / tot:5.1f} %")
    lv = raw[:, 6].sum().item()                      # levels descended (the root level is not stamped: its children came with round 1)
    for i, n in () if not os.environ.get("AQG_LEVELS") else ((11, "chosen -> next request"), (8, "request -> state advanced"), (9, "... -> children arrived"), (10, "... -> child chosen")):
        print(f"      per level below the root: {n:28s} {raw[:, i].sum().item() / max(lv, 1):7.0f} cycles")

    if os.environ.get("AQG_TAIL"):
        mx = full[:, 16]
        order = torch.argsort(mx, descending=True)[:16]
        print(f"   longest step per game: median {mx.median().item():.0f}, max {mx.max().item():.0f} cycles; the 16 slowest games' longest steps, mean per phase:")
        for i, n in enumerate(names):
            print(f"      {n:30s} {full[order, 17 + i].mean().item():9.0f}")
        print(f"      depth {full[order, 22].mean().item():.1f}, terminal {full[order, 23].mean().item():.2f}")
        hist = full[:, 24:64].sum(0)
        cum = torch.cumsum(hist, 0) / hist.sum()
        print("   step length histogram (2,048-cycle buckets; share of steps at or below): " + " ".join(f"{(b + 1) * 2}k:{cum[b].item():.3f}" for b in range(4, 24)))
