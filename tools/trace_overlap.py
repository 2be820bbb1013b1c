#!/usr/bin/env python3
"""Diagnostic: which kernels of different game sets REALLY run side by side.  Builds the library with -DAQG_TRACE (every
workgroup logs its start / end on the 100 MHz s_memrealtime clock), plays a few moves of the benchmark configuration and
prints, for one traced move: launch durations per kernel, the share of wall time with k launches in flight, how the trunk
launches of the sets overlap, and the idle gaps inside one set's chain.  (rocprofv3 --kernel-trace cannot show this: it
serialises the dispatches it intercepts -- the same bench runs 2.5x slower under it.)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = "/tmp/libaqgnn_hip_trace.so"
src = os.path.join(ROOT, "alphaquoridorgnn_amd", "csrc")
subprocess.check_call(f"cd {src} && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DAQG_TRACE "
                      f"legal_mask.hip gcn_forward.hip gcn_train.hip mcts.hip capi.hip host_agents.cpp -o {so} 2>/dev/null", shell=True)
os.environ["AQG_LIB_PATH"] = so
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
from collections import defaultdict
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.engine import MultiSetSelfPlay
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
sets = int(sys.argv[1]) if len(sys.argv) > 1 else 4
games = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
dev = _lib.require_gpu("cuda:0"); lib = _lib.load()
for k, v in (a.split("=") for a in sys.argv[3:]):
    _lib.set_option(k, int(v))
torch.manual_seed(0)
model = GNNNetwork().to(dev).eval()
eng = MultiSetSelfPlay(model, num_games=games, sims=200, num_sets=sets, seed=1)
for _ in range(12):
    eng.move()
eng.sync()
CAP = 3_000_000
buf = torch.zeros((1 + 4 * CAP,), dtype=torch.int64, device=dev)
_lib.check(lib.aqg_debug_trace(_lib.ptr(buf), CAP), "trace")
torch.cuda.synchronize()
eng.move(); eng.sync(); torch.cuda.synchronize()
_lib.check(lib.aqg_debug_trace(None, 0), "trace off")
n = int(buf[0].item()) & 0xFFFFFFFF
raw = buf[1:1 + 4 * min(n, CAP)].cpu().numpy().reshape(-1, 4)
kid, tag, t0 = raw[:, 0] & 0xFF, raw[:, 1], raw[:, 2]
t1 = raw[:, 3] & ((1 << 48) - 1)
print(f"{n} workgroup records; clock 100 MHz (10 ns ticks)")
names = {1: "step", 2: "trunk", 3: "heads"}
tags = {t: i for i, t in enumerate(sorted(set(tag.tolist())))}
# group workgroups into launches: same (kernel, tag), start times within a launch are close; a new launch starts after the previous ended
launches = []
for k in (1, 2, 3):
    for t in tags:
        m = (kid == k) & (tag == t)
        if not m.any():
            continue
        order = np.argsort(t0[m]); s, e = t0[m][order], t1[m][order]
        cs, ce, cnt = s[0], e[0], 1
        for a, b in zip(s[1:], e[1:]):
            if a > ce + 100:                      # > 1 us after everything seen so far ended: next launch
                launches.append((k, tags[t], cs, ce, cnt)); cs, ce, cnt = a, b, 1
            else:
                ce = max(ce, b); cnt += 1
        launches.append((k, tags[t], cs, ce, cnt))
L = np.array(launches, dtype=np.int64)
T0, T1 = L[:, 2].min(), L[:, 3].max()
print(f"traced move: {(T1 - T0) / 100:.0f} us wall, {len(L)} launches ({(T1 - T0) / 100 / 200:.1f} us per simulation round of all {sets} sets)")
for k in (1, 2, 3):
    d = (L[L[:, 0] == k, 3] - L[L[:, 0] == k, 2]) / 100
    print(f"  {names[k]:6s} launches {len(d):5d}  span mean {d.mean():6.2f} us  median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}   sum / wall {d.sum() / ((T1 - T0) / 100):.2f}")
# concurrency profile
ev = sorted([(int(a), 1, int(k)) for k, _, a, b, _ in L] + [(int(b), -1, int(k)) for k, _, a, b, _ in L])
depth = 0; last = T0; hist = defaultdict(int); act = defaultdict(int); mix = defaultdict(int)
for t, d, k in ev:
    hist[depth] += t - last
    mix[tuple(sorted((kk, v) for kk, v in act.items() if v))] += t - last
    last = t; depth += d; act[k] += d
wall = T1 - T0
print("  launches in flight: " + "  ".join(f"{k}: {100 * v / wall:.1f}%" for k, v in sorted(hist.items())))
top = sorted(mix.items(), key=lambda x: -x[1])[:8]
for m, v in top:
    print(f"    {100 * v / wall:5.1f}%  " + (" + ".join(f"{c}x{names[k]}" for k, c in m) or "idle"))
# per-set chain gaps
for s in range(min(sets, 2)):
    ch = L[L[:, 1] == s]; ch = ch[np.argsort(ch[:, 2])]
    gaps = (ch[1:, 2] - ch[:-1, 3]) / 100
    print(f"  set {s}: chain of {len(ch)} launches, gap between consecutive launches mean {gaps.mean():.2f} us median {np.median(gaps):.2f} us; busy {((ch[:, 3] - ch[:, 2]).sum()) / wall * 100:.0f}% of the wall")
