#!/usr/bin/env python3
"""Diagnostic (round 4): one GNN evaluation of AQG_B boards (default 480) as the MCTS makes it -- trunk launch, then heads launch (8 waves
per 16 boards) -- from the -DAQG_TRACE build (every workgroup's start / end on the 100 MHz s_memrealtime clock): first workgroup start ->
last workgroup end of a whole evaluation, the trunk workgroups' durations, the heads workgroups' durations.
(At commit d13f38b this tool also traced the one-launch form -- heads by the trunk workgroup that pools a 16-board group's last board --
whose result is kept in profiles/r04_heads_by_last_finisher_trace.log: 25.7 against 23.4 us per evaluation.)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = "/tmp/libaqgnn_hip_trace.so"
src = os.path.join(ROOT, "alphaquoridorgnn_amd", "csrc")
subprocess.check_call(f"cd {src} && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DAQG_TRACE "
                      f"legal_mask.hip gcn_forward.hip gcn_train.hip mcts.hip capi.hip host_agents.cpp -o {so} 2>/dev/null", shell=True)
os.environ["AQG_LIB_PATH"] = so
import numpy as np, torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
from tools.microbench import synth_states
dev = _lib.require_gpu("cuda:0"); lib = _lib.load()
B = int(os.environ.get("AQG_B", "480"))
model = GNNNetwork().to(dev).eval(); pk = model.packed_weights(dev)
st = synth_states(B)
pooled = torch.empty((B, 128), device=dev); policy = torch.empty((B, 209), device=dev); value = torch.empty((B,), device=dev)
flags = model.gnn_flags(dev); word = model.saturation_word(dev)
def fwd(fused):
    _lib.check(lib.aqg_gcn_forward_boards_guarded(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(pooled), None, _lib.ptr(policy), None, _lib.ptr(value), flags,
                                                  _lib.ptr(word), _lib.stream_ptr(dev)), "fwd")
CAP = 200_000
for fused in (False, False):
    for _ in range(10): fwd(fused)
    torch.cuda.synchronize()
    buf = torch.zeros((1 + 4 * CAP,), dtype=torch.int64, device=dev)
    _lib.check(lib.aqg_debug_trace(_lib.ptr(buf), CAP), "trace")
    torch.cuda.synchronize()
    for _ in range(20):
        fwd(fused)
        torch.cuda.synchronize()
    _lib.check(lib.aqg_debug_trace(None, 0), "trace off")
    n = int(buf[0].item()) & 0xFFFFFFFF
    raw = buf[1:1 + 4 * min(n, CAP)].cpu().numpy().reshape(-1, 4)
    kid = raw[:, 0] & 0xFF
    t0 = raw[:, 2]; t1 = raw[:, 3] & ((1 << 48) - 1)
    order = np.argsort(t0); t0, t1, kid = t0[order], t1[order], kid[order]
    calls = [0]; cur_end = t1[0]
    for i in range(1, len(t0)):
        if t0[i] > cur_end + 300: calls.append(i)              # > 3 us of nothing: the next evaluation (a host synchronisation lies between)
        cur_end = max(cur_end, t1[i])
    calls.append(len(t0))
    span, tdur, hdur, tmax = [], [], [], []
    for a, b in zip(calls[2:-1], calls[3:]):
        s, e, k = t0[a:b], t1[a:b], kid[a:b]
        span.append((e.max() - s.min()) / 100)
        tr = k == 2
        d = (e - s)[tr] / 100
        tdur.append(np.median(d)); tmax.append(d.max())
        if (k == 3).any(): hdur.append(((e - s)[k == 3] / 100).mean())
    name = "one launch (heads by the last finisher)" if fused else "two launches (trunk, heads)"
    print(f"{name:42s} {B} boards: evaluation first start -> last end {np.mean(span):6.2f} us (min {np.min(span):.2f}) | trunk workgroups: median {np.mean(tdur):.2f} us, "
          f"longest {np.mean(tmax):.2f} us" + (f" | heads workgroups {np.mean(hdur):.2f} us" if hdur else ""))
    if fused:
        a, b = calls[5], calls[6]
        d = np.sort((t1[a:b] - t0[a:b]) / 100)
        ng = (B + 15) // 16
        print(f"      one evaluation: the {ng} longest workgroups (they ran a group's heads) {d[-ng:].mean():.2f} us, the others {d[:-ng].mean():.2f} us (median {np.median(d[:-ng]):.2f})")
