#!/usr/bin/env python3
"""Diagnostic: anatomy of ONE trunk launch (AQG_B boards, default 480) from the -DAQG_TRACE build: when each workgroup starts
and ends on the 100 MHz s_memrealtime clock -- how much of the launch is ramp (staggered starts), chain (a workgroup's own
time) and tail."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = "/tmp/libaqgnn_hip_trace.so"
src = os.path.join(ROOT, "alphaquoridorgnn_amd", "csrc")
if not (os.environ.get("AQG_TRACE_REUSE") and os.path.exists(so)):      # (scans: build once)
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
pooled = torch.empty((B, 128), device=dev)
flags = model.gnn_flags(dev); word = model.saturation_word(dev)      # the module's own call: guarded entry, trunk only (no heads)
if os.environ.get("AQG_TRUNK_PRIO"): _lib.set_option("trunk_prio", int(os.environ["AQG_TRUNK_PRIO"]))
def fwd():
    _lib.check(lib.aqg_gcn_forward_boards_guarded(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(pooled), None, None, None, None, flags, _lib.ptr(word), _lib.stream_ptr(dev)), "fwd")
for _ in range(20): fwd()
torch.cuda.synchronize()
CAP = 200_000
buf = torch.zeros((1 + 4 * CAP,), dtype=torch.int64, device=dev)
_lib.check(lib.aqg_debug_trace(_lib.ptr(buf), CAP), "trace")
torch.cuda.synchronize()
for _ in range(20): fwd()
torch.cuda.synchronize()
_lib.check(lib.aqg_debug_trace(None, 0), "trace off")
n = int(buf[0].item()) & 0xFFFFFFFF
raw = buf[1:1 + 4 * min(n, CAP)].cpu().numpy().reshape(-1, 4)
t0 = raw[:, 2]; t1 = raw[:, 3] & ((1 << 48) - 1); wg = raw[:, 3] >> 48
hw = (raw[:, 0] >> 8) & 0xFFFF; xcc = (raw[:, 0] >> 24) & 0xF
cu = (xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)      # XCC, SE, SH, CU
order = np.argsort(t0); t0, t1, wg, cu, xcc = t0[order], t1[order], wg[order], cu[order], xcc[order]
# split into launches: a start more than 2 us after the running max end begins a new launch
starts = [0]; cur_end = t1[0]
for i in range(1, len(t0)):
    if t0[i] > cur_end + 50: starts.append(i); cur_end = t1[i]
    else: cur_end = max(cur_end, t1[i])
starts.append(len(t0))
print(f"{n} workgroup records, {len(starts) - 1} launches of {B} boards (10 ns ticks)")
rows = []
for a, b in zip(starts[:-1], starts[1:]):
    s, e = t0[a:b], t1[a:b]
    rows.append(((e.max() - s.min()) / 100, (s.max() - s.min()) / 100, np.median(e - s) / 100, (e - s).max() / 100, (e - s).min() / 100, (e.max() - np.median(e)) / 100, b - a))
r = np.array(rows[2:])
print("per launch (us):  span  | last start - first start | workgroup duration median / max / min | last end - median end | workgroups")
print("mean:            %5.2f  | %5.2f | %5.2f / %5.2f / %5.2f | %5.2f | %d" % tuple(r.mean(0)))
a, b = starts[5], starts[6]
s, e, w = t0[a:b], t1[a:b], wg[a:b]
print("one launch: start offsets (us) by decile:", np.round(np.percentile(s - s.min(), [0, 10, 25, 50, 75, 90, 100]) / 100, 2))
print("            durations (us) by decile:   ", np.round(np.percentile(e - s, [0, 10, 25, 50, 75, 90, 100]) / 100, 2))
first = (w < 256); print("            workgroups 0..255: mean start %.2f dur %.2f | 256..: mean start %.2f dur %.2f" % ((s[first] - s.min()).mean() / 100, (e - s)[first].mean() / 100, (s[~first] - s.min()).mean() / 100, (e - s)[~first].mean() / 100))
gap = [(t0[starts[i + 1]] - t1[starts[i]:starts[i + 1]].max()) / 100 for i in range(2, len(starts) - 2)]
print("gap between launches (last end -> next first start): mean %.2f us" % np.mean(gap))

# who shares a CU with whom: the older (first started) and the younger workgroup of every pair, and the workgroups that ran alone
lone, older, younger, lag = [], [], [], []
per_xcc = {}
for a, b in zip(starts[2:-1], starts[3:]):
    s, e, c, x = t0[a:b], t1[a:b], cu[a:b], xcc[a:b]
    for k in np.unique(c):
        i = np.nonzero(c == k)[0]
        if len(i) == 1: lone.append((e - s)[i[0]])
        elif len(i) == 2:
            o, y = (i[0], i[1]) if s[i[0]] <= s[i[1]] else (i[1], i[0])
            older.append((e - s)[o]); younger.append((e - s)[y]); lag.append(s[y] - s[o])
    for k in np.unique(x):
        per_xcc.setdefault(int(k), []).append(((e - s)[x == k].mean(), (e[x == k].max() - s.min())))
print("by CU: %d lone workgroups %.2f us | pairs: older %.2f us (max %.2f), younger %.2f us (max %.2f), younger starts %.2f us later" % (
    len(lone), np.mean(lone) / 100 if lone else 0, np.mean(older) / 100, np.max(older) / 100, np.mean(younger) / 100, np.max(younger) / 100, np.mean(lag) / 100))
print("by XCC: mean workgroup duration / last end since launch start (us):", {k: (round(np.mean([v[0] for v in vs]) / 100, 2), round(np.mean([v[1] for v in vs]) / 100, 2)) for k, vs in sorted(per_xcc.items())})
