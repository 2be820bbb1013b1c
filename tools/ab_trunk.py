#!/usr/bin/env python3
"""Developer A/B of trunk builds on one GPU box: every argument is a set of -D flags (e.g. "-DAQG_PREFETCH=1 -DAQG_AF_AT=0"; "-"
= the defaults); each set is compiled into its own library on the box, then the trunk (no heads) is timed at several launch sizes
in ONE process with the variants' launches interleaved round by round (guide rule 24); the pooled outputs are compared as well.
  python tools/ab_trunk.py [--sizes 480,4096,65536] [--rounds 2] [--bench] "<flags A>" "<flags B>" ...
A flag set may start with FILE=<path relative to the repo root> to compile another version of gcn_forward.hip (e.g. last round's).
--bench also runs bench.py's headline generation (2 timed steps, no extra legs) per variant."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "alphaquoridorgnn_amd", "csrc")

_WORKER = r"""
# every variant's library is loaded into THIS process (ctypes handles are independent), launches interleaved round by round
import os, sys, json, ctypes
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork, STATE_DICT_KEYS
from tools.microbench import synth_states
dev = _lib.require_gpu("cuda:0")
_lib.load()                                   # the in-tree library serves synth_states (rules kernels)
paths = sys.argv[3].split(",")
model = GNNNetwork()
sd = model.state_dict()
host = [sd[k].detach().to("cpu", torch.float32).contiguous() for k in STATE_DICT_KEYS]
arr = (ctypes.c_void_p * 14)(*[ctypes.c_void_p(t.data_ptr()) for t in host])
libs = []
for pth in paths:
    L = ctypes.CDLL(pth)
    for name, (res, args) in _lib.SIGNATURES.items():
        if hasattr(L, name):
            fn = getattr(L, name); fn.restype = res; fn.argtypes = args
    out = torch.empty(L.aqg_gcn_packed_floats(9), dtype=torch.float32)
    assert L.aqg_gcn_pack_weights_host(9, arr, ctypes.c_void_p(out.data_ptr())) == 0
    libs.append((L, out.to(dev)))
res = {i: {} for i in range(len(libs))}
stream = _lib.stream_ptr(dev)
FLAGS = int(os.environ.get("AQG_AB_FLAGS", "0"))
word = torch.zeros((1,), dtype=torch.int32, device=dev)
for B in [int(x) for x in sys.argv[2].split(",")]:
    st = synth_states(B); pooled = torch.empty((B, 128), device=dev)
    ref = None
    for rnd in range(int(sys.argv[4])):
        for i, (L, pk) in enumerate(libs):
            def trunk():
                if FLAGS:      # AQG_AB_FLAGS=2: the build without per-value range tracking (AQG_GNN_RANGE_PROVEN needs the guarded entry)
                    assert L.aqg_gcn_forward_boards_guarded(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(pooled), None, None, None, None, FLAGS, _lib.ptr(word), stream) == 0
                else:
                    assert L.aqg_gcn_forward_boards(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(pooled), None, None, None, None, 0, stream) == 0
            for _ in range(10): trunk()
            n = 300 if B <= 4096 else 30
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n): trunk()
            e1.record(); torch.cuda.synchronize()
            res[i].setdefault(B, []).append(e0.elapsed_time(e1) / n * 1e3)
            if rnd == 0:                      # the variants must agree on the numbers, too
                if ref is None: ref = pooled.clone()
                else: res[i].setdefault("maxdiff", []).append(float((pooled - ref).abs().max()))
print("RESULT " + json.dumps(res))
"""


def main():
    args = sys.argv[1:]
    sizes, rounds, bench = "480,4096,65536", 2, False
    while args and args[0].startswith("--"):
        if args[0] == "--sizes":
            sizes = args[1]; args = args[2:]
        elif args[0] == "--rounds":
            rounds = int(args[1]); args = args[2:]
        elif args[0] == "--bench":
            bench = True; args = args[1:]
        else:
            raise SystemExit("unknown option " + args[0])
    variants = args or ["-"]
    libs = []
    procs = []
    cc = "/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC"
    common = ["legal_mask", "gcn_train", "mcts", "capi"]        # the flags only touch gcn_forward.hip: everything else is built once
    for f in common:
        procs.append((f, subprocess.Popen(f"cd {SRC} && {cc} -c {f}.hip -o /tmp/ab_{f}.o", shell=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)))
    procs.append(("host_agents", subprocess.Popen(f"cd {SRC} && g++ -O2 -std=c++17 -fPIC -c host_agents.cpp -o /tmp/ab_host_agents.o", shell=True,
                                                  stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)))
    for i, fl in enumerate(variants):
        flags = "" if fl == "-" else fl
        src = "gcn_forward.hip"
        if flags.startswith("FILE="):                            # another version of the source file (path relative to the repo root)
            path, _, flags = flags[5:].partition(" ")
            src = f"-I{SRC} " + os.path.join(ROOT, path)
        procs.append((f"gcn_forward[{fl}]", subprocess.Popen(f"cd {SRC} && {cc} {flags} -c {src} -o /tmp/ab_gf_{i}.o", shell=True,
                                                             stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)))
    for name, p in procs:
        err = p.communicate()[1].decode()
        if p.returncode != 0:
            raise SystemExit(f"build failed for {name}:\n{err[-2000:]}")
    objs = " ".join(f"/tmp/ab_{f}.o" for f in common + ["host_agents"])
    for i in range(len(variants)):
        so = f"/tmp/libaqgnn_ab_{i}.so"
        subprocess.check_call(f"/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o {so} {objs} /tmp/ab_gf_{i}.o", shell=True)
        libs.append(so)
    print("built", len(libs), "variants", flush=True)
    out = subprocess.run([sys.executable, "-c", _WORKER, ROOT, sizes, ",".join(libs), str(rounds)], capture_output=True, text=True, timeout=900)
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
    if not line:
        raise SystemExit(f"worker failed:\n{out.stdout[-1500:]}{out.stderr[-3000:]}")
    res = {int(k): v for k, v in json.loads(line[0][7:]).items()}
    for i, fl in enumerate(variants):
        cells = "  ".join(f"B={B}: {min(v):8.1f} us ({int(B) / min(v):6.2f} M/s)" for B, v in res[i].items() if B != "maxdiff")
        print(f"[{i}] {fl:45s} {cells}   max |pooled - variant 0| {max(res[i].get('maxdiff', [0.0])):.2e}", flush=True)
    if bench:
        for i, so in enumerate(libs):
            env = dict(os.environ, AQG_LIB_PATH=so)
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra-legs"],
                                 env=env, capture_output=True, text=True, timeout=900)
            try:
                d = json.loads(out.stdout.strip().splitlines()[-1])
                print(f"[{i}] bench: {d['value']:.1f} games/s, trunk launch {d['roofline']['avg_launch_us']:.2f} us, gnn_forward {d['gnn_forward']['boards_per_s'] / 1e6:.2f} M boards/s", flush=True)
            except Exception as ex:
                print(f"[{i}] bench failed: {ex}\n{out.stdout[-800:]}{out.stderr[-800:]}", flush=True)


if __name__ == "__main__":
    main()
