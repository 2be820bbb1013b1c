#!/usr/bin/env python3
"""Cold-launch determinism check for the GCN trunk ("LDS above 128 KB").

Each fresh process uploads weights, poisons the LDS, launches the trunk 40 times on the same 300 boards (300 > 256 CUs,
so some CUs hold a second resident workgroup) and compares every launch with the last one bit-for-bit and the first
one with the exact-f32 kernel (trunk_variant 0) on the same boards.  Run several fresh processes: the failure this guards against showed only on the first
launches of a process.  Usage: cold_launch_check.py [variant=3] [processes=8]
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(variant):
    import numpy as np
    import torch
    from alphaquoridorgnn_amd import _lib
    from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
    from tests import _util as U
    dev = _lib.require_gpu("cuda:0")
    lib = _lib.load()
    _lib.set_option("trunk_variant", variant)
    g = U.golden("walk_9x9.npz")
    B = 300
    recs = g["states"][np.linspace(0, g["states"].shape[0] - 1, B).astype(int)]
    torch.manual_seed(0)
    m = GNNNetwork().to("cuda").eval()
    pk = m.packed_weights(dev)
    st = torch.from_numpy(recs).to(dev)
    word = m.saturation_word(dev)
    rc = 0
    # both builds of the default trunk: flags 0 = range guard by bounds (any weight set), the module's own flags = the build for a
    # weight set whose fp16 range is proven (AQG_GNN_RANGE_PROVEN; what predict / the engine launch for these weights)
    for flags in (0, m.gnn_flags(dev)):
        _lib.set_option("trunk_variant", variant)
        _lib.poison_lds(dev)
        outs = []
        for _ in range(40):
            pooled = torch.full((B, 128), float("nan"), device=dev)
            _lib.check(lib.aqg_gcn_forward_boards_guarded(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(pooled), None, None, None, None,
                                                          flags, _lib.ptr(word), _lib.stream_ptr(dev)), "trunk")
            outs.append(pooled)
        torch.cuda.synchronize()
        bad = [i for i, o in enumerate(outs) if not torch.equal(o, outs[-1])]
        _lib.set_option("trunk_variant", 0)
        ref = torch.empty((B, 128), device=dev)
        _lib.check(lib.aqg_gcn_forward_boards(9, _lib.ptr(st), 0, B, _lib.ptr(pk), _lib.ptr(ref), None, None, None, None, 0, _lib.stream_ptr(dev)), "trunk")
        err = float((outs[0] - ref).abs().max())
        print(f"variant {variant} flags {flags}: launches differing from the last: {bad[:8]} ({len(bad)}/40); first launch vs exact-f32 kernel max|d pooled| = {err:.3e}; guard word {int(word.item())}")
        rc |= 1 if bad or not err < 1e-5 or int(word.item()) else 0
    return rc


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        sys.exit(child(int(sys.argv[2])))
    variant = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    fails = 0
    for _ in range(procs):                       # one GPU process at a time
        fails += subprocess.call([sys.executable, os.path.abspath(__file__), "--child", str(variant)]) != 0
    print(f"cold-launch check: {fails}/{procs} processes failed")
    sys.exit(1 if fails else 0)
