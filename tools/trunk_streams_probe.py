#!/usr/bin/env python3
"""Round-4 premise test for a role-fused / finer-grained launch structure: is the CU-time of a trunk board lower when the two
workgroups that share a CU come from DIFFERENT launches (naturally out of phase) than when they come from the same 480-board launch
(lock step)?  The MCTS loop pays four 480-board launches per round (1,920 boards).  Each case below pushes the same 1,920 boards per
round through the trunk alone (no heads, no step kernels), `rounds` times, on up to four streams, and prints the time per round.
  A  4 streams x 1 launch  of 480 boards   (today's structure)
  B  4 streams x 2 launches of 240 boards  (every CU gets at most one workgroup of a launch)
  C  4 streams x 3 launches of 160 boards
  D  4 streams x 4 launches of 120 boards
  E  1 stream  x 1 launch  of 1,920 boards (the persistent grid walks 3.75 boards per workgroup)
  F  2 streams x 1 launch  of 960 boards
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.pv_network_gnn import GraphPolicyValueNetwork
from tools.microbench import synth_states

dev = _lib.require_gpu("cuda:0")
lib = _lib.load()
torch.manual_seed(0)
model = GraphPolicyValueNetwork().to(dev)
pk = model.packed_weights(dev)
flags = model.gnn_flags(dev)
sat = model.saturation_word(dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
rounds = int(os.environ.get("ROUNDS", "300"))


def case(name, nstreams, per_stream_launches, boards):
    bufs = []
    for s in range(nstreams):
        for l in range(per_stream_launches):
            bd = synth_states(boards, seed=7 * s + l, dev=dev)
            bufs.append((bd, torch.empty((boards, 128), device=dev)))
    torch.cuda.synchronize()

    # one hipGraph per stream holding `rounds` rounds of that stream's launches: the host must not be the bottleneck
    def enqueue(s, n, sp):
        for _ in range(n):
            for l in range(per_stream_launches):
                bd, pooled = bufs[s * per_stream_launches + l]
                _lib.check(lib.aqg_gcn_forward_boards_guarded(9, _lib.ptr(bd), 0, boards, _lib.ptr(pk), _lib.ptr(pooled), None, None, None, None,
                                                              flags, _lib.ptr(sat), sp), "fwd")
    graphs = []
    for s in range(nstreams):
        enqueue(s, 2, streams[s].cuda_stream)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=streams[s], capture_error_mode="thread_local"):
            enqueue(s, rounds, streams[s].cuda_stream)
        graphs.append(g)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        for s in range(nstreams):
            with torch.cuda.stream(streams[s]):
                graphs[s].replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / rounds * 1e6)
    tot = nstreams * per_stream_launches * boards
    print(f"{name}: {nstreams} streams x {per_stream_launches} launches x {boards:4d} boards = {tot} boards/round: {best:7.1f} us/round "
          f"= {best * 256 / tot:5.2f} CU-us per board, {tot / best:6.1f} boards/us", flush=True)


case("A", 4, 1, 480)
case("B", 4, 2, 240)
case("C", 4, 3, 160)
case("D", 4, 4, 120)
case("E", 1, 1, 1920)
case("F", 2, 1, 960)
case("A", 4, 1, 480)
