"""CPU tests (no GPU): host-side logic of the drop-in modules, the C-ABI surface of libaqgnn_hip.so (symbols only --
no compute call is possible without a GPU), and the N>1 exchange step on gloo with world_size 2."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import _util as U

REPO = U.REPO


def test_capi_exports_every_declared_symbol():
    """Every function declared in include/aqgnn.h must be exported by the built library and bound by _lib.py."""
    from alphaquoridorgnn_amd import _lib
    hdr = open(os.path.join(REPO, "include", "aqgnn.h")).read()
    declared = set(re.findall(r"\b(aqg_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("aqg_engine")
    assert len(declared) >= 15
    lib = _lib.load()                     # must exist: built by __graft_entry__.build()
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in aqgnn.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} not bound in _lib.SIGNATURES"
    assert lib.aqg_abi_version() == _lib.ABI_VERSION
    assert lib.aqg_gcn_packed_floats(9) > 64082          # all 64,082 parameters + padding + fragment copies
    assert ctypes.sizeof(_lib.EngineStruct) == 11 * 4 + 2 * 4 + 4 + 34 * 8 + 8   # 13 scalars (+4 pad) + 34 pointers (ABI 10: + 7 of the evaluation cache) + eval_cache_log2 (+4 pad)
    assert ctypes.sizeof(_lib.TrainStruct) == 8 * 4 + 4 * 14 * 8 + 17 * 8      # aqg_train: 8 scalars, 4 x 14 + 17 pointers


def test_no_cpu_fallback_without_gpu():
    """Without a GPU the hot path must fail loudly, never compute on the host."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from alphaquoridorgnn_amd import _lib
    from alphaquoridorgnn_amd.game_logic import State
    from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
    with pytest.raises(_lib.HipLibraryError):
        State().legal_actions()
    with pytest.raises(_lib.HipLibraryError):
        GNNNetwork().predict(State(), "cpu")
    for mod in ("game_logic", "pv_network_gnn", "pv_mcts", "self_play", "engine", "_lib"):
        src = open(os.path.join(REPO, "alphaquoridorgnn_amd", mod + ".py")).read()
        assert "import oracle" not in src and "from oracle" not in src, f"{mod} must not touch the oracle"


def test_state_host_logic_matches_reference_vectors():
    """State bookkeeping (next / rotate / terminal flags / to_array) is host Python like the reference; check it
    against the reference-generated walk."""
    from alphaquoridorgnn_amd.game_logic import State, pack_state72
    g = U.golden("walk_9x9.npz")
    for i in range(0, 3000, 7):
        r = g["states"][i]
        s = State(player=[int(r[0]), int(r[1])], enemy=[int(r[2]), int(r[3])], walls=[int(x) for x in r[4:68]],
                  plies_played=int(r[68]) | (int(r[69]) << 8))
        assert np.array_equal(s.record(), r)
        assert ((1 if s.is_lose() else 0) | (2 if s.is_draw() else 0)) == g["status"][i]
        a = int(g["actions"][i])
        if a >= 0:
            n = s.next(a)
            assert np.array_equal(n.record(), g["next_states"][i])
            assert s.to_array() == [[int(r[0]), int(r[1])], [int(r[2]), int(r[3])], [int(x) for x in r[4:68]]]
    with pytest.raises(ValueError):
        State(board_size=6)
    s0 = State()
    assert s0.player == [76, 10] and s0.enemy == [76, 10] and s0.is_first_player() and not s0.is_done()


def test_module_surface_matches_reference_names():
    import alphaquoridorgnn_amd.pv_mcts as pm
    import alphaquoridorgnn_amd.self_play as sp
    import alphaquoridorgnn_amd.pv_network_gnn as pg
    import alphaquoridorgnn_amd.constants as c
    assert pm.PV_EVALUATE_COUNT == 50 and callable(pm.pv_mcts_policy) and callable(pm.pv_mcts_action)
    assert pm.boltzman([1, 3], 1.0) == [0.25, 0.75]
    assert sp.SP_GAME_COUNT == 50 and sp.SP_TEMPERATURE == 1.0
    assert (pg.NUM_FEATURES, pg.HIDDEN_DIM, pg.NUM_GCN_LAYERS, pg.POLICY_OUTPUT_SIZE) == (6, 128, 3, 209)
    assert (c.BOARD_SIZE, c.NUM_WALLS, c.NUM_PLIES_FOR_DRAW) == (9, 10, 116)
    m = pg.GraphPolicyValueNetwork(6, 128, 3, 209)
    keys = list(m.state_dict().keys())
    assert sorted(keys) == sorted(pg.STATE_DICT_KEYS)
    assert sum(p.numel() for p in m.parameters()) == 64082          # parameter count of the reference net
    assert m.state_dict()["gcn_layers.0.lin.weight"].shape == (128, 6)
    assert float(m.state_dict()["gcn_layers.1.bias"].abs().sum()) == 0.0


def test_weight_packing_layout():
    """aqg_gcn_pack_weights_host is host code: check the kernel layout (transposes, padding, fragment order)."""
    from alphaquoridorgnn_amd import _lib
    from oracle import gnn as og
    lib = _lib.load()
    p = og.init_params(5)
    host = [np.ascontiguousarray(p[k], dtype=np.float32) for k in og.KEYS]
    arr = (ctypes.c_void_p * 14)(*[h.ctypes.data_as(ctypes.c_void_p) for h in host])
    n = lib.aqg_gcn_packed_floats(9)
    out = np.zeros(n, dtype=np.float32)
    assert lib.aqg_gcn_pack_weights_host(9, arr, out.ctypes.data_as(ctypes.c_void_p)) == 0
    W1 = out[:128 * 8].reshape(128, 8)
    assert np.array_equal(W1[:, :6], p["gcn_layers.0.lin.weight"]) and not W1[:, 6:].any()
    off = 128 * 8 + 128
    W2T = out[off:off + 128 * 128].reshape(128, 128)
    assert np.array_equal(W2T, p["gcn_layers.1.lin.weight"].T)
    WF2 = 67396                                                            # offsets documented in include/aqgnn.h / gcn_forward.hip
    WH2 = WF2 + 2 * 128 * 128
    WH1 = WH2 + 2 * 2 * 128 * 128 // 2
    WHH1 = WH1 + 4 * 2 * 64 * 4
    WHP2 = WHH1 + 2 * 8 * 4 * 64 * 4
    TB = WHP2 + 2 * 14 * 2 * 64 * 4
    GUARD = TB + 3 * 5 * 128                                               # the range guard's thresholds: the last four floats
    assert n == GUARD + 4
    tb2 = 15.0 / 16.0 * np.sqrt(5.0) * np.abs(p["gcn_layers.1.bias"].astype(np.float64)).max()
    assert abs(out[GUARD] - (65504.0 - tb2) / 2.07) <= 1e-2 and out[GUARD + 1] == 65504.0 and not out[GUARD + 2:].any()
    # a weight whose fp16 hi half is not finite (|W| / CQ >= 65504, inf, NaN) would give NaN columns that the float maxima of the
    # tracking build skip: such a set is packed with NEGATIVE thresholds -- every board is reported
    for key, idx, val in (("gcn_layers.1.lin.weight", (3, 7), 7e4), ("gcn_layers.2.lin.weight", (100, 1), -7e4), ("gcn_layers.1.lin.weight", (0, 0), np.nan),
                          ("gcn_layers.0.lin.weight", (5, 2), np.inf), ("gcn_layers.1.lin.weight", (3, 7), 6.0e4)):
        pb = {k: v.copy() for k, v in p.items()}
        pb[key][idx] = val
        hb = [np.ascontiguousarray(pb[k], dtype=np.float32) for k in og.KEYS]
        ab = (ctypes.c_void_p * 14)(*[h.ctypes.data_as(ctypes.c_void_p) for h in hb])
        ob = np.zeros(n, dtype=np.float32)
        assert lib.aqg_gcn_pack_weights_host(9, ab, ob.ctypes.data_as(ctypes.c_void_p)) == 0
        if val == 6.0e4:           # 6.0e4 / CQ = 64,000: still a finite fp16
            assert ob[GUARD] > 0 and ob[GUARD + 1] == 65504.0
        else:
            assert ob[GUARD] == -1.0 and ob[GUARD + 1] == -1.0, (key, val)
    CQ = 15.0 / 16.0                                                       # scale of the default trunk's activation image
    wf2 = out[WF2:WF2 + 128 * 128].reshape(4, 2, 8, 64, 4)                 # [wave][ntile][s4][lane][i]
    for (w, j, s4, lane, i) in [(0, 0, 0, 0, 0), (3, 1, 7, 63, 3), (2, 0, 5, 17, 2), (1, 1, 2, 40, 1)]:
        c, q = lane & 15, lane >> 4
        k = (q & 1) * 64 + (q >> 1) * 32 + 4 * s4 + i
        assert wf2[w, j, s4, lane, i] == p["gcn_layers.1.lin.weight"][32 * w + 16 * j + c, k]
    # fp16 2-way split fragments (default trunk) of W / CQ: hi = RNE_f16(w), lo = RNE_f16(w - hi)
    for L, key in ((0, "gcn_layers.1.lin.weight"), (1, "gcn_layers.2.lin.weight")):
        wh = out[WH2 + L * 128 * 128:WH2 + (L + 1) * 128 * 128].view(np.uint16).reshape(2, 4, 2, 4, 64, 8)   # [plane][wave][ntile][kb][lane][8 halves]
        W = (p[key].astype(np.float64) / CQ).astype(np.float32)
        hi = W.astype(np.float16)
        lo = (W - hi.astype(np.float32)).astype(np.float16)
        wv, j, kb, lane, e = np.meshgrid(np.arange(4), np.arange(2), np.arange(4), np.arange(64), np.arange(8), indexing="ij")
        nn, kk = 32 * wv + 16 * j + (lane & 15), 32 * kb + 8 * (lane >> 4) + e
        assert np.array_equal(wh[0], hi[nn, kk].view(np.uint16)) and np.array_equal(wh[1], lo[nn, kk].view(np.uint16))
        rec = hi.astype(np.float64) + lo.astype(np.float64)
        assert np.max(np.abs(rec - W)) <= 2.0 ** -22 * np.max(np.abs(W))
    # layer-1 fragments (aggregate-first layer 1: A operand, rows = output features): k-slots 0..7 and 8..15 = hi(c W1[n][0..5]),0,0
    # 16..23 = lo(c W1[n][0..5]),0,0   24..31 = 0
    w1 = out[WH1:WH1 + 4 * 2 * 64 * 4].view(np.uint16).reshape(4, 2, 4, 16, 8)                        # [wave][ntile][q][c][8 halves]
    W = (p["gcn_layers.0.lin.weight"].astype(np.float64) * CQ).astype(np.float32)                     # [128, 6]
    hi = W.astype(np.float16)
    lo = (W - hi.astype(np.float32)).astype(np.float16)
    cols = (32 * np.arange(4)[:, None, None] + 16 * np.arange(2)[None, :, None] + np.arange(16)[None, None, :])   # [wave][ntile][c]
    assert np.array_equal(w1[:, :, 0, :, :6], hi[cols].view(np.uint16)) and np.array_equal(w1[:, :, 1, :, :6], hi[cols].view(np.uint16))
    assert np.array_equal(w1[:, :, 2, :, :6], lo[cols].view(np.uint16))
    assert not w1[:, :, :3, :, 6:].any() and not w1[:, :, 3].any()
    # bias tables of the default trunk: TB[layer][deg - 1][f] = CQ * b[f] * sqrt(deg)   (random-init biases are zero: use a second set)
    p2 = {k: v.copy() for k, v in p.items()}
    rng = np.random.RandomState(3)
    for L in range(3):
        p2[f"gcn_layers.{L}.bias"] = rng.randn(128).astype(np.float32)
    host2 = [np.ascontiguousarray(p2[k], dtype=np.float32) for k in og.KEYS]
    arr2 = (ctypes.c_void_p * 14)(*[h.ctypes.data_as(ctypes.c_void_p) for h in host2])
    out2 = np.zeros(n, dtype=np.float32)
    assert lib.aqg_gcn_pack_weights_host(9, arr2, out2.ctypes.data_as(ctypes.c_void_p)) == 0
    tb = out2[TB:TB + 3 * 5 * 128].reshape(3, 5, 128)
    for L in range(3):
        for deg in range(1, 6):
            want = (CQ * p2[f"gcn_layers.{L}.bias"].astype(np.float64) * np.sqrt(float(deg))).astype(np.float32)
            assert np.array_equal(tb[L, deg - 1], want)
    # heads: hidden layer A fragments [plane][unit tile][kb][q][c][8 halves], policy B fragments in accumulator k order
    hw = out[WHH1:WHP2].view(np.uint16).reshape(2, 8, 4, 4, 16, 8)
    Wh = np.concatenate([p["policy_head.0.weight"], p["value_head.0.weight"]], 0).astype(np.float32)      # [128 units, 128]
    hi = Wh.astype(np.float16); lo = (Wh - hi.astype(np.float32)).astype(np.float16)
    ut, kb, q, c, e = np.meshgrid(np.arange(8), np.arange(4), np.arange(4), np.arange(16), np.arange(8), indexing="ij")
    assert np.array_equal(hw[0], hi[16 * ut + c, 32 * kb + 8 * q + e].view(np.uint16))
    assert np.array_equal(hw[1], lo[16 * ut + c, 32 * kb + 8 * q + e].view(np.uint16))
    pw = out[WHP2:TB].view(np.uint16).reshape(2, 14, 2, 4, 16, 8)
    Wp = np.zeros((224, 64), np.float32); Wp[:209] = p["policy_head.2.weight"]
    hi = Wp.astype(np.float16); lo = (Wp - hi.astype(np.float32)).astype(np.float16)
    at, kb, q, c, e = np.meshgrid(np.arange(14), np.arange(2), np.arange(4), np.arange(16), np.arange(8), indexing="ij")
    unit = 32 * kb + 16 * (e >> 2) + 4 * q + (e & 3)
    assert np.array_equal(pw[0], hi[16 * at + c, unit].view(np.uint16)) and np.array_equal(pw[1], lo[16 * at + c, unit].view(np.uint16))


_GLOO_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["AQG_REPO"])
import torch, torch.distributed as dist
from alphaquoridorgnn_amd.engine import gather_history
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["AQG_PORT"], rank=int(os.environ["RANK"]), world_size=2)
r = dist.get_rank()
n = 5 if r == 0 else 3                      # ragged: ranks hold different numbers of positions
st = torch.full((n, 72), 10 + r, dtype=torch.uint8)
vis = (torch.arange(n * 209).view(n, 209) % 199 + 1000 * r).to(torch.int16)
z = torch.tensor([(-1) ** i for i in range(n)], dtype=torch.int8) * (1 if r == 0 else -1)
s, v, zz = gather_history(st, vis, z)
assert s.shape == (8, 72) and v.shape == (8, 209) and zz.shape == (8,)
assert (s[:5] == 10).all() and (s[5:] == 11).all()
assert torch.equal(v[:5], (torch.arange(5 * 209).view(5, 209) % 199).to(torch.int16))
assert torch.equal(v[5:], (torch.arange(3 * 209).view(3, 209) % 199 + 1000).to(torch.int16))
assert zz.tolist() == [1, -1, 1, -1, 1, -1, 1, -1]
# empty shard on one rank
s, v, zz = gather_history(st[:0] if r == 1 else st, vis[:0] if r == 1 else vis, z[:0] if r == 1 else z)
assert s.shape[0] == 5
dist.destroy_process_group()
print("ok", r)
'''


def test_gather_history_world_size_2_gloo(tmp_path):
    """the one exchange step per generation -- ragged all-gather of (s, pi, z) -- on 2 gloo ranks."""
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    port = str(29500 + os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), AQG_REPO=REPO, AQG_PORT=port, MASTER_ADDR="127.0.0.1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "ok" in o


_ENTRY_WORKER = r'''
import os, sys, time
import torch, torch.distributed as dist
from alphaquoridorgnn_amd import distributed as aqd
rank, world = aqd.init_from_env()                 # torchrun's environment; backend from AQG_DIST_BACKEND (gloo here)
assert (rank, world) == (int(os.environ["RANK"]), 2) and dist.get_backend() == "gloo"
assert aqd.init_from_env() == (rank, world)       # idempotent
# a long single-rank stage: rank 1 must wait on the store (no collective pending), and only leave after rank 0 published
tag = aqd.next_tag("stage")
assert tag == "stage/1"
t0 = time.time()
if rank == 0:
    time.sleep(1.5)
    aqd.release_ranks(tag)
else:
    aqd.wait_for_rank0(tag)
    assert time.time() - t0 > 1.0
dist.barrier()
t = torch.tensor([rank + 5], device=aqd.collective_device())
dist.broadcast(t, src=0)
assert int(t) == 5 and t.device.type == "cpu"
aqd.shutdown()
assert not dist.is_initialized()
open(f"entry.{rank}.ok", "w").write("ok")         # (the ranks' stdout lines interleave under torchrun)
'''


def test_multi_rank_entry_under_torchrun_gloo(tmp_path):
    """The package's multi-rank entry (alphaquoridorgnn_amd/distributed.py, what `-m alphaquoridorgnn_amd.train_cycle` calls first)
    started the way INTEGRATION.md starts it -- torch.distributed.run, one process per rank, rendezvous on 127.0.0.1 -- with the
    gloo backend on CPU: ranks / world size from the environment, the store-based wait that keeps idle ranks out of a pending
    collective while rank 0 works alone, clean shutdown."""
    script = tmp_path / "entry_worker.py"
    script.write_text(_ENTRY_WORKER)
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""), AQG_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + (os.getpid() + 7) % 2000), str(script)]
    res = subprocess.run(cmd, env=env, cwd=str(tmp_path), timeout=300, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert (tmp_path / "entry.0.ok").exists() and (tmp_path / "entry.1.ok").exists()


_FAIL_WORKER = r"""
import os, sys, time
import torch.distributed as dist
from alphaquoridorgnn_amd import distributed as aqd
rank, world = aqd.init_from_env()
t0 = time.time()
try:
    tag = aqd.next_tag("train")
    if rank == 0:
        with aqd.single_rank_stage(tag):
            time.sleep(0.5)
            raise ValueError("rank 0 failed inside its solitary stage")
    else:
        aqd.wait_for_rank0(tag)
    dist.barrier()
except BaseException as e:
    open(f"fail.{rank}.txt", "w").write(f"{type(e).__name__} {time.time() - t0:.2f}")
    aqd.shutdown(ok=False)
    raise
aqd.shutdown()
"""


def test_multi_rank_entry_failure_in_single_rank_stage(tmp_path):
    """rank 0 raising inside a single-rank stage (training / evaluation) must end the job in seconds with ITS exception:
    the stage key is published with value b'fail' on the way out, the idle rank wakes up in wait_for_rank0 and raises
    Rank0StageFailed, and neither rank enters a barrier on the error path (shutdown(ok=False)) -- before, rank 0 sat in
    shutdown()'s barrier until AQG_DIST_TIMEOUT_S and the others in store.wait for hours."""
    import time
    script = tmp_path / "fail_worker.py"
    script.write_text(_FAIL_WORKER)
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""), AQG_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + (os.getpid() + 13) % 2000), str(script)]
    t0 = time.time()
    res = subprocess.run(cmd, env=env, cwd=str(tmp_path), timeout=300, capture_output=True, text=True)
    assert res.returncode != 0
    assert time.time() - t0 < 120, "the job must not wait for a collective timeout"
    f0, f1 = (tmp_path / "fail.0.txt").read_text().split(), (tmp_path / "fail.1.txt").read_text().split()
    assert f0[0] == "ValueError" and f1[0] == "Rank0StageFailed"
    assert float(f0[1]) < 30 and float(f1[1]) < 30
    assert "rank 0 failed inside its solitary stage" in res.stderr + res.stdout


def test_self_play_sharding_arithmetic():
    """Games split over ranks: every game is played exactly once (self_play.py:81-84 loop, sharded)."""
    for total in (50, 2048, 7):
        for world in (1, 2, 3, 8):
            mine = [total // world + (1 if r < total % world else 0) for r in range(world)]
            assert sum(mine) == total and max(mine) - min(mine) <= 1


def test_history_file_format_interoperates_with_reference(tmp_path, monkeypatch):
    """SURVEY 8 f2.  tests/golden/history_9x9.json is what the REFERENCE's self_play.write_data() pickled for one seeded game
    of its own play() and what its load_data() read back (tools/gen_golden_history.py).  Our writer side (_history_rows: the
    rows self_play() hands to write_data) must produce that list field for field from the engine's tensors, and our reader
    side (train_network.load_data + the unzip of train_network.py:37-46) must turn a file with the reference's content into
    the same training arrays."""
    import json
    import pickle
    from alphaquoridorgnn_amd import self_play as sp, train_network as tn
    from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
    with open(os.path.join(U.GOLDEN, "history_9x9.json")) as f:
        doc = json.load(f)
    hist = doc["history"]
    g = U.golden("games_9x9.npz")
    st, pol, z = g["g2_states"], g["g2_policy"], g["g2_z"]           # the same reference game (seed 77, 24 sims, bias 60)
    assert len(hist) == st.shape[0] == 27
    for i, (s, p, v) in enumerate(hist):
        assert s == [[int(st[i, 0]), int(st[i, 1])], [int(st[i, 2]), int(st[i, 3])], [int(x) for x in st[i, 4:68]]]
        assert p == pol[i].tolist() and v == int(z[i])
        assert len(p) == 209 and abs(sum(p) - 1.0) < 1e-12
    # writer: engine tensors (state72 rows, dense root visit counts, z) -> rows; visit counts = policy * (sims - 1)
    visits = np.rint(pol * 23).astype(np.int16)
    assert np.array_equal(visits.sum(1), np.full(27, 23))
    rows = sp._history_rows(torch.from_numpy(st), torch.from_numpy(visits), torch.from_numpy(z), 9)
    assert rows == hist                                               # ints, float64 quotients and z all equal
    assert all(type(x) is int for r in rows for part in r[0] for x in part) and all(type(r[2]) is int for r in rows)
    # reader: a file with the reference's content (numpy.int64 pawn positions included, as in its pickles)
    ref_like = [[[[np.int64(s[0][0]), s[0][1]], [np.int64(s[1][0]), s[1][1]], list(s[2])], list(p), v] for s, p, v in hist]
    monkeypatch.chdir(tmp_path)
    os.makedirs("data")
    with open(os.path.join("data", doc["file_name_pattern"]), "wb") as f:
        pickle.dump(ref_like, f)
    with open(os.path.join("data", "20000101000000.history"), "wb") as f:      # an older generation: must be ignored (:21)
        pickle.dump(ref_like[:3], f)
    back = tn.load_data()
    assert len(back) == 27
    s, p, v = zip(*back)                                                        # train_network.py:37
    recs = GNNNetwork().preprocess_input(s)
    want = st.copy(); want[:, 68:70] = 0                                        # plies are not part of to_array()
    assert np.array_equal(recs, want)
    assert list(np.array(p).shape) == doc["train_p_shape"] and np.array_equal(np.array(p), pol)
    assert np.array(v).tolist() == doc["train_v"]
    # the six feature planes the reference's featuriser builds from those states sum to the recorded checksum
    from oracle import gnn as og
    assert abs(sum(float(og.node_features(r).sum()) for r in want) - doc["train_planes_sum"]) < 1e-9


@pytest.mark.parametrize("N", [3, 5, 9])
def test_baseline_agents_match_reference(N):
    """random / alpha-beta / rollout-MCTS opponents (agents.py:14-214).  tests/golden/agents_*.npz hold what the
    REAL reference agents answered on positions from random play (tools/gen_golden_agents.py; Python's `random` seeded per
    call): heuristic_eval as a float, alpha_beta_action at depth 1 and 2, random_action and mcts_action under the recorded
    seeds.  Ours are host code over the rule header the kernels compile -- every answer must be identical."""
    import random
    from alphaquoridorgnn_amd import agents
    from alphaquoridorgnn_amd.game_logic import State
    g = U.golden(f"agents_{N}x{N}.npz")
    assert int(g["board"][0]) == N

    def mk(rec):
        return State(board_size=N, player=[int(rec[0]), int(rec[1])], enemy=[int(rec[2]), int(rec[3])],
                     walls=[int(x) for x in rec[4:4 + (N - 1) ** 2]], plies_played=int(rec[68]) | (int(rec[69]) << 8))
    assert agents._max_dist(mk(g["states"][0])) == int(g["max_dist"][0])
    for i, rec in enumerate(g["states"]):
        s = mk(rec)
        assert agents.heuristic_eval(s) == g["heuristic"][i], i
        assert agents.alpha_beta_action(s, 2) == int(g["ab2"][i]) and agents.alpha_beta_action(s, 1) == int(g["ab1"][i]), i
        random.seed(int(g["random_seed"][i]))
        assert agents.random_action(s) == int(g["random"][i]), i
        # the host rules agree with the oracle's on the way (legal list, shortest path >= 0 unless walled in)
        from oracle import quoridor as oq
        assert agents._legal(s) == [int(a) for a in oq.State(rec).legal_actions()]
    for j, i in enumerate(g["mcts_index"]):
        random.seed(int(g["mcts_seed"][j]))
        assert agents.mcts_action(mk(g["states"][i])) == int(g["mcts_action"][j]), i


def test_baseline_agents_play_a_game():
    """evaluate_agents.py-style use: alpha-beta (depth 1) against the random agent on 5x5 until the game ends."""
    import random
    from alphaquoridorgnn_amd import agents
    from alphaquoridorgnn_amd.game_logic import State
    random.seed(3)
    s = State(board_size=5, num_walls=2)
    plies = 0
    while not s.is_done():
        a = agents.alpha_beta_action(s, 1) if s.is_first_player() else agents.random_action(s)
        assert a in agents._legal(s)
        s = s.next(a)
        plies += 1
    assert 4 <= plies <= 28


def test_bench_cpu_baseline_workers():
    """bench.py's cpu_baseline leg: N single-threaded oracle processes playing the reference's sequential self-play loop for a
    bounded time; the rates add up and the record says how many cores were used (no GPU involved)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    r = bench.cpu_baseline(8, 100.0, budget_s=2.0, workers=2)
    assert r["cores"] == 2 and r["kind"] == "port" and r["unit"] == "games/s"
    assert r["value"] > 0 and abs(r["sims_per_s"] - r["value"] * 100.0 * 8) <= 1e-6 * r["sims_per_s"]


def test_static_fp16_range_bound_of_a_weight_set():
    """pv_network_gnn._range_proven (what lets the split trunk drop its per-value range tracking, include/aqgnn.h
    AQG_GNN_RANGE_PROVEN): a rigorous bound from the weights alone.  It must hold for initialisation-scale weights, fail once the
    trunk weights are x3 or a bias is huge or anything is not finite, and -- being a bound -- dominate the activations the fp64
    oracle actually produces on real positions (times the kernel's internal scale factor)."""
    import torch
    from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork, STATE_DICT_KEYS
    from oracle import gnn as og
    params = og.init_params(4)
    m = GNNNetwork()

    def host(p):
        return [torch.from_numpy(np.asarray(p[k], dtype=np.float32).copy()) for k in STATE_DICT_KEYS]

    assert m._range_proven(host(params))
    big = {k: (v * (3.0 if "gcn" in k and "weight" in k else 1.0)) for k, v in params.items()}
    assert not m._range_proven(host(big))
    hb = {k: v.copy() for k, v in params.items()}
    hb["gcn_layers.1.bias"][5] = 3.0e4
    assert not m._range_proven(host(hb))
    nan = {k: v.copy() for k, v in params.items()}
    nan["gcn_layers.2.lin.weight"][0, 0] = np.nan
    assert not m._range_proven(host(nan))
    # the bound dominates reality: the largest layer activation of the oracle on walk positions, with 10 walls in hand
    recs = U.golden("walk_9x9.npz")["states"][::700]
    p64 = {k: np.asarray(v, dtype=np.float64) for k, v in params.items()}
    W = [np.abs(p64[f"gcn_layers.{l}.lin.weight"]) for l in range(3)]
    b = [np.abs(p64[f"gcn_layers.{l}.bias"]) for l in range(3)]
    h = np.array([1.0, 16.0, 1.0, 16.0, 1.0, 1.0])
    bounds = []
    for l in range(3):
        h = (0.2 + 4.0 / np.sqrt(10.0)) * (W[l] @ h) + b[l]
        bounds.append(h.max())
    for rec in recs:
        x, e = og.node_features(rec).astype(np.float64), og.board_edges(rec)
        for l in range(3):
            x = np.maximum(og.gcn_conv(x, e, p64[f"gcn_layers.{l}.lin.weight"], p64[f"gcn_layers.{l}.bias"]), 0.0)
            assert x.max() <= bounds[l]


def test_set_option_names_ranges_and_errors():
    """aqg_set_option is host code: every documented knob is accepted, out-of-range values and unknown names are refused with a
    message in aqg_last_error (no GPU call involved)."""
    from alphaquoridorgnn_amd import _lib
    lib = _lib.load()
    defaults = {"trunk_variant": 3, "heads_prio": 3, "trunk_prio": -1, "trunk_grid": 0, "trunk_phase_delay": 100, "trunk_delay_min_boards": 2048,
                "step_prio": 1, "step_waves": 8, "step_variant": 1, "step_fast_depth": None, "train_fused": 2, "use_graph": 1,
                "profile_trunk": 0}
    header = open(os.path.join(REPO, "include", "aqgnn.h")).read()
    integration = open(os.path.join(REPO, "INTEGRATION.md")).read()
    for name, value in defaults.items():
        assert f'"{name}"' in header, f"{name} is not documented in include/aqgnn.h"
        assert f"`{name}`" in integration, f"{name} is not listed in INTEGRATION.md"
        if value is not None:
            assert lib.aqg_set_option(name.encode(), value) == 0, name
    for name, bad in (("trunk_variant", 2), ("trunk_variant", 4), ("trunk_variant", 7), ("trunk_variant", 8), ("trunk_phase_delay", -1), ("step_fast_depth", 62), ("train_fused", 4)):
        assert lib.aqg_set_option(name.encode(), bad) != 0, (name, bad)
        assert lib.aqg_last_error()
    assert lib.aqg_set_option(b"no_such_option", 1) != 0 and b"no_such_option" in lib.aqg_last_error()
    with pytest.raises(RuntimeError):
        _lib.set_option("no_such_option", 1)
