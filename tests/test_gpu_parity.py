"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(alphaquoridorgnn_amd/_lib.py -> libaqgnn_hip.so), against the oracle and the committed golden vectors.

Bars: bit-exact for legal masks / ordered lists / transitions / MCTS visit counts / self-play histories;
fp32 GNN within atol 1e-5 + rtol 1e-4 of the fp64 oracle on pre-softmax logits and pre-tanh value
(PARITY UNPINNED against PyG itself -- see oracle/gnn.py)."""
import os
import sys

import numpy as np
import pytest
import torch

from tests import _util as U

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

DRAW = {9: 116, 5: 28, 3: 14}


@pytest.fixture(scope="module")
def dev():
    from alphaquoridorgnn_amd import _lib
    _lib.load()          # raises if the HIP library is missing: no fallback
    return _lib.require_gpu()


def _model(seed=0):
    from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
    from oracle import gnn as og
    params = og.init_params(seed)
    m = GNNNetwork()
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    return m.to("cuda").eval(), params


# ------------------------------------------------------------------ K3 legal actions / S7 next / S1 status
@pytest.mark.parametrize("N", [9, 5, 3])
def test_legal_actions_golden(dev, N):
    from alphaquoridorgnn_amd import game_logic as gl
    g = U.golden(f"walk_{N}x{N}.npz")
    recs = torch.from_numpy(g["states"]).to(dev)
    mask, order, count = gl.legal_actions_batch(recs, N)
    count = count.cpu().numpy()
    order = order.cpu().numpy().astype(np.int16)
    order[order == 255] = -1
    assert np.array_equal(count, g["counts"])
    assert np.array_equal(order, g["legal"])
    mask = mask.cpu().numpy()
    assert np.array_equal(mask.sum(1), g["counts"])
    for b in range(0, len(count), 97):
        exp = np.zeros(mask.shape[1], dtype=np.uint8)
        exp[g["legal"][b, :count[b]]] = 1
        assert np.array_equal(mask[b], exp)
    ok = g["actions"] >= 0
    nxt = gl.next_batch(recs[torch.from_numpy(ok).to(dev)], torch.from_numpy(g["actions"][ok]), N).cpu().numpy()
    assert np.array_equal(nxt, g["next_states"][ok])
    st = gl.status_batch(recs, N, DRAW[N]).cpu().numpy()
    assert np.array_equal(st, g["status"])


def test_state_api_known_answers(dev):
    from alphaquoridorgnn_amd.game_logic import State
    k = U.golden("kat_9x9.npz")
    s = State()
    for i, o in [(24, 1), (27, 1), (32, 2), (36, 2), (37, 1), (41, 1), (42, 2), (43, 1)]:
        s.walls[i] = o
    s.player[0] = 40
    s.enemy[0] = 32
    assert s.legal_actions_wall(pos=26) == []            # the reference's own scenario (test_legal_walls.py:21)
    assert s.legal_actions_pos(40) == [49, 39]
    assert s.legal_actions() == [int(x) for x in k["tlw_legal"]]
    s0 = State()
    la = s0.legal_actions()
    assert len(la) == 131 and la[:7] == [67, 75, 77, 81, 145, 82, 146]
    n = s0.next(67)
    assert n.player == [76, 10] and n.enemy == [67, 10] and n.plies_played == 1
    w = State(); w.walls[0] = 1
    la = w.legal_actions()
    assert 81 not in la and 82 not in la and 145 not in la and 146 in la and 153 in la
    nw = State(); nw.player[1] = 0
    assert nw.legal_actions() == [67, 75, 77]
    with pytest.raises(ValueError):
        State(board_size=4)
    g = U.golden("obstacle_9x9.npz")                     # pawn-as-obstacle counter-examples
    for rec, legal, absent in zip(g["states"], g["legal"], g["absent"]):
        st = State(player=[int(rec[0]), int(rec[1])], enemy=[int(rec[2]), int(rec[3])], walls=[int(x) for x in rec[4:68]],
                   plies_played=int(rec[68]))
        la = st.legal_actions()
        assert la == [int(x) for x in legal if x >= 0] and int(absent) not in la


def test_legal_actions_large_batch_properties(dev):
    """Full-size batch (65,536 states): size-independent invariants + oracle spot check; empty batch edge case."""
    from alphaquoridorgnn_amd import game_logic as gl
    from oracle import quoridor as oq
    g = U.golden("walk_9x9.npz")
    rng = np.random.RandomState(0)
    idx = rng.randint(0, g["states"].shape[0], size=65536)
    recs = torch.from_numpy(g["states"][idx]).to(dev)
    mask, order, count = gl.legal_actions_batch(recs, 9)
    m, o, c = mask.cpu().numpy(), order.cpu().numpy(), count.cpu().numpy()
    assert np.array_equal(m.sum(1), c)
    assert np.array_equal(c, g["counts"][idx])
    rows = np.arange(len(c))
    for j in range(int(c.max())):                         # every listed action is set in the mask
        sel = c > j
        assert m[rows[sel], o[sel, j]].all()
    sub = rng.randint(0, 65536, size=512)
    a, cc, mm = oq.legal_actions_batch(g["states"][idx][sub])
    assert np.array_equal(mm, m[sub])
    e_mask, e_order, e_count = gl.legal_actions_batch(recs[:0], 9)
    assert e_count.numel() == 0


def test_rules_and_mcts_on_7x7_vs_oracle(dev):
    """A board size with no reference fixtures (the reference defines 3x3 / 5x5 / 9x9 constants only): GPU legal lists,
    transitions, terminal flags and lock-step MCTS visit counts against the oracle on random wall-heavy play."""
    from alphaquoridorgnn_amd import game_logic as gl
    from alphaquoridorgnn_amd.pv_mcts import pv_mcts_policy_batch
    from oracle import mcts as om, quoridor as oq
    N = 7
    rng = np.random.RandomState(7)
    recs = []
    for game in range(30):
        s = oq.State(N=N)
        for ply in range(50):
            if s.is_done():
                break
            recs.append(s.rec.copy())
            la = s.legal_actions()
            walls = [a for a in la if a >= N * N]
            pick = walls if (walls and rng.rand() < 0.6) else la
            s = s.next(pick[rng.randint(len(pick))])
    recs = np.stack(recs)
    a, c, m = oq.legal_actions_batch(recs)
    d = torch.from_numpy(recs).to(dev)
    mask, order, count = gl.legal_actions_batch(d, N)
    A = N * N + 2 * (N - 1) ** 2
    assert np.array_equal(count.cpu().numpy(), c) and np.array_equal(mask.cpu().numpy(), m[:, :A])
    o = order.cpu().numpy()
    for i in range(len(recs)):
        assert np.array_equal(o[i, :c[i]], a[i, :c[i]])
    first = np.asarray([a[i, rng.randint(c[i])] for i in range(len(recs))], dtype=np.int32)
    assert np.array_equal(gl.next_batch(d, torch.from_numpy(first).to(dev), N).cpu().numpy(), oq.next_batch(recs, first))
    roots = recs[::37][:12]
    pols = pv_mcts_policy_batch(None, roots, 1.0, sims=30, board_size=N, evaluator="fake", fake_bias=11)
    for b in range(len(roots)):
        ref = om.pv_mcts_policy(om.FakeModel(11), oq.State(roots[b]), 1.0, 30)
        assert np.array_equal(np.asarray(pols[b]), np.asarray(ref))


# ------------------------------------------------------------------ K1/K2 GNN forward
@pytest.mark.parametrize("variant", [0, 1, 3, 6])
def test_gnn_forward_boards_vs_fp64_oracle(dev, variant):
    from alphaquoridorgnn_amd import _lib
    from oracle import gnn as og
    _lib.set_option("trunk_variant", variant)
    model, params = _model(0)
    assert not (model.gnn_flags(dev) & _lib.GNN_EXACT_F32)     # the range guard must not have swapped the kernels under test for the exact ones
    g = U.golden("walk_9x9.npz")
    sel = np.linspace(0, g["states"].shape[0] - 1, 300).astype(int)
    recs = g["states"][sel]
    ref = og.forward_states(params, recs)
    _lib.poison_lds(dev)                 # NaN-fill LDS: any read-before-write inside the kernels becomes visible
    policy, value, logits, vpre = model.forward_states(torch.from_numpy(recs).to(dev), want_logits=True)
    logits, vpre = logits.cpu().numpy().astype(np.float64), vpre.cpu().numpy().astype(np.float64)
    assert np.isfinite(logits).all() and np.isfinite(vpre).all()
    # stated tolerance (fp32 MFMA path): atol 1e-5, rtol 1e-4 on logits / pre-tanh value
    np.testing.assert_allclose(logits, ref["logits"], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(vpre, ref["value_pre"], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(policy.cpu().numpy(), ref["policy"], atol=1e-6, rtol=1e-4)
    np.testing.assert_allclose(value[:, 0].cpu().numpy(), ref["value"], atol=1e-5, rtol=1e-4)
    assert np.allclose(policy.sum(1).cpu().numpy(), 1.0, atol=1e-5)
    # ragged / tiny / odd batch sizes give the same rows
    for B in (1, 3, 17):
        _lib.poison_lds(dev)
        p2, v2 = model.forward_states(torch.from_numpy(recs[:B]).to(dev))
        assert torch.equal(p2, policy[:B]) and torch.equal(v2, value[:B])
    _lib.set_option("trunk_variant", 3)


@pytest.mark.parametrize("variant", [1, 6])
def test_gnn_forward_scaled_weights(dev, variant):
    """Weights scaled up so activations are O(10): relative tolerance still holds (catches layout slips that
    small random weights could hide)."""
    from alphaquoridorgnn_amd import _lib
    from oracle import gnn as og
    _lib.set_option("trunk_variant", variant)
    model, params = _model(3)
    big = {k: (v * (3.0 if "gcn" in k and "weight" in k else 1.0)).astype(np.float32) for k, v in params.items()}
    big["gcn_layers.1.bias"] = np.linspace(-0.5, 0.5, 128).astype(np.float32)
    big["gcn_layers.2.bias"] = np.linspace(0.3, -0.3, 128).astype(np.float32)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in big.items()})
    assert not (model.gnn_flags(dev) & _lib.GNN_EXACT_F32)
    g = U.golden("walk_9x9.npz")
    recs = g["states"][5000:5064]
    ref = og.forward_states(big, recs)
    _, _, logits, vpre = model.forward_states(torch.from_numpy(recs).to(dev), want_logits=True)
    np.testing.assert_allclose(logits.cpu().numpy(), ref["logits"], atol=2e-4, rtol=2e-4)
    np.testing.assert_allclose(vpre.cpu().numpy(), ref["value_pre"], atol=2e-4, rtol=2e-4)
    _lib.set_option("trunk_variant", 3)


_WALK_ORACLE = {}


def _walk_oracle(seed):
    """fp64 oracle outputs for ALL 14,000 reference-walk states (dense form, ~12 s once per weight seed), sliced by the tests."""
    from oracle import gnn as og
    if seed not in _WALK_ORACLE:
        params = og.init_params(seed)
        for l in range(3):      # non-zero GCN biases: the folded bias table (TB) and its per-degree rows matter at every node
            params[f"gcn_layers.{l}.bias"] = (np.linspace(-0.3, 0.5, 128) * (1 + l)).astype(np.float32)
        _WALK_ORACLE[seed] = (params, og.forward_states_dense(params, U.golden("walk_9x9.npz")["states"]))
    return _WALK_ORACLE[seed]


@pytest.mark.parametrize("variant,B", [(6, 4096), (6, 1000), (6, 1001), (6, 2049), (6, 513), (3, 8192), (1, 2049)])
def test_gnn_forward_many_boards_per_workgroup(dev, variant, B):
    """BASELINE configs[1] at its own size, and ragged sizes around the launch-size switches: with more than 512 boards a
    workgroup of the persistent trunk walks SEVERAL boards (next-record prefetch, LDS reuse between boards, the conditional end
    barrier), launches of >= 1,024 boards alternate wave priorities, launches of >= 2,048 start the second-resident workgroups
    with an offset.  Every row against the fp64 oracle at the same tolerance as the one-board-per-workgroup test, LDS poisoned
    first; a failing row's index mod 512 tells which loop iteration produced it."""
    from alphaquoridorgnn_amd import _lib
    from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
    params, ref = _walk_oracle(11)
    model = GNNNetwork()
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    model = model.to(dev).eval()
    _lib.set_option("trunk_variant", variant)
    try:
        assert not (model.gnn_flags(dev) & _lib.GNN_EXACT_F32)
        g = U.golden("walk_9x9.npz")
        idx = np.random.RandomState(B + variant).randint(0, g["states"].shape[0], size=B)
        recs = torch.from_numpy(g["states"][idx]).to(dev)
        _lib.poison_lds(dev)
        policy, value, logits, vpre = model.forward_states(recs, want_logits=True)
        lg, vp = logits.cpu().numpy().astype(np.float64), vpre.cpu().numpy().astype(np.float64)
        bad = np.nonzero(~np.isclose(lg, ref["logits"][idx], atol=1e-5, rtol=1e-4).all(1) |
                         ~np.isclose(vp, ref["value_pre"][idx], atol=1e-5, rtol=1e-4))[0]
        assert bad.size == 0, f"{bad.size} rows off, first {bad[:8]} (row mod 512: {bad[:8] % 512})"
        np.testing.assert_allclose(policy.cpu().numpy(), ref["policy"][idx], atol=1e-6, rtol=1e-4)
        np.testing.assert_allclose(value[:, 0].cpu().numpy(), ref["value"][idx], atol=1e-5, rtol=1e-4)
        # a row does not depend on which workgroup / loop iteration computed it: the same boards in another order
        perm = torch.from_numpy(np.random.RandomState(1).permutation(B)).to(dev)
        _lib.poison_lds(dev)
        p2, v2 = model.forward_states(recs[perm].contiguous())
        assert torch.equal(p2, policy[perm]) and torch.equal(v2, value[perm])
    finally:
        _lib.set_option("trunk_variant", 3)


@pytest.mark.parametrize("variant", [6, 1])
def test_engine_masked_trunk_launch(dev, variant):
    """The trunk as the ENGINE launches it: 24-byte packed leaf states (fmt 1) + the leaf_flag mask.  2,048 roots of which 35 %
    are terminal (enemy on its goal row: game_logic.py:43-46, never evaluated, pv_mcts.py:35-42), one simulation: the rows of
    evaluated roots must equal the mask-free forward of the same packed states bit for bit and the oracle within the
    tolerance; the rows of masked-out roots must keep the sentinel the test put there (pooled, policy and value)."""
    from alphaquoridorgnn_amd import _lib
    from alphaquoridorgnn_amd.engine import BatchedSelfPlay
    from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
    params, ref = _walk_oracle(11)
    model = GNNNetwork()
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    model = model.to(dev).eval()
    G = 2048
    g = U.golden("walk_9x9.npz")
    rng = np.random.RandomState(77)
    idx = rng.randint(0, g["states"].shape[0], size=G)
    recs = g["states"][idx].copy()
    dead = rng.rand(G) < 0.35
    dead[:3] = [True, False, True]
    recs[dead, 2] = recs[dead, 2] % 9                     # enemy pawn onto row 0 of its own frame: is_lose()
    _lib.set_option("trunk_variant", variant)
    try:
        eng = BatchedSelfPlay(model, num_games=G, sims=1, record_history=False)
        for name in ("pooled", "policy", "value"):
            eng.t[name].fill_(-7.25)
        _lib.poison_lds(dev)
        eng.search(recs)
        torch.cuda.synchronize()
        live = torch.from_numpy(~dead).to(dev)
        assert int(eng.t["stat_leaf_evals"].sum()) == int((~dead).sum())
        for name in ("pooled", "policy", "value"):
            assert bool((eng.t[name][~live] == -7.25).all()), name
        # mask-free forward of the engine's own packed leaf states (the roots), same kernels: bit-identical rows
        policy, value = model.forward_states(eng.t["leaf_state"][live].contiguous(), state_fmt=1)
        assert torch.equal(eng.t["policy"][live], policy) and torch.equal(eng.t["value"][live], value[:, 0])
        keep = ~dead
        np.testing.assert_allclose(eng.t["policy"][live].cpu().numpy(), ref["policy"][idx][keep], atol=1e-6, rtol=1e-4)
        np.testing.assert_allclose(eng.t["value"][live].cpu().numpy(), ref["value"][idx][keep], atol=1e-5, rtol=1e-4)
        np.testing.assert_allclose(eng.t["pooled"][live].cpu().numpy(), ref["pooled"][idx][keep], atol=2e-6, rtol=1e-4)
    finally:
        _lib.set_option("trunk_variant", 3)


def test_gnn_fp16_range_guard(dev):
    """The default kernels hold activations as fp16 hi + lo pairs: fp32-equivalent only inside fp16 range.  The reference's
    fp32 has no such cliff (pv_network_gnn.py:53-64), so every weight set is checked against the exact f32 kernels on
    calibration boards when it is packed: a sanely scaled net (also x3, with non-zero biases, as above) keeps the fast kernels,
    weights x300 (activations ~1e9) are served by the exact ones -- and still match the fp64 oracle in relative terms."""
    from alphaquoridorgnn_amd import _lib
    from alphaquoridorgnn_amd.engine import BatchedSelfPlay
    from oracle import gnn as og
    model, params = _model(3)
    assert model.gnn_flags(dev) == _lib.GNN_RANGE_PROVEN          # initialisation-scale weights: the static bound holds, no tracking needed
    g = U.golden("walk_9x9.npz")
    recs = g["states"][7000:7048]
    for scale, want_flag in ((3.0, 0), (300.0, _lib.GNN_EXACT_F32)):
        big = {k: (v * (scale if "gcn" in k and "weight" in k else 1.0)).astype(np.float32) for k, v in params.items()}
        big["gcn_layers.0.bias"] = np.linspace(-0.2, 0.4, 128).astype(np.float32)
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in big.items()})
        assert model.gnn_flags(dev) == want_flag, scale
        ref = og.forward_states(big, recs)
        _, _, logits, vpre = model.forward_states(torch.from_numpy(recs).to(dev), want_logits=True)
        lg, vp = logits.cpu().numpy().astype(np.float64), vpre.cpu().numpy().astype(np.float64)
        assert np.isfinite(lg).all() and np.isfinite(vp).all()
        sc = np.abs(ref["logits"]).max()
        np.testing.assert_allclose(lg, ref["logits"], atol=2e-5 * max(sc, 1.0), rtol=2e-4)
        np.testing.assert_allclose(vp, ref["value_pre"], atol=2e-5 * max(np.abs(ref["value_pre"]).max(), 1.0), rtol=2e-4)
    # the flag travels into the engine: a search with the x300 net uses the exact kernels and stays finite
    eng = BatchedSelfPlay(model, num_games=4, sims=6, record_history=False)
    assert eng.e.gnn_flags == _lib.GNN_EXACT_F32
    visits, _, cnt = eng.search(recs[:4])
    assert int(visits.sum()) == 4 * 5


def test_gnn_runtime_saturation_signal(dev):
    """The fp16-split kernels report at RUN TIME when a value leaves fp16 range (the calibration boards cannot see every input).
    Weight sets that pass the calibration -- all its boards have <= 10 walls in hand -- but leave the range on one crafted record
    with 255 walls in hand (the network takes any feature value; pv_network_cnn.py:96 puts the count in a plane as it is):
      A. a post-ReLU activation beyond 65504 (layer 1), B. a layer-2 linear-map output beyond it, negative (-> -inf -> 0 x inf = NaN
      in the aggregation); and C. pooled features beyond it on every board (a layer-3 bias of 1e5: caught by the heads kernel, and
      by the calibration already).
    For each: the guarded C entry sets the caller's word on the crafted record and leaves it alone on an ordinary one; the module
    notices, switches itself to the exact f32-input kernels and returns outputs that match the fp64 oracle; an engine searching
    from the crafted root reports counters()['gnn_saturated']."""
    from alphaquoridorgnn_amd import _lib
    from alphaquoridorgnn_amd.engine import BatchedSelfPlay
    from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
    from oracle import gnn as og
    lib = _lib.load()
    g = U.golden("walk_9x9.npz")
    normal = g["states"][[10, 400, 3000, 9000]].copy()
    crafted = normal.copy()
    crafted[:, 1] = 255                                    # walls in hand of the mover: feature 1 of every node
    base = og.init_params(6)

    def variant(which):
        p = {k: v.copy() for k, v in base.items()}
        if which == "A":
            p["gcn_layers.0.lin.weight"][:, 1] = 200.0     # layer-1 activations ~ 8e3 at 10 walls, ~ 2e5 at 255
        elif which == "B":
            p["gcn_layers.0.lin.weight"][:, 1] = 20.0      # layer 1 stays in range (~ 2e4 at 255 walls) ...
            p["gcn_layers.1.lin.weight"][:] = -0.05        # ... its 128 equal features sum to z ~ -1.3e5 in layer 2
        else:
            p["gcn_layers.2.bias"][:] = 1.0e5              # every pooled feature ~ 1e5 on every board: the heads' fp16 split overflows
        return p

    for which in ("A", "B", "C"):
        params = variant(which)
        model = GNNNetwork()
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
        model = model.to(dev).eval()
        pk = model.packed_weights(dev)
        if which != "C":
            assert model.gnn_flags(dev) == 0, which       # passes the calibration: the fast kernels are in use
        else:
            assert model.gnn_flags(dev) == _lib.GNN_EXACT_F32    # the calibration boards overflow as well: caught when the set is packed
        word = torch.zeros((1,), dtype=torch.int32, device=dev)
        pooled = torch.empty((4, 128), device=dev)
        policy = torch.empty((4, 209), device=dev)
        value = torch.empty((4,), device=dev)
        for recs, want in ((normal, 0), (crafted, 1)):
            word.zero_()
            d = torch.from_numpy(recs).to(dev)
            _lib.check(lib.aqg_gcn_forward_boards_guarded(9, _lib.ptr(d), 0, 4, _lib.ptr(pk), _lib.ptr(pooled), None, _lib.ptr(policy), None,
                                                          _lib.ptr(value), 0, _lib.ptr(word), _lib.stream_ptr(dev)), "guarded")
            got = int(word.item())
            if which == "C" and want == 0:
                continue                                   # (see above)
            assert got == want, (which, want, got)
            if which != "C":
                assert bool(torch.isfinite(policy).all()) and bool(torch.isfinite(value).all())   # activations are clamped, never inf / NaN
        # the module: notices, switches, returns the network's outputs
        ref = og.forward_states(params, crafted)
        _, _, logits, vpre = model.forward_states(torch.from_numpy(crafted).to(dev), want_logits=True)
        assert model.gnn_flags(dev) == _lib.GNN_EXACT_F32, which
        sc = max(float(np.abs(ref["logits"]).max()), 1.0)
        np.testing.assert_allclose(logits.cpu().numpy(), ref["logits"], atol=2e-5 * sc, rtol=2e-4)
        np.testing.assert_allclose(vpre.cpu().numpy(), ref["value_pre"], atol=2e-5 * max(float(np.abs(ref["value_pre"]).max()), 1.0), rtol=2e-4)
        # the engine: a search from the crafted roots on a fresh copy of the weights (fast kernels again) raises its counter
        model2 = GNNNetwork()
        model2.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
        model2 = model2.to(dev).eval()
        if model2.gnn_flags(dev) == 0:
            eng = BatchedSelfPlay(model2, num_games=4, sims=3, record_history=False)
            assert eng.counters()["gnn_saturated"] == 0
            eng.search(crafted, check_saturation=False)
            assert eng.counters()["gnn_saturated"] == 1, which


def test_gnn_range_guard_watches_every_feature(dev):
    """The run-time fp16-range guard of the split kernels sees an excursion in ANY single feature column, whatever its position in a
    lane's group of four (a compiler defect once made it watch one value in four while every test overflowed whole matrices):
    one layer-1 output feature j beyond 65504 (positive), and one layer-2 linear-map output feature j beyond it on the negative
    side, for j in every residue class mod 4 -- on the guarded C entry, for every split trunk form."""
    from alphaquoridorgnn_amd import _lib
    from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
    from oracle import gnn as og
    lib = _lib.load()
    g = U.golden("walk_9x9.npz")
    normal = g["states"][[10, 400, 3000, 9000]].copy()
    crafted = normal.copy()
    crafted[:, 1] = 255
    base = og.init_params(6)
    word = torch.zeros((1,), dtype=torch.int32, device=dev)
    pooled = torch.empty((4, 128), device=dev)
    try:
        for variant in (6,):
            _lib.set_option("trunk_variant", variant)
            for kind in ("positive", "negative"):
                for j in (0, 1, 2, 3, 37, 66, 127):
                    p = {k: v.copy() for k, v in base.items()}
                    if kind == "positive":
                        p["gcn_layers.0.lin.weight"][j, 1] = 200.0           # layer-1 feature j ~ 2e5 at 255 walls in hand
                    else:
                        p["gcn_layers.0.lin.weight"][:, 1] = 20.0            # layer 1 in range (~ 2e4) ...
                        p["gcn_layers.1.lin.weight"][j, :] = -0.05           # ... layer-2 output feature j ~ -1.3e5
                    model = GNNNetwork()
                    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in p.items()})
                    model = model.to(dev).eval()
                    pk = model.packed_weights(dev)
                    for recs, want in ((normal, 0), (crafted, 1)):
                        word.zero_()
                        d = torch.from_numpy(recs).to(dev)
                        _lib.check(lib.aqg_gcn_forward_boards_guarded(9, _lib.ptr(d), 0, 4, _lib.ptr(pk), _lib.ptr(pooled), None, None, None, None,
                                                                      0, _lib.ptr(word), _lib.stream_ptr(dev)), "guarded")
                        assert int(word.item()) == want, (variant, kind, j, want)
        # a weight outside fp16 range: its hi half packs as inf, its column becomes NaN, and the tracking build's float maxima skip
        # NaNs -- the pack function stores negative thresholds for such a set, so a DIRECT caller of the guarded entry with flags 0
        # (the Python wrapper's calibration would have caught it) gets the word on ordinary records too
        import ctypes
        for key, idx, val in (("gcn_layers.1.lin.weight", (33, 70), 7.0e4), ("gcn_layers.2.lin.weight", (2, 5), float("nan")), ("gcn_layers.0.lin.weight", (77, 4), -8.0e4)):
            p = {k: v.copy() for k, v in base.items()}
            p[key][idx] = val
            host = [np.ascontiguousarray(p[k], dtype=np.float32) for k in og.KEYS]
            arr = (ctypes.c_void_p * 14)(*[h.ctypes.data_as(ctypes.c_void_p) for h in host])
            out = np.zeros(lib.aqg_gcn_packed_floats(9), dtype=np.float32)
            assert lib.aqg_gcn_pack_weights_host(9, arr, out.ctypes.data_as(ctypes.c_void_p)) == 0
            pk = torch.from_numpy(out).to(dev)
            word.zero_()
            d = torch.from_numpy(normal).to(dev)
            _lib.check(lib.aqg_gcn_forward_boards_guarded(9, _lib.ptr(d), 0, 4, _lib.ptr(pk), _lib.ptr(pooled), None, None, None, None,
                                                          0, _lib.ptr(word), _lib.stream_ptr(dev)), "guarded")
            assert int(word.item()) == 1, (key, val)
    finally:
        _lib.set_option("trunk_variant", 3)


def test_gnn_range_proven_path(dev):
    """A weight set whose activations are bounded inside fp16 range for EVERY input with at most 16 walls in hand (a static bound
    from the weights, pv_network_gnn._range_proven) is served by the trunk build without per-value range tracking: bit-identical
    results to the tracking build on ordinary positions; a record outside the bound's premise (255 walls in hand) raises the
    guard's word through the per-record wall-count check, and the module then returns the network's outputs from the exact kernels.
    The bound itself: holds for initialisation-scale weights, fails -- as it must, being a bound -- once the trunk weights are x3."""
    from alphaquoridorgnn_amd import _lib
    from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
    from oracle import gnn as og
    lib = _lib.load()
    model, params = _model(9)
    assert model.gnn_flags(dev) == _lib.GNN_RANGE_PROVEN
    pk = model.packed_weights(dev)
    g = U.golden("walk_9x9.npz")
    recs = g["states"][500:1100].copy()
    d = torch.from_numpy(recs).to(dev)
    word = torch.zeros((1,), dtype=torch.int32, device=dev)
    outs = []
    for flags in (0, _lib.GNN_RANGE_PROVEN):
        pooled = torch.empty((recs.shape[0], 128), device=dev)
        policy = torch.empty((recs.shape[0], 209), device=dev)
        _lib.check(lib.aqg_gcn_forward_boards_guarded(9, _lib.ptr(d), 0, recs.shape[0], _lib.ptr(pk), _lib.ptr(pooled), None, _lib.ptr(policy), None,
                                                      None, flags, _lib.ptr(word), _lib.stream_ptr(dev)), "guarded")
        outs.append((pooled, policy))
    assert int(word.item()) == 0
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    crafted = recs[:4].copy()
    crafted[1, 1] = 255                                   # the mover's walls in hand, one record of four
    dc = torch.from_numpy(crafted).to(dev)
    pooled = torch.empty((4, 128), device=dev)
    _lib.check(lib.aqg_gcn_forward_boards_guarded(9, _lib.ptr(dc), 0, 4, _lib.ptr(pk), _lib.ptr(pooled), None, None, None, None,
                                                  _lib.GNN_RANGE_PROVEN, _lib.ptr(word), _lib.stream_ptr(dev)), "guarded")
    assert int(word.item()) == 1
    crafted[1, 1] = 0
    crafted[2, 3] = 17                                    # the enemy's count, just beyond the bound's premise
    word.zero_()
    _lib.check(lib.aqg_gcn_forward_boards_guarded(9, _lib.ptr(torch.from_numpy(crafted).to(dev)), 0, 4, _lib.ptr(pk), _lib.ptr(pooled), None, None, None,
                                                  None, _lib.GNN_RANGE_PROVEN, _lib.ptr(word), _lib.stream_ptr(dev)), "guarded")
    assert int(word.item()) == 1
    crafted[2, 3] = 16                                    # at the premise: fine
    word.zero_()
    _lib.check(lib.aqg_gcn_forward_boards_guarded(9, _lib.ptr(torch.from_numpy(crafted).to(dev)), 0, 4, _lib.ptr(pk), _lib.ptr(pooled), None, None, None,
                                                  None, _lib.GNN_RANGE_PROVEN, _lib.ptr(word), _lib.stream_ptr(dev)), "guarded")
    assert int(word.item()) == 0
    # the module on a record outside the premise: notices, switches, returns the network's outputs
    crafted[1, 1] = 255
    ref = og.forward_states(params, crafted)
    _, _, logits, vpre = model.forward_states(torch.from_numpy(crafted).to(dev), want_logits=True)
    assert model.gnn_flags(dev) == _lib.GNN_EXACT_F32
    sc = max(float(np.abs(ref["logits"]).max()), 1.0)
    np.testing.assert_allclose(logits.cpu().numpy(), ref["logits"], atol=2e-5 * sc, rtol=2e-4)
    # the bound is a bound: x3 on the trunk weights and it no longer holds (the tracking build serves the set)
    big = {k: (v * (3.0 if "gcn" in k and "weight" in k else 1.0)).astype(np.float32) for k, v in params.items()}
    m3 = GNNNetwork()
    m3.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in big.items()})
    assert m3.to(dev).eval().gnn_flags(dev) == 0


def test_gnn_small_boards_forward_and_selfplay(dev):
    """The reference's smaller boards (constants.py:5-20) with the GNN evaluator: the any-size forward (plain kernels)
    against the fp64 oracle on 5x5 fixtures states, and a GNN-driven 5x5 self-play generation on the engine."""
    from alphaquoridorgnn_amd.engine import BatchedSelfPlay
    from alphaquoridorgnn_amd.pv_network_gnn import GraphPolicyValueNetwork
    from oracle import gnn as og, quoridor as oq
    N, A = 5, 25 + 2 * 16
    params = og.init_params(1, N=N)
    model = GraphPolicyValueNetwork(6, 128, 3, A, board_size=N)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    model = model.to(dev).eval()
    g = U.golden("walk_5x5.npz")
    recs = g["states"][::23][:150]
    ref = og.forward_states(params, recs)
    policy, value, logits, vpre = model.forward_states(torch.from_numpy(recs).to(dev), want_logits=True)
    np.testing.assert_allclose(logits.cpu().numpy(), ref["logits"], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(vpre.cpu().numpy(), ref["value_pre"], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(policy.cpu().numpy(), ref["policy"], atol=1e-6, rtol=1e-4)
    eng = BatchedSelfPlay(model, num_games=40, sims=10, board_size=N, seed=2)
    c = eng.play_generation()
    assert c["active"] == 0 and c["finished"] == 40
    st, vis, z = (x.cpu().numpy() for x in eng.history_tensors())
    assert st.shape[0] == vis.shape[0] == z.shape[0] > 40 and (vis.sum(1) == 9).all()
    assert np.array_equal(st[0], oq.init_record(N))
    # ... and the training step on that board: gradients against fp64 autograd
    from alphaquoridorgnn_amd.train_network import GNNTrainer
    from oracle import train as ot
    rng = np.random.RandomState(4)
    tr_recs = recs[:24]
    pi = rng.rand(24, A).astype(np.float32)
    pi /= pi.sum(1, keepdims=True)
    zt = rng.choice([-1.0, 0.0, 1.0], 24).astype(np.float32)
    tr = GNNTrainer(model, max_batch=24)
    from alphaquoridorgnn_amd import _lib
    _lib.poison_lds(dev)                   # the NaN-padding bug of round 2 was found in exactly this step
    tr.step(torch.from_numpy(tr_recs), torch.from_numpy(pi), torch.from_numpy(zt), update=False)
    refg = ot.train_steps(params, [(tr_recs, pi.astype(np.float64), zt.astype(np.float64))])[0]["grads"]
    for k, gt in zip(og.KEYS, tr.grads):
        assert np.abs(gt.cpu().numpy().astype(np.float64) - refg[k]).max() <= 2e-5 * np.abs(refg[k]).max() + 1e-7, k


def test_gnn_forward_generic_graph(dev):
    """forward(x, edge_index, batch) on (a) the board graphs and (b) an arbitrary ragged graph batch."""
    from oracle import gnn as og
    model, params = _model(1)
    g = U.golden("walk_9x9.npz")
    recs = g["states"][100:140]
    xs, es, bs, off = [], [], [], 0
    for b, rec in enumerate(recs):
        x = og.node_features(rec); e = og.board_edges(rec)
        xs.append(x); es.append(e + off); bs.append(np.full(81, b)); off += 81
    x = torch.from_numpy(np.concatenate(xs)).float().to(dev)
    ei = torch.from_numpy(np.concatenate(es, 1)).to(dev)
    bt = torch.from_numpy(np.concatenate(bs)).to(dev)
    policy, value = model(x, ei, bt)
    ref = og.forward_states(params, recs)
    np.testing.assert_allclose(model.last_logits.cpu().numpy(), ref["logits"], atol=1e-5, rtol=1e-4)
    np.testing.assert_allclose(value[:, 0].cpu().numpy(), ref["value"], atol=1e-5, rtol=1e-4)
    # same boards through the fused path agree with the generic path
    p2, v2 = model.forward_states(torch.from_numpy(recs).to(dev))
    np.testing.assert_allclose(policy.cpu().numpy(), p2.cpu().numpy(), atol=1e-6, rtol=1e-4)
    # arbitrary graphs: ragged sizes, isolated nodes, an explicit self loop, duplicate-free random edges
    rng = np.random.RandomState(4)
    sizes = [1, 7, 30, 81, 2]
    xs, es, bs, off = [], [], [], 0
    for gi, n in enumerate(sizes):
        xs.append(rng.randn(n, 6))
        if n > 1:
            pairs = {(int(a), int(b)) for a, b in rng.randint(0, n, size=(3 * n, 2)) if a != b}
            e = np.asarray(sorted(pairs), dtype=np.int64).T
            if gi == 2:
                e = np.concatenate([e, np.asarray([[0], [0]])], 1)   # explicit self loop on node 0
            es.append(e + off)
        bs.append(np.full(n, gi)); off += n
    xn, en, bn = np.concatenate(xs), np.concatenate(es, 1), np.concatenate(bs)
    ref = og.forward_graph(params, xn, en, bn, len(sizes))
    policy, value = model(torch.from_numpy(xn).float().to(dev), torch.from_numpy(en).to(dev), torch.from_numpy(bn).to(dev))
    np.testing.assert_allclose(model.last_logits.cpu().numpy(), ref["logits"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(value[:, 0].cpu().numpy(), ref["value"], atol=2e-5, rtol=1e-4)


def test_predict_contract(dev):
    """P0 (pv_network_cnn.py:117-137): PMF over legal_actions() in order, float32 numpy, python float value."""
    from alphaquoridorgnn_amd.game_logic import State
    from oracle import gnn as og
    model, params = _model(2)
    s = State().next(67).next(81 + 20)
    pol, val = model.predict(s, "cuda")
    la = s.legal_actions()
    assert isinstance(pol, np.ndarray) and pol.dtype == np.float32 and pol.shape == (len(la),)
    assert isinstance(val, float) and abs(float(pol.sum()) - 1) < 1e-5
    ref = og.forward_states(params, s.record()[None])
    exp = ref["policy"][0][la] / ref["policy"][0][la].sum()
    np.testing.assert_allclose(pol, exp, atol=1e-6, rtol=1e-4)
    assert abs(val - ref["value"][0]) < 1e-5


def _root_children(eng):
    """Root children of every game after a search, straight from the tree pool (csrc/mcts.hip NodeRec, 32 bytes:
    f64 w | f32 p | u32 action || i32 n | u32 first_child + (count << 24) | f32 q | f32 C_PUCT * p): (priors, visits, actions) per game."""
    G, cap = eng.G, eng.node_cap
    raw = eng.t["node_rec"].view(torch.uint8).view(G, cap, 32).cpu().numpy()
    out = []
    for g in range(G):
        kids = int(raw[g, 0, 20:24].view(np.uint32)[0])
        first, cnt = kids & 0xFFFFFF, kids >> 24
        ch = raw[g, first:first + cnt]
        out.append((ch[:, 8:12].copy().view(np.float32)[:, 0], ch[:, 16:20].copy().view(np.int32)[:, 0],
                    ch[:, 12:16].copy().view(np.uint32)[:, 0]))
    return out


def test_engine_priors_and_visits_vs_oracle_gnn(dev):
    """P0 INSIDE the engine (pv_network_cnn.py:129-132 as done by game_expand_backup with prior_mode 0: gather the softmax
    output at legal_actions() in order, divide by the sum): the priors stored in the root's children must equal
    OracleModel.predict (fp64 GNN + C rules) to 1e-6, child i must carry legal action i, and a 10-simulation search must
    distribute its visits exactly like oracle.mcts driven by that model (pv_mcts.py:47-57)."""
    from alphaquoridorgnn_amd.engine import BatchedSelfPlay
    from oracle import gnn as og, mcts as om, quoridor as oq
    model, params = _model(4)
    oracle = og.OracleModel(params)
    g = U.golden("walk_9x9.npz")
    idx = [0, 5, 40, 333, 1200, 2600, 5000, 9000]
    recs = np.stack([g["states"][i] for i in idx])
    recs = recs[[not (oq.State(r).is_done()) for r in recs]]
    sims = 10
    eng = BatchedSelfPlay(model, num_games=recs.shape[0], sims=sims, record_history=False)
    eng.search(recs)
    torch.cuda.synchronize()
    for rec, (pri, vis, act) in zip(recs, _root_children(eng)):
        st = oq.State(rec)
        legal = st.legal_actions()
        assert [int(a) for a in act] == [int(a) for a in legal]
        want, _ = oracle.predict(st)
        np.testing.assert_allclose(pri, want, atol=1e-6, rtol=1e-5)
        assert abs(float(pri.sum()) - 1.0) < 1e-5
        root = om.search(oracle, st, sims)
        assert [int(v) for v in vis] == [c.n for c in root.children]


# ------------------------------------------------------------------ K4 MCTS / self-play against reference traces
@pytest.mark.parametrize("N", [9, 5, 3])
def test_mcts_visit_counts_match_reference_traces(dev, N):
    """Golden traces were produced by the REAL reference pv_mcts.py with the integer-hash fake model; the engine's
    `fake` evaluator reproduces that model exactly, so visit distributions must be bit-identical."""
    from alphaquoridorgnn_amd.pv_mcts import pv_mcts_policy_batch
    g = U.golden(f"mcts_{N}x{N}.npz")
    n = int(g["count"][0])
    groups = {}
    for k in range(n):
        sims, bias, T = g[f"t{k}_cfg"]
        groups.setdefault((int(sims), int(bias), float(T)), []).append(k)
    for (sims, bias, T), ks in groups.items():
        recs = np.stack([g[f"t{k}_state"] for k in ks])
        pols = pv_mcts_policy_batch(None, recs, T, sims=sims, board_size=N, evaluator="fake", fake_bias=bias)
        for k, pol in zip(ks, pols):
            assert np.array_equal(np.asarray(pol, dtype=np.float64), g[f"t{k}_policy"]), (k, sims, bias, T)


@pytest.mark.parametrize("N", [9, 5, 3])
def test_selfplay_games_match_reference(dev, N):
    """Whole games through the engine == the reference's self_play.play() (seeded np.random, fake model):
    same states, same visit distributions (float64-exact), same z."""
    from alphaquoridorgnn_amd.engine import BatchedSelfPlay
    g = U.golden(f"games_{N}x{N}.npz")
    for i in range(int(g["count"][0])):
        seed, sims, bias = (int(x) for x in g[f"g{i}_cfg"])
        rng = np.random.RandomState(seed)
        eng = BatchedSelfPlay(None, num_games=1, sims=sims, board_size=N, evaluator="fake", fake_bias=bias)
        u = rng.random_sample(size=(eng.max_plies, 1))     # one uniform per move, like np.random.choice
        eng.play_generation(uniforms=torch.from_numpy(u), check_every=1)
        hist = eng.history()
        st = g[f"g{i}_states"]
        assert len(hist) == st.shape[0], (i, len(hist), st.shape[0])
        for j, (sa, pol, z) in enumerate(hist):
            assert sa[0] == list(st[j, 0:2]) and sa[1] == list(st[j, 2:4]) and sa[2] == list(st[j, 4:4 + (N - 1) ** 2])
            assert np.array_equal(np.asarray(pol, dtype=np.float64), g[f"g{i}_policy"][j]), (i, j)
            assert z == int(g[f"g{i}_z"][j])


@pytest.mark.parametrize("N", [9, 5])
def test_evaluation_games_match_reference(dev, N):
    """evaluate_network row: a two-model game on the engine == the reference's evaluate_network.play()
    with two fake models through pv_mcts_action (tools/gen_golden_eval.py): same action on every ply, same point."""
    from alphaquoridorgnn_amd.evaluate_network import BatchedMatch
    g = U.golden(f"eval_{N}x{N}.npz")
    for i in range(int(g["count"][0])):
        seed, sims, b0, b1 = (int(x) for x in g[f"e{i}_cfg"])
        m = BatchedMatch((b0, b1), 1, sims=sims, board_size=N, evaluator="fake")
        eng = m.engines[0]
        u = np.random.RandomState(seed).random_sample(size=(eng.max_plies, 1))   # one uniform per move (np.random.choice)
        points = m.play(uniforms=(torch.from_numpy(u), None))
        ref_actions = g[f"e{i}_actions"]
        plies = int(eng.t["game_plies"][0])
        assert plies == len(ref_actions)
        assert np.array_equal(eng.t["hist_action"][0, :plies].cpu().numpy().astype(np.int16), ref_actions)
        assert points == [float(g[f"e{i}_point"][0])]


@pytest.mark.parametrize("cfg", [(0, 61), (1, 0), (1, 1), (1, 2), (1, 5)])
def test_step_kernel_variants_bit_identical(dev, cfg):
    """The simulation step exists in two forms (csrc/mcts.hip): `step_variant` 0 = expand / backup through memory, fence,
    select; 1 (default) = one load round + the previous simulation's updates applied in registers to whatever the descent
    loads.  Variant 1 hands over to memory once a path gets deeper than `step_fast_depth` (61 by default, i.e. never in
    practice): with the limit at 0, 1, 2 and 5 every hand-over point is exercised.  All forms must reproduce the
    reference's traces, self-play games and evaluation games bit for bit."""
    from alphaquoridorgnn_amd import _lib
    variant, depth = cfg
    _lib.set_option("step_variant", variant)
    _lib.set_option("step_fast_depth", depth)
    try:
        test_mcts_visit_counts_match_reference_traces(dev, 9)
        test_selfplay_games_match_reference(dev, 9)
        test_evaluation_games_match_reference(dev, 9)
        test_mcts_visit_counts_match_reference_traces(dev, 5)
        test_mcts_many_games_equal_single_game(dev)
        if variant == 1 and depth == 2:
            test_engine_priors_and_visits_vs_oracle_gnn(dev)
    finally:
        _lib.set_option("step_variant", 1)
        _lib.set_option("step_fast_depth", 61)


def test_batched_match_colours_and_points(dev):
    """BatchedMatch bookkeeping: game i has player i % 2 moving first, points are player 0's, and a batch of games equals
    the same games played one at a time (fake evaluator, explicit uniforms)."""
    from alphaquoridorgnn_amd.evaluate_network import BatchedMatch
    rng = np.random.RandomState(3)
    G = 5
    m = BatchedMatch((40, 0), G, sims=8, board_size=5, evaluator="fake")
    ua = rng.random_sample(size=(m.engines[0].max_plies, 3))
    ub = rng.random_sample(size=(m.engines[1].max_plies, 2))
    pts = m.play(uniforms=(torch.from_numpy(ua), torch.from_numpy(ub)))
    assert len(pts) == G and all(p in (0.0, 0.5, 1.0) for p in pts)
    for i in range(G):
        first, col = i % 2, i // 2
        one = BatchedMatch((40, 0) if first == 0 else (0, 40), 1, sims=8, board_size=5, evaluator="fake")
        u = (ua if first == 0 else ub)[:, col:col + 1]
        p = one.play(uniforms=(torch.from_numpy(np.ascontiguousarray(u)), None))[0]
        assert pts[i] == (p if first == 0 else 1.0 - p)


def test_evaluate_network_with_gnn(dev, tmp_path, monkeypatch):
    """evaluate_network() end to end with two random GNNs saved as latest.pth / best.pth: returns a bool and promotes
    by copying the file exactly when it says so (evaluate_network.py:90-94)."""
    from alphaquoridorgnn_amd import evaluate_network as en, pv_mcts
    from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
    monkeypatch.setattr(pv_mcts, "PV_EVALUATE_COUNT", 6)
    monkeypatch.setattr(en, "EN_GAME_COUNT", 6)
    path = str(tmp_path) + "/"
    monkeypatch.setattr(en, "PV_NETWORK_PATH", path)
    torch.manual_seed(1); torch.save(GNNNetwork().state_dict(), path + "latest.pth")
    torch.manual_seed(2); torch.save(GNNNetwork().state_dict(), path + "best.pth")
    before = open(path + "best.pth", "rb").read()
    promoted = en.evaluate_network()
    after = open(path + "best.pth", "rb").read()
    assert isinstance(promoted, bool)
    assert (after == open(path + "latest.pth", "rb").read()) if promoted else (after == before)


def test_mcts_many_games_equal_single_game(dev):
    """Lock-step batching must not couple games: 257 copies of different roots == each searched alone (oracle)."""
    from alphaquoridorgnn_amd.pv_mcts import pv_mcts_policy_batch
    from oracle import mcts as om, quoridor as oq
    g = U.golden("walk_9x9.npz")
    ok = (g["status"] == 0) & (g["counts"] > 0)
    recs = g["states"][ok][::47][:257]
    pols = pv_mcts_policy_batch(None, recs, 1.0, sims=40, board_size=9, evaluator="fake", fake_bias=7)
    for b in range(0, len(recs), 16):
        ref = om.pv_mcts_policy(om.FakeModel(7), oq.State(recs[b]), 1.0, 40)
        assert np.array_equal(np.asarray(pols[b]), np.asarray(ref))


def test_selfplay_generation_with_gnn(dev):
    """End-to-end generation with the real GNN evaluator: structural invariants of the recorded tuples."""
    from alphaquoridorgnn_amd.engine import BatchedSelfPlay
    from oracle import quoridor as oq
    model, _ = _model(0)
    eng = BatchedSelfPlay(model, num_games=96, sims=12, board_size=9, seed=5)
    c = eng.play_generation()
    assert c["active"] == 0 and c["finished"] == 96 and c["dead_ends"] == 0
    st, vis, z = (x.cpu().numpy() for x in eng.history_tensors())
    plies = eng.t["game_plies"].cpu().numpy()
    assert st.shape[0] == plies.sum() == vis.shape[0] == z.shape[0]
    assert (vis.sum(1) == 11).all()                       # root visit counts sum to sims - 1
    acts = eng.t["hist_action"].cpu().numpy()
    off = 0
    for gme in range(96):
        p = plies[gme]
        s = st[off:off + p]
        assert np.array_equal(s[0], oq.init_record(9))
        nxt = oq.next_batch(s[:-1], acts[gme, :p - 1])
        assert np.array_equal(nxt, s[1:])                 # recorded states chain through next()
        _, _, mask = oq.legal_actions_batch(s)
        assert ((vis[off:off + p] > 0) <= (mask > 0)).all()   # visits only on legal actions
        zz = z[off:off + p]
        assert (zz[1:] == -zz[:-1]).all()
        last = oq.next_record(s[-1], acts[gme, p - 1])
        flag = oq.status_batch(last, 116)[0]
        assert flag != 0
        if flag & 1:
            assert zz[0] == (-1 if (p % 2 == 0) else 1)
        else:
            assert zz[0] == 0
        off += p


def test_selfplay_generation_full_config_invariants(dev):
    """BASELINE configs[2] at its full size -- 2048 concurrent games x 200 simulations per move, GNN evaluator, the four
    game sets the benchmark uses -- through the structural invariants of a generation: every game ends; every recorded
    position's visits sit on legal actions only and sum to sims - 1;
    consecutive states chain through next(); z alternates from the terminal result."""
    from alphaquoridorgnn_amd.engine import MultiSetSelfPlay
    from alphaquoridorgnn_amd import game_logic as gl
    model, _ = _model(0)
    G, sims = 2048, 200
    eng = MultiSetSelfPlay(model, num_games=G, sims=sims, num_sets=4, seed=11)
    c = eng.play_generation()
    assert c["finished"] == G and c["active"] == 0 and c["dead_ends"] == 0
    for e in eng.sets:
        plies = e.t["game_plies"].cpu().numpy()
        assert plies.min() >= 8 and plies.max() <= 116
        st = e.t["hist_state72"]; vis = e.t["hist_visits"].to(torch.int32); act = e.t["hist_action"].long()
        hp = st.shape[1]
        valid = torch.arange(hp, device=dev).unsqueeze(0) < e.t["game_plies"].unsqueeze(1)
        flat = st[valid]
        mask, _, _ = gl.legal_actions_batch(flat.contiguous(), 9)
        v = vis[valid]
        assert int((v * (1 - mask.to(torch.int32))).abs().sum()) == 0              # visits only on legal actions
        tot = v.sum(1)
        assert int(tot.max()) == sims - 1 and int(tot.min()) == sims - 1            # every simulation after the first passes one root child
        a = act[valid]
        assert bool((mask.gather(1, a.unsqueeze(1)) == 1).all())                    # the chosen action is legal
        nxt = gl.next_batch(flat.contiguous(), a.to(torch.int32), 9)
        # state at ply p+1 of the same game == next(state at ply p, action at ply p)
        has_next = valid.clone(); has_next[:, :-1] &= valid[:, 1:]; has_next[:, -1] = False
        nn = torch.zeros_like(st); nn[valid] = nxt
        assert torch.equal(nn[has_next], torch.roll(st, -1, 1)[has_next])
    s72, visits, z = eng.history_tensors()
    assert s72.shape[0] == int(sum(int(e.t["game_plies"].sum()) for e in eng.sets))
    assert set(np.unique(z.cpu().numpy()).tolist()) <= {-1, 0, 1}


def test_slot_refill_games_equal_standalone_games(dev):
    """A rank plays a QUOTA of games on G slots (the reference's plain loop over games, self_play.py:81-84): a slot whose
    game has ended takes the next game not yet handed out.  Every game played that way -- whichever slot, whenever it
    started -- must be exactly the game a stand-alone one-game engine plays on the same per-game stream of uniforms
    (its slot's column of the uniform matrix from its first move on): states, visit counts, actions and result."""
    from alphaquoridorgnn_amd.engine import BatchedSelfPlay
    G, Q, sims, bias = 6, 23, 12, 40
    rs = np.random.RandomState(99)
    U_all = torch.from_numpy(rs.random_sample((116 * 4, G)))
    eng = BatchedSelfPlay(None, num_games=G, quota=Q, sims=sims, evaluator="fake", fake_bias=bias)
    c = eng.play_generation(uniforms=U_all, check_every=1)
    assert c["finished"] == Q and c["started"] == Q and c["active"] == 0 and c["dead_ends"] == 0
    plies = eng.t["game_plies"].cpu().numpy(); slot = eng.t["game_slot"].cpu().numpy(); first = eng.t["game_first_move"].cpu().numpy()
    done = eng.t["game_done"].cpu().numpy(); res = eng.t["game_result"].cpu().numpy()
    assert done.all() and (slot >= 0).all() and sorted(slot[:G].tolist()) == list(range(G)) and (first[:G] == 0).all()
    # a slot's games follow each other without a gap: game starts == previous game's first move + its plies
    for g in range(G):
        ks = [k for k in range(Q) if slot[k] == g]
        assert all(first[b] == first[a] + plies[a] for a, b in zip(ks, ks[1:]))
    # the slots never idled before the quota ran out: moves made == the longest slot's total
    assert c["moves"] == max(sum(plies[k] for k in range(Q) if slot[k] == g) for g in range(G))
    one = BatchedSelfPlay(None, num_games=1, sims=sims, evaluator="fake", fake_bias=bias)
    hs, hv, ha = eng.t["hist_state72"].cpu().numpy(), eng.t["hist_visits"].cpu().numpy(), eng.t["hist_action"].cpu().numpy()
    for k in range(Q):
        one.reset()
        one.play_generation(uniforms=U_all[first[k]:first[k] + 116, slot[k]:slot[k] + 1], check_every=1)
        p = int(one.t["game_plies"][0])
        assert p == plies[k], k
        assert np.array_equal(one.t["hist_state72"][0, :p].cpu().numpy(), hs[k, :p]), k
        assert np.array_equal(one.t["hist_visits"][0, :p].cpu().numpy(), hv[k, :p]), k
        assert np.array_equal(one.t["hist_action"][0, :p].cpu().numpy(), ha[k, :p]), k
        assert int(one.t["game_result"][0]) == res[k], k
    st, vis, z = eng.history_tensors()
    assert st.shape[0] == plies.sum()
    # the same through several game sets with the GNN evaluator: every set fills its own quota
    from alphaquoridorgnn_amd.engine import MultiSetSelfPlay
    model, _ = _model(0)
    ms = MultiSetSelfPlay(model, num_games=8, quota=20, sims=6, num_sets=2, seed=3)
    c = ms.play_generation()
    assert c["finished"] == 20 and c["active"] == 0
    assert ms.history_tensors()[0].shape[0] == sum(int(e.t["game_plies"].sum()) for e in ms.sets)


def test_multiset_selfplay_equals_standalone_sets(dev):
    """engine.MultiSetSelfPlay (K game sets on K streams) = K stand-alone BatchedSelfPlay runs, bit for bit:
    the sets share nothing but read-only weights, so concurrency must not change a single recorded tuple."""
    from alphaquoridorgnn_amd.engine import BatchedSelfPlay, MultiSetSelfPlay
    model, _ = _model(0)
    multi = MultiSetSelfPlay(model, num_games=70, sims=10, num_sets=3, seed=9, board_size=9)
    c = multi.play_generation()
    assert c["active"] == 0 and c["finished"] == 70
    st, vis, z = (x.cpu().numpy() for x in multi.history_tensors())
    assert [e.G for e in multi.sets] == [24, 23, 23]
    parts = []
    for k, g in enumerate((24, 23, 23)):
        eng = BatchedSelfPlay(model, num_games=g, sims=10, seed=9 * 64 + k, board_size=9)
        eng.play_generation()
        parts.append(tuple(x.cpu().numpy() for x in eng.history_tensors()))
    assert np.array_equal(st, np.concatenate([p[0] for p in parts]))
    assert np.array_equal(vis, np.concatenate([p[1] for p in parts]))
    assert np.array_equal(z, np.concatenate([p[2] for p in parts]))


# ------------------------------------------------------------------ training row
def _walk_states(N, count, seed):
    """Reference-walk fixture states; 7x7 (no reference constants, no fixture) from the oracle's rules by random wall-heavy play."""
    if N != 7:
        return U.golden(f"walk_{N}x{N}.npz")["states"]
    from oracle import quoridor as oq
    rng = np.random.RandomState(seed)
    recs = []
    while len(recs) < count:
        s = oq.State(N=N)
        for ply in range(40):
            if s.is_done():
                break
            recs.append(s.rec.copy())
            la = s.legal_actions()
            walls = [a for a in la if a >= N * N]
            pick = walls if (walls and rng.rand() < 0.5) else la
            s = s.next(pick[rng.randint(len(pick))])
    return np.stack(recs)


KINK_MARGIN = 5e-7
KINK_DROP_BOUND = 0.25      # largest share of drawn positions the kink filter may discard (measured: 5-17 % at initialisation-scale weights)
_kink_dropped = {}          # (B, seed, N) -> (drawn, dropped) of the last filtered draw: read by the tests that report it


def _train_batch(B, seed, N=9, params=None):
    """B positions with random targets.  With `params`: only positions whose every ReLU pre-activation under those weights is
    at least KINK_MARGIN away from zero (oracle/gnn.py::relu_margins) -- nearer than that the ReLU branch, and with it a whole
    element of the backward pass, is decided by rounding in any fp32 implementation (the kernels' pre-activations are within
    ~5e-8 of the fp64 oracle's), so a gradient comparison there measures coin flips, not arithmetic.  How many positions that
    throws away is recorded in _kink_dropped and bounded: a position has ~31,000 pre-activations of magnitude ~0.1-1, so
    ~10-15 % of all positions have one within 5e-7 of zero; more than KINK_DROP_BOUND fails the test."""
    states = _walk_states(N, 4 * B, seed + 100)
    rng = np.random.RandomState(seed)
    if params is not None:
        from oracle import gnn as og
        states = states[rng.choice(states.shape[0], min(2 * B, states.shape[0]), replace=False)]
        drawn = states.shape[0]
        states = states[og.relu_margins(params, states) >= KINK_MARGIN]
        _kink_dropped[(B, seed, N)] = (drawn, drawn - states.shape[0])
        assert drawn - states.shape[0] <= KINK_DROP_BOUND * drawn, f"kink filter dropped {drawn - states.shape[0]} of {drawn} positions"
        assert states.shape[0] >= B
    A = N * N + 2 * (N - 1) ** 2
    recs = states[rng.choice(states.shape[0], B, replace=False)]
    pi = rng.rand(B, A) * (rng.rand(B, A) < 0.2)
    pi[:, 0] += 1e-3
    pi = pi / pi.sum(1, keepdims=True)
    z = rng.choice([-1.0, 0.0, 1.0], B)
    return recs, pi.astype(np.float32), z.astype(np.float32)


TRAIN_FUSED_DEFAULT = 2


@pytest.mark.parametrize("fused", [2, 1, 0, 3])
def test_train_step_gradients_vs_autograd(dev, fused):
    """aqg_gcn_train_step against torch autograd in fp64 (oracle/train.py): forward outputs, both losses and all 14
    gradients.  Stated tolerance: gradients within 2e-5 * max|g| + 1e-7 per tensor (fp32 accumulation over 10,368 nodes),
    losses within 1e-5 relative.  Every form of the step: one workgroup per position with the contractions in fp16 split
    precision on the 16-bit matrix pipe (2, the default on 9x9), the same with f32-input MFMA (1), the six-launch
    column-split chain (0), and the split form with every position sent through its f32 fallback (3)."""
    from alphaquoridorgnn_amd import _lib
    from alphaquoridorgnn_amd.train_network import GNNTrainer
    from oracle import gnn as og, train as ot
    lib = _lib.load()
    _lib.set_option("train_fused", fused)
    model, params = _model(2)
    recs, pi, z = _train_batch(48, 0, params=params)
    tr = GNNTrainer(model, max_batch=64)
    _lib.poison_lds(dev)                   # NaN-fill LDS first (read-before-write defence)
    lib.aqg_gcn_train_fallbacks(1)
    pl, vl = tr.step(torch.from_numpy(recs), torch.from_numpy(pi), torch.from_numpy(z), update=False)
    assert lib.aqg_gcn_train_fallbacks(1) == (48 if fused == 3 else 0)     # ordinary weights never leave fp16 range
    ref = ot.train_steps(params, [(recs, pi.astype(np.float64), z.astype(np.float64))])[0]
    pol, val = tr.outputs(48)
    np.testing.assert_allclose(pol.cpu().numpy(), ref["policy"], atol=1e-6, rtol=1e-4)
    np.testing.assert_allclose(val.cpu().numpy(), ref["value"], atol=1e-5, rtol=1e-4)
    assert abs(float(pl) - ref["policy_loss"]) <= 1e-5 * abs(ref["policy_loss"])
    assert abs(float(vl) - ref["value_loss"]) <= 1e-5 * abs(ref["value_loss"]) + 1e-7
    _lib.set_option("train_fused", TRAIN_FUSED_DEFAULT)
    for k, gt in zip(og.KEYS, tr.grads):
        r = ref["grads"][k]
        tol = 2e-5 * np.abs(r).max() + 1e-7
        assert np.abs(gt.cpu().numpy().astype(np.float64) - r).max() <= tol, k


@pytest.mark.parametrize("fused", [2, 1])
def test_train_step_gradients_at_reference_batch_size(dev, fused):
    """The step where the reference runs it (train_network.py:15 BATCH_SIZE = 128, :49 DataLoader keeps the short last batch, :72-95
    the step): a full batch of 128 and the short batch that follows it in a 165-position epoch (37), both against fp64 autograd
    (oracle/train.py), same bar as the 48-position test -- every gradient within 2e-5 max|g| + 1e-7, losses within 1e-5 -- for the
    split-precision form (2) and the f32-input MFMA form (1), LDS poisoned first.  At 128 the board kernel fills 128 CUs and
    train_final_kernel sums 32 boards per group (12 at batch 48): same code, different fill.  Prints how many drawn positions the
    ReLU-kink filter discarded (bounded by KINK_DROP_BOUND inside _train_batch)."""
    from alphaquoridorgnn_amd import _lib
    from alphaquoridorgnn_amd.train_network import GNNTrainer, BATCH_SIZE
    from oracle import gnn as og, train as ot
    assert BATCH_SIZE == 128
    lib = _lib.load()
    model, params = _model(2)
    tr = GNNTrainer(model, max_batch=BATCH_SIZE)
    for B, seed in ((128, 30), (37, 31)):
        recs, pi, z = _train_batch(B, seed, params=params)
        drawn, dropped = _kink_dropped[(B, seed, 9)]
        print(f"train_fused {fused}, batch {B}: kink filter dropped {dropped} of {drawn} drawn positions ({100.0 * dropped / drawn:.1f} %)")
        _lib.set_option("train_fused", fused)
        _lib.poison_lds(dev)
        lib.aqg_gcn_train_fallbacks(1)
        pl, vl = tr.step(torch.from_numpy(recs), torch.from_numpy(pi), torch.from_numpy(z), update=False)
        _lib.set_option("train_fused", TRAIN_FUSED_DEFAULT)
        assert lib.aqg_gcn_train_fallbacks(1) == 0
        ref = ot.train_steps(params, [(recs, pi.astype(np.float64), z.astype(np.float64))])[0]
        pol, val = tr.outputs(B)
        np.testing.assert_allclose(pol.cpu().numpy(), ref["policy"], atol=1e-6, rtol=1e-4)
        np.testing.assert_allclose(val.cpu().numpy(), ref["value"], atol=1e-5, rtol=1e-4)
        assert abs(float(pl) - ref["policy_loss"]) <= 1e-5 * abs(ref["policy_loss"])
        assert abs(float(vl) - ref["value_loss"]) <= 1e-5 * abs(ref["value_loss"]) + 1e-7
        for k, gt in zip(og.KEYS, tr.grads):
            r = ref["grads"][k]
            assert np.abs(gt.cpu().numpy().astype(np.float64) - r).max() <= 2e-5 * np.abs(r).max() + 1e-7, (B, k)


@pytest.mark.parametrize("N", [3, 5, 7])
def test_train_step_gradients_small_boards(dev, N):
    """The same gradient parity on the reference's smaller boards (constants.py:5-20) and on 7x7: 9 / 25 / 49 nodes = 1 / 2 / 4
    row tiles of the per-board kernels, their own policy sizes; both forms of the step, LDS poisoned first (the padding-row bug
    of round 2 depended on the board size)."""
    from alphaquoridorgnn_amd import _lib
    from alphaquoridorgnn_amd.pv_network_gnn import GraphPolicyValueNetwork
    from alphaquoridorgnn_amd.train_network import GNNTrainer
    from oracle import gnn as og, train as ot
    A = N * N + 2 * (N - 1) ** 2
    params = og.init_params(4, N=N)
    model = GraphPolicyValueNetwork(policy_output_size=A, board_size=N)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    model = model.to(dev)
    recs, pi, z = _train_batch(40, 2, N=N, params=params)
    ref = ot.train_steps(params, [(recs, pi.astype(np.float64), z.astype(np.float64))])[0]
    for fused in (1, 0):
        _lib.set_option("train_fused", fused)
        tr = GNNTrainer(model, max_batch=64)
        _lib.poison_lds(dev)               # NaN-fill LDS: padding rows a kernel forgot to clear poison the weight gradients
        pl, vl = tr.step(torch.from_numpy(recs), torch.from_numpy(pi), torch.from_numpy(z), update=False)
        _lib.set_option("train_fused", TRAIN_FUSED_DEFAULT)
        assert abs(float(pl) - ref["policy_loss"]) <= 1e-5 * abs(ref["policy_loss"])
        assert abs(float(vl) - ref["value_loss"]) <= 1e-5 * abs(ref["value_loss"]) + 1e-7
        for k, gt in zip(og.KEYS, tr.grads):
            r = ref["grads"][k]
            assert np.abs(gt.cpu().numpy().astype(np.float64) - r).max() <= 2e-5 * np.abs(r).max() + 1e-7, (N, fused, k)


def test_train_split_step_falls_back_to_f32_out_of_fp16_range(dev):
    """The split-precision step range-checks every value it is about to split; a position that meets |x| > 65504 is redone by the
    f32 body inside the same launch.  Weights the reference's fp32 handles without blinking -- gcn_layers.1 scaled by 1e6 (its
    entries alone leave fp16 range), the heads' first layers by 1e-6 so that losses and gradients stay ordinary -- must give, for
    EVERY output, bit for bit what the f32 form gives, with every position counted as a fallback."""
    from alphaquoridorgnn_amd import _lib
    from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
    from alphaquoridorgnn_amd.train_network import GNNTrainer
    from oracle import gnn as og
    lib = _lib.load()
    recs, pi, z = _train_batch(40, 5)
    for case in ("whole layer", "one output feature"):
        params = og.init_params(12)
        if case == "whole layer":
            params["gcn_layers.1.lin.weight"] = params["gcn_layers.1.lin.weight"] * np.float32(1e6)
            for k in ("policy_head.0.weight", "value_head.0.weight"):
                params[k] = params[k] * np.float32(1e-6)
        else:                                              # a single row (feature 5 of layer 2): its column alone leaves the range
            params["gcn_layers.1.lin.weight"][5, :] *= np.float32(3e6)
            params["gcn_layers.2.lin.weight"][:, 5] *= np.float32(1e-6)
        _check_split_fallback(dev, params, recs, pi, z)


def _check_split_fallback(dev, params, recs, pi, z):
    from alphaquoridorgnn_amd import _lib
    from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
    from alphaquoridorgnn_amd.train_network import GNNTrainer
    lib = _lib.load()
    outs = {}
    for fused in (1, 2):
        model = GNNNetwork()
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
        model = model.to(dev)
        _lib.set_option("train_fused", fused)
        tr = GNNTrainer(model, max_batch=64)
        lib.aqg_gcn_train_fallbacks(1)
        pl, vl = tr.step(torch.from_numpy(recs), torch.from_numpy(pi), torch.from_numpy(z), update=False)
        pol, val = tr.outputs(40)
        outs[fused] = [float(pl), float(vl), pol.cpu().numpy().copy(), val.cpu().numpy().copy()] + [g.cpu().numpy().copy() for g in tr.grads]
        assert lib.aqg_gcn_train_fallbacks(1) == (40 if fused == 2 else 0)
    _lib.set_option("train_fused", TRAIN_FUSED_DEFAULT)
    assert all(np.isfinite(x).all() for x in outs[2][2:])
    assert max(np.abs(g).max() for g in outs[2][4:10]) > 0          # the trunk's gradients are not degenerate
    for a, b in zip(outs[1], outs[2]):
        assert np.array_equal(np.asarray(a), np.asarray(b))


def test_train_adam_steps_vs_torch(dev):
    """Three Adam steps (LambdaLR factors 1.0, 0.5, 0.25 as at epochs 0 / 50 / 80) from the same start: parameters after
    each step against torch.optim.Adam driven by fp64 autograd.  Adam's update lr * m / (sqrt(v) + eps) is ~lr for EVERY
    element whose gradient is well above eps = 1e-8 and ill-conditioned below that, so the tight tolerance (1e-5 absolute
    = 1 % of updates of ~1e-3) is asserted on the elements whose reference gradient stayed above 1e-3 of the tensor's largest
    in every step so far; all elements must stay within 25 % of the learning rate."""
    from alphaquoridorgnn_amd.train_network import GNNTrainer, LEARNING_RATE, lr_lambda
    from oracle import gnn as og, train as ot
    model, params = _model(6)
    epochs = [0, 50, 80]
    batches, ref = [], None
    for i in range(3):         # every batch is drawn away from the ReLU kinks of the weights it will meet (_train_batch)
        batches.append(_train_batch(32, 10 + i, params=params if i == 0 else ref[i - 1]["params_after"]))
        ref = ot.train_steps(params, [(r, p.astype(np.float64), zz.astype(np.float64)) for r, p, zz in batches], epoch_of_step=epochs[:i + 1])
    tr = GNNTrainer(model, max_batch=32)
    well = {k: np.ones_like(ref[0]["grads"][k], dtype=bool) for k in og.KEYS}
    checked = 0
    for i, (recs, pi, z) in enumerate(batches):
        tr.step(torch.from_numpy(recs), torch.from_numpy(pi), torch.from_numpy(z), lr=LEARNING_RATE * lr_lambda(epochs[i]))
        sd = model.state_dict()
        for k in og.KEYS:
            gref = np.abs(ref[i]["grads"][k])
            well[k] &= gref >= 1e-3 * gref.max()
            d = np.abs(sd[k].cpu().numpy().astype(np.float64) - ref[i]["params_after"][k])
            assert d.max() <= 0.25 * LEARNING_RATE, (i, k)
            if well[k].any():
                assert d[well[k]].max() <= 1e-5, (i, k, float(d[well[k]].max()))
                checked += int(well[k].sum())
    assert checked > 50000
    # the inference path sees the updated weights (packed copy refreshed)
    pol, _ = model.forward_states(torch.from_numpy(batches[0][0]).to(dev))
    chk = og.forward_states({k: v for k, v in ref[2]["params_after"].items()}, batches[0][0])
    np.testing.assert_allclose(pol.cpu().numpy(), chk["policy"], atol=2e-6, rtol=1e-3)


def test_train_run_epoch_equals_single_steps(dev):
    """aqg_gcn_train_steps (a whole epoch in one call: on-device gather by the shuffled order, short last batch kept,
    losses summed on the device) takes bit-identically the steps GNNTrainer.step takes on the same batches -- on the 9x9
    board and, for the odd tile counts, on 5x5 (25 nodes = 2 row tiles) with its own parameter shapes."""
    from alphaquoridorgnn_amd.train_network import GNNTrainer
    from alphaquoridorgnn_amd.pv_network_gnn import GraphPolicyValueNetwork
    for N, n, batch, pre_shuffle in ((9, 150, 64, True), (9, 150, 64, False), (5, 70, 32, False)):
        A = N * N + 2 * (N - 1) ** 2
        recs, pi, z = _train_batch(n, 21, N=N)
        order = torch.from_numpy(np.random.RandomState(3).permutation(n))
        S, P, Z = torch.from_numpy(recs).to(dev), torch.from_numpy(pi).to(dev), torch.from_numpy(z).to(dev)
        torch.manual_seed(5)
        ma = GraphPolicyValueNetwork(policy_output_size=A, board_size=N).to(dev)
        mb = GraphPolicyValueNetwork(policy_output_size=A, board_size=N).to(dev)
        mb.load_state_dict(ma.state_dict())
        ta, tb = GNNTrainer(ma, max_batch=batch), GNNTrainer(mb, max_batch=batch)
        sums = ta.run_epoch(S, P, Z, order, lr=7e-4, pre_shuffle=pre_shuffle)  # both ways of applying the order
        ref = torch.zeros(2, device=dev)
        for i in range(0, n, batch):
            idx = order[i:i + batch].to(dev)
            pl, vl = tb.step(S[idx], P[idx], Z[idx], lr=7e-4)
            ref += torch.stack([pl, vl])
        assert ta.step_count == tb.step_count == (n + batch - 1) // batch
        for (k, a), b in zip(ma.state_dict().items(), mb.state_dict().values()):
            assert torch.equal(a, b), (N, k)
        np.testing.assert_allclose(sums.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5)
        assert all(torch.isfinite(v).all() for v in ma.state_dict().values())


_DP_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["AQG_REPO"])
import numpy as np, torch, torch.distributed as dist
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
from alphaquoridorgnn_amd.train_network import GNNTrainer
from oracle import gnn as og
rank = int(os.environ["RANK"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + os.environ["AQG_PORT"], rank=rank, world_size=2)
d = np.load(os.environ["AQG_DATA"])
m = GNNNetwork(); m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in og.init_params(8).items()}); m = m.to("cuda")
tr = GNNTrainer(m, max_batch=32)
losses = []
for i in range(3):
    sl = slice(rank, None, 2) if i < 2 else (slice(0, None) if rank == 0 else slice(0, 0))   # step 2: rank 1 holds nothing
    pl, vl = tr.step(torch.from_numpy(d[f"s{i}"][sl]), torch.from_numpy(d[f"p{i}"][sl]), torch.from_numpy(d[f"z{i}"][sl]))
    losses.append([float(pl), float(vl)])
sd = m.state_dict()
np.savez(os.environ["AQG_OUT"] + f".{rank}.npz", losses=np.asarray(losses), **{k: sd[k].cpu().numpy() for k in og.KEYS})
dist.destroy_process_group()
'''


def test_train_data_parallel_equals_single_process(dev, tmp_path):
    """Data-parallel training step (GNNTrainer.step under torch.distributed): two ranks, each with its share of every
    batch (incl. a ragged batch in which one rank holds nothing), one gradient all-reduce per step == the single-process
    step on the whole batch.  gloo on one GPU here (RCCL needs one GPU per rank); the exchange code path is the same."""
    import subprocess
    from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
    from alphaquoridorgnn_amd.train_network import GNNTrainer
    from oracle import gnn as og
    batches = [_train_batch(32, 30 + i) for i in range(3)]
    data = {}
    for i, (r, p, z) in enumerate(batches):
        data[f"s{i}"], data[f"p{i}"], data[f"z{i}"] = r, p, z
    np.savez(tmp_path / "data.npz", **data)
    (tmp_path / "worker.py").write_text(_DP_WORKER)
    env = dict(os.environ, AQG_REPO=REPO, AQG_PORT=str(29600 + os.getpid() % 300), AQG_DATA=str(tmp_path / "data.npz"),
               AQG_OUT=str(tmp_path / "out"))
    procs = [subprocess.Popen([sys.executable, str(tmp_path / "worker.py")], env=dict(env, RANK=str(r))) for r in range(2)]
    assert all(p.wait(timeout=300) == 0 for p in procs)
    m = GNNNetwork(); m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in og.init_params(8).items()}); m = m.to(dev)
    tr = GNNTrainer(m, max_batch=32)
    ref_losses = []
    for r, p, z in batches:
        pl, vl = tr.step(torch.from_numpy(r), torch.from_numpy(p), torch.from_numpy(z))
        ref_losses.append([float(pl), float(vl)])
    sd = m.state_dict()
    outs = [np.load(str(tmp_path / "out") + f".{r}.npz") for r in range(2)]
    for k in og.KEYS:
        assert np.array_equal(outs[0][k], outs[1][k]), k                       # replicas stay identical
        d = np.abs(outs[0][k].astype(np.float64) - sd[k].cpu().numpy().astype(np.float64))
        assert d.max() <= 0.25 * 1e-3, k                                        # Adam is ill-conditioned where |g| ~ eps ...
        assert np.quantile(d, 0.99) <= 1e-5, (k, float(np.quantile(d, 0.99)))   # ... everywhere else the steps coincide
    np.testing.assert_allclose(outs[0]["losses"], np.asarray(ref_losses), rtol=1e-5, atol=1e-7)
    assert np.array_equal(outs[0]["losses"], outs[1]["losses"])


def test_train_network_end_to_end(dev, tmp_path, monkeypatch):
    """train_network() on a .history written by this build's self_play: best.pth -> latest.pth, loss goes down."""
    from alphaquoridorgnn_amd import train_network as tn, self_play, pv_mcts
    from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
    monkeypatch.chdir(tmp_path)
    path = str(tmp_path) + "/"
    monkeypatch.setattr(tn, "PV_NETWORK_PATH", path)
    monkeypatch.setattr(tn, "NUM_EPOCH", 4)
    monkeypatch.setattr(pv_mcts, "PV_EVALUATE_COUNT", 8)
    torch.manual_seed(3)
    model = GNNNetwork()
    torch.save(model.state_dict(), path + "best.pth")
    self_play.self_play(model.to(dev), games=6)                     # writes ./data/*.history
    hist = tn.load_data()
    assert len(hist) > 20 and len(hist[0][1]) == 209
    tn.train_network()
    new = torch.load(path + "latest.pth", map_location="cpu", weights_only=True)
    old = torch.load(path + "best.pth", map_location="cpu", weights_only=True)
    assert set(new) == set(old) and any(not torch.equal(new[k], old[k]) for k in new)
    assert all(torch.isfinite(v).all() for v in new.values())


def test_train_cycle_one_iteration(dev, tmp_path, monkeypatch):
    """train_cycle.py:22-41 end to end in a scratch directory: create_network -> self_play -> train_network ->
    evaluate_network, every stage on the GPU path, files in the reference's places."""
    from alphaquoridorgnn_amd import train_cycle as tc, self_play as sp, train_network as tn, evaluate_network as en, pv_mcts
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(pv_mcts, "PV_EVALUATE_COUNT", 6)
    monkeypatch.setattr(sp, "SP_GAME_COUNT", 5)
    monkeypatch.setattr(tn, "NUM_EPOCH", 2)
    monkeypatch.setattr(en, "EN_GAME_COUNT", 4)
    promoted = tc.train_cycle(num_cycles=1)
    assert len(promoted) == 1 and isinstance(promoted[0], bool)
    from alphaquoridorgnn_amd.constants import PV_NETWORK_PATH
    assert os.path.exists(PV_NETWORK_PATH + "best.pth") and os.path.exists(PV_NETWORK_PATH + "latest.pth")
    assert len(list((tmp_path / "data").glob("*.history"))) == 1


_CYCLE_5X5 = r'''
import os, sys
sys.path.insert(0, os.environ["AQG_REPO"])
from alphaquoridorgnn_amd import constants, pv_mcts, self_play as sp, train_network as tn, evaluate_network as en, train_cycle as tc
assert constants.BOARD_SIZE == 5 and constants.NUM_WALLS == 2 and constants.NUM_PLIES_FOR_DRAW == 28
pv_mcts.PV_EVALUATE_COUNT = 8
sp.SP_GAME_COUNT = 12
tn.NUM_EPOCH = 2
en.EN_GAME_COUNT = 4
print("PROMOTED", tc.train_cycle(num_cycles=1))
'''


def test_train_cycle_on_5x5_board(dev, tmp_path):
    """The whole learning loop with the module constants switched to the reference's 5x5 block (AQG_BOARD_SIZE=5, the
    counterpart of editing constants.py:13-16): self-play, training and evaluation all run the GNN on the small board."""
    import subprocess
    (tmp_path / "cycle.py").write_text(_CYCLE_5X5)
    env = dict(os.environ, AQG_REPO=REPO, AQG_BOARD_SIZE="5")
    r = subprocess.run([sys.executable, str(tmp_path / "cycle.py")], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "PROMOTED [" in r.stdout
    assert (tmp_path / "models" / "GNN" / "5x5" / "latest.pth").exists() and len(list((tmp_path / "data").glob("*.history"))) == 1


def test_multiset_engines_share_one_set_of_streams(dev):
    """Every MultiSetSelfPlay of a process runs on the same K streams (engine._SET_STREAMS): torch's stream pool is never
    destroyed and the runtime maps streams onto 8 hardware queues, so a second engine on NEW streams would share queues with
    the first one's idle streams and serialise two of its game sets (measured: 35 % slower generations from the second
    self_play() of a train_cycle on).  Sharing the streams must not couple the engines' results."""
    from alphaquoridorgnn_amd.engine import MultiSetSelfPlay
    a = MultiSetSelfPlay(None, num_games=12, sims=8, num_sets=4, seed=3, evaluator="fake", fake_bias=5)
    b = MultiSetSelfPlay(None, num_games=12, sims=8, num_sets=4, seed=3, evaluator="fake", fake_bias=5)
    assert all(x is y for x, y in zip(a.streams, b.streams)) and len({s.cuda_stream for s in a.streams}) == 4
    for _ in range(3):                     # interleaved moves of two live engines on the shared streams
        a.move(); b.move()
    ra, rb = a.play_generation(), b.play_generation()
    assert ra["finished"] == rb["finished"] == 12
    for x, y in zip(a.history_tensors(), b.history_tensors()):
        assert torch.equal(x, y)


def test_self_play_generations_differ_unless_seeded(dev, tmp_path, monkeypatch):
    """The reference samples from the unseeded global numpy RNG (self_play.py:57): two generations with unchanged weights
    never repeat.  Same here -- and a fixed seed (argument or AQG_SELFPLAY_SEED) reproduces a generation exactly."""
    import pickle
    from alphaquoridorgnn_amd import self_play as sp, pv_mcts
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(pv_mcts, "PV_EVALUATE_COUNT", 8)
    monkeypatch.setattr(sp, "write_data", lambda h: h)                  # keep the rows, skip the one-file-per-second naming
    model, _ = _model(1)
    a = sp.self_play(model, games=6)
    b = sp.self_play(model, games=6)
    assert a != b, "two unseeded generations replayed the same uniform stream"
    c = sp.self_play(model, games=6, seed=77)
    d = sp.self_play(model, games=6, seed=77)
    assert c == d
    monkeypatch.setenv("AQG_SELFPLAY_SEED", "77")
    assert sp.self_play(model, games=6) == c


def _torchrun_two_ranks(tmp_path, module, *args, **extra_env):
    """Two ranks through the package's own multi-rank entry, exactly as INTEGRATION.md launches it (torch.distributed.run with one
    process per rank; here backend gloo and both ranks on this one GPU -- RCCL needs a GPU per rank, the host code path is the
    same): `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P -m <module> ...`."""
    import subprocess
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""), AQG_DIST_BACKEND="gloo", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29900 + os.getpid() % 90), "-m", module] + [str(a) for a in args]
    res = subprocess.run(cmd, env=env, cwd=str(tmp_path), timeout=900, capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]


def _write_best(tmp_path, seed):
    from oracle import gnn as og
    d = tmp_path / "models" / "GNN" / "9x9"
    d.mkdir(parents=True, exist_ok=True)
    torch.save({k: torch.from_numpy(v.copy()) for k, v in og.init_params(seed).items()}, str(d / "best.pth"))


def test_sharded_self_play_two_ranks_equals_standalone_engines(dev, tmp_path):
    """BASELINE configs[3] on two ranks (gloo, both on this one GPU -- RCCL needs a GPU per rank; the host code path is the
    same): self_play(games=7) shards 4 + 3, the ranks exchange their (s, pi, z) rows once, rank 0 writes ONE history file.
    That file must be the concatenation, in rank order, of what two stand-alone engines with the ranks' seeds produce."""
    import pickle
    from alphaquoridorgnn_amd.engine import MultiSetSelfPlay
    from alphaquoridorgnn_amd import self_play as sp
    _write_best(tmp_path, 2)
    _torchrun_two_ranks(tmp_path, "alphaquoridorgnn_amd.self_play", "--games", 7, "--sims", 8, "--seed", 500)   # rank 0: 4 games, rank 1: 3
    files = sorted((tmp_path / "data").glob("*.history"))
    assert len(files) == 1
    with open(files[0], "rb") as f:
        rows = pickle.load(f)
    model, _ = _model(2)
    want = []
    for rank, games in ((0, 4), (1, 3)):
        eng = MultiSetSelfPlay(model, num_games=games, sims=8, num_sets=1, seed=500 + rank)
        eng.play_generation()
        want += sp._history_rows(*eng.history_tensors(), 9)
    assert len(rows) == len(want) and rows == want


def test_rccl_world_size_1_self_play_equals_no_group(dev, tmp_path):
    """The RCCL path on the one GPU this pool gives a builder: `-m alphaquoridorgnn_amd.self_play` started through
    torch.distributed.run with ONE rank and AQG_DIST_FORCE_GROUP=1 -- distributed.init_from_env creates the backend-`nccl` process
    group (device_id = this GPU), engine.gather_history runs its count all-gather and its all_gather_into_tensor on DEVICE tensors
    through the RCCL communicator (and self_play's barrier after it) -- must write exactly the rows the group-free run writes with
    the same seed: 300 games on the four game-set streams (the configuration in which an extra stream could collide with the sets')."""
    import pickle
    import subprocess
    import torch.distributed as dist
    if not dist.is_nccl_available():
        pytest.skip("torch built without nccl/RCCL")
    _write_best(tmp_path, 2)
    files = {}
    for mode in ("nccl_ws1", "no_group"):
        wd = tmp_path / mode
        wd.mkdir()
        (wd / "models").symlink_to(tmp_path / "models")
        env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
        env.pop("AQG_DIST_BACKEND", None)
        if mode == "nccl_ws1":
            env["AQG_DIST_FORCE_GROUP"] = "1"
            env["AQG_DIST_LOG"] = "1"
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                   "--master-port", str(29800 + os.getpid() % 90), "-m", "alphaquoridorgnn_amd.self_play"]
        else:
            cmd = [sys.executable, "-m", "alphaquoridorgnn_amd.self_play"]
        res = subprocess.run(cmd + ["--games", "300", "--sims", "8", "--seed", "500"], env=env, cwd=str(wd), timeout=900,
                             capture_output=True, text=True)
        assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
        if mode == "nccl_ws1":
            assert "backend=nccl world=1 collectives=forced" in res.stdout, res.stdout[-2000:]
        fs = sorted((wd / "data").glob("*.history"))
        assert len(fs) == 1
        with open(fs[0], "rb") as f:
            files[mode] = pickle.load(f)
    assert len(files["nccl_ws1"]) > 300 * 8
    assert files["nccl_ws1"] == files["no_group"]


@pytest.mark.parametrize("data_parallel", ["0", "1"])
def test_train_cycle_two_ranks(dev, tmp_path, data_parallel):
    """Two whole cycles under torch.distributed (2 ranks, gloo): sharded self-play, training on the file rank 0 wrote (nobody
    reads it early) -- by rank 0 alone (the default: the batch-128 step is latency-bound, sharding it cannot speed it up) or
    data-parallel (AQG_TRAIN_DATA_PARALLEL=1) --, evaluation by rank 0 with the decision broadcast.  Both ranks must agree on
    the promotions and hold identical weights; every cycle leaves one history file."""
    import json
    _torchrun_two_ranks(tmp_path, "alphaquoridorgnn_amd.train_cycle", "--cycles", 2, "--games", 6, "--sims", 8, "--epochs", 2,
                        "--eval-games", 4, "--result-dir", ".", AQG_TRAIN_DATA_PARALLEL=data_parallel)
    outs = []
    for r in range(2):
        with open(tmp_path / f"train_cycle.rank{r}.json") as f:
            outs.append(json.load(f))
    assert [o["rank"] for o in outs] == [0, 1] and all(o["world"] == 2 for o in outs)
    assert outs[0]["promoted"] == outs[1]["promoted"] and len(outs[0]["promoted"]) == 2
    assert outs[0]["latest_sha256"] == outs[1]["latest_sha256"]
    # one history file per generation, named by the second like the reference's (self_play.py:33-35): two of these tiny
    # generations can share a name
    assert 1 <= len(list((tmp_path / "data").glob("*.history"))) <= 2


# ------------------------------------------------------------------ external evaluator (any BaseNetwork-style model)
class _OracleFakeAdapter:
    """oracle.mcts.FakeModel behind the reference's predict(state, device) contract, fed with OUR game_logic.State."""

    def __init__(self, bias):
        from oracle import mcts as om
        self.m = om.FakeModel(bias)

    def predict(self, state, device=None):
        from oracle import quoridor as oq
        return self.m.predict(oq.State(state.record()))


def test_external_evaluator_reproduces_reference_traces(dev):
    """prior_mode 2: the engine selects / expands / backs up, the CALLER's model.predict (BaseNetwork.py:36-40, called once
    per simulation like pv_mcts.py:47) supplies priors and value from the host.  Driven by the integer-hash model it must
    give the reference's visit distributions bit for bit, like the in-kernel fake evaluator does."""
    from alphaquoridorgnn_amd.pv_mcts import pv_mcts_policy_batch
    g = U.golden("mcts_9x9.npz")
    n = int(g["count"][0])
    done = 0
    for k in range(n):
        sims, bias, T = g[f"t{k}_cfg"]
        if int(sims) != 50 or done >= 6:
            continue
        pol = pv_mcts_policy_batch(_OracleFakeAdapter(int(bias)), g[f"t{k}_state"][None], float(T), sims=50, board_size=9,
                                   evaluator="external")[0]
        assert np.array_equal(np.asarray(pol, dtype=np.float64), g[f"t{k}_policy"]), k
        done += 1
    assert done == 6
    # a whole reference game (self_play.play) through the external path: same states, policies and z
    gg = U.golden("games_9x9.npz")
    from alphaquoridorgnn_amd.engine import BatchedSelfPlay
    seed, sims, bias = (int(x) for x in gg["g2_cfg"])
    rs = np.random.RandomState(seed)
    eng = BatchedSelfPlay(_OracleFakeAdapter(bias), num_games=1, sims=sims, evaluator="external")
    uni = torch.from_numpy(rs.random_sample(116).reshape(116, 1))
    eng.play_generation(uniforms=uni, check_every=1)
    h = eng.history()
    assert len(h) == gg["g2_states"].shape[0]
    for i, (sa, pol, z) in enumerate(h):
        assert np.array_equal(np.asarray(pol), gg["g2_policy"][i]) and z == int(gg["g2_z"][i])
        assert sa[0] == [int(x) for x in gg["g2_states"][i, 0:2]] and sa[2] == [int(x) for x in gg["g2_states"][i, 4:68]]


class _StockCNN(torch.nn.Module):
    """A stock-PyTorch policy-value CNN of the shape the reference wires (pv_network_cnn.py:50-86: 3x3 conv + BN stem, 16
    residual blocks of 128 filters, global average pool, Linear->Softmax policy over 209 actions, Linear->Tanh value) with
    the BaseNetwork contract (name / predict / preprocess_input).  Own code, random weights, CPU: plumbing only."""

    def __init__(self, filters=128, blocks=16, board=9):
        super().__init__()
        nn = torch.nn
        self.board, self.actions = board, board * board + 2 * (board - 1) ** 2

        def unit(cin):
            return nn.Sequential(nn.Conv2d(cin, filters, 3, padding=1, bias=False), nn.BatchNorm2d(filters))
        self.stem = unit(6)
        self.tower = nn.ModuleList([nn.ModuleList([unit(filters), unit(filters)]) for _ in range(blocks)])
        self.policy = nn.Linear(filters, self.actions)
        self.value = nn.Linear(filters, 1)
        self.name = "CNN"

    def forward(self, x):
        x = torch.relu(self.stem(x))
        for a, b in self.tower:
            x = torch.relu(b(torch.relu(a(x))) + x)
        x = x.mean(dim=(2, 3))
        return torch.softmax(self.policy(x), dim=1), torch.tanh(self.value(x))

    def preprocess_input(self, arrays):
        N = self.board
        out = np.zeros((len(arrays), 6, N, N), dtype=np.float32)
        for i, (player, enemy, walls) in enumerate(arrays):
            out[i, 0].flat[player[0]] = 1; out[i, 1] = player[1]
            out[i, 2].flat[enemy[0]] = 1; out[i, 3] = enemy[1]
            for slot, w in enumerate(walls):
                if w:
                    out[i, 3 + w, slot // (N - 1), slot % (N - 1)] = 1
        return out

    def predict(self, state, device="cpu"):
        x = torch.from_numpy(self.preprocess_input([state.to_array()]))
        with torch.inference_mode():
            p, v = self(x)
        p = p[0][list(state.legal_actions())]
        p = p / (p.sum() if p.sum() else 1)
        return p.numpy(), float(v.item())


def test_play_with_stock_cnn_is_plumbing_compatible(dev, monkeypatch):
    """BASELINE configs[0]: one self_play.play() game with a CNN of the reference's shape on the CPU.  The model is NOT ours
    (no packed_weights): pv_mcts / self_play must accept anything with the reference's predict() and run the search on the
    engine with the model as external evaluator."""
    from alphaquoridorgnn_amd import pv_mcts, self_play
    from alphaquoridorgnn_amd.game_logic import State
    torch.manual_seed(0)
    model = _StockCNN().eval()
    assert sum(p.numel() for p in model.parameters()) > 4_000_000            # the 4.76 M-parameter tower of the reference
    monkeypatch.setattr(pv_mcts, "PV_EVALUATE_COUNT", 4)
    s = State()
    pol = pv_mcts.pv_mcts_policy(model, s, 1.0, "cpu")
    assert len(pol) == len(s.legal_actions()) == 131 and abs(sum(pol) - 1.0) < 1e-9
    a = pv_mcts.pv_mcts_action(model, 1.0, "cpu")(s)
    assert a in s.legal_actions()
    hist = self_play.play(model, "cpu")
    assert 8 <= len(hist) <= 116 and len(hist[0][1]) == 209 and hist[0][0] == State().to_array()
    zs = [h[2] for h in hist]
    assert all(zs[i + 1] == -zs[i] for i in range(len(zs) - 1)) and zs[0] in (-1, 0, 1)


# ------------------------------------------------------------------ drop-in surface (reference-shaped calls)
def test_dropin_surface_play_and_policy(dev, tmp_path, monkeypatch):
    """The reference's call surface end to end on the GPU: pv_mcts_policy / pv_mcts_action on a State, self_play.play()
    history schema (self_play.py:51-54,:63-66), write_data() pickle (self_play.py:30-37), create_network()."""
    import pickle
    from alphaquoridorgnn_amd import pv_mcts, self_play, pv_network_gnn
    from alphaquoridorgnn_amd.game_logic import State
    model, _ = _model(0)
    monkeypatch.setattr(pv_mcts, "PV_EVALUATE_COUNT", 16)
    s = State()
    la = s.legal_actions()
    pol = pv_mcts.pv_mcts_policy(model, s, 1.0, "cuda")
    assert len(pol) == len(la) and abs(sum(pol) - 1.0) < 1e-12 and min(pol) >= 0
    assert round(sum(p * 15 for p in pol)) == 15                      # visit counts sum to sims - 1
    onehot = pv_mcts.pv_mcts_policy(model, s, 0, "cuda")
    assert sorted(onehot)[-1] == 1 and sum(onehot) == 1
    a = pv_mcts.pv_mcts_action(model, 1.0, "cuda")(s)
    assert int(a) in la
    monkeypatch.setattr(pv_mcts, "PV_EVALUATE_COUNT", 6)
    hist = self_play.play(model, "cuda")
    assert 1 <= len(hist) <= 116
    st0, pi0, z0 = hist[0]
    assert st0 == State().to_array() and len(pi0) == 209 and abs(sum(pi0) - 1.0) < 1e-12 and z0 in (-1, 0, 1)
    assert all(h[2] == (z0 if i % 2 == 0 else -z0) for i, h in enumerate(hist))
    monkeypatch.chdir(tmp_path)
    path = self_play.write_data(hist)
    with open(path, "rb") as f:                                       # our own file: plain pickle of python lists
        back = pickle.load(f)
    assert back == hist
    monkeypatch.setattr(pv_network_gnn, "PV_NETWORK_PATH", str(tmp_path / "models/GNN/9x9") + "/")
    pv_network_gnn.create_network()
    net = pv_network_gnn.GNNNetwork()
    net.prep_for_inference(str(tmp_path / "models/GNN/9x9/best.pth"))
    p, v = net.predict(State(), "cuda")
    assert p.shape == (131,) and abs(float(p.sum()) - 1) < 1e-5 and -1 <= v <= 1
    assert net.name == "GNN" and net.preprocess_input([State().to_array()]).shape == (1, 72)


def test_bench_two_ranks_on_one_gpu(dev, tmp_path):
    """The N > 1 path of bench.py (what the driver launches with torchrun, backend nccl = RCCL) rehearsed with two ranks sharing
    this GPU over gloo: every rank plays its own games, the (s, pi, z) rows are all-gathered, rank 0 prints ONE JSON line whose
    value counts the games of both ranks."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29500 + (os.getpid() % 400)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--games", "64", "--sims", "16", "--sets", "2", "--backend", "gloo", "--no-extra-legs", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["unit"] == "games/s" and d["steps"] == 1
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 128) < 1e-3          # 2 ranks x 64 games, all finished (the line carries six significant digits)
    assert len(lines[0]) < 6000
    assert d["positions_gathered_per_step"] > 128 and d["roofline"]["launches"] > 0


def test_graph_cache_eviction_keeps_results(dev):
    """A move's 3 x sims + 2 launches are replayed from a captured hipGraph, cached per engine (csrc/mcts.hip run_sims: 64 entries,
    FIFO).  More engines than entries alternate on a side stream, so execs are evicted (waiting for THEIR last replay only) and
    re-captured: every engine's games must still equal the same engine played with plain launches."""
    from alphaquoridorgnn_amd import _lib
    from alphaquoridorgnn_amd.engine import BatchedSelfPlay
    side = torch.cuda.Stream(device=dev)
    n = 70

    def run(use_graph):
        _lib.set_option("use_graph", use_graph)
        hist = []
        with torch.cuda.stream(side):
            engines = [BatchedSelfPlay(None, num_games=2, sims=6, evaluator="fake", fake_bias=3, seed=100 + i) for i in range(n)]
            for rnd in range(3):                       # round-robin: engine 0's graph is long evicted when its turn comes again
                for e in engines:
                    e.move()
            side.synchronize()
            for e in engines:
                hist.append(tuple(x.cpu() for x in (e.t["hist_action"][:, :3], e.t["root_state"])))
        return hist

    try:
        a, b = run(1), run(0)
    finally:
        _lib.set_option("use_graph", 1)
    for (ha, ra), (hb, rb) in zip(a, b):
        assert torch.equal(ha, hb) and torch.equal(ra, rb)


def test_generation_replays_on_exact_kernels_after_range_guard(dev):
    """The engine side of the fp16-range guard: when the split kernels report a value outside fp16 range during a generation
    (counters[5]; forced here, since positions reached by play look like the calibration boards), play_generation switches the
    engine and the model to the exact f32-input kernels and plays the generation again from the start -- also with several game
    sets -- instead of keeping moves that were searched with clamped evaluations."""
    from alphaquoridorgnn_amd import _lib
    from alphaquoridorgnn_amd.engine import BatchedSelfPlay, MultiSetSelfPlay
    model, _ = _model(5)
    assert not (model.gnn_flags(dev) & _lib.GNN_EXACT_F32)
    eng = BatchedSelfPlay(model, num_games=6, sims=6, seed=3)
    for _ in range(3):
        eng.move()
    eng.t["counters"][5] = 1                      # what a saturating launch would have done
    c = eng.play_generation()
    assert eng.e.gnn_flags == _lib.GNN_EXACT_F32 and model.gnn_flags(dev) == _lib.GNN_EXACT_F32
    assert c["finished"] == 6 and c["active"] == 0 and c["gnn_saturated"] == 0
    st, vis, z = eng.history_tensors()
    assert st.shape[0] == vis.shape[0] == z.shape[0] and (vis.sum(1) == 5).all()
    # several sets: one set's counter is enough
    model2, _ = _model(5)
    ms = MultiSetSelfPlay(model2, num_games=8, sims=6, num_sets=2, seed=4)
    ms.move()
    ms.sets[1].t["counters"][5] = 1
    c = ms.play_generation()
    assert all(e.e.gnn_flags == _lib.GNN_EXACT_F32 for e in ms.sets) and c["finished"] == 8 and c["gnn_saturated"] == 0


def test_search_and_match_replay_on_exact_kernels_after_range_guard(dev):
    """the other two consumers of the engine honour the fp16-range guard the way play_generation does.
    (1) BatchedSelfPlay.search -- the path behind pv_mcts_policy / pv_mcts_action and the drop-in surface: a weight set that passes
    the calibration but leaves fp16 range on a root with 255 walls in hand (test_gnn_runtime_saturation_signal's set A) is searched
    again on the exact f32-input kernels inside the same call: the visit counts equal those of an engine that was exact from the start,
    the model is marked, and a cached engine does not inherit the word.  (2) BatchedMatch.play -- evaluate_network's promotion
    decision: a guard word raised in mid-match (forced after the third ply, positions reached by play look like the calibration
    boards) makes the match switch both players to the exact kernels and replay from ply 0; the points are those of a match that was
    exact from the start (same uniforms)."""
    from alphaquoridorgnn_amd import _lib, pv_mcts
    from alphaquoridorgnn_amd.engine import BatchedSelfPlay
    from alphaquoridorgnn_amd.evaluate_network import BatchedMatch
    from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
    from oracle import gnn as og
    g = U.golden("walk_9x9.npz")
    crafted = g["states"][[10, 400, 3000, 9000]].copy()
    crafted[:, 1] = 255
    params = og.init_params(6)
    params["gcn_layers.0.lin.weight"][:, 1] = 200.0

    def fresh():
        m = GNNNetwork()
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
        return m.to(dev).eval()
    exact = fresh()
    exact.packed_weights(dev)
    exact.mark_saturated(dev)
    want = BatchedSelfPlay(exact, num_games=4, sims=12, record_history=False).search(crafted)
    model = fresh()
    assert model.gnn_flags(dev) == 0
    eng = BatchedSelfPlay(model, num_games=4, sims=12, record_history=False)
    got = eng.search(crafted)
    assert model.gnn_flags(dev) == _lib.GNN_EXACT_F32 and eng.e.gnn_flags == _lib.GNN_EXACT_F32
    assert eng.counters()["gnn_saturated"] == 0
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    # through the reference-shaped surface, on a cached engine
    model = fresh()
    pv_mcts._engines.clear()
    pols = pv_mcts.pv_mcts_policy_batch(model, crafted, 1.0, sims=12, board_size=9)
    assert model.gnn_flags(dev) == _lib.GNN_EXACT_F32
    v, _, cnt = want
    for b in range(4):
        n = int(cnt[b])
        assert np.array_equal(np.asarray(pols[b]), np.asarray(pv_mcts.boltzman([int(x) for x in v[b, :n].cpu()], 1.0)))
    pv_mcts._engines.clear()
    # (2) the match
    rng = np.random.RandomState(4)
    uni = [rng.random_sample((116, 3)), rng.random_sample((116, 3))]
    m0, m1 = _model(7)[0], _model(8)[0]
    for m in (m0, m1):
        m.packed_weights(dev)
        m.mark_saturated(dev)
    want_points = BatchedMatch((m0, m1), 6, sims=6, seed=1).play(uni)
    m0, m1 = _model(7)[0], _model(8)[0]
    match = BatchedMatch((m0, m1), 6, sims=6, seed=1)
    assert not any(f & _lib.GNN_EXACT_F32 for f in match._flags)
    eng0, calls = match.engines[0], [0]
    plain_move = eng0.move

    def move_then_raise_word(u=None):
        plain_move(u)
        calls[0] += 1
        if calls[0] == 3:
            eng0.t["counters"][5] = 1              # what a saturating launch of this ply would have done
    eng0.move = move_then_raise_word
    points = match.play(uni)
    assert calls[0] > 3 + 3                        # three plies, then the whole match again
    assert all(f == _lib.GNN_EXACT_F32 for f in match._flags)
    assert m0.gnn_flags(dev) == _lib.GNN_EXACT_F32 and m1.gnn_flags(dev) == _lib.GNN_EXACT_F32
    assert points == want_points


@pytest.mark.gpu
# ------------------------------------------------------------------ evaluation cache (ABI 10)
def _generation(model, slots, **kw):
    from alphaquoridorgnn_amd.engine import BatchedSelfPlay
    eng = BatchedSelfPlay(model, eval_cache_slots=slots, **kw)
    c = eng.play_generation()
    rows = tuple(x.cpu() for x in eng.history_tensors())
    per_game = tuple(eng.t[k].cpu() for k in ("game_plies", "game_result", "game_slot", "game_first_move", "hist_action"))
    return eng, c, rows + per_game


@pytest.mark.parametrize("N,slots", [(9, 4096), (9, 64), (5, 256)])
def test_eval_cache_generation_bit_identical(dev, N, slots):
    """The evaluation cache (include/aqgnn.h `eval_cache_keys`; the reference re-predicts every leaf, pv_mcts.py:47, and builds a new
    tree per move, :84) must not change one byte of a generation: history rows, results, actions, plies and the number of LOGICAL
    evaluations are those of the cache-less engine -- with a roomy table, with a table so small (64 entries = one probe window) that
    entries are replaced all the time, and on a small board (the any-size forward takes the same mask)."""
    model, _ = _model(2) if N == 9 else (None, None)
    if N != 9:
        from alphaquoridorgnn_amd.pv_network_gnn import GraphPolicyValueNetwork
        from oracle import gnn as og
        model = GraphPolicyValueNetwork(6, 128, 3, N * N + 2 * (N - 1) ** 2, board_size=N)
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in og.init_params(1, N=N).items()})
        model = model.to(dev).eval()
    kw = dict(num_games=40, sims=24, board_size=N, seed=11)
    _, c0, ref = _generation(model, 0, **kw)
    eng, c1, got = _generation(model, slots, **kw)
    assert c0["cache_hits"] == 0 and c1["cache_hits"] > 0.2 * c1["leaf_evals"], (c0, c1)
    for k in ("finished", "leaf_evals", "terminal_sims", "dead_ends"):
        assert c0[k] == c1[k], k
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    # leaf_flag 2 / eval_mask: what the GNN launches of the last simulation were asked to evaluate is a subset of the leaves
    lf, em = eng.t["leaf_flag"].cpu().numpy(), eng.t["eval_mask"].cpu().numpy()
    assert ((em == 1) <= (lf == 1)).all() and not ((lf == 2) & (em != 0)).any()


def test_eval_cache_search_served_from_table_vs_oracle(dev):
    """A search whose every position is already in the table (the same roots searched twice) must still expand its leaves with the
    oracle's priors in legal_actions() order and distribute its visits like oracle.mcts -- the cached row IS the evaluation."""
    from alphaquoridorgnn_amd.engine import BatchedSelfPlay
    from oracle import gnn as og, mcts as om, quoridor as oq
    model, params = _model(4)
    oracle = og.OracleModel(params)
    g = U.golden("walk_9x9.npz")
    recs = np.stack([g["states"][i] for i in [0, 5, 40, 333, 1200, 2600, 5000, 9000]])
    recs = recs[[not (oq.State(r).is_done()) for r in recs]]
    sims = 10
    eng = BatchedSelfPlay(model, num_games=recs.shape[0], sims=sims, record_history=False, eval_cache_slots=256)
    eng.search(recs)
    first = eng.counters()
    eng.search(recs)
    torch.cuda.synchronize()
    second = eng.counters()
    # every evaluation of the second search is a hit (same roots, same deterministic search)
    assert second["cache_hits"] - first["cache_hits"] == second["leaf_evals"] - first["leaf_evals"] > 0
    for rec, (pri, vis, act) in zip(recs, _root_children(eng)):
        st = oq.State(rec)
        assert [int(a) for a in act] == [int(a) for a in st.legal_actions()]
        want, _ = oracle.predict(st)
        np.testing.assert_allclose(pri, want, atol=1e-6, rtol=1e-5)
        root = om.search(oracle, st, sims)
        assert [int(v) for v in vis] == [c.n for c in root.children]


def test_eval_cache_refill_weights_and_range_guard(dev):
    """The table over a slot's lifetime: (1) slot refill -- a slot keeps its table over its games (positions stay valid) and every game
    is the cache-less one; (2) new weights: refresh_weights() empties the table (its rows are the OLD network's outputs) -- a
    generation after the update equals a fresh engine's; (3) the fp16-range guard: the replay on the exact kernels starts from an
    empty table (the clamped evaluations must not survive) and equals the cache-less replay; (4) several game sets."""
    from alphaquoridorgnn_amd import _lib
    from alphaquoridorgnn_amd.engine import BatchedSelfPlay, MultiSetSelfPlay
    model, _ = _model(6)
    kw = dict(num_games=6, quota=15, sims=10, seed=2)
    _, c0, ref = _generation(model, 0, **kw)
    for slots in (512, 64):                       # 64: one probe window per slot, entries of earlier games replaced all the time
        _, c1, got = _generation(model, slots, **kw)
        assert c0["finished"] == c1["finished"] == 15 and c1["cache_hits"] > 0
        for a, b in zip(ref, got):
            assert torch.equal(a, b)
    # (2)
    eng = BatchedSelfPlay(model, num_games=8, sims=10, seed=5, eval_cache_slots=512)
    eng.play_generation()
    with torch.no_grad():
        for prm in model.parameters():
            prm.mul_(1.25)
    model.invalidate_packed() if hasattr(model, "invalidate_packed") else None
    eng.refresh_weights()
    eng.reset()
    eng.gen.manual_seed(5)
    eng.play_generation()
    fresh = BatchedSelfPlay(model, num_games=8, sims=10, seed=5)
    fresh.play_generation()
    for a, b in zip(eng.history_tensors(), fresh.history_tensors()):
        assert torch.equal(a, b)
    # (3) the same forced replay with and without the table (two copies of the weight set: the first replay marks its model)
    hist = []
    for slots in (256, 0):
        model3, _ = _model(5)
        a = BatchedSelfPlay(model3, num_games=6, sims=6, seed=3, eval_cache_slots=slots)
        for _ in range(3):
            a.move()
        a.t["counters"][5] = 1
        ca = a.play_generation()
        assert a.e.gnn_flags == _lib.GNN_EXACT_F32 and ca["finished"] == 6 and (ca["cache_hits"] > 0) == (slots > 0)
        hist.append(a.history_tensors())
    for x, y in zip(*hist):
        assert torch.equal(x, y)
    # (4)
    model4, _ = _model(7)
    m0 = MultiSetSelfPlay(model4, num_games=24, sims=8, num_sets=3, seed=9)
    m1 = MultiSetSelfPlay(model4, num_games=24, sims=8, num_sets=3, seed=9, eval_cache_slots=256)
    c0, c1 = m0.play_generation(), m1.play_generation()
    assert c1["cache_hits"] > 0 and c0["leaf_evals"] == c1["leaf_evals"]
    for x, y in zip(m0.history_tensors(), m1.history_tensors()):
        assert torch.equal(x, y)
    with pytest.raises(ValueError):
        BatchedSelfPlay(None, num_games=2, sims=4, evaluator="fake", eval_cache_slots=64)
    with pytest.raises(ValueError):
        BatchedSelfPlay(model4, num_games=2, sims=4, eval_cache_slots=100)


def test_eval_cache_through_the_program_surface(dev, tmp_path, monkeypatch):
    """AQG_EVAL_CACHE_SLOTS switches the table on for whole programs: self_play() writes the same rows, and pv_mcts_policy -- whose
    engine is kept between calls, so that the table carries the previous moves' evaluations into the next search, the tree reuse the
    reference does not have (pv_mcts.py:84) -- returns the same policies along a game, while evaluate_network's match engines (two
    weight sets taking turns on one engine) stay without a table."""
    from alphaquoridorgnn_amd import self_play as sp, pv_mcts
    from alphaquoridorgnn_amd.evaluate_network import BatchedMatch
    from alphaquoridorgnn_amd.game_logic import State
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(pv_mcts, "PV_EVALUATE_COUNT", 12)
    monkeypatch.setattr(sp, "write_data", lambda h: h)
    model, _ = _model(1)

    def walk():
        pv_mcts._engines.clear()
        st, out = State(), []
        for _ in range(6):
            pol = pv_mcts.pv_mcts_policy(model, st, 1.0)
            out.append(list(pol))
            st = st.next(st.legal_actions()[int(np.argmax(pol))])
        eng = next(iter(pv_mcts._engines.values()))
        return out, eng

    rows0 = sp.self_play(model, games=6, seed=5)
    pol0, eng0 = walk()
    assert eng0.eval_cache_slots == 0
    monkeypatch.setenv("AQG_EVAL_CACHE_SLOTS", "256")
    rows1 = sp.self_play(model, games=6, seed=5)
    pol1, eng1 = walk()
    assert rows0 == rows1 and pol0 == pol1
    assert eng1.eval_cache_slots == 256 and eng1.counters()["cache_hits"] > 0      # later searches found the earlier ones' positions
    m2, _ = _model(2)
    bm = BatchedMatch([model, m2], num_games=4, sims=6)
    assert all(e is None or e.eval_cache_slots == 0 for e in bm.engines)


def test_graft_entry_smoke():
    """The driver's smoke step (legal mask, GNN forward, MCTS, a tiny generation against the oracle) stays runnable."""
    import __graft_entry__
    __graft_entry__.smoke()
