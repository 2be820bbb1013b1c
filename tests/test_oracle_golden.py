"""CPU tests: pin the oracle (oracle/) and the host build of the kernels' rule header against golden
vectors generated from the REAL reference (tools/gen_golden.py, numpy 2.2.6).  Bit-exact bar."""
import numpy as np
import pytest

from oracle import quoridor as oq
from oracle import mcts as om
from tests import _util as U

DRAW = {9: 116, 5: 28, 3: 14}


def test_reference_known_answers_9x9():
    k = U.golden("kat_9x9.npz")
    # the reference's own scenario (test_legal_walls.py:3-21): both orientations illegal at slot 26
    assert list(k["tlw_wall26"]) == [] and list(k["tlw_pos40"]) == [49, 39] and len(k["tlw_legal"]) == 99
    rec = k["tlw_state"]
    assert oq.legal_actions_wall(rec, 26) == []
    assert oq.legal_actions_pos(rec, 40) == [49, 39]
    assert oq.legal_actions(rec) == [int(x) for x in k["tlw_legal"]]
    acts, cnt = U.hc_legal(9, rec)
    assert list(acts[0, :cnt[0]]) == list(k["tlw_legal"])
    # initial position: 131 actions, documented prefix
    init = [int(x) for x in k["init_legal"]]
    assert len(init) == 131 and init[:7] == [67, 75, 77, 81, 145, 82, 146]
    assert oq.legal_actions(k["init_state"]) == init
    assert np.array_equal(oq.init_record(9), k["init_state"])
    # pawn-jump known answers
    for rec, exp in zip(k["jump_states"], k["jump_moves"]):
        exp = [int(x) for x in exp if x >= 0]
        assert oq.legal_actions_pos(rec, int(rec[0])) == exp
        a, c = U.hc_legal(9, rec)
        assert [int(x) for x in a[0, :len(exp)]] == exp
    # transitions
    assert np.array_equal(oq.next_batch(np.repeat(k["init_state"][None], 3, 0), k["next_actions"]), k["next_states"])
    assert np.array_equal(U.hc_next(9, np.repeat(k["init_state"][None], 3, 0), k["next_actions"]), k["next_states"])


def test_survey_documented_jumps():
    # SURVEY 8c item 3 (player[0], enemy[0] own-frame, walls) -> legal_actions_pos
    cases = [(40, 49, {}, [22, 49, 39, 41]), (40, 49, {20: 1}, [30, 32, 49, 39, 41]),
             (13, 76, {}, [3, 5, 22, 12, 14]), (40, 41, {}, [31, 49, 38, 41])]
    for pp, ep, walls, exp in cases:
        rec = oq.init_record(9)
        rec[0], rec[2] = pp, ep
        for s, o in walls.items():
            rec[4 + s] = o
        assert oq.legal_actions_pos(rec, pp) == exp


@pytest.mark.parametrize("N", [9, 5, 3])
def test_walk_states_bit_exact(N):
    g = U.golden(f"walk_{N}x{N}.npz")
    recs, legal, counts = g["states"], g["legal"], g["counts"]
    # oracle (C restatement of the reference algorithm)
    acts, cnt, mask = oq.legal_actions_batch(recs)
    assert np.array_equal(cnt, counts)
    assert np.array_equal(acts[:, :U.MAX_LEGAL], legal)
    assert mask.sum() == counts.sum()
    # host build of the kernels' bitboard rules
    hacts, hcnt = U.hc_legal(N, recs)
    assert np.array_equal(hcnt, counts)
    assert np.array_equal(hacts, legal)
    # next() and terminal flags
    ok = g["actions"] >= 0
    assert np.array_equal(oq.next_batch(recs[ok], g["actions"][ok]), g["next_states"][ok])
    assert np.array_equal(U.hc_next(N, recs[ok], g["actions"][ok]), g["next_states"][ok])
    assert np.array_equal(oq.status_batch(recs, DRAW[N]), g["status"])
    assert np.array_equal(U.hc_status(N, recs, DRAW[N]), g["status"])


def test_pawn_obstacle_counter_examples():
    # reference says ILLEGAL where a pawn-free flood fill says legal
    g = U.golden("obstacle_9x9.npz")
    for rec, legal, absent in zip(g["states"], g["legal"], g["absent"]):
        exp = [int(x) for x in legal if x >= 0]
        assert int(absent) not in exp
        assert oq.legal_actions(rec) == exp
        a, c = U.hc_legal(9, rec)
        assert [int(x) for x in a[0, :c[0]]] == exp


@pytest.mark.parametrize("N", [9, 5, 3])
def test_mcts_traces_match_reference(N):
    g = U.golden(f"mcts_{N}x{N}.npz")
    n = int(g["count"][0])
    checked = 0
    for k in range(n):
        sims, bias, T = g[f"t{k}_cfg"]
        if sims > 50 and k % 3:      # keep the CPU suite short; 200-sim traces sampled
            continue
        st = oq.State(g[f"t{k}_state"])
        pol = om.pv_mcts_policy(om.FakeModel(int(bias)), st, float(T), int(sims))
        assert np.array_equal(np.asarray(pol, dtype=np.float64), g[f"t{k}_policy"]), f"trace {k}"
        checked += 1
    assert checked >= 10


@pytest.mark.parametrize("N", [9, 5, 3])
def test_full_games_match_reference(N):
    g = U.golden(f"games_{N}x{N}.npz")
    for i in range(int(g["count"][0])):
        seed, sims, bias = (int(x) for x in g[f"g{i}_cfg"])
        if N == 9 and i == 1:
            continue  # 116-ply draw game: covered on the GPU engine test; too slow for the CPU suite
        rng = np.random.RandomState(seed)   # == np.random.seed(seed) + global np.random.choice in the reference
        hist = om.play(om.FakeModel(bias), sims, 1.0, N=N, rng=rng)
        st = g[f"g{i}_states"]
        assert len(hist) == st.shape[0]
        for j, (sa, pol, z) in enumerate(hist):
            assert sa[0] == list(st[j, 0:2]) and sa[1] == list(st[j, 2:4])
            assert sa[2] == list(st[j, 4:4 + (N - 1) ** 2])
            assert np.array_equal(np.asarray(pol, dtype=np.float64), g[f"g{i}_policy"][j])
            assert z == int(g[f"g{i}_z"][j])


@pytest.mark.parametrize("N", [9, 5])
def test_evaluation_games_match_reference(N):
    """evaluate_network.play() of the reference (two fake models through pv_mcts_action, tools/gen_golden_eval.py)."""
    g = U.golden(f"eval_{N}x{N}.npz")
    for i in range(int(g["count"][0])):
        seed, sims, b0, b1 = (int(x) for x in g[f"e{i}_cfg"])
        if N == 9 and len(g[f"e{i}_actions"]) > 30:
            continue  # the longer 9x9 games run on the GPU engine test
        rng = np.random.RandomState(seed)
        point, actions = om.evaluate_play(om.FakeModel(b0), om.FakeModel(b1), sims, 1.0, N=N, rng=rng)
        assert actions == [int(a) for a in g[f"e{i}_actions"]]
        assert point == float(g[f"e{i}_point"][0])


def test_training_oracle_forward_equals_gnn_oracle():
    """oracle/train.py (torch fp64, dense adjacency, autograd) restates the same network as oracle/gnn.py."""
    from oracle import gnn as og, train as ot
    g = U.golden("walk_9x9.npz")
    recs = g["states"][::700][:12]
    p = og.init_params(4)
    pol, val = ot.TorchGNN(p)(recs)
    ref = og.forward_states(p, recs)
    assert np.abs(pol.detach().numpy() - ref["policy"]).max() < 1e-14
    assert np.abs(val.detach().numpy()[:, 0] - ref["value"]).max() < 1e-14
    assert [ot.lr_lambda(e) for e in (0, 49, 50, 79, 80, 99)] == [1.0, 1.0, 0.5, 0.5, 0.25, 0.25]   # train_network.py:59-65


@pytest.mark.parametrize("N", [7, 9, 5, 3])
def test_host_rule_header_equals_oracle_on_random_play(N):
    """Beyond the fixtures: the header the HIP kernels compile (bitboards, interleaved flood fills, mask-algebra prefilter)
    against the C oracle (array/queue restatement, pinned on 3x3 / 5x5 / 9x9) on fresh random play, wall-heavy, including
    the 7x7 board for which the reference defines no constants.  Ordered legal lists, transitions, terminal flags."""
    rng = np.random.RandomState(100 + N)
    recs = []
    for game in range(40):
        s = oq.State(N=N)
        for ply in range(60):
            if s.is_done():
                break
            recs.append(s.rec.copy())
            la = s.legal_actions()
            walls = [a for a in la if a >= N * N]
            pick = walls if (walls and rng.rand() < 0.6) else la
            s = s.next(pick[rng.randint(len(pick))])
    recs = np.stack(recs)
    a, c, _ = oq.legal_actions_batch(recs)
    ha, hc = U.hc_legal(N, recs)
    assert np.array_equal(hc, c)
    for i in range(len(recs)):
        assert np.array_equal(ha[i, :c[i]], a[i, :c[i]]), i
    first = np.asarray([a[i, rng.randint(c[i])] for i in range(len(recs))], dtype=np.int32)
    assert np.array_equal(U.hc_next(N, recs, first), oq.next_batch(recs, first))
    draw = oq.BOARDS[N][1]
    assert np.array_equal(U.hc_status(N, recs, draw), oq.status_batch(recs, draw))


def test_choice_index_matches_numpy():
    rng = np.random.RandomState(5)
    for _ in range(200):
        n = rng.randint(1, 40)
        p = rng.randint(0, 7, size=n).astype(np.float64)
        if p.sum() == 0:
            p[0] = 1
        p = p / p.sum()
        r1 = np.random.RandomState(9)
        r2 = np.random.RandomState(9)
        assert r1.choice(n, p=p) == om.choice_index(p, r2.random_sample())


@pytest.mark.parametrize("N", [9, 5, 3])
def test_gnn_oracle_dense_form_equals_edge_list_form(N):
    """oracle/gnn.py holds the GNN forward twice: the edge-list statement (gcn_conv over board_edges, PyG's scatter form) and a
    dense-adjacency vectorisation the GPU tests use at BASELINE sizes.  They must agree to fp64 rounding on reference-walk
    states (wall-heavy ones included) with non-zero biases."""
    from oracle import gnn as og
    g = U.golden(f"walk_{N}x{N}.npz")
    p = og.init_params(5, N=N)
    for l in range(3):
        p[f"gcn_layers.{l}.bias"] = np.linspace(-0.4, 0.6, 128).astype(np.float32) * (l + 1)
    nwalls = (g["states"][:, 4:68] != 0).sum(1)
    pick = np.concatenate([np.argsort(-nwalls, kind="stable")[:40], np.arange(0, g["states"].shape[0], 97)[:80]])
    recs = g["states"][pick]
    a, b = og.forward_states(p, recs), og.forward_states_dense(p, recs, chunk=50)
    for k in a:
        assert np.abs(a[k] - b[k]).max() < 1e-12, k


def test_relu_margins_match_the_edge_list_form():
    """oracle/gnn.py::relu_margins (dense form, what the gradient parity tests filter their positions with) against the smallest
    non-zero |pre-activation| recomputed layer by layer with the edge-list GCNConv."""
    from oracle import gnn as og
    params = og.init_params(3)
    recs = U.golden("walk_9x9.npz")["states"][5:400:60]
    got = og.relu_margins(params, recs)
    p = {k: np.asarray(v, dtype=np.float64) for k, v in params.items()}
    for i, rec in enumerate(recs):
        h, e = og.node_features(rec).astype(np.float64), og.board_edges(rec)
        smallest = np.inf
        for l in range(3):
            pre = og.gcn_conv(h, e, p[f"gcn_layers.{l}.lin.weight"], p[f"gcn_layers.{l}.bias"])
            smallest = min(smallest, np.abs(pre[pre != 0]).min())
            h = np.maximum(pre, 0.0)
        g = h.mean(0)
        for head in ("policy_head", "value_head"):
            pre = g @ p[f"{head}.0.weight"].T + p[f"{head}.0.bias"]
            smallest = min(smallest, np.abs(pre[pre != 0]).min())
        assert abs(got[i] - smallest) <= 1e-12 + 1e-6 * smallest
