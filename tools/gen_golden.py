#!/usr/bin/env python3
"""Generate golden input/output vectors by IMPORTING the real reference (read-only, /root/reference).

Generation-time tooling only: runs in the build container (where /root/reference exists), never on
the GPU box.  It writes small data fixtures (inputs + expected outputs) to tests/golden/; no reference
source text is copied anywhere.  Recipe follows SURVEY.md Appendix A:

  * stub modules for imports the hot path never uses (snakeviz, zmq.backend.first, torchsummary,
    torch_tensorrt) -- none of them touch arithmetic;
  * a scratch `constants.py` holding the reference's own commented 9x9 block (constants.py:17-23)
    (or the committed 5x5 block for --board 5) placed ahead of /root/reference on sys.path;
  * `train_network.preprocess_input = None` to satisfy the unused import at pv_mcts.py:14.

Usage:  python tools/gen_golden.py --board 9 [--states 12000]
        python tools/gen_golden.py --board 5
"""
import argparse
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
BOARDS = {9: (10, 116), 5: (2, 28), 3: (1, 14)}  # constants.py:5-20


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference(board):
    walls, draw = BOARDS[board]
    scratch = tempfile.mkdtemp(prefix="aqg_golden_")
    with open(os.path.join(scratch, "constants.py"), "w") as f:
        f.write(f"BOARD_SIZE = {board}\nNUM_WALLS = {walls}\nNUM_PLIES_FOR_DRAW = {draw}\n"
                f"PV_NETWORK_NAME = 'CNN'\nPV_NETWORK_PATH = f'models/CNN/{board}x{board}/'\n")
    _stub("snakeviz")
    _stub("snakeviz.cli", main=lambda *a, **k: None)
    zb = _stub("zmq.backend", first=None)
    _stub("zmq", backend=zb)
    _stub("torchsummary", summary=lambda *a, **k: None)

    class _Input:  # torch_tensorrt.Input placeholder
        def __init__(self, *a, **k):
            pass
    _stub("torch_tensorrt", compile=lambda m, **k: m, Input=_Input)
    sys.path[:0] = [scratch, REF]
    os.chdir(scratch)
    import train_network  # noqa
    train_network.preprocess_input = None
    import game_logic, pv_mcts, self_play, pv_network_cnn  # noqa
    assert game_logic.State().N == board
    return game_logic, pv_mcts, self_play, pv_network_cnn


# ---------------------------------------------------------------- state <-> state72 records
def rec_of(state):
    r = np.zeros(72, dtype=np.uint8)
    r[0], r[1] = state.player
    r[2], r[3] = state.enemy
    w = np.asarray(state.walls, dtype=np.uint8)
    r[4:4 + len(w)] = w
    r[68] = state.plies_played & 0xFF
    r[69] = state.plies_played >> 8
    r[70] = state.N
    return r


# ---------------------------------------------------------------- fake model (network-independent MCTS traces)
def fnv1a(rec68, plies):
    h = 0x811C9DC5
    for b in list(rec68) + [plies & 0xFF, (plies >> 8) & 0xFF]:
        h ^= int(b)
        h = (h * 0x01000193) & 0xFFFFFFFF
    return h


class FakeModel:
    """predict() is a deterministic integer hash of the state -> exactly reproducible f32 priors/value.

    prior weight of the i-th legal action a:  r = ((h ^ (a+1)*0x9E3779B1) * 0x85EBCA6B mod 2^32 >> 22) + 1
    (1..1024), multiplied by (1+bias) when a is a pawn move to a smaller row (forward);
    p = f32(r) / f32(sum r)   (one correctly-rounded f32 division);
    value = ((h * 0xC2B2AE35 mod 2^32 >> 16) - 32768) / 32768  (exact dyadic rational).
    """

    def __init__(self, bias=0):
        self.bias = bias
        self.calls = 0

    def predict(self, state, device):
        self.calls += 1
        rec = rec_of(state)
        h = fnv1a(rec[:68], state.plies_played)
        legal = state.legal_actions()
        N = state.N
        rs = []
        for a in legal:
            a = int(a)  # np.random.choice hands np.int64 actions to State.next inside play()
            r = ((((h ^ ((a + 1) * 0x9E3779B1)) & 0xFFFFFFFF) * 0x85EBCA6B) & 0xFFFFFFFF) >> 22
            r += 1
            if a < N * N and (a // N) < (int(state.player[0]) // N):
                r *= (1 + self.bias)
            rs.append(r)
        tot = np.float32(sum(rs))
        policy = (np.asarray(rs, dtype=np.float32) / tot).astype(np.float32)
        v = ((((h * 0xC2B2AE35) & 0xFFFFFFFF) >> 16) - 32768) / 32768.0
        return policy, float(np.float32(v))


# ---------------------------------------------------------------- generators
def random_walk_states(gl, n_states, seed, p_wall=0.3):
    """Random legal play from the start position; P(wall action)=p_wall when the mover has walls."""
    rng = np.random.RandomState(seed)
    out = []
    while len(out) < n_states:
        s = gl.State()
        while not s.is_done() and len(out) < n_states:
            la = s.legal_actions()
            out.append((s, la))
            if not la:
                break
            N2 = s.N * s.N
            pawn = [a for a in la if a < N2]
            wall = [a for a in la if a >= N2]
            if wall and (not pawn or rng.random_sample() < p_wall):
                a = wall[rng.randint(len(wall))]
            else:
                a = pawn[rng.randint(len(pawn))]
            s = s.next(a)
    return out


def pack_lists(lists, width):
    arr = -np.ones((len(lists), width), dtype=np.int16)
    for i, l in enumerate(lists):
        arr[i, :len(l)] = l
    return arr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--board", type=int, default=9)
    ap.add_argument("--states", type=int, default=12000)
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    args = ap.parse_args()
    board = args.board
    gl, pv_mcts, self_play, cnn = import_reference(board)
    os.makedirs(args.out, exist_ok=True)
    State = gl.State
    tag = f"{board}x{board}"
    A = board * board + 2 * (board - 1) ** 2
    meta = dict(numpy=np.__version__, board=board)

    # ---- 1. known-answer vectors
    kat = {}
    if board == 9:
        s = State()
        for i, o in [(24, 1), (27, 1), (32, 2), (36, 2), (37, 1), (41, 1), (42, 2), (43, 1)]:
            s.walls[i] = o          # the reference's own scenario, test_legal_walls.py:5-16
        s.player[0] = 40
        s.enemy[0] = 32
        kat["tlw_state"] = rec_of(s)
        kat["tlw_wall26"] = np.asarray(s.legal_actions_wall(26), dtype=np.int16)
        kat["tlw_pos40"] = np.asarray(s.legal_actions_pos(40), dtype=np.int16)
        kat["tlw_legal"] = np.asarray(s.legal_actions(), dtype=np.int16)
        jumps = []
        for (pp, ep, walls) in [(40, 49, {}), (40, 49, {20: 1}), (13, 76, {}), (40, 41, {}),
                                (4, 76, {}), (36, 43, {}), (44, 37, {}), (40, 31, {27: 1}),
                                (40, 39, {27: 2}), (9, 70, {}), (76, 13, {})]:
            s = State()
            s.player[0] = pp
            s.enemy[0] = ep
            for k, v in walls.items():
                s.walls[k] = v
            jumps.append((rec_of(s), s.legal_actions_pos(pp)))
        kat["jump_states"] = np.stack([j[0] for j in jumps])
        kat["jump_moves"] = pack_lists([j[1] for j in jumps], 8)
    s0 = State()
    kat["init_state"] = rec_of(s0)
    kat["init_legal"] = np.asarray(s0.legal_actions(), dtype=np.int16)
    nxt = []
    for a in [s0.legal_actions()[0], board * board + 3, board * board + (board - 1) ** 2 + min(10, (board - 1) ** 2 - 1)]:
        nxt.append((a, rec_of(s0.next(a))))
    kat["next_actions"] = np.asarray([n[0] for n in nxt], dtype=np.int32)
    kat["next_states"] = np.stack([n[1] for n in nxt])
    np.savez_compressed(os.path.join(args.out, f"kat_{tag}.npz"), **kat)
    print("kat done", {k: v.shape for k, v in kat.items()})

    # ---- 2. random-walk states: ordered legal list, next() for one sampled action, status flags
    n_states = args.states if board == 9 else min(args.states, 4000)
    walk = random_walk_states(gl, n_states, seed=2)
    # also include walls-exhausted and p_wall-heavy walks for coverage
    walk += random_walk_states(gl, n_states // 6, seed=7, p_wall=0.8)
    recs = np.stack([rec_of(s) for s, _ in walk])
    legal = pack_lists([la for _, la in walk], 136)
    counts = np.asarray([len(la) for _, la in walk], dtype=np.int32)
    rng = np.random.RandomState(11)
    acts = np.asarray([la[rng.randint(len(la))] if la else -1 for _, la in walk], dtype=np.int32)
    nexts = np.stack([rec_of(s.next(int(a))) if a >= 0 else rec_of(s) for (s, _), a in zip(walk, acts)])
    status = np.asarray([(1 if s.is_lose() else 0) | (2 if s.is_draw() else 0) for s, _ in walk], dtype=np.uint8)
    np.savez_compressed(os.path.join(args.out, f"walk_{tag}.npz"), states=recs, legal=legal, counts=counts,
                        actions=acts, next_states=nexts, status=status)
    print("walk done", recs.shape, "mean legal", counts.mean(), "min", counts.min(), "max", counts.max())

    # ---- 3. pawn-as-obstacle counter-examples: reference stricter than a plain flood fill
    if board == 9:
        B = [
            ([40, 2], [56, 0], [0,2,1,0,2,1,2,0, 0,0,0,0,1,0,0,0, 0,2,1,0,1,0,0,0, 1,0,0,0,0,1,0,2, 0,1,0,0,0,0,0,0, 2,1,0,0,0,0,1,0, 0,0,0,0,0,0,0,2, 0,0,0,0,0,0,0,1], 104),
            ([64, 4], [80, 3], [2,1,0,0,0,0,0,0, 0,0,0,2,0,0,0,0, 2,0,0,0,0,0,0,2, 0,1,0,1,0,2,0,0, 0,0,0,0,0,0,0,0, 0,0,0,0,0,1,2,1, 0,1,0,0,0,0,0,0, 0,0,2,0,0,0,0,0], 181),
            ([80, 3], [65, 4], [0,0,0,0,0,2,0,0, 0,0,0,0,0,0,1,0, 1,2,1,0,0,0,0,0, 0,0,0,0,0,0,0,0, 0,0,2,0,1,0,1,0, 2,0,0,0,0,0,0,2, 0,0,0,0,2,0,0,0, 0,0,0,0,0,0,1,2], 172),
        ]
        st, ll, absent = [], [], []
        for p, e, w, a in B:
            s = State(player=list(p), enemy=list(e), walls=list(w), plies_played=20)
            la = s.legal_actions()
            assert a not in la
            st.append(rec_of(s)); ll.append(la); absent.append(a)
        np.savez_compressed(os.path.join(args.out, f"obstacle_{tag}.npz"), states=np.stack(st),
                            legal=pack_lists(ll, 136), absent=np.asarray(absent, dtype=np.int32))
        print("obstacle done")

    # ---- 4. MCTS traces with the fake model (network independent)
    traces = {}
    roots = [State()]
    mids = [s for s, la in walk[:: max(1, len(walk) // 40)] if not s.is_done() and la][:10]
    late = [s for s, la in walk if s.plies_played > (60 if board == 9 else 12) and not s.is_done() and la][:4]
    roots += mids + late
    sims_list = [50, 200] if board == 9 else [50]
    k = 0
    for sims in sims_list:
        pv_mcts.PV_EVALUATE_COUNT = sims
        for bias in (0, 30):
            fm = FakeModel(bias)
            for ri, s in enumerate(roots if sims == 50 else roots[:5]):
                for T in ((1.0, 0) if ri < 3 else (1.0,)):
                    pol = pv_mcts.pv_mcts_policy(fm, State(player=list(s.player), enemy=list(s.enemy),
                                                         walls=list(s.walls), plies_played=s.plies_played), T, "cpu")
                    traces[f"t{k}_state"] = rec_of(s)
                    traces[f"t{k}_cfg"] = np.asarray([sims, bias, T], dtype=np.float64)
                    traces[f"t{k}_policy"] = np.asarray(pol, dtype=np.float64)
                    traces[f"t{k}_legal"] = np.asarray(s.legal_actions(), dtype=np.int16)
                    k += 1
        print("mcts traces sims", sims, "->", k)
    traces["count"] = np.asarray([k])
    np.savez_compressed(os.path.join(args.out, f"mcts_{tag}.npz"), **traces)

    # ---- 5. full seeded games through the reference's self_play.play() (history schema, z signs)
    games = {}
    g = 0
    for seed, sims, bias in ([(123, 16, 30), (321, 12, 0), (77, 24, 60)] if board == 9 else [(123, 30, 10), (5, 20, 0)]):
        pv_mcts.PV_EVALUATE_COUNT = sims
        np.random.seed(seed)
        hist = self_play.play(FakeModel(bias), "cpu")
        st = np.zeros((len(hist), 72), dtype=np.uint8)
        for i, (sa, pol, z) in enumerate(hist):
            st[i, 0:2] = sa[0]; st[i, 2:4] = sa[1]; st[i, 4:4 + len(sa[2])] = sa[2]
            st[i, 68] = i & 0xFF; st[i, 69] = i >> 8; st[i, 70] = board
        games[f"g{g}_cfg"] = np.asarray([seed, sims, bias], dtype=np.int64)
        games[f"g{g}_states"] = st
        games[f"g{g}_policy"] = np.asarray([h[1] for h in hist], dtype=np.float64)
        games[f"g{g}_z"] = np.asarray([h[2] for h in hist], dtype=np.int8)
        print("game", g, "plies", len(hist), "z0", hist[0][2])
        g += 1
    games["count"] = np.asarray([g])
    np.savez_compressed(os.path.join(args.out, f"games_{tag}.npz"), **games)

    # ---- 6. featuriser (pv_network_cnn.preprocess_input, :88-114) on a sample of walk states
    net = cnn.CNNNetwork.__new__(cnn.CNNNetwork)  # preprocess_input uses no instance state
    idx = np.linspace(0, len(walk) - 1, 200).astype(int)
    triples = [walk[i][0].to_array() for i in idx]
    feats = cnn.CNNNetwork.preprocess_input(net, triples)
    np.savez_compressed(os.path.join(args.out, f"feat_{tag}.npz"), states=recs[idx], planes=feats.astype(np.float32))
    print("feat done", feats.shape, meta)


if __name__ == "__main__":
    main()
