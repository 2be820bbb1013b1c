"""Python restatement of the reference's PV-MCTS and self-play loop -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/pv_mcts.py:20-109 and /root/reference/self_play.py:22-68, written iteratively
(explicit path + backup loop instead of the reference's recursion) with the NumPy-2 scalar promotion the
reference relies on spelled out:

    U = ((f32(1.25) * p) * f32(sqrt(t))) / f32(1 + n)     all float32, left to right   (pv_mcts.py:74)
    Q = -w / n  in float64 (python floats), rounded to float32 at the add, 0.0 when n == 0
    child = first index of the maximum of the float32 scores                          (pv_mcts.py:78)

Pinned by tests/golden/mcts_*.npz and games_*.npz (generated from the real reference with the FakeModel
below and seeded numpy RNG, numpy 2.2.6).
"""
from math import sqrt

import numpy as np

from . import quoridor

C_PUCT = 1.25  # pv_mcts.py:71


def fnv1a(rec68, plies):
    h = 0x811C9DC5
    for b in list(rec68) + [plies & 0xFF, (plies >> 8) & 0xFF]:
        h ^= int(b)
        h = (h * 0x01000193) & 0xFFFFFFFF
    return h


class FakeModel:
    """Deterministic integer-hash evaluator (same definition as tools/gen_golden.py FakeModel and the
    `fake` evaluator of the HIP engine): exactly reproducible f32 priors and value, network independent."""

    def __init__(self, bias=0):
        self.bias = bias

    def predict(self, state, device=None):
        rec = state.rec
        h = fnv1a(rec[:68], state.plies_played)
        legal = state.legal_actions()
        N = state.N
        prow = int(rec[0]) // N
        rs = []
        for a in legal:
            r = (((((h ^ ((a + 1) * 0x9E3779B1)) & 0xFFFFFFFF) * 0x85EBCA6B) & 0xFFFFFFFF) >> 22) + 1
            if a < N * N and (a // N) < prow:
                r *= 1 + self.bias
            rs.append(r)
        policy = (np.asarray(rs, dtype=np.float32) / np.float32(sum(rs))).astype(np.float32)
        v = ((((h * 0xC2B2AE35) & 0xFFFFFFFF) >> 16) - 32768) / 32768.0
        return policy, float(np.float32(v))


class _Node:
    __slots__ = ("state", "p", "w", "n", "children")

    def __init__(self, state, p):
        self.state, self.p, self.w, self.n, self.children = state, p, 0, 0, None


def _select(node):
    """pv_mcts.py:69-78."""
    t = 0
    for c in node.children:
        t += c.n
    st = np.float32(sqrt(t))
    best, best_i = None, 0
    for i, c in enumerate(node.children):
        u = ((np.float32(C_PUCT) * np.float32(c.p)) * st) / np.float32(1 + c.n)
        q = np.float32(-c.w / c.n) if c.n else np.float32(0.0)
        s = np.float32(q + u)
        if best is None or s > best:
            best, best_i = s, i
    return node.children[best_i]


def search(model, state, sims, device=None, trace=None):
    """Run `sims` simulations from `state`; returns the root node (pv_mcts.py:81-85, :33-66)."""
    root = _Node(state, 0)
    for _ in range(sims):
        path = [root]
        node = root
        while True:
            if node.state.is_done():                         # :35-42
                value = -1 if node.state.is_lose() else 0
                break
            if not node.children:                            # :45-57 (empty list re-predicts, as the reference)
                prior, value = model.predict(node.state, device)
                legal = node.state.legal_actions()
                node.children = [_Node(node.state.next(a), p) for a, p in zip(legal, prior)]
                break
            node = _select(node)                             # :60
            path.append(node)
        if trace is not None:
            trace.append(len(path))
        v = value
        for nd in reversed(path):                            # w += value; n += 1 up the path, sign flips per ply
            nd.w += v
            nd.n += 1
            v = -v
    return root


def boltzman(xs, temperature):
    """pv_mcts.py:106-109."""
    xs = [x ** (1 / temperature) for x in xs]
    tot = sum(xs)
    return [x / tot for x in xs]


def pv_mcts_policy(model, state, temperature, sims, device=None):
    """pv_mcts.py:20-95."""
    root = search(model, state, sims, device)
    visits = [c.n for c in root.children]
    if temperature == 0:
        pol = np.zeros(len(visits))
        pol[int(np.argmax(visits))] = 1
        return pol
    return boltzman(visits, temperature)


def choice_index(p, u):
    """np.random.choice(a, p=p) index for one uniform u: cdf = cumsum(p); cdf /= cdf[-1]; searchsorted right."""
    cdf = np.cumsum(np.asarray(p, dtype=np.float64))
    cdf /= cdf[-1]
    return int(np.searchsorted(cdf, u, side="right"))


def first_player_value(ended_state):
    """self_play.py:22-27."""
    if ended_state.is_lose():
        return -1 if ended_state.is_first_player() else 1
    return 0


def play(model, sims, temperature=1.0, N=9, uniforms=None, rng=None, max_plies=None):
    """self_play.py:40-68.  `uniforms` (iterable of floats) or `rng` (np.random.RandomState) supply the one
    uniform consumed per move by np.random.choice (self_play.py:57).  `max_plies` bounds the sample (bench)."""
    A = N * N + 2 * (N - 1) ** 2
    history = []
    state = quoridor.State(N=N)
    it = iter(uniforms) if uniforms is not None else None
    while not state.is_done():
        if max_plies is not None and len(history) >= max_plies:
            break
        scores = pv_mcts_policy(model, state, temperature, sims)
        legal = state.legal_actions()
        policy = [0] * A
        for a, s in zip(legal, scores):
            policy[a] = s
        history.append([state.to_array(), policy, None])
        u = next(it) if it is not None else rng.random_sample()
        state = state.next(legal[choice_index(scores, u)])
    value = first_player_value(state)
    for h in history:
        h[2] = value
        value = -value
    return history


def first_player_point(ended_state):
    """evaluate_network.py:18-22: 1 first player wins, 0 loses, 0.5 draw."""
    if ended_state.is_lose():
        return 0 if ended_state.is_first_player() else 1
    return 0.5


def evaluate_play(model_first, model_second, sims, temperature=1.0, N=9, uniforms=None, rng=None):
    """evaluate_network.py:25-44 `play(next_actions)` with next_actions = (pv_mcts_action(model_first),
    pv_mcts_action(model_second)) (pv_mcts.py:98-104): the mover's own model searches from a fresh tree, the action is
    np.random.choice(legal, p=scores) -- one uniform per move.  Returns (first player's point, list of actions)."""
    state = quoridor.State(N=N)
    it = iter(uniforms) if uniforms is not None else None
    actions = []
    while not state.is_done():
        model = model_first if state.is_first_player() else model_second
        scores = pv_mcts_policy(model, state, temperature, sims)
        legal = state.legal_actions()
        u = next(it) if it is not None else rng.random_sample()
        a = legal[choice_index(scores, u)]
        actions.append(int(a))
        state = state.next(a)
    return first_player_point(state), actions
