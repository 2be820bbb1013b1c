"""Baseline Quoridor agents -- drop-in for the reference's agents.py (random, alpha-beta, rollout MCTS).

They are CPU opponents for strength tracking (evaluate_agents.py:62-89), CPU code in the reference too.  Each is a function
of the game state that returns an action (agents.py:1-4).  The rules they walk -- legal_actions(), next(), the jump-aware
shortest path -- come from the host build of the SAME rule header the GPU kernels compile (csrc/host_agents.cpp over
csrc/quoridor_core.hpp, inside libaqgnn_hip.so), so no GPU launch per call and no second copy of the rules.
The random streams are consumed exactly like the reference's: `random.randint` once per random move (agents.py:17).
"""
import ctypes
import math
import random

import numpy as np

from . import _lib
from .constants import NUM_PLIES_FOR_DRAW, NUM_WALLS, board_params

MAX_DIST_FROM_GOAL = NUM_PLIES_FOR_DRAW // 2 - NUM_WALLS   # agents.py:11


def _rec(state):
    return np.ascontiguousarray(state.record() if hasattr(state, "record") else state, dtype=np.uint8)


def _legal(state):
    """State.legal_actions() on the host (ordered like the reference: pawn moves, then per slot H, V)."""
    rec = _rec(state)
    out = (ctypes.c_uint8 * _lib.MAX_LEGAL)()
    n = _lib.load().aqg_host_legal_actions(int(rec[70]), rec.ctypes.data_as(ctypes.c_void_p), out)
    if n < 0:
        raise _lib.HipLibraryError("aqg_host_legal_actions failed")
    return [int(out[i]) for i in range(n)]


def _board(state):
    return int(state.N) if hasattr(state, "N") else int(_rec(state)[70])


def _max_dist(state):
    walls, draw = board_params(_board(state))
    return draw // 2 - walls


def _draw(state):
    return board_params(_board(state))[1]


# ------------------------------------------------------------------ random (agents.py:14-18)
def random_action(state):
    """Selects a uniformly random legal action."""
    legal_actions = _legal(state)
    return legal_actions[random.randint(0, len(legal_actions) - 1)]


# ------------------------------------------------------------------ alpha-beta (agents.py:22-108)
def shortest_path(state):
    """Plies the mover needs to reach its goal row with the other pawn frozen (jumps allowed); -1 if walled in (agents.py:27-41)."""
    rec = _rec(state)
    return int(_lib.load().aqg_host_shortest_path(int(rec[70]), rec.ctypes.data_as(ctypes.c_void_p)))


def heuristic_eval(state):
    """(enemy's shortest path - mover's shortest path) / MAX_DIST_FROM_GOAL (agents.py:22-54)."""
    rec = _rec(state)
    return float(_lib.load().aqg_host_heuristic_eval(int(rec[70]), rec.ctypes.data_as(ctypes.c_void_p), _max_dist(state)))


def alpha_beta_action(state, max_depth=2):
    """The action with the maximum depth-limited negamax value, first best wins (agents.py:90-108).  The whole search runs
    natively: the reference's pure-Python version needs ~131^3 evaluations per move on an open 9x9 board."""
    rec = _rec(state)
    a = int(_lib.load().aqg_host_alpha_beta_action(int(rec[70]), rec.ctypes.data_as(ctypes.c_void_p), _draw(state), _max_dist(state), int(max_depth)))
    return None if a < 0 else a


# ------------------------------------------------------------------ rollout MCTS (agents.py:112-214)
def playout(state):
    """Random playout to the end of the game: -1 loss, 0 draw, from the point of view of `state`'s mover (agents.py:112-122)."""
    sign = 1
    while True:                                   # (the reference recurses; a loop keeps 116-ply games off the Python stack)
        if state.is_lose():
            return -sign
        if state.is_draw():
            return 0
        state = state.next(random_action(state))
        sign = -sign


def argmax(collection):
    return collection.index(max(collection))


class _Tree:
    """The reference's recursive Node.evaluate (agents.py:133-197) as an arena walked iteratively: nodes are rows of parallel
    lists (state, w, n, first child, child count), children of a node are consecutive rows, one simulation = one descent that
    records its path + one backup loop with the sign flipping per ply.  Same decisions in the same order, hence the same
    consumption of the `random` stream (the goldens replay it): an untried child first, else the first maximum of UCB1
    (agents.py:177-196); a playout at a childless node, which is expanded on its tenth visit (:153-163)."""

    def __init__(self, state):
        self.state, self.w, self.n, self.first, self.count = [state], [0], [0], [-1], [0]
        self.expand(0)

    def expand(self, i):
        self.first[i] = len(self.state)
        for a in _legal(self.state[i]):
            self.state.append(self.state[i].next(a))
            self.w.append(0); self.n.append(0); self.first.append(-1); self.count.append(0)
        self.count[i] = len(self.state) - self.first[i]

    def select(self, i):
        kids = range(self.first[i], self.first[i] + self.count[i])
        t = 0
        for k in kids:
            if self.n[k] == 0:
                return k
            t += self.n[k]
        best, best_k = None, -1
        for k in kids:
            u = -self.w[k] / self.n[k] + 2 * (2 * math.log(t) / self.n[k]) ** 0.5
            if best is None or u > best:          # strict: the first maximum wins, like list.index(max(...))
                best, best_k = u, k
        return best_k

    def simulate(self):
        path = [0]
        while True:
            i = path[-1]
            st = self.state[i]
            if st.is_done():
                value = -1 if st.is_lose() else 0
                break
            if self.count[i] == 0:
                value = playout(st)
                if self.n[i] + 1 == 10:
                    self.expand(i)
                break
            path.append(self.select(i))
        for i in reversed(path):                  # value is the leaf's own view; every step up the path negates it (:167)
            self.w[i] += value
            self.n[i] += 1
            value = -value


def mcts_action(state, evaluations=100):
    """Plain Monte-Carlo tree search with random playouts; the most visited root action (agents.py:130-214)."""
    tree = _Tree(state)
    for _ in range(evaluations):
        tree.simulate()
    visits = tree.n[tree.first[0]:tree.first[0] + tree.count[0]]
    return _legal(state)[argmax(visits)]


if __name__ == '__main__':
    from .game_logic import State
    s = State()
    while not s.is_done():
        s = s.next(random_action(s))
    print(s)
