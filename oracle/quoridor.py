"""ctypes front-end of oracle/quoridor_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libquoridor_oracle.so")
MAXA = 209
BOARDS = {9: (10, 116), 5: (2, 28), 3: (1, 14), 7: (6, 70)}  # N -> (walls, plies_for_draw); constants.py:5-20 (7x7 is ours)


def build(force=False):
    src = os.path.join(_HERE, "quoridor_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-Wall", "-Wno-comment", "-shared", "-o", _SO, src])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.qo_bfs_runs.restype = ctypes.c_longlong
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def init_record(N=9, num_walls=None):
    rec = np.zeros(72, dtype=np.uint8)
    lib().qo72_init(_p(rec), N, BOARDS[N][0] if num_walls is None else num_walls)
    return rec


def legal_actions_batch(recs):
    recs = np.ascontiguousarray(recs, dtype=np.uint8).reshape(-1, 72)
    B = recs.shape[0]
    acts = np.empty((B, MAXA), dtype=np.int16)
    cnt = np.empty(B, dtype=np.int32)
    mask = np.empty((B, MAXA), dtype=np.uint8)
    lib().qo72_legal_actions_batch(_p(recs), B, _p(acts), _p(cnt), _p(mask))
    return acts, cnt, mask


def legal_actions(rec):
    a, c, _ = legal_actions_batch(rec)
    return [int(x) for x in a[0, :c[0]]]


def legal_actions_pos(rec, pos):
    rec = np.ascontiguousarray(rec, dtype=np.uint8)
    out = np.empty(8, dtype=np.int32)
    c = lib().qo72_legal_actions_pos(_p(rec), int(pos), _p(out))
    return [int(x) for x in out[:c]]


def legal_actions_wall(rec, pos):
    rec = np.ascontiguousarray(rec, dtype=np.uint8)
    out = np.empty(2, dtype=np.int32)
    c = lib().qo72_legal_actions_wall(_p(rec), int(pos), _p(out))
    return [int(x) for x in out[:c]]


def next_batch(recs, actions):
    recs = np.ascontiguousarray(recs, dtype=np.uint8).reshape(-1, 72)
    actions = np.ascontiguousarray(actions, dtype=np.int32).reshape(-1)
    out = np.empty_like(recs)
    lib().qo72_next_batch(_p(recs), _p(actions), recs.shape[0], _p(out))
    return out


def next_record(rec, action):
    return next_batch(rec, [action])[0]


def status_batch(recs, plies_for_draw):
    recs = np.ascontiguousarray(recs, dtype=np.uint8).reshape(-1, 72)
    out = np.empty(recs.shape[0], dtype=np.uint8)
    lib().qo72_status_batch(_p(recs), recs.shape[0], int(plies_for_draw), _p(out))
    return out


class State:
    """Reference-shaped state (game_logic.py:15-40) backed by the C oracle; used by oracle/mcts.py."""

    __slots__ = ("rec", "draw")

    def __init__(self, rec=None, N=9, plies_for_draw=None):
        self.rec = init_record(N) if rec is None else np.array(rec, dtype=np.uint8)
        self.draw = BOARDS[int(self.rec[70])][1] if plies_for_draw is None else plies_for_draw

    @property
    def N(self):
        return int(self.rec[70])

    @property
    def player(self):
        return [int(self.rec[0]), int(self.rec[1])]

    @property
    def enemy(self):
        return [int(self.rec[2]), int(self.rec[3])]

    @property
    def walls(self):
        return [int(x) for x in self.rec[4:4 + (self.N - 1) ** 2]]

    @property
    def plies_played(self):
        return int(self.rec[68]) | (int(self.rec[69]) << 8)

    def is_lose(self):
        return self.enemy[0] // self.N == 0

    def is_draw(self):
        return self.plies_played >= self.draw

    def is_done(self):
        return self.is_lose() or self.is_draw()

    def is_first_player(self):
        return self.plies_played % 2 == 0

    def to_array(self):
        return [self.player, self.enemy, self.walls]

    def legal_actions(self):
        return legal_actions(self.rec)

    def next(self, action):
        return State(next_record(self.rec, int(action)), plies_for_draw=self.draw)
