"""New-parameter evaluation -- drop-in for the reference's evaluate_network.py, on the batched HIP engine.

Call surface kept (evaluate_network.py:13-94): EN_GAME_COUNT, EN_TEMPERATURE, first_player_point, play(next_actions),
update_best_player, evaluate_network().  `evaluate_network()` plays the EN_GAME_COUNT games of latest vs best
CONCURRENTLY: the games in which `latest` moves first form one engine batch, the games in which `best` moves first a
second one, and before every ply each batch is pointed at the weights of the model whose turn it is -- per game exactly
the reference's loop, in which the mover's own model searches from a fresh tree (pv_mcts.py:98-104) and the colours
alternate with the game index (evaluate_network.py:69-74).
"""
from shutil import copy

import numpy as np
import torch

from . import _lib
from . import pv_mcts
from .constants import PV_NETWORK_PATH, BOARD_SIZE
from .engine import BatchedSelfPlay
from .game_logic import State
from .pv_network_gnn import GNNNetwork

EN_GAME_COUNT = 15    # Number of games per evaluation (evaluate_network.py:14; originally 400)
EN_TEMPERATURE = 1.0  # Temperature of the Boltzmann distribution (evaluate_network.py:15)


def first_player_point(ended_state):
    """1: first player wins, 0: first player loses, 0.5: draw (evaluate_network.py:18-22)."""
    if ended_state.is_lose():
        return 0 if ended_state.is_first_player() else 1
    return 0.5


def play(next_actions):
    """Execute one game with two action functions (evaluate_network.py:25-44); host loop, reference-shaped."""
    state = State()
    while True:
        if state.is_done():
            break
        next_action = next_actions[0] if state.is_first_player() else next_actions[1]
        action = next_action(state)
        state = state.next(action)
    return first_player_point(state)


def update_best_player():
    """Replace the best player (evaluate_network.py:47-49)."""
    copy(PV_NETWORK_PATH + 'latest.pth', PV_NETWORK_PATH + 'best.pth')
    print('Latest model is better than current best. Replacing best model with latest.')


class BatchedMatch:
    """`num_games` games of player 0 vs player 1 on the batched engine; game i has player (i % 2) moving first
    (evaluate_network.py:69-74).  Players are models (evaluator='gnn') or integer biases of the parity tests' hash
    evaluator (evaluator='fake')."""

    def __init__(self, players, num_games, sims=None, board_size=BOARD_SIZE, temperature=EN_TEMPERATURE,
                 evaluator="gnn", seed=0, device=None):
        self.players = players
        self.evaluator = evaluator
        sims = pv_mcts.PV_EVALUATE_COUNT if sims is None else sims
        counts = [(num_games + 1) // 2, num_games // 2]          # games with player 0 first / player 1 first
        self.engines = []
        for first, g in enumerate(counts):
            if g == 0:
                self.engines.append(None)
                continue
            # (no evaluation cache here: the two players' weights take turns on one engine, a table would mix their outputs)
            kw = dict(num_games=g, sims=sims, board_size=board_size, temperature=temperature, seed=2 * int(seed) + first,
                      device=device, eval_cache_slots=0)
            if evaluator == "gnn":
                eng = BatchedSelfPlay(players[first], **kw)
            else:
                eng = BatchedSelfPlay(None, evaluator="fake", fake_bias=int(players[first]), **kw)
            self.engines.append(eng)
        if evaluator == "gnn":
            dev = next(e for e in self.engines if e is not None).dev
            self._packed = [m.packed_weights(dev) for m in players]
            self._flags = [int(m.gnn_flags(dev)) for m in players]

    def _point_at(self, eng, mover):
        if self.evaluator == "gnn":
            eng.t["packed_weights"] = self._packed[mover]
            eng.e.packed_weights = self._packed[mover].data_ptr()
            eng.e.gnn_flags = self._flags[mover]
        else:
            eng.e.fake_bias = int(self.players[mover])

    def play(self, uniforms=None):
        """Play every game to the end.  uniforms: optional pair of float64 arrays [max_plies, G_first] (parity tests).
        Returns the per-game points of player 0 in game order.
        fp16-range guard: the engines' counters are read after every ply anyway; if a split-kernel launch of either player met a value
        outside fp16 range (counters()['gnn_saturated']) the moves so far were searched with clamped evaluations, so BOTH players are
        marked (mark_saturated), every evaluation switches to the exact f32-input kernels and the match is replayed from ply 0 -- the
        promotion decision of evaluate_network (evaluate_network.py:90-94) is never taken on evaluations that are not the networks'."""
        while True:
            points = self._play_once(uniforms)
            if points is not None:
                return points

    def _switch_to_exact_kernels(self):
        for m in self.players:
            if hasattr(m, "mark_saturated"):
                m.mark_saturated()
        self._flags = [_lib.GNN_EXACT_F32 for _ in self.players]
        for eng in self.engines:
            if eng is not None:
                eng._gnn_flags = _lib.GNN_EXACT_F32
                eng.e.gnn_flags = _lib.GNN_EXACT_F32
                eng.reset()

    def _play_once(self, uniforms):
        live = [e is not None for e in self.engines]
        ply = 0
        while any(live):
            for first, eng in enumerate(self.engines):
                if not live[first]:
                    continue
                self._point_at(eng, first if ply % 2 == 0 else 1 - first)
                eng.move(None if uniforms is None else uniforms[first][ply])
            ply += 1
            for first, eng in enumerate(self.engines):
                if not live[first]:
                    continue
                c = eng.counters()
                if self.evaluator == "gnn" and c["gnn_saturated"] and not all(f & _lib.GNN_EXACT_F32 for f in self._flags):
                    self._switch_to_exact_kernels()
                    return None
                if c["active"] == 0 or ply >= eng.max_plies:
                    live[first] = False
        points = []
        per = []
        for first, eng in enumerate(self.engines):
            if eng is None:
                per.append(np.zeros((0,)))
                continue
            z0 = eng.t["game_result"].cpu().numpy().astype(np.float64)     # first mover's result: +1 / -1 / 0
            fp = (z0 + 1.0) / 2.0                                          # first_player_point
            per.append(fp if first == 0 else 1.0 - fp)                    # evaluate_network.py:71-74
        n = len(per[0]) + len(per[1])
        for i in range(n):
            points.append(float(per[i % 2][i // 2]))
        return points


def evaluate_network():
    """Network evaluation (evaluate_network.py:52-94): latest vs best, promote when the average point exceeds 0.5."""
    model0 = GNNNetwork()
    model0.prep_for_inference(PV_NETWORK_PATH + 'latest.pth')
    model1 = GNNNetwork()
    model1.prep_for_inference(PV_NETWORK_PATH + 'best.pth')
    match = BatchedMatch((model0, model1), EN_GAME_COUNT, temperature=EN_TEMPERATURE,
                         seed=int(np.random.randint(0, 2 ** 30)))
    points = match.play()
    print('Evaluating latest model against current best ({} games, concurrent)'.format(EN_GAME_COUNT))
    average_point = sum(points) / EN_GAME_COUNT
    print('Average points of latest model against current best:', average_point)
    del model0
    del model1
    del match
    torch.cuda.empty_cache()
    if average_point > 0.5:
        update_best_player()
        return True
    return False


if __name__ == '__main__':
    evaluate_network()
