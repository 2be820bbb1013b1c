"""Policy-Value MCTS -- drop-in for the reference's pv_mcts.py on top of the batched HIP engine.

Same call surface (pv_mcts.py:18-109): PV_EVALUATE_COUNT, pv_mcts_policy(model, state, temperature, device),
pv_mcts_action(model, temperature=0, device='cpu'), boltzman(xs, temperature).  A single-state call is a batch
of one game on the GPU engine (csrc/mcts.hip); thousands of states go through `pv_mcts_policy_batch`.
"""
import numpy as np
import torch

from . import _lib
from .engine import BatchedSelfPlay

PV_EVALUATE_COUNT = 50  # Number of simulations per inference (pv_mcts.py:18; "original is 1600")

_engines = {}


def _engine_for(model, num_games, sims, board_size, evaluator="gnn", fake_bias=0):
    key = (id(model), num_games, sims, board_size, evaluator, fake_bias)
    eng = _engines.get(key)
    if eng is None:
        if len(_engines) > 4:
            _engines.clear()
        eng = _engines[key] = BatchedSelfPlay(model, num_games=num_games, sims=sims, board_size=board_size,
                                              evaluator=evaluator, fake_bias=fake_bias, record_history=False)
    elif evaluator == "gnn":
        eng.refresh_weights()
    return eng


def boltzman(xs, temperature):
    """Boltzmann distribution (pv_mcts.py:106-109)."""
    xs = [x ** (1 / temperature) for x in xs]
    return [x / sum(xs) for x in xs]


def _policy_from_visits(visits, temperature):
    if temperature == 0:                               # pv_mcts.py:89-92
        pol = np.zeros(len(visits))
        pol[int(np.argmax(visits))] = 1
        return pol
    return boltzman(visits, temperature)               # pv_mcts.py:93-95


def pv_mcts_policy_batch(model, states72, temperature, sims=None, board_size=None, evaluator="gnn", fake_bias=0):
    """states72: uint8 [B,72] -> list of B policies (each aligned with that state's legal_actions())."""
    states72 = torch.as_tensor(states72, dtype=torch.uint8)
    B = states72.shape[0]
    N = int(states72[0, 70]) if board_size is None else board_size
    eng = _engine_for(model, B, PV_EVALUATE_COUNT if sims is None else sims, N, evaluator, fake_bias)
    visits, actions, count = eng.search(states72)
    visits, count = visits.cpu().numpy(), count.cpu().numpy()
    return [_policy_from_visits([int(v) for v in visits[b, :count[b]]], temperature) for b in range(B)]


def evaluator_of(model):
    """'gnn' for the network the HIP kernels evaluate themselves; 'external' for any other object with the reference's
    predict(state, device) (BaseNetwork.py:36-40) -- e.g. the CNN the reference wires (self_play.py:16,78)."""
    return "gnn" if hasattr(model, "packed_weights") else "external"


def pv_mcts_policy(model, state, temperature, device=None):
    """PUCT MCTS from `state`; returns the improved policy over state.legal_actions() (pv_mcts.py:20-95)."""
    rec = torch.from_numpy(state.record()).unsqueeze(0)
    return pv_mcts_policy_batch(model, rec, temperature, PV_EVALUATE_COUNT, state.N, evaluator=evaluator_of(model))[0]


def pv_mcts_action(model, temperature=0, device='cpu'):
    """Returns a function of the game state that selects an action based on PV-MCTS (pv_mcts.py:98-103)."""
    def pv_mcts_action(state):
        policy = pv_mcts_policy(model, state, temperature, device)
        return np.random.choice(state.legal_actions(), p=policy)
    return pv_mcts_action
