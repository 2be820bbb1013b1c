"""Self-play -- drop-in for the reference's self_play.py, running whole generations on the batched HIP engine.

Call surface kept (self_play.py:19-95): SP_GAME_COUNT, SP_TEMPERATURE, first_player_value, write_data, play,
self_play.  `self_play()` plays SP_GAME_COUNT games CONCURRENTLY on this rank's GPU (sharded over ranks when
torch.distributed is initialised: rank r plays its share, then one all-gather of (s, pi, z) over RCCL/xGMI and
rank 0 writes the single history file, like self_play.py:84,:91).
"""
import os
import pickle
from datetime import datetime

import numpy as np
import torch

from . import distributed as aqd
from . import pv_mcts
from .constants import PV_NETWORK_PATH, BOARD_SIZE
from .engine import BatchedSelfPlay, MultiSetSelfPlay, gather_history
from .pv_network_gnn import GNNNetwork, POLICY_OUTPUT_SIZE

SP_GAME_COUNT = 50    # Number of games for self-play (self_play.py:19; 25000 in the original version)
SP_TEMPERATURE = 1.0  # Temperature parameter for Boltzmann distribution (self_play.py:20)


def first_player_value(ended_state):
    """1: first player wins, -1: first player loses, 0: draw (self_play.py:22-27)."""
    if ended_state.is_lose():
        return -1 if ended_state.is_first_player() else 1
    return 0


def write_data(history):
    """Save training data to ./data/YYYYMMDDhhmmss.history (self_play.py:30-37)."""
    now = datetime.now()
    os.makedirs('./data/', exist_ok=True)
    path = './data/{:04}{:02}{:02}{:02}{:02}{:02}.history'.format(
        now.year, now.month, now.day, now.hour, now.minute, now.second)
    with open(path, mode='wb') as f:
        pickle.dump(history, f)
    return path


def _history_rows(states72, visits, z, board_size):
    nw = (board_size - 1) ** 2
    st, vis, zz = states72.cpu().numpy(), visits.cpu().numpy().astype(np.float64), z.cpu().numpy()
    out = []
    for s, v, r in zip(st, vis, zz):
        tot = v.sum()
        pol = (v / tot).tolist() if tot > 0 else [0.0] * v.shape[0]
        out.append([[[int(s[0]), int(s[1])], [int(s[2]), int(s[3])], [int(x) for x in s[4:4 + nw]]], pol, int(r)])
    return out


def play(model, device=None, uniforms=None):
    """Execute one self-play game (self_play.py:40-68) -- a generation of one game on the engine.
    Returns [[state_array, policy list[POLICY_OUTPUT_SIZE], z], ...]."""
    eng = BatchedSelfPlay(model, num_games=1, sims=pv_mcts.PV_EVALUATE_COUNT, board_size=BOARD_SIZE,
                          temperature=SP_TEMPERATURE, seed=int(np.random.randint(0, 2 ** 31 - 1)),
                          evaluator=pv_mcts.evaluator_of(model))
    eng.play_generation(uniforms=uniforms, check_every=1)
    return eng.history()


def _fresh_seed():
    """Base seed of one self_play() call.  The reference draws every move from the unseeded global numpy RNG
    (self_play.py:57), so two generations never repeat; here the engine's uniform stream is seeded per call from
    the same global RNG (or from AQG_SELFPLAY_SEED for reproducible runs).  Under torch.distributed rank 0's draw
    is broadcast, and every rank adds its rank, so the ranks' streams differ and a rerun with the same override
    reproduces the same generation."""
    import torch.distributed as dist
    env = os.environ.get("AQG_SELFPLAY_SEED")
    seed = int(env) if env is not None else int(np.random.randint(0, 2 ** 31 - 1))
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([seed], dtype=torch.int64, device=aqd.collective_device())
        dist.broadcast(t, src=0)
        seed = int(t.item())
    return seed


def self_play(model=None, games=None, seed=None):
    """Perform self-play games and save the training data (self_play.py:71-95).  `seed` (tests) fixes the uniform
    stream; by default every call draws a fresh one, like the reference's unseeded np.random.choice."""
    import torch.distributed as dist
    if model is None:
        model = GNNNetwork()
        model.prep_for_inference(model_path=PV_NETWORK_PATH + 'best.pth')
    total = SP_GAME_COUNT if games is None else games
    distributed = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(), dist.get_world_size()) if distributed else (0, 1)
    base = _fresh_seed() if seed is None else int(seed)
    mine = total // world + (1 if rank < total % world else 0)
    dev = aqd.device()                                   # this rank's GPU (LOCAL_RANK under torchrun), never a bare 'cuda'
    st = torch.zeros((0, 72), dtype=torch.uint8, device=dev)
    vis = torch.zeros((0, POLICY_OUTPUT_SIZE), dtype=torch.int16, device=dev)
    z = torch.zeros((0,), dtype=torch.int8, device=dev)
    if mine > 0:
        # >= 256 games: independent game sets on their own streams fill the holes of each other's serial kernel chains
        eng = MultiSetSelfPlay(model, num_games=mine, sims=pv_mcts.PV_EVALUATE_COUNT, num_sets=None if mine >= 256 else 1,
                               board_size=BOARD_SIZE, temperature=SP_TEMPERATURE, seed=(base + rank) % (2 ** 31 - 1), device=dev)
        c = eng.play_generation()
        print(f'\rSelf-play (rank {rank}: {c["finished"]}/{mine} games)', end='')
        st, vis, z = eng.history_tensors()
    if distributed and os.environ.get("AQG_DIST_LOG") == "1" and rank == 0:
        print(f'exchange step: backend={dist.get_backend()} world={world} collectives={"forced" if aqd.force_group() else "as needed"}')
    if distributed and dist.get_backend() != 'nccl':          # gloo (CPU tests): the exchange runs on host tensors
        st, vis, z = (x.cpu() for x in gather_history(st.cpu(), vis.cpu(), z.cpu()))
    else:
        st, vis, z = gather_history(st, vis, z)
    print('')
    path = None
    if rank == 0:
        path = write_data(_history_rows(st, vis, z, BOARD_SIZE))
    if distributed:
        dist.barrier()          # no rank may go on to train_network.load_data() before rank 0 has finished the file
    del model
    torch.cuda.empty_cache()
    return path


def main(argv=None):
    """`python -m alphaquoridorgnn_amd.self_play` -- also the per-rank program of
    `python -m torch.distributed.run --nproc-per-node N ... -m alphaquoridorgnn_amd.self_play` (distributed.init_from_env binds
    the rank to its GPU and joins the group; one generation sharded over the ranks, one all-gather, rank 0 writes the file)."""
    import argparse
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--games", type=int, default=None, help="games of the generation over ALL ranks (default SP_GAME_COUNT)")
    ap.add_argument("--sims", type=int, default=None, help="simulations per move (default pv_mcts.PV_EVALUATE_COUNT)")
    ap.add_argument("--seed", type=int, default=None, help="fix the uniform streams (default: fresh per call, like the reference)")
    args = ap.parse_args(argv)
    aqd.init_from_env()
    if args.sims is not None:
        pv_mcts.PV_EVALUATE_COUNT = args.sims
    try:
        path = self_play(games=args.games, seed=args.seed)
    except BaseException:
        aqd.shutdown(ok=False)
        raise
    aqd.shutdown()
    return path


if __name__ == '__main__':
    main()
