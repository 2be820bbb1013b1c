"""The learning loop, every stage on the MI355X path -- drop-in for the reference's train_cycle.py:10-45.

    create_network()  ->  repeat NUM_TRAIN_CYCLE times:  self_play()  ->  train_network()  ->  evaluate_network()

(The reference's commented-out evaluate_best_player stage -- CPU baseline agents, SURVEY 8f.4 -- is not part of this build.)
"""
from . import constants
from .evaluate_network import evaluate_network
from .pv_network_gnn import create_network
from .self_play import self_play
from .train_network import train_network

NUM_TRAIN_CYCLE = 1000   # train_cycle.py:18

_STAGES = (("self-play", self_play), ("parameter update", train_network), ("evaluation of the new parameters", evaluate_network))


def _dist():
    import torch.distributed as dist
    on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    return dist, on, (dist.get_rank() if on else 0)


def _create_network_once():
    """create_network() draws random weights: under torch.distributed only rank 0 may write best.pth (every rank would
    otherwise save a different random init to the same path); the others wait for the file."""
    dist, on, rank = _dist()
    if rank == 0:
        create_network()
    if on:
        dist.barrier()


def _evaluate_once():
    """evaluate_network() samples its games from the global numpy RNG: rank 0 plays the match and decides, the decision
    is broadcast, and nobody reads best.pth again before the copy is done."""
    import torch
    dist, on, rank = _dist()
    promoted = evaluate_network() if rank == 0 else False
    if on:
        t = torch.tensor([1 if promoted else 0], dtype=torch.int64, device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
        dist.broadcast(t, src=0)
        promoted = bool(int(t.item()))
        dist.barrier()
    return promoted


def train_cycle(num_cycles=None):
    """Run the cycle; returns, per iteration, whether `latest` was promoted to `best`."""
    total = NUM_TRAIN_CYCLE if num_cycles is None else int(num_cycles)
    print(f'{constants.PV_NETWORK_NAME} network, {constants.BOARD_SIZE}x{constants.BOARD_SIZE} board, {total} training cycle(s)')
    _create_network_once()
    promoted = []
    rank = _dist()[2]
    for cycle in range(1, total + 1):
        outcome = None
        for title, stage in _STAGES:
            if rank == 0:
                print(f'\n[cycle {cycle}/{total}] {title}')
            outcome = _evaluate_once() if stage is evaluate_network else stage()
        promoted.append(bool(outcome))
    return promoted


if __name__ == '__main__':
    train_cycle()
