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


def train_cycle(num_cycles=None):
    """Run the cycle; returns, per iteration, whether `latest` was promoted to `best`."""
    total = NUM_TRAIN_CYCLE if num_cycles is None else int(num_cycles)
    print(f'{constants.PV_NETWORK_NAME} network, {constants.BOARD_SIZE}x{constants.BOARD_SIZE} board, {total} training cycle(s)')
    create_network()
    promoted = []
    for cycle in range(1, total + 1):
        outcome = None
        for title, stage in _STAGES:
            print(f'\n[cycle {cycle}/{total}] {title}')
            outcome = stage()
        promoted.append(bool(outcome))
    return promoted


if __name__ == '__main__':
    train_cycle()
