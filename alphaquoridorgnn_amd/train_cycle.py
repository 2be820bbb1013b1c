"""Execution of the learning cycle -- drop-in for the reference's train_cycle.py:10-45 with every stage on the MI355X path:
create_network -> [self_play -> train_network -> evaluate_network] x NUM_TRAIN_CYCLE.  (The reference's commented-out
evaluate_best_player stage -- CPU baseline agents, SURVEY 8f.4 -- is not part of this build.)
"""
from .constants import BOARD_SIZE, PV_NETWORK_NAME
from .evaluate_network import evaluate_network
from .pv_network_gnn import create_network
from .self_play import self_play
from .train_network import train_network

NUM_TRAIN_CYCLE = 1000  # Number of training cycles (train_cycle.py:18)


def train_cycle(num_cycles=None):
    print(f'Model {PV_NETWORK_NAME} on board size {BOARD_SIZE}')
    create_network()
    promoted = []
    n = NUM_TRAIN_CYCLE if num_cycles is None else num_cycles
    for i in range(n):
        print(f'\nBegin training cycle {i + 1}/{n} ====================')
        print('\nBegin self-play ====================')
        self_play()
        print('\nUpdate network parameters ====================')
        train_network()
        print('\nEvaluate new parameters ====================')
        promoted.append(evaluate_network())
    return promoted


if __name__ == '__main__':
    train_cycle()
