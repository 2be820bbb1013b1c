"""Configuration constants -- mirror of the reference's constants.py:13-23, defaulting to the 9x9 block
(constants.py:17-20) that every BASELINE.json config is quoted on.  Override with the environment variable
AQG_BOARD_SIZE (3, 5, 7 or 9) before import, the way the reference is switched by editing its file."""
import os

_BOARDS = {3: (1, 14), 5: (2, 28), 7: (6, 70), 9: (10, 116)}  # N -> (NUM_WALLS, NUM_PLIES_FOR_DRAW); 7x7 is ours

BOARD_SIZE = int(os.environ.get("AQG_BOARD_SIZE", "9"))
if BOARD_SIZE not in _BOARDS:
    raise ValueError("AQG_BOARD_SIZE must be one of 3, 5, 7, 9")
NUM_WALLS, NUM_PLIES_FOR_DRAW = _BOARDS[BOARD_SIZE]

PV_NETWORK_NAME = "GNN"  # which network to use (the reference ships 'CNN'; this build wires the GNN)
PV_NETWORK_PATH = f"models/{PV_NETWORK_NAME}/{BOARD_SIZE}x{BOARD_SIZE}/"  # path for network weights


def board_params(board_size):
    return _BOARDS[board_size]
