"""alphaquoridorgnn_amd -- MI355X-native AlphaZero-Quoridor self-play hot path.

Host-side mirror of the reference's module surface (game_logic / pv_network_gnn / pv_mcts / self_play /
constants) over libaqgnn_hip.so (hand-written gfx950 kernels, C ABI in include/aqgnn.h).  Importing the
package does not touch the GPU; the first hot-path call loads the library and fails loudly if it is missing.
"""
import os as _os

# engine.MultiSetSelfPlay drives up to 4 game-set streams next to the default stream; the HIP runtime maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue serialise.  Read once at HIP init, so it
# has to be in the environment before the first GPU call; an explicit setting by the user wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__all__ = ["constants", "game_logic", "pv_network_gnn", "pv_mcts", "self_play", "evaluate_network", "train_network", "train_cycle", "engine", "agents"]
