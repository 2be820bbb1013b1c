"""alphaquoridorgnn_amd -- MI355X-native AlphaZero-Quoridor self-play hot path.

Host-side mirror of the reference's module surface (game_logic / pv_network_gnn / pv_mcts / self_play /
constants) over libaqgnn_hip.so (hand-written gfx950 kernels, C ABI in include/aqgnn.h).  Importing the
package does not touch the GPU; the first hot-path call loads the library and fails loudly if it is missing.
"""
__all__ = ["constants", "game_logic", "pv_network_gnn", "pv_mcts", "self_play", "engine"]
