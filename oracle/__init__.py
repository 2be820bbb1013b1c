"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

A CPU restatement of the reference's hot path, used solely as the checker:
  * quoridor_oracle.c / quoridor.py  : game rules (game_logic.py), plain C + ctypes
  * gnn.py                           : GraphPolicyValueNetwork forward in numpy fp64 (pv_network_gnn.py + PyG GCNConv)
  * mcts.py                          : pv_mcts.py / self_play.play restated iteratively in Python

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Nothing under alphaquoridorgnn_amd/ imports it; the product path is the HIP library and fails loudly
if that library is missing.

Parity pin: quoridor + mcts are pinned by golden vectors generated from the real reference
(tools/gen_golden.py -> tests/golden/).  gnn.py is "parity unpinned": torch_geometric is absent from
this image and the reference holds no outputs for pv_network_gnn.py, so the GNN oracle restates PyG's
published GCNConv definition (see gnn.py header).
"""
