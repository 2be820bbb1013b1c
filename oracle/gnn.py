"""numpy fp64 restatement of the GNN policy/value forward -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

PARITY UNPINNED: torch_geometric is not installed in this image (no network), the reference pins no
version of it (requirements.txt:1-5 names neither torch nor torch_geometric), and the reference holds no
outputs for pv_network_gnn.py.  This file therefore restates

  * structure: /root/reference/pv_network_gnn.py:23-64 (3x GCNConv+ReLU -> global_mean_pool ->
    policy MLP 128-64-209 + Softmax, value MLP 128-64-1 + Tanh);
  * the layer: PyG's published GCNConv with its defaults (improved=False, add_self_loops=True,
    normalize=True, bias=True):  x' = x W^T (no bias in `lin`);  gcn_norm: every edge weight 1, one
    self-loop of weight 1 added per node that has none, deg[i] = sum of weights of edges INTO i,
    w_e = deg[src]^-1/2 * deg[dst]^-1/2 (inf -> 0);  out[i] = sum_{e=(j->i)} w_e x'[j];  out += bias;
  * global_mean_pool: per-graph mean of node rows;
  * node features: pv_network_cnn.py:88-114 (the 6 planes read as [81, 6] node features);
  * the board graph (absent from the reference): nodes = tiles, directed edges both ways
    between 4-adjacent tiles not separated by a wall under game_logic.py:145-167.

Tolerance used by the tests against this oracle: fp32-MFMA path atol 1e-5 / rtol 1e-4 on the
pre-softmax policy logits and pre-tanh value (stated again in tests/test_gnn_gpu.py).
"""
import numpy as np

KEYS = ["gcn_layers.0.lin.weight", "gcn_layers.0.bias", "gcn_layers.1.lin.weight", "gcn_layers.1.bias",
        "gcn_layers.2.lin.weight", "gcn_layers.2.bias", "policy_head.0.weight", "policy_head.0.bias",
        "policy_head.2.weight", "policy_head.2.bias", "value_head.0.weight", "value_head.0.bias",
        "value_head.2.weight", "value_head.2.bias"]


def node_features(rec):
    """pv_network_cnn.py:88-114 as [V, 6] float64."""
    N = int(rec[70])
    V, S = N * N, N - 1
    x = np.zeros((V, 6), dtype=np.float64)
    x[int(rec[0]), 0] = 1.0
    x[:, 1] = float(rec[1])
    x[int(rec[2]), 2] = 1.0          # enemy pawn in the ENEMY's own frame (:101)
    x[:, 3] = float(rec[3])
    for i in range(S * S):
        w = int(rec[4 + i])
        if w:
            t = N * (i // S) + (i % S)  # :107
            x[t, 3 + w] = 1.0
    return x


def blocked(rec, x, y, nx, ny):
    """game_logic.py:145-167 is_wall_blocking."""
    N = int(rec[70])
    S = N - 1
    w = rec[4:68]
    if nx > x:
        return (y < S and w[x * S + y] == 1) or (y > 0 and w[x * S + y - 1] == 1)
    if nx < x:
        return (y < S and w[(x - 1) * S + y] == 1) or (y > 0 and w[(x - 1) * S + y - 1] == 1)
    if ny > y:
        return (x < S and w[x * S + y] == 2) or (x > 0 and w[(x - 1) * S + y] == 2)
    if ny < y:
        return (x < S and w[x * S + y - 1] == 2) or (x > 0 and w[(x - 1) * S + y - 1] == 2)
    return False


def board_edges(rec):
    """Directed edge list [2, E] (src, dst) of the wall-cut grid graph, no self loops."""
    N = int(rec[70])
    src, dst = [], []
    for x in range(N):
        for y in range(N):
            for dx, dy in ((-1, 0), (1, 0), (0, -1), (0, 1)):
                nx, ny = x + dx, y + dy
                if 0 <= nx < N and 0 <= ny < N and not blocked(rec, x, y, nx, ny):
                    src.append(x * N + y)
                    dst.append(nx * N + ny)
    return np.asarray([src, dst], dtype=np.int64)


def gcn_conv(x, edge_index, W, b):
    """One GCNConv (PyG defaults) in fp64.  x [V,F], edge_index [2,E] (src,dst), W [out,in], b [out]."""
    V = x.shape[0]
    src, dst = edge_index[0], edge_index[1]
    has_loop = np.zeros(V, dtype=bool)
    has_loop[src[src == dst]] = True
    extra = np.nonzero(~has_loop)[0]
    src = np.concatenate([src, extra])
    dst = np.concatenate([dst, extra])
    deg = np.zeros(V, dtype=np.float64)
    np.add.at(deg, dst, 1.0)
    with np.errstate(divide="ignore"):
        dis = deg ** -0.5
    dis[np.isinf(dis)] = 0.0
    w = dis[src] * dis[dst]
    xw = x @ W.T
    out = np.zeros((V, W.shape[0]), dtype=np.float64)
    np.add.at(out, dst, w[:, None] * xw[src])
    return out + b


def forward_graph(params, x, edge_index, batch, num_graphs=None):
    """pv_network_gnn.py:53-64 on an arbitrary (x, edge_index, batch).  Returns dict of fp64 arrays."""
    p = {k: np.asarray(v, dtype=np.float64) for k, v in params.items()}
    h = np.asarray(x, dtype=np.float64)
    for l in range(3):
        h = np.maximum(gcn_conv(h, edge_index, p[f"gcn_layers.{l}.lin.weight"], p[f"gcn_layers.{l}.bias"]), 0.0)
    B = int(batch.max()) + 1 if num_graphs is None else num_graphs
    g = np.zeros((B, h.shape[1]), dtype=np.float64)
    cnt = np.zeros(B, dtype=np.float64)
    np.add.at(g, batch, h)
    np.add.at(cnt, batch, 1.0)
    g = g / np.maximum(cnt, 1.0)[:, None]
    hp = np.maximum(g @ p["policy_head.0.weight"].T + p["policy_head.0.bias"], 0.0)
    logits = hp @ p["policy_head.2.weight"].T + p["policy_head.2.bias"]
    hv = np.maximum(g @ p["value_head.0.weight"].T + p["value_head.0.bias"], 0.0)
    vpre = hv @ p["value_head.2.weight"].T + p["value_head.2.bias"]
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    return dict(pooled=g, logits=logits, value_pre=vpre[:, 0], policy=e / e.sum(axis=1, keepdims=True),
                value=np.tanh(vpre[:, 0]))


def forward_states(params, recs):
    """Forward over a batch of state72 records using the board graph."""
    recs = np.asarray(recs, dtype=np.uint8).reshape(-1, 72)
    xs, es, bs = [], [], []
    off = 0
    for b, rec in enumerate(recs):
        x = node_features(rec)
        e = board_edges(rec)
        xs.append(x)
        es.append(e + off)
        bs.append(np.full(x.shape[0], b, dtype=np.int64))
        off += x.shape[0]
    return forward_graph(params, np.concatenate(xs), np.concatenate(es, axis=1), np.concatenate(bs), len(recs))


def forward_states_dense(params, recs, chunk=1024):
    """The same forward as forward_states(), vectorised over boards with dense [V, V] adjacencies (one fp64 batched matmul per
    layer instead of a Python loop per tile): what the GPU parity tests use at the full BASELINE sizes (4,096 boards in
    seconds).  Same definitions, re-derived independently of the edge-list code above: tests/test_oracle_golden.py checks
    that the two agree to 1e-12 on reference-walk states, so the slow form stays the statement and this one the tool."""
    recs = np.asarray(recs, dtype=np.uint8).reshape(-1, 72)
    p = {k: np.asarray(v, dtype=np.float64) for k, v in params.items()}
    outs, margins = [], []
    for c0 in range(0, recs.shape[0], chunk):
        r = recs[c0:c0 + chunk]
        B = r.shape[0]
        N = int(r[0, 70])
        assert (r[:, 70] == N).all()
        V, S = N * N, N - 1
        W = r[:, 4:4 + S * S].reshape(B, S, S)
        Hh = np.zeros((B, S, N + 1), dtype=bool)          # horizontal wall at slot (x, y), columns padded on both sides
        Hh[:, :, 1:N] = W == 1
        Vv = np.zeros((B, N + 1, S), dtype=bool)          # vertical wall at slot (x, y), rows padded on both sides
        Vv[:, 1:N, :] = W == 2
        # game_logic.py:145-167: the step (x, y) -> (x + 1, y) is blocked by a horizontal wall at slot (x, y) or (x, y - 1);
        # the step (x, y) -> (x, y + 1) by a vertical wall at slot (x, y) or (x - 1, y)
        down = ~(Hh[:, :, 1:] | Hh[:, :, :N])             # [B, S, N] open between rows x and x + 1 at column y
        right = ~(Vv[:, 1:, :] | Vv[:, :N, :])            # [B, N, S] open between columns y and y + 1 at row x
        A = np.zeros((B, V, V), dtype=np.float64)
        bi = np.arange(B)[:, None, None]
        t = (np.arange(S)[:, None] * N + np.arange(N)[None, :])[None]           # tile (x, y), x < S
        A[bi, t, t + N] = down
        A[bi, t + N, t] = down
        t = (np.arange(N)[:, None] * N + np.arange(S)[None, :])[None]           # tile (x, y), y < S
        A[bi, t, t + 1] = right
        A[bi, t + 1, t] = right
        A[:, np.arange(V), np.arange(V)] = 1.0            # gcn_norm: one self loop of weight 1 per node
        dis = A.sum(2) ** -0.5                            # deg[i] = weights of edges INTO i (symmetric here), >= 1
        An = dis[:, :, None] * A * dis[:, None, :]
        x = np.zeros((B, V, 6), dtype=np.float64)         # pv_network_cnn.py:88-114
        rows = np.arange(B)
        x[rows, r[:, 0].astype(int), 0] = 1.0
        x[:, :, 1] = r[:, 1].astype(np.float64)[:, None]
        x[rows, r[:, 2].astype(int), 2] = 1.0
        x[:, :, 3] = r[:, 3].astype(np.float64)[:, None]
        tw = (N * (np.arange(S * S) // S) + np.arange(S * S) % S)               # :107
        x[:, tw, 4] = (r[:, 4:4 + S * S] == 1)
        x[:, tw, 5] = (r[:, 4:4 + S * S] == 2)
        h = x
        margin = np.full(B, np.inf)
        for l in range(3):
            pre = An @ (h @ p[f"gcn_layers.{l}.lin.weight"].T) + p[f"gcn_layers.{l}.bias"]
            margin = np.minimum(margin, _kink_margin(pre.reshape(B, -1)))
            h = np.maximum(pre, 0.0)
        outs.append(h.mean(1))
        margins.append(margin)
    g = np.concatenate(outs, 0) if outs else np.zeros((0, 128))
    pre_p = g @ p["policy_head.0.weight"].T + p["policy_head.0.bias"]
    hp = np.maximum(pre_p, 0.0)
    logits = hp @ p["policy_head.2.weight"].T + p["policy_head.2.bias"]
    pre_v = g @ p["value_head.0.weight"].T + p["value_head.0.bias"]
    hv = np.maximum(pre_v, 0.0)
    vpre = hv @ p["value_head.2.weight"].T + p["value_head.2.bias"]
    e = np.exp(logits - logits.max(axis=1, keepdims=True))
    margin = np.concatenate(margins) if margins else np.zeros((0,))
    if margin.size:
        margin = np.minimum(margin, np.minimum(_kink_margin(pre_p), _kink_margin(pre_v)))
    return dict(pooled=g, logits=logits, value_pre=vpre[:, 0], policy=e / e.sum(axis=1, keepdims=True),
                value=np.tanh(vpre[:, 0]), relu_margin=margin)


def _kink_margin(pre):
    """Per row: the smallest NON-ZERO |pre-activation| (an exact zero takes the same ReLU branch in every arithmetic)."""
    a = np.abs(pre)
    a = np.where(a == 0.0, np.inf, a)
    return a.min(axis=1)


def relu_margins(params, recs):
    """Distance of every position from its nearest ReLU kink: min non-zero |pre-activation| over the three GCN layers and the
    two head hidden layers (fp64).  Below ~1e-7 the branch -- and with it one whole element of the backward pass -- is decided
    by rounding in ANY fp32 implementation, the reference's own included; the gradient parity tests draw their positions
    from those at least 1e-6 away (tests/test_gpu_parity.py::_train_batch)."""
    return forward_states_dense(params, recs)["relu_margin"]


def init_params(seed=0, N=9, num_features=6, hidden=128, layers=3):
    """Random-init weights of the reference architecture: GCN `lin` Glorot-uniform, GCN bias zero (PyG),
    heads default nn.Linear init (kaiming-uniform a=sqrt(5) => U(-1/sqrt(fan_in), 1/sqrt(fan_in)))."""
    rng = np.random.RandomState(seed)
    A = N * N + 2 * (N - 1) ** 2
    p = {}
    fin = num_features
    for l in range(layers):
        a = np.sqrt(6.0 / (fin + hidden))
        p[f"gcn_layers.{l}.lin.weight"] = rng.uniform(-a, a, size=(hidden, fin)).astype(np.float32)
        p[f"gcn_layers.{l}.bias"] = np.zeros(hidden, dtype=np.float32)
        fin = hidden
    for name, (o, i) in {"policy_head.0": (hidden // 2, hidden), "policy_head.2": (A, hidden // 2),
                         "value_head.0": (hidden // 2, hidden), "value_head.2": (1, hidden // 2)}.items():
        a = 1.0 / np.sqrt(i)
        p[name + ".weight"] = rng.uniform(-a, a, size=(o, i)).astype(np.float32)
        p[name + ".bias"] = rng.uniform(-a, a, size=(o,)).astype(np.float32)
    return p


class OracleModel:
    """predict() contract of pv_network_cnn.py:117-137 on top of the fp64 forward above (used by the tests and by
    bench.py's cpu_baseline leg together with oracle/mcts.py)."""

    def __init__(self, params):
        self.params = params
        self.calls = 0

    def predict(self, state, device=None):
        self.calls += 1
        out = forward_states(self.params, state.rec[None])
        legal = state.legal_actions()
        pol = out["policy"][0][legal].astype(np.float32)
        s = pol.sum()
        pol = pol / (s if s else 1)
        return pol.astype(np.float32), float(np.float32(out["value"][0]))
