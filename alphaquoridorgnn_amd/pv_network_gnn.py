"""Policy-Value GNN -- drop-in for the reference's pv_network_gnn.py, executed by hand-written gfx950 kernels.

Call surface kept (pv_network_gnn.py:17-80): NUM_FEATURES, HIDDEN_DIM, NUM_GCN_LAYERS, POLICY_OUTPUT_SIZE,
`GraphPolicyValueNetwork(num_features, hidden_dim, num_gcn_layers, policy_output_size)` with submodules
`gcn_layers`, `policy_head`, `value_head`, `forward(x, edge_index, batch) -> (policy, value)`, `create_network()`.
state_dict keys follow PyG's GCNConv (`gcn_layers.i.lin.weight [out,in]`, `gcn_layers.i.bias [out]`).

Added (absent from the reference): `forward_states(states72)` -- the fused board-graph path the
self-play engine uses -- and `GNNNetwork`, the BaseNetwork-style wrapper (BaseNetwork.py:9-54) giving
`predict / prep_for_inference / preprocess_input / name`.

All arithmetic runs in libaqgnn_hip.so (fp32 data; fp16-split or f32-input MFMA, see csrc/gcn_forward.hip); there is no
torch/CPU forward in this file.
"""
import ctypes
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .constants import BOARD_SIZE, PV_NETWORK_PATH
from . import game_logic

NUM_FEATURES = 6      # pv_network_gnn.py:17
HIDDEN_DIM = 128      # :18
NUM_GCN_LAYERS = 3    # :19
POLICY_OUTPUT_SIZE = BOARD_SIZE ** 2 + 2 * (BOARD_SIZE - 1) ** 2  # :20

STATE_DICT_KEYS = [
    "gcn_layers.0.lin.weight", "gcn_layers.0.bias", "gcn_layers.1.lin.weight", "gcn_layers.1.bias",
    "gcn_layers.2.lin.weight", "gcn_layers.2.bias", "policy_head.0.weight", "policy_head.0.bias",
    "policy_head.2.weight", "policy_head.2.bias", "value_head.0.weight", "value_head.0.bias",
    "value_head.2.weight", "value_head.2.bias",
]


class GCNConv(nn.Module):
    """Parameter container with PyG's GCNConv naming and initialisation (lin: Glorot-uniform, no bias; bias: zeros).
    The layer arithmetic is fused into the network-level HIP kernels, so it has no stand-alone forward."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        a = math.sqrt(6.0 / (in_channels + out_channels))
        with torch.no_grad():
            self.lin.weight.uniform_(-a, a)

    def forward(self, x, edge_index):
        raise NotImplementedError("GCNConv is evaluated inside GraphPolicyValueNetwork.forward (fused HIP kernels)")


class GraphPolicyValueNetwork(nn.Module):
    def __init__(self, num_features=NUM_FEATURES, hidden_dim=HIDDEN_DIM, num_gcn_layers=NUM_GCN_LAYERS,
                 policy_output_size=POLICY_OUTPUT_SIZE, board_size=BOARD_SIZE):
        super().__init__()
        if (num_features, hidden_dim, num_gcn_layers) != (NUM_FEATURES, HIDDEN_DIM, NUM_GCN_LAYERS):
            raise ValueError("the HIP kernels are built for the reference configuration 6/128/3 (pv_network_gnn.py:17-19)")
        self.num_features = num_features
        self.hidden_dim = hidden_dim
        self.num_gcn_layers = num_gcn_layers
        self.policy_output_size = policy_output_size
        self.board_size = board_size

        self.gcn_layers = nn.ModuleList()
        self.gcn_layers.append(GCNConv(num_features, hidden_dim))
        for _ in range(num_gcn_layers - 1):
            self.gcn_layers.append(GCNConv(hidden_dim, hidden_dim))
        self.policy_head = nn.Sequential(nn.Linear(hidden_dim, hidden_dim // 2), nn.ReLU(),
                                         nn.Linear(hidden_dim // 2, policy_output_size), nn.Softmax(dim=1))
        self.value_head = nn.Sequential(nn.Linear(hidden_dim, hidden_dim // 2), nn.ReLU(),
                                        nn.Linear(hidden_dim // 2, 1), nn.Tanh())
        self._packed = None
        self._packed_key = None
        self._gnn_flags = 0
        self._sat_words = {}          # device -> int32 [1]: the runtime fp16-range guard's word (aqg_gcn_forward_boards_guarded)

    # ---------------------------------------------------------------- weight packing
    def packed_weights(self, device):
        """float32 device buffer in the kernel layout (include/aqgnn.h); rebuilt when a parameter changes.  Every rebuild
        also re-runs the fp16-range check of this weight set (`gnn_flags`)."""
        sd = self.state_dict()
        key = (str(device),) + tuple((sd[k].data_ptr(), sd[k]._version) for k in STATE_DICT_KEYS)
        if self._packed is None or key != self._packed_key:
            lib = _lib.load()
            host = [sd[k].detach().to("cpu", torch.float32).contiguous() for k in STATE_DICT_KEYS]
            arr = (ctypes.c_void_p * 14)(*[ctypes.c_void_p(t.data_ptr()) for t in host])
            out = torch.empty(lib.aqg_gcn_packed_floats(self.board_size), dtype=torch.float32)
            _lib.check(lib.aqg_gcn_pack_weights_host(self.board_size, arr, ctypes.c_void_p(out.data_ptr())),
                       "aqg_gcn_pack_weights_host")
            self._packed = out.to(device)
            self._packed_key = key
            for w in self._sat_words.values():
                w.zero_()
            self._gnn_flags = self._calibrate(self._packed, device)
            if self._gnn_flags == 0 and self._range_proven(host):
                self._gnn_flags = _lib.GNN_RANGE_PROVEN
        return self._packed

    def _range_proven(self, host):
        """A static bound over ALL inputs (any board graph, any record with at most GNN_PROVEN_MAX_WALLS walls in hand per player) on
        every value the split trunk holds as an fp16 pair (include/aqgnn.h, AQG_GNN_RANGE_PROVEN).  GCNConv is
        P = A_hat (H W^T) + b with A_hat >= 0 and H >= 0 after the ReLU.  A row of A_hat sums to
        1/d_n + sum_{k ~ n} 1/sqrt(d_n d_k) <= 1/d_n + (d_n - 1)/sqrt(2 d_n) <= R = 0.2 + 4/sqrt(10) = 1.465  (closed degrees d <= 5 on
        the grid, and a neighbour of n has d_k >= 2: itself and n), so  |Z_l| <= |W_l| h_{l-1} =: z_l  and  |P_l|, H_l <= R z_l + |b_l|
        =: h_l,  with h_0 = the feature maxima.  The kernel's own images are these times its internal scales: planes
        CQ sqrt(d) H <= 2.10 h, linear-map outputs sqrt(d) Z <= 2.24 z, weights W / CQ <= 1.07 |W|: proven iff
        2.3 max(z_l, h_l, |W_l|) < 65504.  True for the weights, not for a sample of boards: when it holds, the kernels' per-value
        range tracking is redundant and is switched off (the records' wall counts are still checked, on the scalar unit)."""
        if self.board_size != 9:
            return False
        if not all(bool(torch.isfinite(t).all()) for t in host[:6]):
            return False                                  # (Python's max() below would skip a NaN)
        W1, b1, W2, b2, W3, b3 = (t.double().abs() for t in host[:6])
        R = 0.2 + 4.0 / 10.0 ** 0.5
        wmax = float(_lib.GNN_PROVEN_MAX_WALLS)
        h = torch.tensor([1.0, wmax, 1.0, wmax, 1.0, 1.0], dtype=torch.float64)     # pv_network_cnn.py:88-114: one-hot, count, one-hot, count, bit, bit
        worst = 0.0
        for W, b in ((W1, b1), (W2, b2), (W3, b3)):
            z = W @ h
            h = R * z + b
            worst = max(worst, float(z.max()), float(h.max()), float(W.max()))
        return bool(np.isfinite(worst)) and 2.3 * worst < 65504.0

    def invalidate_packed(self):
        """Call after the parameters were changed behind torch's back (train_network.GNNTrainer updates them in place from
        a HIP kernel, which does not bump the tensors' version counters)."""
        self._packed = None
        self._packed_key = None

    def gnn_flags(self, device):
        """Flags every GNN forward of this weight set must carry: 0; _lib.GNN_RANGE_PROVEN when a static bound shows that
        nothing can leave fp16 range (_range_proven: the split kernels then skip their per-value range tracking); or
        _lib.GNN_EXACT_F32 when the fp16-split kernels cannot represent its activations -- found on the calibration boards when
        the set is packed (_calibrate), or at run time by the kernels' range guard (mark_saturated)."""
        self.packed_weights(device)
        return self._gnn_flags

    def saturation_word(self, device):
        """The int32 device word the split kernels OR 1 into when they meet a value outside fp16 range (include/aqgnn.h,
        aqg_gcn_forward_boards_guarded); one per device, zero until that happens."""
        key = str(device)
        if key not in self._sat_words:
            self._sat_words[key] = torch.zeros((1,), dtype=torch.int32, device=device)
        return self._sat_words[key]

    def mark_saturated(self, device=None):
        """The range guard fired for this weight set (a forward of this module, or an engine evaluating with it): from now on --
        until the parameters change -- it is served by the exact f32-input kernels, like a set that fails calibration."""
        self._gnn_flags = _lib.GNN_EXACT_F32
        for w in self._sat_words.values():
            w.zero_()

    _calib_boards = {}

    @classmethod
    def _calibration_boards(cls, device):
        """64 synthetic 9x9 boards spanning the feature range (pawns anywhere, 0..10 walls in hand, 0..20 walls on the
        board, not necessarily legal positions: only the network sees them), plus the start position."""
        key = str(device)
        if key not in cls._calib_boards:
            rng = np.random.RandomState(20250117)
            recs = np.zeros((64, 72), dtype=np.uint8)
            for i in range(64):
                recs[i, 0], recs[i, 2] = rng.randint(0, 81, 2)
                recs[i, 1], recs[i, 3] = (10, 10) if i < 8 else rng.randint(0, 11, 2)
                nw = 0 if i == 0 else rng.randint(0, 21)
                recs[i, 4 + rng.choice(64, nw, replace=False)] = rng.randint(1, 3, nw)
                recs[i, 70] = 9
            recs[0, 0] = recs[0, 2] = 76
            cls._calib_boards[key] = torch.from_numpy(recs).to(device)
        return cls._calib_boards[key]

    def _calibrate(self, packed, device):
        """The default trunk / heads hold every activation as TWO fp16 numbers (hi + lo: fp32-equivalent products on the
        16-bit matrix pipe).  That is exact-enough only while activations stay inside fp16 range; beyond it the kernels
        saturate at 65504 instead of overflowing -- finite, but no longer the network.  The reference's fp32 has no such
        limit, so each weight set is checked once: both kernel families run on the calibration boards, and if their logits
        or values disagree beyond rounding the weight set is served by the exact f32-input MFMA kernels from then on."""
        if self.board_size != 9:
            return 0                                      # smaller boards run the plain f32 kernels anyway
        lib = _lib.load()
        boards = self._calibration_boards(device)
        B, A = boards.shape[0], self.policy_output_size
        res = []
        word = torch.zeros((1,), dtype=torch.int32, device=device)
        for flags in (0, _lib.GNN_EXACT_F32):
            pooled = torch.empty((B, HIDDEN_DIM), dtype=torch.float32, device=device)
            logits = torch.empty((B, A), dtype=torch.float32, device=device)
            vpre = torch.empty((B,), dtype=torch.float32, device=device)
            _lib.check(lib.aqg_gcn_forward_boards_guarded(9, _lib.ptr(boards), 0, B, _lib.ptr(packed), _lib.ptr(pooled), _lib.ptr(logits),
                                                          None, _lib.ptr(vpre), None, flags, _lib.ptr(word), _lib.stream_ptr(device)),
                       "calibration forward")
            res.append(torch.cat([logits, vpre.unsqueeze(1)], 1).double())
        split, exact = res
        ok = bool(((split - exact).abs() <= 1e-4 + 1e-3 * exact.abs()).all())    # NaN compares False
        ok = ok and int(word.item()) == 0                                         # the kernels' own range guard, on the same boards
        return 0 if ok else _lib.GNN_EXACT_F32

    # ---------------------------------------------------------------- fused board path
    def forward_states(self, states72, want_logits=False, state_fmt=0, check_saturation=True):
        """states72: uint8 [B,72] device tensor (state_fmt=0) -> (policy [B,A] softmaxed, value [B,1] tanh'ed);
        with want_logits also returns (logits [B,A], value_pre [B]).
        check_saturation: after a forward on the fp16-split kernels the range guard's word is read back (one 4-byte copy, a
        host sync); if a value left fp16 range the weight set is marked (mark_saturated) and the call is repeated on the
        exact f32-input kernels, so the caller always receives the network's outputs -- the reference's fp32 has no cliff
        (pv_network_gnn.py:53-64).  False skips the read-back (the engine has its own counter, counters()['gnn_saturated'])."""
        dev = _lib.require_gpu(states72.device)
        lib = _lib.load()
        B = states72.shape[0]
        A = self.policy_output_size
        f32 = dict(dtype=torch.float32, device=dev)
        pooled = torch.empty((B, HIDDEN_DIM), **f32)
        policy = torch.empty((B, A), **f32)
        value = torch.empty((B,), **f32)
        logits = torch.empty((B, A), **f32) if want_logits else None
        vpre = torch.empty((B,), **f32) if want_logits else None
        if self.board_size == 9:
            word = self.saturation_word(dev)
            for attempt in range(2):
                flags = self.gnn_flags(dev)
                _lib.check(lib.aqg_gcn_forward_boards_guarded(self.board_size, _lib.ptr(states72), state_fmt, B,
                                                              _lib.ptr(self.packed_weights(dev)), _lib.ptr(pooled), _lib.ptr(logits),
                                                              _lib.ptr(policy), _lib.ptr(vpre), _lib.ptr(value), flags,
                                                              _lib.ptr(word), _lib.stream_ptr(dev)),
                           "aqg_gcn_forward_boards_guarded")
                if (flags & _lib.GNN_EXACT_F32) or not check_saturation or B == 0 or int(word.item()) == 0:
                    break
                self.mark_saturated(dev)          # outside fp16 range: repeat on the exact kernels, and stay there
        else:   # the reference's smaller boards (constants.py:5-20): plain kernels over a caller-owned workspace
            nws = lib.aqg_gcn_boards_any_workspace_floats(self.board_size, B)
            ws = torch.empty((max(int(nws), 1),), **f32)
            _lib.check(lib.aqg_gcn_forward_boards_any(self.board_size, _lib.ptr(states72), state_fmt, B,
                                                      _lib.ptr(self.packed_weights(dev)), _lib.ptr(ws), nws, _lib.ptr(pooled),
                                                      _lib.ptr(logits), _lib.ptr(policy), _lib.ptr(vpre), _lib.ptr(value), 0,
                                                      _lib.stream_ptr(dev)), "aqg_gcn_forward_boards_any")
        if want_logits:
            return policy, value.unsqueeze(1), logits, vpre
        return policy, value.unsqueeze(1)

    # ---------------------------------------------------------------- generic (x, edge_index, batch) path
    @staticmethod
    def _build_csr(edge_index, num_nodes):
        """gcn_norm (PyG defaults: weight 1, add remaining self loops, symmetric normalisation) -> CSR by destination."""
        dev = edge_index.device
        src, dst = edge_index[0].long(), edge_index[1].long()
        has_loop = torch.zeros(num_nodes, dtype=torch.bool, device=dev)
        has_loop[src[src == dst]] = True
        extra = torch.nonzero(~has_loop).flatten()
        src = torch.cat([src, extra])
        dst = torch.cat([dst, extra])
        deg = torch.zeros(num_nodes, dtype=torch.float32, device=dev).index_add_(0, dst, torch.ones_like(dst, dtype=torch.float32))
        dis = deg.pow(-0.5)
        dis[torch.isinf(dis)] = 0
        w = dis[src] * dis[dst]
        order = torch.argsort(dst, stable=True)
        counts = torch.bincount(dst, minlength=num_nodes)
        ptr = torch.zeros(num_nodes + 1, dtype=torch.int32, device=dev)
        ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
        return ptr, src[order].to(torch.int32).contiguous(), w[order].contiguous()

    def forward(self, x, edge_index, batch):
        """pv_network_gnn.py:53-64.  x [sum V, 6] float32, edge_index [2,E] int64, batch [sum V] int64 (sorted)."""
        dev = _lib.require_gpu(x.device)
        lib = _lib.load()
        x = x.to(torch.float32).contiguous()
        n = x.shape[0]
        if x.shape[1] != NUM_FEATURES:
            raise ValueError(f"x must have {NUM_FEATURES} features")
        batch = batch.long()
        if n > 1 and bool((batch[1:] < batch[:-1]).any()):
            raise ValueError("batch must be sorted (PyG Batch convention)")
        G = int(batch.max().item()) + 1 if n else 0
        gptr = torch.zeros(G + 1, dtype=torch.int32, device=dev)
        gptr[1:] = torch.cumsum(torch.bincount(batch, minlength=G), 0).to(torch.int32)
        ptr, csr_src, csr_w = self._build_csr(edge_index.to(dev), n)
        A = self.policy_output_size
        f32 = dict(dtype=torch.float32, device=dev)
        w0, w1 = torch.empty((n, HIDDEN_DIM), **f32), torch.empty((n, HIDDEN_DIM), **f32)
        pooled = torch.empty((G, HIDDEN_DIM), **f32)
        policy, value = torch.empty((G, A), **f32), torch.empty((G,), **f32)
        logits, vpre = torch.empty((G, A), **f32), torch.empty((G,), **f32)
        _lib.check(lib.aqg_gcn_forward_graph(NUM_FEATURES, A, _lib.ptr(x), n, _lib.ptr(ptr), _lib.ptr(csr_src), _lib.ptr(csr_w),
                                             _lib.ptr(gptr), G, _lib.ptr(self.packed_weights(dev)), _lib.ptr(w0), _lib.ptr(w1),
                                             _lib.ptr(pooled), _lib.ptr(logits), _lib.ptr(policy), _lib.ptr(vpre), _lib.ptr(value),
                                             _lib.stream_ptr(dev)), "aqg_gcn_forward_graph")
        self.last_logits, self.last_value_pre = logits, vpre
        return policy, value.unsqueeze(1)


class GNNNetwork(GraphPolicyValueNetwork):
    """BaseNetwork-style wrapper (BaseNetwork.py:9-54; reference implementation of the contract:
    pv_network_cnn.py:50-140) around the GNN.  `optimised_model` of the reference (TensorRT) has no counterpart:
    the HIP kernels are the inference path."""

    def __init__(self):
        super().__init__(NUM_FEATURES, HIDDEN_DIM, NUM_GCN_LAYERS, POLICY_OUTPUT_SIZE)
        self._name = "GNN"

    @property
    def name(self):
        return self._name

    def prep_for_inference(self, model_path):
        """BaseNetwork.py:21-32 minus the TensorRT compile."""
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        self.load_state_dict(torch.load(model_path, map_location=device, weights_only=True))
        self.eval()
        self.to(device)
        if device.type == "cuda":
            self.packed_weights(device)

    def preprocess_input(self, game_state_arrays):
        """List of State.to_array() triples -> uint8 [n,72] state records (the input the GNN kernels accept).
        plies_played is not part of to_array() and is not a network input; it is stored as 0."""
        out = np.zeros((len(game_state_arrays), 72), dtype=np.uint8)
        for i, (player, enemy, walls) in enumerate(game_state_arrays):
            out[i] = game_logic.pack_state72(player, enemy, walls, 0, self.board_size)
        return out

    def predict_batch(self, states72):
        """Batched predict: device uint8 [B,72] -> (policy [B,A] over ALL actions, value [B])."""
        policy, value = self.forward_states(states72)
        return policy, value[:, 0]

    def predict(self, state, device=None):
        """pv_network_cnn.py:117-137: PMF over state.legal_actions() (in that order) as float32 numpy + python float."""
        dev = _lib.require_gpu()
        rec = torch.from_numpy(state.record() if hasattr(state, "record") else
                               game_logic.pack_state72(state.player, state.enemy, state.walls, state.plies_played, state.N)
                               ).to(dev).unsqueeze(0)
        with torch.inference_mode():
            policy, value = self.forward_states(rec)
            _, order, count = game_logic.legal_actions_batch(rec, self.board_size, want_mask=False)
            n = int(count.item())
            pol = policy[0][order[0, :n].long()]
            s = torch.sum(pol)
            pol = pol / (s if s else 1)
        return pol.cpu().numpy(), value.item()

    def train_model(self, data_loader, optimizer, loss_fn, device='cpu', num_epochs=10):
        pass  # stub in the reference as well (pv_network_cnn.py:139-140); training is SURVEY 8(f1), a later row


def create_network():
    """pv_network_gnn.py:68-80 (path taken from constants.PV_NETWORK_PATH like pv_network_cnn.py:144-155)."""
    model_path = PV_NETWORK_PATH + 'best.pth'
    if os.path.exists(model_path):
        return
    model = GraphPolicyValueNetwork(NUM_FEATURES, HIDDEN_DIM, NUM_GCN_LAYERS, POLICY_OUTPUT_SIZE)
    os.makedirs(PV_NETWORK_PATH, exist_ok=True)
    torch.save(model.state_dict(), model_path)


if __name__ == '__main__':
    create_network()
