"""Batched self-play engine: thousands of concurrent PV-MCTS games per MI355X, lock-step simulations.

Host side of aqg_engine_* (include/aqgnn.h).  All state lives in HBM as torch tensors owned by this object;
every simulation of every game is enqueued by one C call per move (no per-simulation Python, no host sync
inside a move).  Per-game semantics equal the reference's sequential `pv_mcts_policy` + `play()`
(pv_mcts.py:20-95, self_play.py:40-68); see csrc/mcts.hip.

Sharding: games are independent, so rank r of W simply owns its own BatchedSelfPlay with its own
uniform stream; `gather_history` is the single exchange step per generation (all-gather over RCCL/xGMI).
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from .constants import BOARD_SIZE, board_params


class BatchedSelfPlay:
    def __init__(self, model=None, num_games=2048, sims=50, board_size=BOARD_SIZE, device=None, temperature=1.0,
                 c_puct=1.25, evaluator="gnn", fake_bias=0, seed=0, record_history=True, quota=None, eval_cache_slots=None):
        """model: GraphPolicyValueNetwork/GNNNetwork (evaluator='gnn'); evaluator='fake' runs the integer-hash
        evaluator used by the parity tests (oracle/mcts.py FakeModel); evaluator='external' calls `model.predict(state,
        device)` -- ANY object honouring the reference's BaseNetwork contract (BaseNetwork.py:36-40), e.g. a stock CNN --
        once per simulation and game from the host, exactly like pv_mcts.py:47 (plumbing path: one host round trip per
        simulation).
        eval_cache_slots (evaluator='gnn' only; a power of two >= 64, 0 = off; None = the environment's AQG_EVAL_CACHE_SLOTS, else
        off): entries per game slot of the evaluation cache
        (include/aqgnn.h, `eval_cache_keys`): a leaf whose position this slot has already sent through the network is expanded from
        the stored priors / value / legal list -- bit-identical searches and game records, fewer network evaluations (736 bytes of
        HBM per entry)."""
        self.dev = _lib.require_gpu(device)
        self.lib = _lib.load()
        self.N = board_size
        self.A = board_size ** 2 + 2 * (board_size - 1) ** 2
        self.num_walls, self.plies_for_draw = board_params(board_size)
        self.G, self.sims = int(num_games), int(sims)
        # quota > G: the slots are refilled -- a slot whose game has ended takes the next game not yet handed out (in slot
        # order, deterministic) until `quota` games have been started: the reference's loop over games (self_play.py:81-84)
        # on G concurrent slots instead of lock-step generations that idle every finished slot until the longest game ends
        self.quota = self.G if quota is None else int(quota)
        if self.quota < self.G:
            raise ValueError("quota must be >= num_games")
        self.node_cap = 1 + self.sims * _lib.MAX_LEGAL
        self.max_plies = self.plies_for_draw
        self.model = model
        self.evaluator = evaluator
        self.gen = torch.Generator(device=self.dev)
        self.gen.manual_seed(int(seed))
        G, cap, dev, Q = self.G, self.node_cap, self.dev, self.quota

        def z(shape, dtype):
            return torch.zeros(shape, dtype=dtype, device=dev)

        t = self.t = {}
        t["node_rec"] = z((G * cap, 4), torch.float64)     # 32-byte node records (csrc/mcts.hip NodeRec)
        t["node_count"] = z((G,), torch.int32)
        t["root_state"] = z((G, 24), torch.uint8)
        t["path"] = z((G, self.sims + 2), torch.int32)
        t["path_len"] = z((G,), torch.int32)
        t["leaf_flag"] = z((G,), torch.uint8)
        t["leaf_state"] = z((G, 24), torch.uint8)
        t["game_active"] = z((G,), torch.uint8)
        t["slot_game"] = z((G,), torch.int32)
        t["game_plies"] = z((Q,), torch.int32)
        t["game_result"] = z((Q,), torch.int8)
        t["game_done"] = z((Q,), torch.uint8)
        t["game_slot"] = z((Q,), torch.int32)
        t["game_first_move"] = z((Q,), torch.int32)
        t["legal_order"] = z((G, _lib.MAX_LEGAL), torch.uint8)
        t["legal_count"] = z((G,), torch.int32)
        t["pooled"] = z((G, 128), torch.float32)
        t["policy"] = z((G, self.A), torch.float32)   # also holds the fake evaluator's legal-ordered priors (count <= A)
        t["value"] = z((G,), torch.float32)
        hp = self.max_plies if record_history else 1
        t["hist_state72"] = z((Q, hp, 72), torch.uint8)
        t["hist_visits"] = z((Q, hp, self.A), torch.int16)
        t["hist_action"] = z((Q, hp), torch.uint8)
        t["counters"] = z((8,), torch.int32)
        t["stat_leaf_evals"] = z((G,), torch.int32)
        t["stat_terminal_sims"] = z((G,), torch.int32)
        if evaluator == "gnn":
            if model is None:
                raise ValueError("evaluator='gnn' needs a model")
            t["packed_weights"] = model.packed_weights(dev)
            self._gnn_flags = model.gnn_flags(dev)
            if self.N != 9:      # smaller boards run the any-size forward, which needs a caller-owned workspace
                t["gnn_workspace"] = z((int(self.lib.aqg_gcn_boards_any_workspace_floats(self.N, G)),), torch.float32)
        else:
            if evaluator == "external" and (model is None or not hasattr(model, "predict")):
                raise ValueError("evaluator='external' needs a model with predict(state, device)")
            t["packed_weights"] = z((4,), torch.float32)
            self._gnn_flags = 0

        if eval_cache_slots is None:              # opt-in for whole programs (self_play, train_cycle, pv_mcts): one environment variable
            eval_cache_slots = int(os.environ.get("AQG_EVAL_CACHE_SLOTS", "0")) if evaluator == "gnn" else 0
        self.eval_cache_slots = int(eval_cache_slots)
        if self.eval_cache_slots:
            if evaluator != "gnn":
                raise ValueError("eval_cache_slots needs evaluator='gnn' (the table stores network outputs)")
            if self.eval_cache_slots < 64 or self.eval_cache_slots & (self.eval_cache_slots - 1) or self.eval_cache_slots > (1 << 20):
                raise ValueError("eval_cache_slots must be a power of two in 64 .. 2**20")
            t["eval_cache_keys"] = z((G * self.eval_cache_slots, 32), torch.uint8)
            t["eval_cache_rows"] = torch.empty((G * self.eval_cache_slots, 704), dtype=torch.uint8, device=dev)
            t["eval_cache_slot"] = torch.full((G,), -1, dtype=torch.int32, device=dev)
            t["eval_mask"] = z((G,), torch.uint8)
            t["stat_cache_hits"] = z((G,), torch.int32)
            t["eval_list"] = z((G,), torch.int32)
            t["eval_count"] = z((self.sims + 1,), torch.int32)

        e = self.e = _lib.EngineStruct()
        e.board_size, e.num_walls, e.plies_for_draw = self.N, self.num_walls, self.plies_for_draw
        e.num_games, e.quota, e.sims, e.node_cap = G, Q, self.sims, cap
        e.max_plies = hp if record_history else 0
        e.prior_mode = {"gnn": 0, "fake": 1, "external": 2}[evaluator]
        e.fake_bias = int(fake_bias)
        e.gnn_flags = int(self._gnn_flags)
        e.c_puct, e.temperature = float(c_puct), float(temperature)
        for name in ("node_rec", "node_count", "root_state", "path",
                     "path_len", "leaf_flag", "leaf_state", "game_active", "slot_game", "game_plies", "game_result", "game_done",
                     "game_slot", "game_first_move", "legal_order",
                     "legal_count", "pooled", "policy", "value", "hist_state72", "hist_visits", "hist_action", "counters",
                     "stat_leaf_evals", "stat_terminal_sims", "packed_weights"):
            setattr(e, name, t[name].data_ptr())
        e.gnn_workspace = t["gnn_workspace"].data_ptr() if "gnn_workspace" in t else None
        if self.eval_cache_slots:
            for name in ("eval_cache_keys", "eval_cache_rows", "eval_cache_slot", "eval_mask", "stat_cache_hits", "eval_list", "eval_count"):
                setattr(e, name, t[name].data_ptr())
            e.eval_cache_log2 = self.eval_cache_slots.bit_length() - 1
        self.record_history = record_history
        self.moves_done = 0
        self.reset()

    # ------------------------------------------------------------------
    def _stream(self):
        return _lib.stream_ptr(self.dev)

    def reset(self):
        _lib.check(self.lib.aqg_engine_reset(ctypes.byref(self.e), self._stream()), "aqg_engine_reset")
        self.moves_done = 0

    def refresh_weights(self):
        if self.evaluator == "gnn":
            new = self.model.packed_weights(self.dev)        # the SAME tensor object while no parameter has changed
            flags = int(self.model.gnn_flags(self.dev))
            changed = new is not self.t["packed_weights"] or flags != int(self.e.gnn_flags)
            self.t["packed_weights"] = new
            self.e.packed_weights = new.data_ptr()
            self.e.gnn_flags = flags
            if self.eval_cache_slots and changed:      # the table holds the OLD weights' (or the other kernel build's) outputs
                _lib.check(self.lib.aqg_engine_clear_eval_cache(ctypes.byref(self.e), self._stream()), "aqg_engine_clear_eval_cache")

    def move(self, uniforms=None):
        """One move for every active game.  uniforms: float64 [G] in [0,1) (default: device RNG stream)."""
        if uniforms is None:
            uniforms = torch.rand((self.G,), dtype=torch.float64, device=self.dev, generator=self.gen)
        else:
            uniforms = torch.as_tensor(uniforms, dtype=torch.float64).to(self.dev).contiguous()
        self._u = uniforms  # keep alive until the stream has consumed it
        if self.evaluator == "external":
            self._external_sims()
            _lib.check(self.lib.aqg_engine_finish_move(ctypes.byref(self.e), _lib.ptr(uniforms), self._stream()), "aqg_engine_finish_move")
        else:
            _lib.check(self.lib.aqg_engine_move(ctypes.byref(self.e), _lib.ptr(uniforms), self._stream()), "aqg_engine_move")
        self.moves_done += 1

    # ------------------------------------------------------------------ external evaluator (prior_mode 2)
    def _leaf_states(self, idx):
        """game_logic.State objects of the leaves of games `idx` (24-byte packed records: wall masks, pawns, plies)."""
        from .game_logic import State
        raw = self.t["leaf_state"][idx].cpu().numpy().view(np.uint64).reshape(-1, 3)
        nw = (self.N - 1) ** 2
        out = []
        for hw, vw, m in raw:
            hw, vw, m = int(hw), int(vw), int(m)
            walls = [(1 if (hw >> i) & 1 else 0) + (2 if (vw >> i) & 1 else 0) for i in range(nw)]
            out.append(State(board_size=self.N, player=[m & 0xFF, (m >> 8) & 0xFF], enemy=[(m >> 16) & 0xFF, (m >> 24) & 0xFF],
                             walls=walls, plies_played=(m >> 32) & 0xFFFF))
        return out

    def _external_sims(self, device=None):
        """pv_mcts.py:84-85 with the caller's model: per simulation the engine selects a leaf per game, the host asks
        model.predict(state, device) for each (pv_mcts.py:47) and hands the PMF over the leaf's legal actions + the value back."""
        e, st = ctypes.byref(self.e), self._stream()
        _lib.check(self.lib.aqg_engine_begin_move(e, st), "aqg_engine_begin_move")
        for sim in range(self.sims):
            _lib.check(self.lib.aqg_engine_step(e, 1 if sim else 0, 1, st), "aqg_engine_step")
            idx = torch.nonzero(self.t["leaf_flag"] == 1).flatten()
            if idx.numel() == 0:
                continue
            counts = self.t["legal_count"][idx].cpu().numpy()
            pol = torch.zeros((idx.numel(), self.A), dtype=torch.float32)
            val = torch.zeros((idx.numel(),), dtype=torch.float32)
            for j, state in enumerate(self._leaf_states(idx)):
                p, v = self.model.predict(state, device if device is not None else "cpu")
                p = np.asarray(p, dtype=np.float32)
                if p.shape[0] != counts[j]:
                    raise ValueError("model.predict must return one prior per legal action (BaseNetwork.py:36-40)")
                pol[j, :p.shape[0]] = torch.from_numpy(p)
                val[j] = float(v)
            self.t["policy"][idx] = pol.to(self.dev)
            self.t["value"][idx] = val.to(self.dev)
        _lib.check(self.lib.aqg_engine_step(e, 1, 0, st), "aqg_engine_step")

    def counters(self):
        c = self.t["counters"].cpu().numpy()
        return dict(active=int(c[0]), finished=int(c[1]), dead_ends=int(c[2]), started=int(c[3]), moves=int(c[4]),
                    gnn_saturated=int(c[5]),      # the split kernels' fp16-range guard fired during this generation (aqgnn.h counters[5])
                    leaf_evals=int(self.t["stat_leaf_evals"].sum().item()),
                    cache_hits=int(self.t["stat_cache_hits"].sum().item()) if self.eval_cache_slots else 0,
                    terminal_sims=int(self.t["stat_terminal_sims"].sum().item()))

    def _fall_back_to_exact_kernels(self):
        """The range guard fired: this weight set leaves fp16 range on positions of this generation.  Its evaluations so far
        were finite but not the network's, so the generation is played again from the start on the exact f32-input kernels
        (the reference's fp32 has no such cliff, pv_network_gnn.py:53-64)."""
        if self.model is not None and hasattr(self.model, "mark_saturated"):
            self.model.mark_saturated(self.dev)
        self._gnn_flags = _lib.GNN_EXACT_F32
        self.e.gnn_flags = _lib.GNN_EXACT_F32
        self.reset()

    def play_generation(self, uniforms=None, check_every=4):
        """Play the whole quota (== every slot once when quota == num_games: one self_play generation's worth of games
        on this rank).  uniforms: optional float64 [moves, G] (parity tests).  Returns the counters dict."""
        ply = 0
        limit = self.max_plies * (-(-self.quota // self.G))          # every slot plays at most ceil(quota / G) games
        while True:
            self.move(None if uniforms is None else uniforms[ply])
            ply += 1
            if ply >= limit or ply % check_every == 0:
                c = self.counters()
                if c["gnn_saturated"] and not (self.e.gnn_flags & _lib.GNN_EXACT_F32):
                    self._fall_back_to_exact_kernels()
                    ply = 0
                    continue
                if c["active"] == 0 or ply >= limit:
                    break
        return self.counters()

    # ------------------------------------------------------------------ search only (pv_mcts_policy)
    def search(self, root_states72, check_saturation=True):
        """pv_mcts_policy for G roots at once: (visits i32 [G, MAX_LEGAL], actions u8, count i32).
        check_saturation (GNN evaluator): the fp16-range guard's word (counters[5]) is cleared before the search and read back after it
        (one 4-byte copy: the callers read the visit counts back anyway); if a launch of this search met a value outside fp16 range its
        evaluations were finite but not the network's, so the weight set is marked (model.mark_saturated), the engine switches to the
        exact f32-input kernels and the search is repeated -- the caller never receives visit counts of clamped evaluations."""
        roots = torch.as_tensor(root_states72, dtype=torch.uint8).to(self.dev).contiguous().view(self.G, 72)
        self._roots = roots
        for attempt in range(2):
            guarded = check_saturation and self.evaluator == "gnn" and not (self.e.gnn_flags & _lib.GNN_EXACT_F32)
            if guarded:
                self.t["counters"][5] = 0          # a cached engine (pv_mcts._engines) must not inherit an earlier search's word
            if self.evaluator == "external":
                _lib.check(self.lib.aqg_engine_set_roots(ctypes.byref(self.e), _lib.ptr(roots), self._stream()), "aqg_engine_set_roots")
                self._external_sims()
            else:
                _lib.check(self.lib.aqg_engine_search(ctypes.byref(self.e), _lib.ptr(roots), self._stream()), "aqg_engine_search")
            if not guarded or int(self.t["counters"][5].item()) == 0:
                break
            if self.model is not None and hasattr(self.model, "mark_saturated"):
                self.model.mark_saturated(self.dev)
            self._gnn_flags = _lib.GNN_EXACT_F32
            self.e.gnn_flags = _lib.GNN_EXACT_F32
            self.t["counters"][5] = 0
            if self.eval_cache_slots:      # the table holds the clamped evaluations of the split kernels
                _lib.check(self.lib.aqg_engine_clear_eval_cache(ctypes.byref(self.e), self._stream()), "aqg_engine_clear_eval_cache")
        visits = torch.empty((self.G, _lib.MAX_LEGAL), dtype=torch.int32, device=self.dev)
        actions = torch.empty((self.G, _lib.MAX_LEGAL), dtype=torch.uint8, device=self.dev)
        count = torch.empty((self.G,), dtype=torch.int32, device=self.dev)
        _lib.check(self.lib.aqg_engine_root_visits(ctypes.byref(self.e), _lib.ptr(visits), _lib.ptr(actions), _lib.ptr(count),
                                                   self._stream()), "aqg_engine_root_visits")
        return visits, actions, count

    # ------------------------------------------------------------------ history
    def history_tensors(self):
        """(states72 u8 [P,72], visits i16 [P,A], z i8 [P]) for all finished games on this rank, game-major (game index =
        order in which the games were started)."""
        plies = self.t["game_plies"].long()
        hp = self.t["hist_state72"].shape[1]
        idx = torch.arange(hp, device=self.dev).unsqueeze(0)
        valid = (idx < plies.unsqueeze(1)) & (self.t["game_done"] != 0).unsqueeze(1)
        z0 = self.t["game_result"].to(torch.int8).unsqueeze(1)
        sign = torch.where(idx % 2 == 0, 1, -1).to(torch.int8)      # z alternates down the history (self_play.py:63-66)
        z = (z0 * sign)[valid]
        return self.t["hist_state72"][valid], self.t["hist_visits"][valid], z

    def history(self):
        """Reference-format history: list of [[player, enemy, walls], policy list[A] (python floats), z]
        (self_play.py:51-54,:63-66).  policy_i = n_i / sum(n) in float64, exactly like boltzman at T=1."""
        st, vis, z = (x.cpu().numpy() for x in self.history_tensors())
        nw = (self.N - 1) ** 2
        out = []
        for s, v, zz in zip(st, vis, z):
            v = v.astype(np.float64)
            tot = v.sum()
            pol = (v / tot).tolist() if tot > 0 else [0.0] * self.A
            out.append([[[int(s[0]), int(s[1])], [int(s[2]), int(s[3])], [int(x) for x in s[4:4 + nw]]], pol, int(zz)])
        return out


_SET_STREAMS = {}     # (device index, number of sets) -> the streams every MultiSetSelfPlay of that shape runs on


class MultiSetSelfPlay:
    """K independent BatchedSelfPlay sets of a rank's games, each on its own HIP stream.

    One set is a strictly serial chain per simulation (step -> trunk -> heads): while its step / heads kernels run, most of
    the chip idles, and the trunk's last boards leave CUs empty.  Games are independent, so splitting them into K sets whose
    move() calls are enqueued round-robin on K streams lets the GPU fill those holes with another set's kernels (2048 games x
    200 sims, hipGraph replay on: K = 1 / 2 / 3 / 4: ~1,000 / 1,330 / 1,360 / 1,590 games/s).  K = 4 needs
    GPU_MAX_HW_QUEUES >= 5 (set to 8 by the package unless the user chose a value): with the runtime's default of 4 hardware
    queues two of the streams share a queue and serialise; K >= 5 collapses (~550-600 games/s): the GPU runs four compute
    queues concurrently.  What the overlap can and cannot do is measured in DESIGN.md section 4 K4 (a round of the four sets
    costs about the SUM of their trunk launches: two resident trunk workgroups fill a CU's register file, so the other
    sets' kernels run in the trunks' shadow).  Set k is bit-identical to a stand-alone
    BatchedSelfPlay(num_games_k, seed = seed * 64 + k): nothing is shared but the read-only packed weights (and the K
    streams, which every engine of the process reuses -- see _SET_STREAMS)."""

    def __init__(self, model=None, num_games=2048, sims=50, num_sets=None, seed=0, device=None, quota=None, **kw):
        self.dev = _lib.require_gpu(device)
        if num_sets is None:                      # one hardware queue per set + the default stream, or fall back to 2
            num_sets = 4 if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) >= 5 else 2
        k = max(1, min(int(num_sets), int(num_games)))
        sizes = [num_games // k + (1 if i < num_games % k else 0) for i in range(k)]
        quota = num_games if quota is None else int(quota)
        quotas = [quota // k + (1 if i < quota % k else 0) for i in range(k)]      # each set refills its own slots
        # The K streams are created once per device and reused by every MultiSetSelfPlay: torch hands out streams from a pool
        # that is never destroyed, and the runtime maps streams onto GPU_MAX_HW_QUEUES (8) hardware queues -- a second engine
        # with four NEW streams (e.g. the next generation's) would share queues with the first one's idle streams, two of its
        # sets would serialise, and the generation would run 35 % slower (measured: 1,030 vs 1,530 games/s).
        key = (self.dev.index, k)
        if key not in _SET_STREAMS:
            _SET_STREAMS[key] = [torch.cuda.Stream(device=self.dev) for _ in sizes]
        self.streams = _SET_STREAMS[key]
        if model is not None and kw.get("evaluator", "gnn") == "gnn":
            model.packed_weights(self.dev)        # pack + calibrate (two forwards and a host sync) on the caller's stream, not inside set 0's
        self.sets = []
        ready = torch.cuda.current_stream(self.dev).record_event()   # e.g. the model's weight upload on the caller's stream
        for i, g in enumerate(sizes):
            with torch.cuda.stream(self.streams[i]):
                self.streams[i].wait_event(ready)
                self.sets.append(BatchedSelfPlay(model, num_games=g, sims=sims, seed=int(seed) * 64 + i, device=self.dev,
                                                 quota=max(quotas[i], g), **kw))
        self.G, self.sims = int(num_games), int(sims)
        # quota > G: the slots are refilled -- a slot whose game has ended takes the next game not yet handed out (in slot
        # order, deterministic) until `quota` games have been started: the reference's loop over games (self_play.py:81-84)
        # on G concurrent slots instead of lock-step generations that idle every finished slot until the longest game ends
        self.quota = self.G if quota is None else int(quota)
        if self.quota < self.G:
            raise ValueError("quota must be >= num_games")
        self.max_plies = self.sets[0].max_plies
        self.A, self.N = self.sets[0].A, self.sets[0].N
        self._live = [True] * k
        self.sync()

    def _each(self, live_only=False):
        for i, (st, eng) in enumerate(zip(self.streams, self.sets)):
            if live_only and not self._live[i]:
                continue
            with torch.cuda.stream(st):
                yield i, eng

    def sync(self):
        for st in self.streams:
            st.synchronize()

    def reset(self):
        for _, eng in self._each():
            eng.reset()
        self._live = [True] * len(self.sets)

    def move(self):
        """One move for every active game of every live set (a set whose games have all ended is skipped)."""
        for _, eng in self._each(live_only=True):
            eng.move()

    def move_exclusive(self, k, before=None, after=None):
        """Measurement aid (bench.py): one move in which set k runs ALONE on the GPU -- every stream is drained, set k
        makes its move (callbacks `before(set)` / `after(set)` run on its stream around it), is drained again, and only
        then do the other live sets move.  Per-kernel durations taken inside are those of the kernel itself, not of a
        kernel sharing the chip with the other sets' launches.  Results are identical to move()."""
        self.sync()
        k %= len(self.sets)
        if self._live[k]:
            with torch.cuda.stream(self.streams[k]):
                if before:
                    before(self.sets[k])
                self.sets[k].move()
                if after:
                    after(self.sets[k])
            self.streams[k].synchronize()
        for i, eng in self._each(live_only=True):
            if i != k:
                eng.move()
        return self._live[k]

    def counters(self):
        tot = dict(active=0, finished=0, dead_ends=0, started=0, moves=0, gnn_saturated=0, leaf_evals=0, cache_hits=0, terminal_sims=0)
        for i, eng in self._each():
            c = eng.counters()                    # .cpu() inside synchronises this set's stream only
            self._live[i] = self._live[i] and c["active"] > 0
            for key in tot:
                tot[key] += c[key]
        return tot

    def play_generation(self, check_every=4):
        ply = 0
        limit = self.max_plies * max(-(-e.quota // e.G) for e in self.sets)
        while True:
            self.move()
            ply += 1
            if ply >= limit or ply % check_every == 0:
                c = self.counters()
                if c["gnn_saturated"] and any(not (e.e.gnn_flags & _lib.GNN_EXACT_F32) for e in self.sets):
                    for _, eng in self._each():          # fp16-range guard: replay the generation on the exact f32 kernels
                        eng._fall_back_to_exact_kernels()
                    self._live = [True] * len(self.sets)
                    ply = 0
                    continue
                if c["active"] == 0 or ply >= limit:
                    break
        return self.counters()

    def history_tensors(self):
        parts = []
        for _, eng in self._each():
            parts.append(eng.history_tensors())
        self.sync()
        cur = torch.cuda.current_stream(self.dev)
        for st in self.streams:
            cur.wait_stream(st)
        for p in parts:
            for x in p:
                x.record_stream(cur)              # the caching allocator must not recycle them under the concatenation
        return tuple(torch.cat([p[j] for p in parts], 0) for j in range(3))


def gather_history(states72, visits, z, group=None, force_collective=None):
    """The one exchange step per generation: all-gather the ragged (s, pi, z) tuples of every rank.
    Counts are gathered first, payloads are padded to the max count (RCCL needs equal sizes) and trimmed after.
    Works on any torch.distributed backend (nccl == RCCL on ROCm; gloo in the CPU tests).
    force_collective (default: AQG_DIST_FORCE_GROUP=1): run the two collectives at world size 1 as well instead of returning the
    inputs -- the RCCL code path on the one GPU a developer box has (the result is the same rows, bit for bit)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return states72, visits, z
    if force_collective is None:
        from .distributed import force_group
        force_collective = force_group()
    if dist.get_world_size(group) == 1 and not force_collective:
        return states72, visits, z
    W = dist.get_world_size(group)
    dev = states72.device
    n = torch.tensor([states72.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(W)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    m = max(counts)
    A = visits.shape[1]
    row = 72 + 2 * A + 1
    buf = torch.zeros((m, row), dtype=torch.uint8, device=dev)
    k = states72.shape[0]
    if k:
        buf[:k, :72] = states72
        buf[:k, 72:72 + 2 * A] = visits.contiguous().view(torch.uint8).view(k, 2 * A)
        buf[:k, 72 + 2 * A] = z.view(torch.uint8)
    out = torch.empty((W * m, row), dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, buf, group=group)
    out = out.view(W, m, row)
    parts = [out[r, :counts[r]] for r in range(W)]
    allrows = torch.cat(parts, 0)
    s = allrows[:, :72].contiguous()
    v = allrows[:, 72:72 + 2 * A].contiguous().view(torch.int16).view(-1, A)
    zz = allrows[:, 72 + 2 * A].contiguous().view(torch.int8)
    return s, v, zz
