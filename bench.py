#!/usr/bin/env python3
"""bench.py -- self-play games/s (whole job) + GNN boards/s on MI355X, with roofline and CPU baseline.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched by torch.distributed.run with one
rank per GPU (RCCL).  One STEP = one self-play generation on every rank: `--games` concurrent 9x9 games per GPU
(default 2048, BASELINE.json configs[2]) played to termination with PV-MCTS at `--sims` simulations per move
(default 200) and the random-weight GNN as evaluator, followed by the generation's single exchange step -- the
all-gather of the (s, pi, z) tuples over RCCL/xGMI (configs[3]).  Weak scaling: per-GPU work is fixed.

`--backend gloo` swaps RCCL for gloo (host tensors) so that the N > 1 path can be rehearsed with several ranks on one GPU.

Rank 0 prints ONE JSON line, numbers only (< 6 KB, so that a driver keeping an 8 KB tail keeps every leg); what each key means, how
it is measured and what bounds it is written HERE and in DESIGN.md section 5 (`notes_ref`).  Besides the contract keys it carries
  gnn_forward  : configs[1] -- pv_network_gnn forward (trunk + heads) at B = 4,096 synthetic boards, boards/s from HIP events; the same
                 at 16,384 and 65,536 boards; `tracking_build` = the trunk build a weight set WITHOUT a proven fp16
                 range runs (flags 0: float-maximum range guard), `proven_build` = the one this run's initialisation-scale weights run
                 (AQG_GNN_RANGE_PROVEN: the headline uses it); `f32_mfma_exact` = the exact f32-input MFMA trunk; mfma_frac against the
                 fp16 / 3 split roof, hbm_frac_survey_formula = boards/s x 169,760 B / 8 TB/s (effective: activations stay in LDS)
  rccl_group_alive : the headline generation in three fresh processes -- no process group / a world-size-1 backend-nccl (RCCL) group
                 created BEFORE the four game-set streams / created AFTER them -- with the generation's exchange step running through
                 the communicator (engine.gather_history, collectives forced): games/s each, and the ratios to `none`
  legal_mask   : SURVEY 8(d)'s second kernel -- batched State.legal_actions() at 4,096 and 65,536 states: states/s from per-launch
                 HIP event pairs (median / min / max), hbm_frac = states/s x 100 B / 8 TB/s (<< 1, stated), the CPU oracle beside it
  step_kernel  : the fused MCTS step (engine_step_fast_kernel): latency-bound, so its entry is us per launch at 512 and
                 4,096 games (HIP events over 160 back-to-back launches on mid-game trees), wavefronts per SIMD and
                 game-steps/s instead of a bandwidth fraction
  slot_refill  : the same slots with refill (quota = 3 x games: a finished game's slot takes the next game) against the
                 lock-step generation of the headline number
  train_step   : SURVEY 8(f).1, one epoch call of 200 steps at batch 128, with its own matrix-pipe roofline
  roofline     : the dominant kernel (gcn_trunk_boards_mm_kernel) over the timed region: algorithmic FLOP of the
                 boards it processed / its summed launch durations (HIP event pairs recorded inside the library around
                 the trunk launches of sampled moves -- one game set running alone, plain launches --, on the launch stream), against the matrix-pipe roof of the
                 fp32-equivalent fp16-split algorithm; the SURVEY 8(d) HBM figure (169,760 B/board against 8 TB/s) is
                 reported beside it as hbm_frac_survey_formula
  eval_cache   : round 4 -- the evaluation cache (include/aqgnn.h eval_cache_keys): a small generation with the table off
                 and on compared byte for byte in this process (identity_check), then the headline configuration and the large_batch
                 configuration WITH the table, with hit rate, network evaluations/s and the ratio to the cache-less number of the same
                 run.  `value` itself never uses the table.
  cpu_baseline : the oracle (CPU restatement, kind "port") on a bounded sample of the same workload -- one sequential self-play
                 loop per host core of the box's share (16), rates summed --, rank 0, N=1.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # 4 game-set streams + the default stream, one hardware queue each (read at HIP init)

import numpy as np
import torch
import torch.distributed as dist

TRUNK_FLOP_PER_BOARD = 2 * 81 * 6 * 128 + 2 * (2 * 81 * 128 * 128)   # 5,432,832: the three GCN contractions
FWD_FLOP_PER_BOARD = 5492480                                         # trunk + both heads
HBM_BYTES_PER_BOARD = 169760                                         # layer-granular algorithmic bytes
PEAK_F32_MFMA = 157.3e12                                             # MI355X_MICROARCH.md: f32-input MFMA
PEAK_F16_MFMA = 2500.0e12                                            # MI355X_MICROARCH.md: dense fp16/bf16 MFMA
SPLIT_TERMS = 3                                                      # hi*hi + hi*lo + lo*hi per f32 product
TRUNK_MFMA_PER_BOARD = 8 * (6 + 2 * 72 + 2 * 20)                     # 1,520 16x16x32 fp16 MFMAs issued per board (layer 1, linear maps, aggregations, padding)
PEAK_HBM = 8.0e12
# Compulsory global traffic of the default trunk per board: the 72-byte record (24 inside the engine) in, the 512-byte pooled
# row out; the weight fragments are served by the L2 (hit rate 97.5-99.5 %, profiles/r03_pmc_summary.csv).  PMC counters
# cannot be collected from inside this process; what a rocprofv3 pass of THIS round measured is in profiles/r03_pmc_summary.csv
# and quoted in DESIGN.md section 5 -- `roofline.traffic` stays null in the line rather than carrying a constant.
TRUNK_COMPULSORY_BYTES_PER_BOARD = 72 + 512
LEGAL_BYTES_PER_STATE = 100                                          # 68 B in + 27 B (209 bits) out ~ 100 B/state
REF_LEGAL_MS_PER_STATE_PY = 4.95                                     # the reference's State.legal_actions(), 1 core (python)
# training step, per position: forward trunk + the three weight gradients (same contractions) + the two data gradients
# (layers 3, 2) + six aggregations (5 terms x 81 nodes x 128 columns) + the heads forward and twice backward
TRAIN_FLOP_PER_POSITION = 2 * TRUNK_FLOP_PER_BOARD + 2 * (2 * 81 * 128 * 128) + 6 * (2 * 81 * 5 * 128) + 3 * (FWD_FLOP_PER_BOARD - TRUNK_FLOP_PER_BOARD)


_CPU_WORKER = r"""
import sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
from oracle import gnn as og, mcts as om, quoridor as oq
sims, budget, seed = int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
oq.lib()
model = og.OracleModel(og.init_params(0))
rng = np.random.RandomState(seed)
state = oq.State(N=9)
t0 = time.time()
plies = 0
while time.time() - t0 < budget:
    if state.is_done():
        state = oq.State(N=9)
    scores = om.pv_mcts_policy(model, state, 1.0, sims)
    legal = state.legal_actions()
    state = state.next(legal[om.choice_index(scores, rng.random_sample())])
    plies += 1
print(plies, time.time() - t0)
"""


_CPU_LEGAL_WORKER = r"""
import sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
from oracle import quoridor as oq
oq.lib()
recs = np.load(sys.argv[2])
oq.legal_actions_batch(recs[:64])
t = time.time()
oq.legal_actions_batch(recs)
print(len(recs) / (time.time() - t))
"""


def cpu_legal_baseline(sample):
    """oracle/quoridor_oracle.c (the reference's array/queue legal_actions in C) on `sample` states, one core: states/s."""
    import subprocess
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"aqg_legal_sample_{os.getpid()}.npy")
    np.save(path, sample)
    try:
        out = subprocess.run([sys.executable, "-c", _CPU_LEGAL_WORKER, ROOT, path], capture_output=True, text=True, timeout=300)
        return float(out.stdout.split()[-1])
    finally:
        if os.path.exists(path):
            os.remove(path)


def cpu_baseline(sims, mean_plies, budget_s=15.0, workers=None):
    """Oracle (oracle/mcts.py + oracle/gnn.py fp64 + C rules) timed on the host: every worker process plays the reference's
    sequential loop (one game after the other, self_play.py:81-84) for `budget_s` seconds on one core; the workers' rates add up
    (the reference itself is single-threaded; running one copy per core is the most it could do on this host)."""
    import subprocess
    if workers is None:
        workers = max(1, min(16, os.cpu_count() or 1))        # a one-GPU box's CPU share
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, "-c", _CPU_WORKER, ROOT, str(sims), str(budget_s), str(i)], stdout=subprocess.PIPE,
                              stderr=subprocess.DEVNULL, env=env, text=True) for i in range(workers)]
    plies_per_s = 0.0
    total_plies = 0
    for p in procs:
        out, _ = p.communicate(timeout=budget_s * 4 + 120)
        pl, dt = out.split()[-2:]
        plies_per_s += int(pl) / float(dt)
        total_plies += int(pl)
    return {"value": plies_per_s / max(mean_plies, 1.0), "unit": "games/s", "cores": workers, "kind": "port",
            "sims_per_s": plies_per_s * sims,
            "sample": f"{workers} procs x {budget_s:.0f} s sequential {sims}-sims/move self-play ({total_plies} plies), oracle GNN fp64 + C rules"}


_RCCL_ORDERS = ("none", "before", "after")


def rccl_child(order, games, sims, sets):
    """One fresh process of the `rccl_group_alive` leg: the headline generation with no process group (`none`), or with a
    world-size-1 backend-nccl group whose RCCL communicator is created BEFORE / AFTER the engine's game-set streams; the exchange
    step of every generation goes through engine.gather_history (with the group: two real collectives on device tensors).
    Prints one small JSON line (not the contract line)."""
    if order != "none":
        os.environ["AQG_DIST_FORCE_GROUP"] = "1"
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from alphaquoridorgnn_amd import distributed as aqd
    from alphaquoridorgnn_amd.engine import MultiSetSelfPlay, gather_history
    from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork

    def make_group():
        aqd.init_from_env()                                     # backend nccl, device_id = this GPU (AQG_DIST_FORCE_GROUP)
        t = torch.ones((4,), device=dev)
        out = torch.empty((4,), device=dev)
        dist.all_gather_into_tensor(out, t)                     # the communicator (and its streams) exist from here on
        torch.cuda.synchronize()
    if order == "before":
        make_group()
    torch.manual_seed(0)
    model = GNNNetwork().to(dev).eval()
    eng = MultiSetSelfPlay(model, num_games=games, sims=sims, num_sets=sets, seed=1000)
    if order == "after":
        make_group()

    def generation():
        eng.reset()
        c = eng.play_generation()
        st, vis, z = gather_history(*eng.history_tensors())
        return c["finished"], int(st.shape[0])
    generation()                                                # untimed: graph capture, clocks
    torch.cuda.synchronize()
    t0 = time.time()
    fin = 0
    for _ in range(2):
        f, rows = generation()
        fin += f
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(json.dumps({"order": order, "games_per_s": fin / dt, "rows": rows,
                      "backend": dist.get_backend() if dist.is_initialized() else None}))
    if dist.is_initialized():
        aqd.shutdown()


def rccl_group_alive_leg(games, sims, sets):
    """Three fresh processes (see rccl_child), one after the other, BEFORE this process touches the GPU."""
    import subprocess
    leg = {"games_per_s": {}}
    for order in _RCCL_ORDERS:
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--rccl-child", order, "--games", str(games), "--sims", str(sims),
                                  "--sets", str(sets)], capture_output=True, text=True, timeout=600)
            rec = json.loads(out.stdout.strip().splitlines()[-1])
            leg["games_per_s"][order] = rec["games_per_s"]
            leg["rows"] = rec["rows"]
            if order != "none":
                leg["backend"] = rec["backend"]
        except Exception as e:                                  # the leg must never take the contract line down with it
            leg["games_per_s"][order] = None
            leg["error"] = f"{order}: {type(e).__name__}"
    base = leg["games_per_s"].get("none")
    if base:
        leg["ratio_to_none"] = {k: (v / base if v else None) for k, v in leg["games_per_s"].items() if k != "none"}
    return leg


def legal_mask_leg(dev, lib, _lib, synth_states):
    """SURVEY 8(d) second kernel: batched State.legal_actions() (game_logic.py:103-117, BFS :309-348) -- legal_actions_kernel<9>,
    one wavefront per state, mask + ordered list + count written.  Every launch sits between its own HIP event pair on the
    launch stream, so a host hiccup between launches shows up as ONE long sample (max_us) instead of inflating the mean
    (round 2's unexplained '481 us at B = 512' was the mean of 200 back-to-back launches behind one event pair)."""
    leg = {"kernel": "legal_actions_kernel<9>", "bound": "integer ALU / latency (not HBM)", "batches": {}}
    for B in (4096, 65536):
        st = synth_states(B, seed=1, dev=dev)
        mask = torch.empty((B, 209), dtype=torch.uint8, device=dev)
        order = torch.empty((B, 136), dtype=torch.uint8, device=dev)
        count = torch.empty((B,), dtype=torch.int32, device=dev)

        def legal():
            _lib.check(lib.aqg_legal_actions(9, _lib.ptr(st), B, _lib.ptr(mask), _lib.ptr(order), _lib.ptr(count), _lib.stream_ptr(dev)), "legal")
        for _ in range(10):
            legal()
        torch.cuda.synchronize()
        n = 100
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        ev[0].record()
        for i in range(n):
            legal()
            ev[i + 1].record()
        torch.cuda.synchronize()
        us = np.asarray([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)])
        med = float(np.median(us))
        leg["batches"][str(B)] = {"us_per_launch_median": med, "us_min": float(us.min()), "us_max": float(us.max()),
                                  "states_per_s": B / (med * 1e-6), "hbm_frac": B / (med * 1e-6) * LEGAL_BYTES_PER_STATE / PEAK_HBM,
                                  "mean_legal_actions": float(count.float().mean()),
                                  "mean_walls_on_board": float((st[:, 4:68] != 0).sum(1).float().mean())}
    leg["_sample"] = synth_states(4096, seed=1, dev=dev).cpu().numpy()     # handed to the cpu_baseline leg, removed before printing
    leg["reference_python_states_per_s_per_core"] = 1e3 / REF_LEGAL_MS_PER_STATE_PY
    return leg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--games", type=int, default=2048, help="concurrent games per GPU")
    ap.add_argument("--sims", type=int, default=200, help="MCTS simulations per move")
    ap.add_argument("--sets", type=int, default=4, help="independent game sets per GPU, one HIP stream each (engine.MultiSetSelfPlay)")
    ap.add_argument("--gnn-batch", type=int, default=4096)
    ap.add_argument("--trunk-variant", type=int, default=3, help="developer knob: aqg_set_option trunk_variant (3 = per-launch choice)")
    ap.add_argument("--trunk-grid", type=int, default=0, help="developer knob: cap the trunk's persistent grid (0 = default 512)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"),
                    help="torch.distributed backend for N > 1: nccl = RCCL over xGMI (default); gloo = host tensors, for rehearsing "
                         "several ranks on one GPU")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the step-kernel / slot-refill / training legs (profiling runs)")
    ap.add_argument("--eval-cache-slots", type=int, default=8192,
                    help="entries per game slot of the evaluation cache in the `eval_cache` LEG (0 = skip the leg); the headline `value` always runs without it")
    ap.add_argument("--large-games", type=int, default=16384,
                    help="extra single-GPU leg: one generation at this many concurrent games (north star: >= 10k); 0 = skip")
    ap.add_argument("--rccl-child", default=None, choices=_RCCL_ORDERS, help="internal: one process of the rccl_group_alive leg")
    args = ap.parse_args()
    if args.rccl_child:
        return rccl_child(args.rccl_child, args.games, args.sims, args.sets)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    rccl_leg = None
    if world == 1 and not args.no_extra_legs and dist.is_nccl_available():
        rccl_leg = rccl_group_alive_leg(args.games, args.sims, args.sets)     # before this process creates its own GPU context
    local = local % max(torch.cuda.device_count(), 1)        # gloo rehearsal: more ranks than GPUs share the card
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    from alphaquoridorgnn_amd import _lib
    from alphaquoridorgnn_amd.engine import BatchedSelfPlay, MultiSetSelfPlay, gather_history
    from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
    lib = _lib.load()                    # no HIP library -> hard failure, there is no fallback path

    if args.trunk_grid:
        _lib.set_option("trunk_grid", args.trunk_grid)
    for opt in ("step_waves", "trunk_prio", "step_prio", "heads_prio"):                       # developer knobs through the environment (tools/*_scan.sh)
        if os.environ.get("AQG_" + opt.upper()):
            _lib.set_option(opt, int(os.environ["AQG_" + opt.upper()]))
    if args.trunk_variant != 3:
        _lib.set_option("trunk_variant", args.trunk_variant)
    torch.manual_seed(0)                 # random-init weights of the reference architecture (synthetic; no checkpoints)
    model = GNNNetwork().to(dev).eval()
    eng = MultiSetSelfPlay(model, num_games=args.games, sims=args.sims, num_sets=args.sets, seed=1000 + rank)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sampled_boards = torch.zeros((1,), dtype=torch.int64, device=dev)
    SAMPLE_PLIES = (27, 83)           # plies of a generation whose trunk launches carry HIP event pairs.  On such a ply ONE
                                      # game set (round-robin) makes its move alone on the GPU with plain launches -- the
                                      # other plies replay captured hipGraphs with all sets overlapping, where an event pair
                                      # would time the sharing, not the kernel.  Costs ~1 % of the timed region (two plies: 800 bracketed
                                      # launches per run of two steps; four plies cost twice that and told nothing more).
    sample_no = [0]
    last_plies = [0]

    def _before(e):
        sampled_boards.sub_(e.t["stat_leaf_evals"].sum())    # boards evaluated by the sampled launches, counted on the device
        _lib.set_option("profile_trunk", 1)

    def _after(e):
        _lib.set_option("profile_trunk", 0)
        sampled_boards.add_(e.t["stat_leaf_evals"].sum())

    def one_step(profile):
        eng.reset()
        ply = 0
        while True:
            if profile and ply in SAMPLE_PLIES:
                eng.move_exclusive(sample_no[0], _before, _after)
                sample_no[0] += 1
            else:
                eng.move()
            ply += 1
            if ply % 4 == 0 or ply >= eng.max_plies:
                if profile:
                    eng.sync()                        # events of every set's stream are complete before they are read
                    _lib.profile_collect()
                if eng.counters()["active"] == 0 or ply >= eng.max_plies:
                    break
        st, vis, z = eng.history_tensors()
        st, vis, z = gather_history(st, vis, z)       # the generation's one exchange step (RCCL all-gather)
        last_plies[0] = ply
        return eng.counters(), int(st.shape[0])

    for _ in range(args.warmup):
        one_step(False)
    _lib.profile_collect(reset=True)
    sync_all()
    t0 = time.time()
    games = positions = leaf_evals = 0
    for _ in range(args.steps):
        c, npos = one_step(True)
        games += c["finished"]
        leaf_evals += c["leaf_evals"]
        positions = npos
    sync_all()
    elapsed = time.time() - t0
    trunk_ms, trunk_launches, trunk_rows = _lib.profile_collect(reset=True)
    trunk_boards = int(sampled_boards.sum().item())    # boards evaluated inside the event-bracketed launches

    tt = torch.tensor([elapsed, float(games), float(leaf_evals)], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if world > 1:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
    total_games, total_evals = float(tt[1]), float(tt[2])
    mean_plies = positions / max(total_games / max(args.steps, 1), 1.0)   # gathered positions of the last step / games per step

    # ---- configs[1]: GNN forward (trunk + heads), B = 4096 synthetic boards, HIP events on the launch stream (rank-local); the same
    #      at 16,384 and 65,536 boards
    from tools.microbench import synth_states, time_ms
    B = args.gnn_batch
    pk = model.packed_weights(dev)
    gnn_flags = model.gnn_flags(dev)          # what GraphPolicyValueNetwork.forward_states passes for this weight set (range guard: proven bound / tracking)
    sat_word = model.saturation_word(dev)

    def forward_fn(nb, flags):
        bd = synth_states(nb, seed=0, dev=dev)
        pooled = torch.empty((nb, 128), device=dev)
        policy = torch.empty((nb, 209), device=dev)
        value = torch.empty((nb,), device=dev)

        def fwd():
            _lib.check(lib.aqg_gcn_forward_boards_guarded(9, _lib.ptr(bd), 0, nb, _lib.ptr(pk), _lib.ptr(pooled), None, _lib.ptr(policy), None,
                                                          _lib.ptr(value), flags, _lib.ptr(sat_word), _lib.stream_ptr(dev)), "fwd")
        fwd.keep = (bd, pooled, policy, value)
        return fwd
    boards = synth_states(B, seed=0, dev=dev)
    # the builds are timed three times in interleaved order and the best time kept: a single pass right after the pools were freed
    # under-reported the exact-f32 kernel by 3.4x in the round-1 driver run (its scratch arena grows on first use)
    builds = {"proven_build": forward_fn(B, _lib.GNN_RANGE_PROVEN), "tracking_build": forward_fn(B, 0), "f32_mfma_exact": forward_fn(B, _lib.GNN_EXACT_F32)}
    samples = {k: [] for k in builds}
    for rep in range(3):
        for k, fn in builds.items():
            samples[k].append(time_ms(fn, 60, warmup=10))
    build_rate = {k: B / (min(v) * 1e-3) for k, v in samples.items()}
    fwd = forward_fn(B, gnn_flags)
    fwd_ms = min(time_ms(fwd, 200, warmup=20) for _ in range(2))      # what this weight set runs, after the clocks have settled on this workload
    fwd_rates = {str(B): B / (fwd_ms * 1e-3)}
    for nb in (16384, 65536):
        if world == 1:
            f2 = forward_fn(nb, gnn_flags)
            fwd_rates[str(nb)] = nb / (min(time_ms(f2, 30, warmup=5) for _ in range(2)) * 1e-3)
            del f2
    del builds

    step_leg = refill_leg = train_leg = legal_leg = graph_leg = None
    if world == 1 and not args.no_extra_legs:
        legal_leg = legal_mask_leg(dev, lib, _lib, synth_states)
        # ---- the reference's literal operator forward(x, edge_index, batch) (pv_network_gnn.py:53-64) on the same boards given as
        #      a PyG-style batch: gcn_norm / CSR built per call with torch ops, then the generic graph kernels (no board structure
        #      assumed: any graph batch goes through here)
        from tools.microbench import board_graph_batch
        gx, gei, gb = board_graph_batch(boards)
        with torch.no_grad():
            pol_g, _ = model(gx, gei, gb)
            gms = time_ms(lambda: model(gx, gei, gb), 10, warmup=2)
            pol_b, _ = model.forward_states(boards)
        graph_leg = {"boards": B, "nodes": int(gx.shape[0]), "directed_edges": int(gei.shape[1]), "boards_per_s": B / (gms * 1e-3), "ms": gms,
                     "max_abs_diff_vs_board_path": float((pol_g - pol_b).abs().max())}
        # (every MultiSetSelfPlay of a process runs on the same four streams, engine._SET_STREAMS: a second engine on four NEW streams
        # would share hardware queues with the first one's idle streams and ran 35 % slower)
        del eng
        torch.cuda.empty_cache()
        # ---- the fused MCTS step kernel alone: one wavefront per game, one dependent chain of loads per tree level
        step_leg = {"kernel": "engine_step_fast_kernel<9>", "bound": "latency", "us_per_launch": {}, "game_steps_per_s": {}}
        for G in (512, 4096):
            # the real loop, one game set alone on the GPU with plain launches, HIP event pairs around the step launches
            # (aqg_set_option("profile_trunk", 2)): plies 24-31 of a GNN-evaluated generation
            e1 = BatchedSelfPlay(model, num_games=G, sims=args.sims, seed=5, record_history=False)
            for _ in range(24):
                e1.move()
            torch.cuda.synchronize()
            _lib.profile_collect(reset=True)
            _lib.set_option("profile_trunk", 2)
            for _ in range(8):
                e1.move()
                torch.cuda.synchronize()
                _lib.profile_collect()
            _lib.set_option("profile_trunk", 0)
            ms, n, _ = _lib.profile_collect(reset=True)
            us = ms / max(n, 1) * 1e3
            step_leg["us_per_launch"][str(G)] = us
            step_leg["launches_timed"] = int(n)
            step_leg["game_steps_per_s"][str(G)] = G / (us * 1e-6)
            del e1
        # ---- slot refill: the same 2048 slots, 3 x 2048 games; a finished game's slot takes the next game
        torch.cuda.empty_cache()
        er = MultiSetSelfPlay(model, num_games=args.games, sims=args.sims, num_sets=args.sets, seed=4242, quota=3 * args.games)
        er.move(); er.sync(); torch.cuda.synchronize()
        er.reset()
        t1 = time.time()
        cr = er.play_generation()
        torch.cuda.synchronize()
        dtr = time.time() - t1
        rpos = int(er.history_tensors()[0].shape[0])
        rplies = max(e2.moves_done for e2 in er.sets)
        refill_leg = {"games": cr["finished"], "slots": args.games, "games_per_s": cr["finished"] / dtr, "s": dtr, "positions": rpos, "plies_played": rplies,
                      "slot_utilisation": rpos / max(1.0, float(args.games) * rplies),
                      "lockstep_slot_utilisation": positions / max(1.0, float(args.games * world) * last_plies[0])}
        del er
        torch.cuda.empty_cache()
        # the same comparison where refill is meant to pay: SHORT games of very different lengths (fake evaluator with a strong
        # forward bias: pawns race to the goal), lock-step generations vs refilled slots, same slots, same number of games
        def short_games(quota_mult, generations):
            es = MultiSetSelfPlay(None, num_games=args.games, sims=args.sims, num_sets=args.sets, seed=9, quota=quota_mult * args.games,
                                  evaluator="fake", fake_bias=40)
            es.move(); es.sync(); torch.cuda.synchronize()
            fin = pos = ev = plies = 0
            t2 = time.time()
            for _ in range(generations):
                es.reset()
                c2 = es.play_generation()
                fin += c2["finished"]; ev += c2["leaf_evals"]
                pos += int(es.history_tensors()[0].shape[0])
                plies += max(e2.moves_done for e2 in es.sets)
            torch.cuda.synchronize()
            d2 = time.time() - t2
            return {"games_per_s": fin / d2, "leaf_evals_per_s": ev / d2, "games": fin, "mean_plies": pos / max(fin, 1),
                    "slot_utilisation": pos / max(1.0, float(args.games) * plies)}
        sg_lock, sg_refill = short_games(1, 3), short_games(3, 1)
        refill_leg["short_games"] = {"lockstep": sg_lock, "refill": sg_refill,
                                     "refill_over_lockstep_games_per_s": sg_refill["games_per_s"] / sg_lock["games_per_s"],
                                     "games_per_leaf_eval_refill_over_lockstep": (sg_refill["games_per_s"] / sg_refill["leaf_evals_per_s"]) /
                                                                                 (sg_lock["games_per_s"] / sg_lock["leaf_evals_per_s"])}
        torch.cuda.empty_cache()
        # ---- next-row leg: train_network.py's optimisation step on the GNN, batch 128 (train_network.py:15)
        from alphaquoridorgnn_amd.train_network import GNNTrainer, BATCH_SIZE
        tr_model = GNNNetwork().to(dev)
        trainer = GNNTrainer(tr_model, max_batch=BATCH_SIZE)
        nsteps = 200
        tb = synth_states(BATCH_SIZE * nsteps, seed=3, dev=dev)
        tpi = torch.softmax(torch.randn((BATCH_SIZE * nsteps, 209), device=dev), dim=1)
        tz = torch.randint(-1, 2, (BATCH_SIZE * nsteps,), device=dev).float()
        order = torch.randperm(BATCH_SIZE * nsteps, device=dev)
        trainer.run_epoch(tb, tpi, tz, order[:BATCH_SIZE * 10])
        torch.cuda.synchronize()
        t1 = time.time()
        trainer.run_epoch(tb, tpi, tz, order)
        torch.cuda.synchronize()
        train_ms = (time.time() - t1) / nsteps * 1e3
        step_ms = time_ms(lambda: trainer.step(tb[:BATCH_SIZE], tpi[:BATCH_SIZE], tz[:BATCH_SIZE]), 50, warmup=5)
        tflops = BATCH_SIZE * TRAIN_FLOP_PER_POSITION / (train_ms * 1e-3) / 1e12
        fallbacks = int(_lib.load().aqg_gcn_train_fallbacks(1))
        _lib.set_option("train_fused", 1)                         # the exact-f32 form of the same step (last round's default), for the record
        trainer.run_epoch(tb, tpi, tz, order[:BATCH_SIZE * 10])
        torch.cuda.synchronize()
        t1 = time.time()
        trainer.run_epoch(tb, tpi, tz, order)
        torch.cuda.synchronize()
        train_ms_f32 = (time.time() - t1) / nsteps * 1e3
        _lib.set_option("train_fused", 2)
        train_leg = {"batch": BATCH_SIZE, "steps_per_call": nsteps, "ms_per_step": train_ms, "positions_per_s": BATCH_SIZE / (train_ms * 1e-3),
                     "ms_per_step_single_calls": step_ms, "ms_per_step_f32_input_mfma_form": train_ms_f32, "positions_redone_in_f32": fallbacks,
                     "launches_per_step": 2,
                     "roofline": {"kernel": "train_board_split_kernel + train_final_kernel", "bound": "mfma", "achieved": tflops,
                                  "peak": PEAK_F16_MFMA / 3 / 1e12, "unit": "TFLOP/s", "frac": tflops * 1e12 / (PEAK_F16_MFMA / 3),
                                  "frac_vs_f32_input_mfma_peak": tflops * 1e12 / PEAK_F32_MFMA, "flop_per_position": TRAIN_FLOP_PER_POSITION}}
        del trainer, tr_model

    large = None
    if world == 1 and args.large_games > 0 and not args.no_extra_legs:
        torch.cuda.empty_cache()
        eng = MultiSetSelfPlay(model, num_games=args.large_games, sims=args.sims, num_sets=args.sets, seed=77)
        for _ in range(2):                    # two untimed moves: graph capture, first-touch of the 14 GB tree pools
            eng.move()
        eng.sync()
        torch.cuda.synchronize()
        t1 = time.time()
        c, _ = one_step(False)
        torch.cuda.synchronize()
        dt = time.time() - t1
        large = {"concurrent_games": args.large_games, "sims_per_move": args.sims, "games_per_s": c["finished"] / dt, "s_per_generation": dt,
                 "leaf_evals_per_s": c["leaf_evals"] / dt}

    # ---- evaluation cache (include/aqgnn.h eval_cache_keys; engine.BatchedSelfPlay(eval_cache_slots=...)): the SAME generations with the
    #      per-slot table of already-evaluated positions on.  Reported as a leg, never as `value`: the games and records are bit-identical
    #      (checked here on a small generation, and in tests/test_gpu_parity.py against the oracle), the network simply is not asked twice.
    cache_leg = None
    if world == 1 and args.eval_cache_slots > 0 and not args.no_extra_legs:
        del eng
        torch.cuda.empty_cache()
        small = {}
        for slots in (0, 1024):
            es = MultiSetSelfPlay(model, num_games=256, sims=50, num_sets=args.sets, seed=31, eval_cache_slots=slots)
            cs = es.play_generation()
            small[slots] = ([x.cpu() for x in es.history_tensors()], cs)
            del es
        identical = all(torch.equal(a, b) for a, b in zip(small[0][0], small[1024][0])) and small[0][1]["leaf_evals"] == small[1024][1]["leaf_evals"]
        cache_leg = {"slots_per_game": args.eval_cache_slots, "bytes_per_slot": 736,
                     "identity_check": {"games": 256, "sims": 50, "rows": int(small[0][0][0].shape[0]), "rows_identical_to_cache_off": bool(identical)}}
        del small
        for label, games, slots in (("headline_config", args.games, args.eval_cache_slots), ("large_batch", args.large_games, args.eval_cache_slots)):
            if games <= 0:
                continue
            torch.cuda.empty_cache()
            free = torch.cuda.mem_get_info(dev)[0]
            while slots > 64 and games * slots * 736 > 0.5 * free:      # the table may take half of the free HBM (16,384 games x 8,192 entries = 99 GB of 288)
                slots //= 2
            eng = MultiSetSelfPlay(model, num_games=games, sims=args.sims, num_sets=args.sets, seed=1000 if label == "headline_config" else 77, eval_cache_slots=slots)
            for _ in range(2):
                eng.move()
            eng.sync()
            torch.cuda.synchronize()
            t1 = time.time()
            c, _ = one_step(False)
            torch.cuda.synchronize()
            dt = time.time() - t1
            cache_leg[label] = {"concurrent_games": games, "slots_per_game": slots, "games_per_s": c["finished"] / dt, "s_per_generation": dt,
                                "leaf_evals_per_s": c["leaf_evals"] / dt, "hit_rate": c["cache_hits"] / max(c["leaf_evals"], 1),
                                "network_evals_per_s": (c["leaf_evals"] - c["cache_hits"]) / dt}
            del eng
        torch.cuda.empty_cache()

    if rank == 0:
        achieved = trunk_boards * TRUNK_FLOP_PER_BOARD / (trunk_ms * 1e-3) if trunk_ms > 0 else 0.0   # rank 0's sampled launches
        boards_per_s_kernel = trunk_boards / (trunk_ms * 1e-3) if trunk_ms > 0 else 0.0
        pmc = None                               # matrix-pipe busy share: a RECORDED rocprofv3 --pmc measurement (separate passes), read from
        pmc_path = os.path.join(ROOT, "profiles", "r04_pmc_trunk.json")      # the file the profiling recipe wrote, never typed in here
        if os.path.exists(pmc_path):
            with open(pmc_path) as f:
                pmc = dict(json.load(f), source="profiles/r04_pmc_trunk.json")
        fwd_rate = fwd_rates[str(B)]
        out = {
            "metric": "self-play games/sec, 9x9 Quoridor (PV-MCTS, GNN evaluator), whole job",
            "value": total_games / elapsed,
            "unit": "games/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (f16x3-split MFMA products, f32 accumulate)",
            "data": "synthetic",
            "notes_ref": "bench.py docstring + DESIGN.md section 5",
            "config": {"workload": "BASELINE configs[2]/[3]: one self-play generation per step (lock-step PV-MCTS, random-weight GNN) + all-gather of (s,pi,z)",
                       "games_per_gpu": args.games, "game_sets_per_gpu": args.sets, "sims_per_move": args.sims, "board": "9x9", "walls": 10, "plies_for_draw": 116,
                       "temperature": 1.0, "c_puct": 1.25, "parallelism": f"games sharded over {world} rank(s), 1 all-gather per generation"},
            "leaf_evals_per_s": total_evals / elapsed,
            "mean_plies_per_game": mean_plies,
            "positions_gathered_per_step": positions,
            "gnn_forward": {"workload": "BASELINE configs[1]: pv_network_gnn forward (trunk + heads), synthetic boards", "batch": B,
                            "fp16_range_guard": "proven" if gnn_flags & _lib.GNN_RANGE_PROVEN else ("exact_f32" if gnn_flags & _lib.GNN_EXACT_F32 else "tracking"),
                            "range_guard_word_after_run": int(sat_word.item()),
                            "boards_per_s": fwd_rate, "ms": fwd_ms, "boards_per_s_by_batch": fwd_rates, "boards_per_s_by_build": build_rate,
                            "mfma_frac": fwd_rate * FWD_FLOP_PER_BOARD / (PEAK_F16_MFMA / SPLIT_TERMS),
                            "frac_vs_f32_input_mfma_peak": fwd_rate * FWD_FLOP_PER_BOARD / PEAK_F32_MFMA,
                            "hbm_frac_survey_formula": fwd_rate * HBM_BYTES_PER_BOARD / PEAK_HBM},
            "roofline": {"kernel": "gcn_trunk_boards_mm_kernel<TRACK> (TRACK 0: fp16 range proven for this weight set)", "bound": "mfma",
                         "achieved": achieved / 1e12, "peak": PEAK_F16_MFMA / SPLIT_TERMS / 1e12,
                         "unit": "TFLOP/s", "frac": achieved / (PEAK_F16_MFMA / SPLIT_TERMS),
                         "traffic": None,
                         "compulsory_bytes_per_board": TRUNK_COMPULSORY_BYTES_PER_BOARD,
                         "compulsory_hbm_frac": boards_per_s_kernel * TRUNK_COMPULSORY_BYTES_PER_BOARD / PEAK_HBM,
                         "frac_vs_dense_f16_peak": achieved / PEAK_F16_MFMA,
                         "mfma_busy_pmc": pmc,
                         "launches": trunk_launches, "avg_launch_us": trunk_ms / max(trunk_launches, 1) * 1e3,
                         "boards_per_launch_avg": trunk_boards / max(trunk_launches, 1),
                         "flop_per_board": TRUNK_FLOP_PER_BOARD,
                         "hbm_frac_survey_formula": boards_per_s_kernel * HBM_BYTES_PER_BOARD / PEAK_HBM,
                         "frac_vs_f32_input_mfma_peak": achieved / PEAK_F32_MFMA,
                         "f16_mfma_issued_tflops": boards_per_s_kernel * TRUNK_MFMA_PER_BOARD * 16384 / 1e12},
        }
        if rccl_leg is not None:
            out["rccl_group_alive"] = rccl_leg
        if cache_leg is not None:
            for k in ("headline_config", "large_batch"):
                if k in cache_leg:
                    base = out["value"] if k == "headline_config" else (large or {}).get("games_per_s")
                    cache_leg[k]["ratio_to_cache_off"] = cache_leg[k]["games_per_s"] / base if base else None
            out["eval_cache"] = cache_leg
            if "headline_config" in cache_leg:      # the same workload with the (bit-identical) evaluation cache on: NOT the headline, shown beside it
                out["value_with_eval_cache"] = cache_leg["headline_config"]["games_per_s"]
        legal_sample = legal_leg.pop("_sample", None) if legal_leg is not None else None
        if legal_leg is not None:
            out["legal_mask"] = legal_leg
        if graph_leg is not None:
            out["generic_graph_forward"] = graph_leg
        if step_leg is not None:
            out["step_kernel"] = step_leg
        if refill_leg is not None:
            out["slot_refill"] = refill_leg
        if train_leg is not None:
            out["train_step"] = train_leg
        if large is not None:
            out["large_batch"] = large
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.sims, mean_plies)
            out["cpu_baseline"]["host_cpus"] = os.cpu_count()
            if legal_sample is not None:
                out["cpu_baseline"]["legal_mask_states_per_s_per_core"] = cpu_legal_baseline(legal_sample)
        def _round(x):                     # six significant digits: the line stays well under the 8 KB tail a driver keeps
            if isinstance(x, float):
                return float(f"{x:.6g}")
            if isinstance(x, dict):
                return {k: _round(v) for k, v in x.items()}
            if isinstance(x, (list, tuple)):
                return [_round(v) for v in x]
            return x
        print(json.dumps(_round(out), separators=(",", ":")))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
