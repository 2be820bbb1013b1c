#!/usr/bin/env python3
"""Round 4: how does the evaluation cache behave on a network that is not random any more?  Runs a few train_cycle iterations in a scratch
directory (CYCLES x {GAMES self-play games at SIMS_TRAIN sims, EPOCHS epochs, evaluation}), then for the random-weight network it started
from and for the network it ended with: one 2,048-game x 200-sim generation with the table off and on (games/s, hit rate, mean plies,
share of draws)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
work = tempfile.mkdtemp(prefix="aqg_trained_")
os.chdir(work)
from alphaquoridorgnn_amd import _lib, constants, pv_mcts, self_play as sp, train_network as tn, evaluate_network as en, train_cycle as tc
from alphaquoridorgnn_amd.engine import MultiSetSelfPlay
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
dev = _lib.require_gpu("cuda:0")
CYCLES = int(os.environ.get("CYCLES", "4"))
pv_mcts.PV_EVALUATE_COUNT = int(os.environ.get("SIMS_TRAIN", "50"))
sp.SP_GAME_COUNT = int(os.environ.get("GAMES", "1024"))
tn.NUM_EPOCH = int(os.environ.get("EPOCHS", "20"))
en.EN_GAME_COUNT = 20


def measure(tag, model):
    for slots in (0, 8192):
        eng = MultiSetSelfPlay(model, num_games=2048, sims=200, num_sets=4, seed=1000, eval_cache_slots=slots)
        for rep in range(2):
            eng.reset()
            eng.sync(); t0 = time.perf_counter()
            c = eng.play_generation()
            eng.sync(); dt = time.perf_counter() - t0
        st, vis, z = eng.history_tensors()
        plies = float(st.shape[0]) / 2048
        res = torch.cat([e.t["game_result"] for e in eng.sets]).float()
        print(f"{tag}: slots {slots:5d}: {c['finished'] / dt:8.1f} games/s  hit rate {c['cache_hits'] / max(c['leaf_evals'], 1):.3f}  mean plies {plies:6.1f}  "
              f"draws {(res == 0).float().mean().item():.2f}  fp16 guard {'exact kernels' if eng.sets[0].e.gnn_flags & _lib.GNN_EXACT_F32 else 'split kernels'}", flush=True)
        del eng
        torch.cuda.empty_cache()


t0 = time.time()
if os.environ.get("MODEL_PATH") and os.path.exists(os.environ["MODEL_PATH"]):      # measure a saved network only (A/B of library builds)
    m = GNNNetwork()
    m.prep_for_inference(os.environ["MODEL_PATH"])
    measure("saved network  ", m)
    sys.exit(0)
print("PROMOTED", tc.train_cycle(num_cycles=CYCLES), f"({time.time() - t0:.0f} s)", flush=True)
if os.environ.get("SAVE_LATEST"):
    import shutil
    shutil.copy(constants.PV_NETWORK_PATH + "latest.pth", os.environ["SAVE_LATEST"])
rand = GNNNetwork().to(dev).eval()
torch.manual_seed(0)
measure("random weights ", GNNNetwork().to(dev).eval())
for name in ("best", "latest"):
    m = GNNNetwork()
    m.prep_for_inference(constants.PV_NETWORK_PATH + name + ".pth")
    measure(f"after {CYCLES} cycles ({name})", m)
