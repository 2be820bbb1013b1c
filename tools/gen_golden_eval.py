#!/usr/bin/env python3
"""Golden vectors for the evaluate_network row: the reference's own evaluate_network.play()
(/root/reference/evaluate_network.py:25-44) driven by two integer-hash fake models through the reference's
pv_mcts_action (pv_mcts.py:98-104), seeded np.random.  Generation-time only; imports the reference exactly like
tools/gen_golden.py (same stubs), writes tests/golden/eval_<N>x<N>.npz:
  e<k>_cfg      [seed, sims, bias_first, bias_second]
  e<k>_actions  the action of every ply (recorded by wrapping the two action functions)
  e<k>_point    first player's point as returned by play() (1 / 0 / 0.5)
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--board", type=int, default=9)
    ap.add_argument("--out", default=os.path.join(gg.REPO, "tests", "golden"))
    args = ap.parse_args()
    board = args.board
    gl, pv_mcts, self_play, cnn = gg.import_reference(board)
    import evaluate_network as en  # the reference module (imports game_logic / pv_mcts / pv_network_cnn)
    out = {}
    cfgs = [(11, 12, 30, 0), (12, 16, 0, 60), (13, 10, 20, 20)] if board == 9 else [(11, 20, 10, 0), (12, 16, 0, 25), (13, 24, 5, 5)]
    for k, (seed, sims, b0, b1) in enumerate(cfgs):
        pv_mcts.PV_EVALUATE_COUNT = sims
        np.random.seed(seed)
        acts = []

        def wrap(f):
            def g(state):
                a = f(state)
                acts.append(int(a))
                return a
            return g
        na = (wrap(pv_mcts.pv_mcts_action(gg.FakeModel(b0), en.EN_TEMPERATURE, "cpu")),
              wrap(pv_mcts.pv_mcts_action(gg.FakeModel(b1), en.EN_TEMPERATURE, "cpu")))
        point = en.play(na)
        out[f"e{k}_cfg"] = np.asarray([seed, sims, b0, b1], dtype=np.int64)
        out[f"e{k}_actions"] = np.asarray(acts, dtype=np.int16)
        out[f"e{k}_point"] = np.asarray([point], dtype=np.float64)
        print("eval game", k, "plies", len(acts), "point", point, flush=True)
    out["count"] = np.asarray([len(cfgs)])
    np.savez_compressed(os.path.join(args.out, f"eval_{board}x{board}.npz"), **out)


if __name__ == "__main__":
    main()
