#!/usr/bin/env python3
"""Golden vectors for the baseline agents from the REAL reference agents.py -- generation-time tooling only
(build container; /root/reference imported read-only through tools/gen_golden.py's harness).  One board size per run:

    python tools/gen_golden_agents.py --board 5        -> tests/golden/agents_5x5.npz

Stored per state: the state72 record, heuristic_eval, both shortest paths, alpha_beta_action(max_depth 2 and 1),
random_action / mcts_action under a recorded `random.seed`.  On 9x9 the reference's pure-Python alpha-beta is only feasible
on positions with few legal actions (no walls left), so those are what the 9x9 file holds."""
import argparse, os, random, sys
sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import gen_golden as gg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--board", type=int, default=5)
    args = ap.parse_args()
    board = args.board
    gl, pv_mcts, self_play, cnn = gg.import_reference(board)
    import agents
    rng = np.random.RandomState(7)
    states = []
    # random legal play; keep positions along the way (on 9x9 only wall-less ones: see the docstring)
    for game in range(40 if board == 9 else 12):
        s = gl.State()
        while not s.is_done():
            la = s.legal_actions()
            walls = [a for a in la if a >= board * board]
            a = walls[rng.randint(len(walls))] if walls and rng.rand() < (0.9 if board == 9 else 0.4) else la[rng.randint(len(la))]
            s = s.next(a)
            if s.is_done():
                break
            if board == 9:
                if s.player[1] == 0 and s.enemy[1] == 0 and rng.rand() < 0.25:
                    states.append(s)
            elif rng.rand() < 0.3:
                states.append(s)
    states = states[:30]
    out = {"board": np.asarray([board]), "max_dist": np.asarray([agents.MAX_DIST_FROM_GOAL])}
    recs, heur, spp, ab2, ab1, rnd, rnd_seed = [], [], [], [], [], [], []
    for i, s in enumerate(states):
        recs.append(gg.rec_of(s))
        heur.append(agents.heuristic_eval(s))
        ab2.append(agents.alpha_beta_action(s, 2))
        ab1.append(agents.alpha_beta_action(s, 1))
        random.seed(100 + i)
        rnd.append(agents.random_action(s)); rnd_seed.append(100 + i)
        print(i, "legal", len(s.legal_actions()), "heur", heur[-1], "ab2", ab2[-1], flush=True)
    out.update(states=np.stack(recs), heuristic=np.asarray(heur, dtype=np.float64), ab2=np.asarray(ab2, dtype=np.int16),
               ab1=np.asarray(ab1, dtype=np.int16), random=np.asarray(rnd, dtype=np.int16), random_seed=np.asarray(rnd_seed))
    m_idx, m_act, m_seed = [], [], []
    for i in range(0, len(states), max(1, len(states) // (3 if board >= 5 else 6))):
        random.seed(500 + i)
        m_act.append(agents.mcts_action(states[i])); m_idx.append(i); m_seed.append(500 + i)
        print("mcts", i, m_act[-1], flush=True)
    out.update(mcts_index=np.asarray(m_idx), mcts_action=np.asarray(m_act, dtype=np.int16), mcts_seed=np.asarray(m_seed))
    path = os.path.join(REPO, "tests", "golden", f"agents_{board}x{board}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(states), "states")


if __name__ == "__main__":
    main()
