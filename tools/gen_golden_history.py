#!/usr/bin/env python3
"""Golden fixture for the ON-DISK training-data format: what the reference's self_play.write_data() pickles
and how its train_network.load_data() + train_network() read it back.

Generation-time tooling only (build container; /root/reference is imported read-only through tools/gen_golden.py's
harness, never copied).  One seeded game of the reference's own self_play.play() (FakeModel evaluator, as in games_9x9.npz)
is (1) kept as the in-memory `history` list, (2) written by the reference's write_data() into a scratch directory and read
back by the reference's load_data() -- the round trip must give the same list --, and (3) unzipped exactly as
train_network.py:37-46 does (s, p, v = zip(*history); np.array(p); np.array(v)).  The fixture is DATA only, as JSON:
the nested list (ints and floats keep their Python types in JSON) plus the arrays the reference derives from it.
"""
import glob
import json
import os
import sys

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import gen_golden as gg  # noqa: E402


def main():
    board = 9
    gl, pv_mcts, self_play, cnn = gg.import_reference(board)     # chdir()s into a scratch directory
    import train_network
    seed, sims, bias = 77, 24, 60                                # == games_9x9.npz game 2 (27 plies)
    pv_mcts.PV_EVALUATE_COUNT = sims
    np.random.seed(seed)
    hist = self_play.play(gg.FakeModel(bias), "cpu")
    self_play.write_data(hist)                                   # ./data/YYYYMMDDhhmmss.history in the scratch directory
    files = glob.glob("./data/*.history")
    assert len(files) == 1
    back = train_network.load_data()                             # sorted(glob)[-1] -> pickle.load
    assert back == hist
    # JSON keeps int vs float: the reference leaves int 0 on illegal actions and python floats on legal ones
    doc = {
        "about": "reference self_play.play() history as pickled by write_data (self_play.py:30-37,51-54,63-66), seed 77, 24 sims, FakeModel(60), numpy " + np.__version__,
        "file_name_pattern": os.path.basename(files[0]),
        # (positions reached through np.random.choice carry numpy.int64 pawn positions inside the pickled lists; JSON stores
        #  their values, `numpy_scalars_in_states` records that the reference's file contains such scalars)
        "history": json.loads(json.dumps(back, default=lambda o: o.item())),
        "numpy_scalars_in_states": bool(any(isinstance(x, np.generic) for h in back for part in h[0] for x in part)),
    }
    s, p, v = zip(*back)                                         # train_network.py:37
    net = cnn.CNNNetwork.__new__(cnn.CNNNetwork)
    planes = cnn.CNNNetwork.preprocess_input(net, s)             # train_network.py:39 (6 feature planes per position)
    doc["train_p_shape"] = list(np.array(p).shape)               # train_network.py:41
    doc["train_v"] = np.array(v).tolist()                        # train_network.py:42
    doc["train_planes_sum"] = float(np.asarray(planes, dtype=np.float64).sum())
    out = os.path.join(REPO, "tests", "golden", "history_9x9.json")
    with open(out, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
    print("wrote", out, os.path.getsize(out), "bytes;", len(back), "positions; file", doc["file_name_pattern"])


if __name__ == "__main__":
    main()
