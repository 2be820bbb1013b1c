"""The learning loop, every stage on the MI355X path -- drop-in for the reference's train_cycle.py:10-45.

    create_network()  ->  repeat NUM_TRAIN_CYCLE times:  self_play()  ->  train_network()  ->  evaluate_network()

(The reference's commented-out evaluate_best_player stage -- CPU baseline agents, SURVEY 8f.4 -- is not part of this build.)
"""
from . import constants
from . import distributed as aqd
from .evaluate_network import evaluate_network
from .pv_network_gnn import create_network
from .self_play import self_play
from .train_network import train_network

NUM_TRAIN_CYCLE = 1000   # train_cycle.py:18

_STAGES = (("self-play", self_play), ("parameter update", train_network), ("evaluation of the new parameters", evaluate_network))


def _dist():
    import torch.distributed as dist
    on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    return dist, on, (dist.get_rank() if on else 0)


def _create_network_once():
    """create_network() draws random weights: under torch.distributed only rank 0 may write best.pth (every rank would
    otherwise save a different random init to the same path); the others wait for the file."""
    dist, on, rank = _dist()
    if rank == 0:
        create_network()
    if on:
        dist.barrier()


def _evaluate_once():
    """evaluate_network() samples its games from the global numpy RNG: rank 0 plays the match and decides, the decision
    is broadcast, and nobody reads best.pth again before the copy is done."""
    import torch
    dist, on, rank = _dist()
    tag = aqd.next_tag("evaluate")
    if rank == 0:
        with aqd.single_rank_stage(tag):           # the key is published even if the match raises: idle ranks wake up and raise too
            promoted = evaluate_network()
    else:
        promoted = False
        aqd.wait_for_rank0(tag)          # host-side wait: no collective is pending while rank 0 plays the match
    if on:
        t = torch.tensor([1 if promoted else 0], dtype=torch.int64, device=aqd.collective_device())
        dist.broadcast(t, src=0)
        promoted = bool(int(t.item()))
        dist.barrier()
    return promoted


def train_cycle(num_cycles=None):
    """Run the cycle; returns, per iteration, whether `latest` was promoted to `best`."""
    total = NUM_TRAIN_CYCLE if num_cycles is None else int(num_cycles)
    print(f'{constants.PV_NETWORK_NAME} network, {constants.BOARD_SIZE}x{constants.BOARD_SIZE} board, {total} training cycle(s)')
    _create_network_once()
    promoted = []
    rank = _dist()[2]
    for cycle in range(1, total + 1):
        outcome = None
        for title, stage in _STAGES:
            if rank == 0:
                print(f'\n[cycle {cycle}/{total}] {title}')
            outcome = _evaluate_once() if stage is evaluate_network else stage()
        promoted.append(bool(outcome))
    return promoted


def main(argv=None):
    """`python -m alphaquoridorgnn_amd.train_cycle` -- also the per-rank program of
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P -m
    alphaquoridorgnn_amd.train_cycle`: every rank binds to GPU LOCAL_RANK and joins the RCCL group (distributed.init_from_env)
    before anything touches a device; self-play is sharded over the ranks, rank 0 trains / evaluates (module docstrings)."""
    import argparse
    import json
    import os
    from . import evaluate_network as en, pv_mcts, self_play as sp, train_network as tn
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--cycles", type=int, default=None, help="training cycles (default NUM_TRAIN_CYCLE = 1000, train_cycle.py:18)")
    ap.add_argument("--games", type=int, default=None, help="self-play games per generation over ALL ranks (default SP_GAME_COUNT)")
    ap.add_argument("--sims", type=int, default=None, help="simulations per move (default pv_mcts.PV_EVALUATE_COUNT)")
    ap.add_argument("--epochs", type=int, default=None, help="epochs per parameter update (default train_network.NUM_EPOCH)")
    ap.add_argument("--eval-games", type=int, default=None, help="games per evaluation (default EN_GAME_COUNT)")
    ap.add_argument("--result-dir", default=None, help="every rank writes train_cycle.rank<r>.json (promotions, sha256 of latest.pth) here")
    args = ap.parse_args(argv)
    rank, world = aqd.init_from_env()
    if args.games is not None:
        sp.SP_GAME_COUNT = args.games
    if args.sims is not None:
        pv_mcts.PV_EVALUATE_COUNT = args.sims
    if args.epochs is not None:
        tn.NUM_EPOCH = args.epochs
    if args.eval_games is not None:
        en.EN_GAME_COUNT = args.eval_games
    try:
        promoted = train_cycle(args.cycles)
        if args.result_dir:
            import hashlib
            with open(constants.PV_NETWORK_PATH + 'latest.pth', 'rb') as f:
                digest = hashlib.sha256(f.read()).hexdigest()
            import torch
            with open(os.path.join(args.result_dir, f"train_cycle.rank{rank}.json"), "w") as f:
                json.dump({"rank": rank, "world": world, "promoted": promoted, "latest_sha256": digest,
                           "device": torch.cuda.current_device()}, f)
    except BaseException:
        aqd.shutdown(ok=False)                      # no barrier on the way out of an exception: the other ranks may never reach one
        raise
    aqd.shutdown()
    return promoted


if __name__ == '__main__':
    main()
