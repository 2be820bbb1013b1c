"""One process per GPU: the multi-rank entry of the learning loop (the reference's train_cycle.py:21-39 is a single-process script).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
        -m alphaquoridorgnn_amd.train_cycle            (or  -m alphaquoridorgnn_amd.self_play)

`init_from_env()` is what those entry points call first: it binds the rank to ITS GPU (LOCAL_RANK) before anything touches the
device, then joins the process group -- backend "nccl" (= RCCL over xGMI on ROCm) by default, AQG_DIST_BACKEND=gloo for a
rehearsal with several ranks on one GPU (host tensors).  Every module of this package allocates on `device()`, the rank's
current device, never on a bare 'cuda'.

Long single-rank stages (rank 0 trains / evaluates alone, train_network.py and train_cycle.py) must not leave the other ranks
parked inside a collective: with RCCL a pending barrier is bounded by the process-group watchdog (AQG_DIST_TIMEOUT_S, default
two hours here instead of torch's ten minutes) and would abort the job once a stage outgrows it (25,000-game generations, the
reference's original constants).  `wait_for_rank0(tag)` therefore parks the idle ranks on a key of the rendezvous store -- a
host-side wait with no collective in flight -- until rank 0 publishes it with `release_ranks(tag)`; only then do all ranks meet
in the (now immediate) barrier.
"""
import datetime
import os

import torch

_state = {"initialised_here": False, "seq": 0}


def device():
    """The rank's GPU as an explicit torch.device (never the bare 'cuda' alias, whose meaning is 'GPU 0' until set_device ran)."""
    return torch.device("cuda", torch.cuda.current_device())


def is_distributed():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend=None):
    """Bind this process to its GPU and join the process group described by torchrun's environment (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR, MASTER_PORT).  A plain `python -m ...` start (no WORLD_SIZE, or WORLD_SIZE=1) stays single-process.
    Returns (rank, world_size)."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpu = torch.cuda.device_count()                       # does not initialise the runtime
    if n_gpu > 0:
        torch.cuda.set_device(local % n_gpu)                # before ANY other GPU call; ranks beyond the GPU count share (rehearsals)
    if world <= 1 or (dist.is_available() and dist.is_initialized()):
        return rank_world()
    backend = backend or os.environ.get("AQG_DIST_BACKEND", "nccl")
    timeout = datetime.timedelta(seconds=int(os.environ.get("AQG_DIST_TIMEOUT_S", "7200")))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend == "nccl":
        if n_gpu == 0:
            raise RuntimeError("backend nccl (RCCL) needs a GPU per rank")
        dist.init_process_group("nccl", device_id=device(), timeout=timeout)
    else:
        dist.init_process_group(backend, timeout=timeout)
    _state["initialised_here"] = True
    return dist.get_rank(), dist.get_world_size()


def shutdown():
    import torch.distributed as dist
    if _state["initialised_here"] and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
        _state["initialised_here"] = False


def collective_device():
    """Where the small control tensors of a collective live: the rank's GPU under RCCL, the host under gloo."""
    import torch.distributed as dist
    return device() if dist.get_backend() == "nccl" else torch.device("cpu")


def _store():
    import torch.distributed as dist
    try:
        return dist.distributed_c10d._get_default_store()
    except Exception:
        return None


def release_ranks(tag):
    """Rank 0: publish that the single-rank stage `tag` is complete (see wait_for_rank0)."""
    if not is_distributed():
        return
    st = _store()
    if st is not None:
        st.set(f"aqg/{tag}", b"1")


def wait_for_rank0(tag, poll_hours=240):
    """Ranks other than 0: host-side wait (rendezvous store, no collective in flight, so no watchdog) until rank 0 has
    called release_ranks(tag).  Falls back to nothing when the store is unavailable -- the barrier that follows then waits, bounded
    by AQG_DIST_TIMEOUT_S."""
    if not is_distributed():
        return
    st = _store()
    if st is not None:
        st.wait([f"aqg/{tag}"], datetime.timedelta(hours=poll_hours))


def next_tag(prefix):
    """A tag every rank derives identically (call sites are reached in lock-step)."""
    _state["seq"] += 1
    return f"{prefix}/{_state['seq']}"
