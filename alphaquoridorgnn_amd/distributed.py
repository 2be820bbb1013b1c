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
in the (now immediate) barrier.  A stage that FAILS publishes the key too (value b"fail", `single_rank_stage`): the idle ranks wake
up and raise within seconds, and `shutdown(ok=False)` tears the group down without a barrier, so the job ends with rank 0's own
exception instead of a barrier timeout hours later.

AQG_DIST_FORCE_GROUP=1 creates the process group even for a single rank (WORLD_SIZE absent or 1) and makes `engine.gather_history`
run its collectives at world size 1: the RCCL communicator, its streams and the all-gather code path execute on ONE GPU
(tests/test_gpu_parity.py::test_rccl_world_size_1_self_play_equals_no_group, bench.py leg `rccl_group_alive`).
"""
import contextlib
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
    if dist.is_available() and dist.is_initialized():
        return rank_world()
    if world <= 1:
        if not force_group():
            return rank_world()
        os.environ.setdefault("RANK", "0")                  # a one-rank group on this GPU (see the module docstring)
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
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


def force_group():
    """AQG_DIST_FORCE_GROUP=1: the group exists and the exchange step's collectives run even with one rank."""
    return os.environ.get("AQG_DIST_FORCE_GROUP", "0") == "1"


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def shutdown(ok=True):
    """Leave the group.  ok=True (clean exit): all ranks meet in a barrier first.  ok=False (this rank is unwinding an exception):
    NO barrier -- the other ranks may never reach one -- the group is destroyed (aborted where the backend offers it) and the caller
    re-raises, so the launcher sees the failure at once."""
    import torch.distributed as dist
    if _state["initialised_here"] and dist.is_initialized():
        if ok:
            dist.barrier()
            dist.destroy_process_group()
        else:
            try:
                pg = dist.distributed_c10d._get_default_group()
                if hasattr(pg, "abort"):
                    pg.abort()                              # RCCL: drop pending work instead of draining it
                dist.destroy_process_group()
            except Exception:
                pass
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


class Rank0StageFailed(RuntimeError):
    """Raised on the idle ranks when rank 0's single-rank stage raised (its own traceback is rank 0's)."""


def release_ranks(tag, ok=True):
    """Rank 0: publish that the single-rank stage `tag` is complete -- or, ok=False, that it failed (see wait_for_rank0)."""
    if not is_distributed():
        return
    st = _store()
    if st is not None:
        st.set(f"aqg/{tag}", b"1" if ok else b"fail")


@contextlib.contextmanager
def single_rank_stage(tag):
    """Rank 0 wraps its solitary work in this: the stage key is published on the way out WHATEVER happened, so the ranks parked in
    wait_for_rank0(tag) never outlive a failure (they raise Rank0StageFailed)."""
    try:
        yield
    except BaseException:
        release_ranks(tag, ok=False)
        raise
    release_ranks(tag, ok=True)


def wait_for_rank0(tag, poll_hours=240):
    """Ranks other than 0: host-side wait (rendezvous store, no collective in flight, so no watchdog) until rank 0 has
    published the stage `tag`; raises Rank0StageFailed if rank 0 failed in it.  Falls back to nothing when the store is
    unavailable -- the barrier that follows then waits, bounded by AQG_DIST_TIMEOUT_S."""
    if not is_distributed():
        return
    st = _store()
    if st is not None:
        st.wait([f"aqg/{tag}"], datetime.timedelta(hours=poll_hours))
        if bytes(st.get(f"aqg/{tag}")) == b"fail":
            raise Rank0StageFailed(f"rank 0 failed in stage {tag}")


def next_tag(prefix):
    """A tag every rank derives identically (call sites are reached in lock-step)."""
    _state["seq"] += 1
    return f"{prefix}/{_state['seq']}"
