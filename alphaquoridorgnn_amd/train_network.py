"""Parameter update -- drop-in for the reference's train_network.py, on the GNN, with the whole optimisation step in HIP.

Call surface kept (train_network.py:14-107): NUM_EPOCH, BATCH_SIZE, load_data, train_network.  One step = one C call
(`aqg_gcn_train_step`, csrc/gcn_train.hip): forward, the reference's losses (CrossEntropyLoss on the already-softmaxed
policy + MSELoss, train_network.py:54-55,85-86), backward, Adam (:56,:90-92), all fp32.  The parameters updated are the
model's own state_dict tensors; torch is used for memory, shuffling and the .pth / .history files only.
"""
import ctypes
import pickle
from pathlib import Path

import numpy as np
import torch

from . import _lib
from .constants import PV_NETWORK_PATH, BOARD_SIZE
from .pv_network_gnn import GNNNetwork, STATE_DICT_KEYS, HIDDEN_DIM, NUM_FEATURES

NUM_EPOCH = 100    # train_network.py:14
BATCH_SIZE = 128   # train_network.py:15
LEARNING_RATE = 0.001   # train_network.py:56


def load_data():
    """Load the latest training data (train_network.py:19-23).  The file is this build's own self_play.write_data()
    output (same schema as the reference's)."""
    history_path = sorted(Path('./data').glob('*.history'))[-1]
    with history_path.open(mode='rb') as f:
        return pickle.load(f)


def lr_lambda(epoch):
    """train_network.py:59-65."""
    if epoch >= 80:
        return 0.25
    elif epoch >= 50:
        return 0.5
    return 1.0


class GNNTrainer:
    """Adam state + workspace for optimisation steps of up to `max_batch` positions on `model` (a GNNNetwork on the GPU)."""

    def __init__(self, model, max_batch=BATCH_SIZE, betas=(0.9, 0.999), eps=1e-8):
        self.model = model
        self.lib = _lib.load()
        sd = model.state_dict()
        self.params = [sd[k] for k in STATE_DICT_KEYS]
        self.dev = _lib.require_gpu(self.params[0].device)
        for p in self.params:
            if p.dtype != torch.float32 or not p.is_contiguous():
                raise ValueError("training needs contiguous float32 parameters")
        # all 14 gradient tensors are views of ONE flat buffer: the data-parallel exchange is a single all-reduce
        self.flat_grads = torch.zeros((sum(p.numel() for p in self.params),), dtype=torch.float32, device=self.dev)
        self.grads, off = [], 0
        for p in self.params:
            self.grads.append(self.flat_grads[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.adam_m = [torch.zeros_like(p) for p in self.params]
        self.adam_v = [torch.zeros_like(p) for p in self.params]
        self.N = model.board_size
        self.V = self.N * self.N
        self.A = model.policy_output_size
        self.max_batch = int(max_batch)
        self.step_count = 0
        self.betas, self.eps = betas, eps
        B, V, A, H = self.max_batch, self.V, self.A, HIDDEN_DIM
        f = dict(dtype=torch.float32, device=self.dev)
        w = self.ws = dict(
            h1=torch.empty((B * 96, H), **f), h2=torch.empty((B * 96, H), **f),    # 96 rows per position: include/aqgnn.h
            h3=torch.empty((B * V, H), **f),
            zbuf=torch.empty((B * V, H), **f), dh=torch.empty((B * V, H), **f),
            g=torch.empty((B, H), **f), dg=torch.empty((B, H), **f), hp=torch.empty((B, H // 2), **f),
            hv=torch.empty((B, H // 2), **f), dhp=torch.empty((B, H // 2), **f), dhv=torch.empty((B, H // 2), **f),
            lg=torch.empty((B, A), **f), pol=torch.empty((B, A), **f), vp=torch.empty((B,), **f), val=torch.empty((B,), **f),
            loss=torch.empty((B, 2), **f), part=torch.empty((B * _lib.TRAIN_PART_FLOATS,), **f))
        t = self.t = _lib.TrainStruct()
        t.board_size, t.policy_size = self.N, self.A
        t.beta1, t.beta2, t.eps = float(betas[0]), float(betas[1]), float(eps)
        for name, tensors in (("params", self.params), ("grads", self.grads), ("adam_m", self.adam_m), ("adam_v", self.adam_v)):
            arr = getattr(t, name)
            for i, x in enumerate(tensors):
                arr[i] = x.data_ptr()
        for name, x in w.items():
            setattr(t, name, x.data_ptr())

    def _call(self, states72, pi_target, z_target, mode):
        _lib.check(self.lib.aqg_gcn_train_step(ctypes.byref(self.t), _lib.ptr(states72), _lib.ptr(pi_target), _lib.ptr(z_target),
                                               mode, _lib.stream_ptr(self.dev)), "aqg_gcn_train_step")

    def step(self, states72, pi_target, z_target, lr=LEARNING_RATE, update=True, group=None):
        """One optimisation step.  states72 uint8 [B,72], pi_target float32 [B,A], z_target float32 [B] (device tensors).
        Returns (policy_loss, value_loss) as 0-dim device tensors -- no host synchronisation.

        Data parallel (torch.distributed initialised, world > 1): every rank passes ITS shard of the global batch; the
        local mean-loss gradients are weighted by B_local / B_global, summed with ONE all-reduce of the flat gradient buffer
        (RCCL over xGMI with backend nccl) and applied by every rank, so all replicas take the step a single process would
        take on the whole batch (up to fp32 summation order)."""
        import torch.distributed as dist
        B = int(states72.shape[0])
        if B > self.max_batch:
            raise ValueError("batch larger than the trainer's workspace")
        states72 = states72.to(self.dev, torch.uint8).contiguous()
        pi_target = pi_target.to(self.dev, torch.float32).contiguous()
        z_target = z_target.to(self.dev, torch.float32).contiguous()
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if update:
            self.step_count += 1
        self.t.batch, self.t.step, self.t.lr = B, max(self.step_count, 1), float(lr)
        if world == 1:
            self._call(states72, pi_target, z_target, 1 if update else 0)
            loss = self.ws["loss"][:B].mean(dim=0) if B else torch.zeros((2,), device=self.dev)
        else:
            if B:
                self._call(states72, pi_target, z_target, 0)
                lsum = self.ws["loss"][:B].sum(dim=0)
            else:                                                  # a rank may hold no position of a ragged last batch
                self.flat_grads.zero_()
                lsum = torch.zeros((2,), device=self.dev)
            tot = torch.cat([lsum, torch.tensor([float(B)], device=self.dev)])
            dist.all_reduce(tot, group=group)                      # global loss sums and global batch size
            self.flat_grads.mul_(float(B) / tot[2])                # local mean-loss gradient -> its share of the global mean
            dist.all_reduce(self.flat_grads, group=group)
            if update:
                self._call(states72, pi_target, z_target, 2)
            loss = tot[:2] / tot[2]
        if update:
            self.model.invalidate_packed()
        return loss[0], loss[1]

    def outputs(self, B):
        """(policy [B,A], value [B]) of the last step's forward pass."""
        return self.ws["pol"][:B], self.ws["val"][:B]

    def run_epoch(self, states72, pi_target, z_target, order, lr=LEARNING_RATE, batch=None, pre_shuffle=True):
        """All optimisation steps of one epoch in ONE library call (single process): step i trains on the positions
        order[i*batch:(i+1)*batch] of the resident device arrays (train_network.py:72-95; the short last batch is kept, as
        DataLoader does).  Returns the epoch's summed (policy_loss, value_loss) as a device tensor [2] -- no host sync."""
        batch = self.max_batch if batch is None else int(batch)
        if batch > self.max_batch:
            raise ValueError("batch larger than the trainer's workspace")
        n = int(order.shape[0])
        sums = torch.zeros((2,), dtype=torch.float32, device=self.dev)
        if n == 0:
            return sums
        # The shuffle is applied ONCE per epoch (three gathers, ~1 KB per position) and the steps then index the shuffled
        # copies directly: every kernel of a step starts with cold loads, and reading order[] first would put one more
        # HBM round trip in front of each of them.  (The C entry point also takes the order itself: order != NULL.)
        order = order.to(self.dev, torch.int64).contiguous()
        if pre_shuffle:
            states72, pi_target, z_target = (x.index_select(0, order).contiguous() for x in (states72, pi_target, z_target))
        self.t.batch, self.t.step, self.t.lr = batch, self.step_count + 1, float(lr)
        _lib.check(self.lib.aqg_gcn_train_steps(ctypes.byref(self.t), _lib.ptr(states72), _lib.ptr(pi_target), _lib.ptr(z_target),
                                                None if pre_shuffle else _lib.ptr(order), n, _lib.ptr(sums),
                                                _lib.stream_ptr(self.dev)), "aqg_gcn_train_steps")
        self.step_count += (n + batch - 1) // batch
        self.model.invalidate_packed()
        return sums


def train_network():
    """train_network.py:26-107 on the GNN: best.pth -> NUM_EPOCH epochs over the newest .history -> latest.pth.

    Under torch.distributed rank 0 trains ALONE by default and the others wait at the barrier: a step at the reference's batch
    size (128) is a latency-bound 0.083 ms on one GPU; dealt over W ranks every rank would still run its (shorter-gridded but
    equally long) board kernel and add two collectives per step, i.e. data parallelism makes this loop slower, not faster.
    AQG_TRAIN_DATA_PARALLEL=1 selects the data-parallel form anyway (same shuffle on all ranks, the positions of every batch
    dealt out over the ranks, one all-reduce of the flat gradient buffer per step, GNNTrainer.step); rank 0 writes latest.pth."""
    import os
    import torch.distributed as dist
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
    if world > 1 and os.environ.get("AQG_TRAIN_DATA_PARALLEL", "0") != "1":
        from . import distributed as aqd
        tag = aqd.next_tag("train")
        if rank == 0:
            with aqd.single_rank_stage(tag):       # published on failure too (value b"fail"): the idle ranks raise instead of waiting
                _train_single_process()
        else:
            aqd.wait_for_rank0(tag)     # host-side wait on the rendezvous store: no collective is pending while rank 0 trains, so
                                        # the stage may outlast the process group's watchdog timeout (distributed.py)
        dist.barrier()          # latest.pth is complete before any rank moves on to the evaluation stage
        return
    _train_loop(rank, world)


def _train_single_process():
    _train_loop(0, 1)


def _train_loop(rank, world):
    import torch.distributed as dist
    from . import distributed as aqd
    dev = aqd.device()                                                             # this rank's GPU, explicitly
    model = GNNNetwork()
    model.load_state_dict(torch.load(PV_NETWORK_PATH + 'best.pth', map_location=dev, weights_only=True))
    model = model.to(dev)
    history = load_data()
    s, p, v = zip(*history)
    s = torch.from_numpy(model.preprocess_input(s)).to(dev)                        # uint8 [n,72]
    p = torch.tensor(np.array(p), dtype=torch.float32, device=dev)                 # policy targets
    v = torch.tensor(np.array(v), dtype=torch.float32, device=dev)                 # value targets
    n = s.shape[0]
    trainer = GNNTrainer(model, max_batch=BATCH_SIZE)
    for epoch in range(NUM_EPOCH):
        lr = LEARNING_RATE * lr_lambda(epoch)                                      # LambdaLR, stepped once per epoch (:98)
        perm = torch.randperm(n, device=dev)                                    # DataLoader(shuffle=True), last batch kept
        if world > 1:
            if dist.get_backend() == 'nccl':
                dist.broadcast(perm, src=0)
            else:                                                                  # gloo (tests): host tensors only
                perm_h = perm.cpu()
                dist.broadcast(perm_h, src=0)
                perm = perm_h.to(dev)
        if world == 1:
            epoch_policy_loss, epoch_value_loss = trainer.run_epoch(s, p, v, perm, lr=lr)
        else:
            epoch_policy_loss = torch.zeros((), device=dev)
            epoch_value_loss = torch.zeros((), device=dev)
            for i in range(0, n, BATCH_SIZE):
                idx = perm[i:i + BATCH_SIZE][rank::world]
                pl, vl = trainer.step(s[idx], p[idx], v[idx], lr=lr)
                epoch_policy_loss += pl
                epoch_value_loss += vl
        if rank == 0:
            print(f"\rEpoch {epoch + 1}/{NUM_EPOCH} | Policy Loss: {float(epoch_policy_loss):.4f} | Value Loss: {float(epoch_value_loss):.4f}", end='')
    if rank == 0:
        print('')
        torch.save(model.state_dict(), PV_NETWORK_PATH + 'latest.pth')
    if world > 1:
        dist.barrier()          # latest.pth is complete before any rank moves on to the evaluation stage


if __name__ == '__main__':
    train_network()
