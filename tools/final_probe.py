import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from alphaquoridorgnn_amd import _lib
from alphaquoridorgnn_amd.pv_network_gnn import GNNNetwork
from alphaquoridorgnn_amd.train_network import GNNTrainer, BATCH_SIZE
from tools.microbench import synth_states
dev = _lib.require_gpu("cuda:0")
model = GNNNetwork().to(dev); tr = GNNTrainer(model, max_batch=BATCH_SIZE)
st = synth_states(BATCH_SIZE); A = model.policy_output_size
pi = torch.rand((BATCH_SIZE, A), device=dev); pi = pi / pi.sum(1, keepdim=True)
z = torch.randint(-1, 2, (BATCH_SIZE,), device=dev).float()
for _ in range(50):
    tr.step(st, pi, z, update=False)        # board kernel + final(compute, no update)
    tr.t.step = 1
    tr._call(st, pi, z, 2)                   # final(update only)
    tr._call(st, pi, z, 2)
torch.cuda.synchronize()
print("done")
