"""torch-autograd fp64 restatement of ONE training step of the reference's loop -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/train_network.py:
  :54  policy_loss_fn = nn.CrossEntropyLoss()      applied at :85 to the model's ALREADY-softmaxed policy and to
                                                   probability targets (the double softmax is the reference's, kept)
  :55  value_loss_fn  = nn.MSELoss()                applied at :86 to value_pred.squeeze() (tanh output)
  :56  optim.Adam(model.parameters(), lr=0.001)     torch defaults betas (0.9, 0.999), eps 1e-8
  :59-66 LambdaLR factors 1.0 / 0.5 (epoch >= 50) / 0.25 (epoch >= 80)
  :89-92 loss = policy_loss + value_loss; zero_grad; backward; step
on GraphPolicyValueNetwork (/root/reference/pv_network_gnn.py:23-64) with the GCN layer of oracle/gnn.py (PyG defaults,
"parity unpinned" against PyG itself, see oracle/gnn.py).  The network arithmetic is restated with dense normalised
adjacency matrices so that torch autograd provides the gradients; losses and optimizer ARE torch's own, i.e. what the
reference calls.  The forward of this file is checked against oracle/gnn.py in tests/test_oracle_golden.py.
"""
import numpy as np
import torch

from . import gnn as og

KEYS = og.KEYS


def lr_lambda(epoch):
    """train_network.py:59-65."""
    if epoch >= 80:
        return 0.25
    if epoch >= 50:
        return 0.5
    return 1.0


def dense_adjacency(rec):
    """D^-1/2 (A + I) D^-1/2 of one board graph as a dense [V, V] float64 array (PyG gcn_norm, oracle/gnn.py:76-92)."""
    V = int(rec[70]) ** 2
    e = og.board_edges(rec)
    A = np.zeros((V, V), dtype=np.float64)
    A[e[1], e[0]] = 1.0                       # row = destination
    A[np.arange(V), np.arange(V)] = 1.0       # one self loop per node
    deg = A.sum(axis=1)
    dis = deg ** -0.5
    return dis[:, None] * A * dis[None, :]


class TorchGNN(torch.nn.Module):
    def __init__(self, params):
        super().__init__()
        self.p = torch.nn.ParameterDict({k.replace(".", "__"): torch.nn.Parameter(torch.tensor(np.asarray(v), dtype=torch.float64))
                                         for k, v in params.items()})

    def w(self, key):
        return self.p[key.replace(".", "__")]

    def forward(self, recs):
        recs = np.asarray(recs, dtype=np.uint8).reshape(-1, 72)
        x = torch.tensor(np.stack([og.node_features(r) for r in recs]), dtype=torch.float64)            # [B, V, 6]
        adj = torch.tensor(np.stack([dense_adjacency(r) for r in recs]), dtype=torch.float64)           # [B, V, V]
        h = x
        for l in range(3):
            h = torch.relu(adj @ (h @ self.w(f"gcn_layers.{l}.lin.weight").T) + self.w(f"gcn_layers.{l}.bias"))
        g = h.mean(dim=1)
        hp = torch.relu(g @ self.w("policy_head.0.weight").T + self.w("policy_head.0.bias"))
        policy = torch.softmax(hp @ self.w("policy_head.2.weight").T + self.w("policy_head.2.bias"), dim=1)   # pv_network_gnn.py:42
        hv = torch.relu(g @ self.w("value_head.0.weight").T + self.w("value_head.0.bias"))
        value = torch.tanh(hv @ self.w("value_head.2.weight").T + self.w("value_head.2.bias"))               # [B, 1]
        return policy, value


def train_steps(params, batches, lr=0.001, epoch_of_step=None):
    """Run len(batches) optimisation steps (each batch = (recs, pi [B,A], z [B])) from `params` (dict of arrays).
    Returns a list with, per step: dict(grads, params_after, policy_loss, value_loss, policy, value)."""
    model = TorchGNN(params)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    ce, mse = torch.nn.CrossEntropyLoss(), torch.nn.MSELoss()
    out = []
    for i, (recs, pi, z) in enumerate(batches):
        if epoch_of_step is not None:
            for gparam in opt.param_groups:
                gparam["lr"] = lr * lr_lambda(epoch_of_step[i])
        policy, value = model(recs)
        pl = ce(policy, torch.tensor(np.asarray(pi), dtype=torch.float64))
        vl = mse(value.squeeze(), torch.tensor(np.asarray(z), dtype=torch.float64))
        loss = pl + vl
        opt.zero_grad()
        loss.backward()
        grads = {k: model.w(k).grad.detach().numpy().copy() for k in KEYS}
        opt.step()
        out.append(dict(grads=grads, params_after={k: model.w(k).detach().numpy().copy() for k in KEYS},
                        policy_loss=float(pl.detach()), value_loss=float(vl.detach()), policy=policy.detach().numpy().copy(),
                        value=value.detach().numpy()[:, 0].copy()))
    return out
