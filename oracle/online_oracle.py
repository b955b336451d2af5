"""CPU ORACLE for the online-learning step.  TEST INFRASTRUCTURE ONLY (see gnn_oracle.py).

Restates graphnet/graph_score_online.py:9-15 and :62-77: the GNN parameters under torch.optim.Adam(lr, weight_decay),
loss = gnn_score - kw_score + improvement with gnn_score = torch.max(scores, 0) (first maximum) and kw_score the score of
the KW decision, loss.backward(), optimizer.step().  The forward is oracle_forward (gnn_oracle.py) with the parameters as
autograd leaves; the backward pass is torch.autograd's, as in the reference.

Parity pin: tests/golden/*_online.npz hold gradients and parameters produced by the REFERENCE's own
GraphChoice.online_learning (oracle/make_golden_online.py); tests/test_online.py checks this file against them.
"""
from collections import OrderedDict

import numpy as np
import torch

from .gnn_oracle import oracle_forward


class OnlineOracle:
    def __init__(self, state, lr=1e-4, wd=1e-4, T=2):
        self.T = T
        self.params = OrderedDict((k, torch.nn.Parameter(torch.as_tensor(np.asarray(v)).float().clone())) for k, v in state.items())
        self.opt = torch.optim.Adam(list(self.params.values()), lr=lr, weight_decay=wd)       # graph_score_online.py:15

    def blob(self):
        return np.concatenate([p.detach().numpy().reshape(-1) for p in self.params.values()])

    def grad_blob(self):
        return np.concatenate([(p.grad if p.grad is not None else torch.zeros_like(p)).numpy().reshape(-1) for p in self.params.values()])

    def step(self, forward_args, kw_flat, improvement, apply=True):
        """forward_args: the argument tuple of GraphNet.forward for B subproblems; kw_flat (B) flat ReLU indices;
        improvement (B).  Returns (loss per subproblem, ragged scores)."""
        lbs, ubs, duals, prim, x_lp, layers, masks = forward_args
        scores = oracle_forward(self.params, lbs, ubs, duals, prim, x_lp, layers, masks, T=self.T)
        losses = []
        for b, s in enumerate(scores):
            gnn_score, _ = torch.max(s, 0)                                                    # :41
            kw_index = len(masks[b][:int(kw_flat[b])].nonzero())                              # :69
            losses.append(gnn_score - s[kw_index] + float(improvement[b]))                    # :73
        self.opt.zero_grad()                                                                  # :72
        torch.stack(losses).sum().backward()                                                  # :74
        if apply:
            self.opt.step()                                                                   # :75
        return np.array([float(l.detach()) for l in losses], np.float32), [s.detach() for s in scores]
