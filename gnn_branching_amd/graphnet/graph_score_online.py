"""``graphnet.graph_score_online`` surface of the reference: the online-learning ``GraphChoice``.

Same constructor, ``decision``, ``online_learning`` and ``del_score`` as reference graph_score_online.py:7-90, so
``plnn/relu_conv_online.py`` (call sites :109, :117, :208, :230, :239) can use it unchanged.  ``decision`` is the scorer's
fused forward; ``online_learning`` re-runs the forward of the LAST decision in training form on the device, walks it
backwards and applies one Adam step (libgnnb.so ``gnnb_online_step``, csrc/gnnb_train.h) -- the reference keeps the
autograd graph of the decision alive instead (``self.scores``), which ``del_score`` then frees.
"""
import sys
import time

import numpy as np
import torch

from .graph_conv import GraphNet
from .graph_score import _load_state


class GraphChoice:

    def __init__(self, init_mask, model_name, lr=1e-4, wd=1e-4, linear=False):
        model = GraphNet(2, 64)                                   # graph_score_online.py:10
        model.load_state_dict(_load_state(model_name))            # :12
        model.eval()
        self.model = model
        self.lr, self.wd = lr, wd                                 # torch.optim.Adam(..., lr=lr, weight_decay=wd)  :15
        self._engine = None
        trans_len, temp = [], 0
        for i in init_mask:                                       # :16-21
            temp += len(i)
            trans_len.append(temp)
        self.trans_len = torch.tensor(trans_len)
        self.verbose = True
        self._last = None

    def _eng(self):
        eng = self.model.engine()
        if eng is not self._engine:                               # first use (or parameters replaced from outside): new optimizer
            eng.online_create(self.lr, self.wd)
            self._engine = eng
        return eng

    def decision(self, lower_bounds_all, upper_bounds_all, dual_vars, primal_input, primals, layers, mask):
        """[dec_lay, dec_idx] of the highest-scoring undecided ReLU (graph_score_online.py:23-59)."""
        mask_1d = torch.cat([(i == -1).float().reshape(-1) for i in mask], 0).unsqueeze(0)       # :24-26
        self.mask_1d = mask_1d
        start = time.time()
        args = (lower_bounds_all, upper_bounds_all, dual_vars, primals, primal_input, layers, mask_1d)
        host = not any(torch.is_tensor(t) and t.is_cuda for grp in (lower_bounds_all, upper_bounds_all, dual_vars, primals) for t in grp) \
            and not (torch.is_tensor(primal_input) and primal_input.is_cuda)
        with torch.no_grad():
            if host:                                              # CPU tensors (the reference's pattern): one pinned transfer
                d, sc = self._eng().forward_host(*args, want_scores=True)
                dec = d[0].tolist()
                ragged = [torch.from_numpy(sc[0][mask_1d[0].cpu().numpy() != 0])]
            else:
                res = self._eng().forward(*args).check()
                dec = res.decisions[0].tolist()
                ragged = res.ragged()
        end = time.time()
        if self.verbose:
            print(f'graph requires: {end-start}')                 # :36
        if dec[0] < 0:
            print("[gnn_branching_amd] GraphChoice.decision: no undecided ReLU in the mask", file=sys.stderr)
            raise RuntimeError("GraphChoice.decision: no undecided ReLU in the mask")
        self._last = args
        self.scores = ragged                                      # :34 (values only: the tape is rebuilt by online_learning)
        self.gnn_score = self.scores[0].max() if self.scores[0].numel() else None
        return [int(dec[0]), int(dec[1])]

    def online_learning(self, kw_decision, improvement):
        """One step on loss = gnn_score - kw_score + improvement (graph_score_online.py:62-77)."""
        if self._last is None:
            raise RuntimeError("online_learning: no decision to learn from (call decision first)")
        partial_len = 0 if kw_decision[0] == 0 else int(self.trans_len[kw_decision[0] - 1])          # :63-66
        if self.verbose:
            print('updating the trained model')                   # :67
        flat = partial_len + int(kw_decision[1])                  # :68
        eng = self._eng()
        loss, _ = eng.online_step(self._last, [flat], [float(improvement)])
        self.model.load_blob(eng.get_weights())                   # the nn.Module mirrors the device parameters (state_dict(), save)
        self.last_loss = float(loss[0])

    def del_score(self):                                          # :79-83
        for name in ("scores", "gnn_score", "mask_1d"):
            if hasattr(self, name):
                delattr(self, name)
        self._last = None
