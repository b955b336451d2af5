"""Batched counterpart of the reference's BaB call pattern.

``relu_gnn`` (reference plnn/relu_conv_gnnkwthreshold.py) scores ONE subproblem per ``graph.decision`` call: the
root at :117 and the two children of every branch at :230 and :239, each time re-marshalling ~14 tensors and 8 Python
lists (``torch.tensor(list)``, graph_score.py:30).  Here the subproblems that are live at the same time -- the two
children, or a whole frontier -- are collated once and scored in ONE batched forward on the MI355X:

    g = BatchedGraphChoice(init_mask, model_path)
    decisions = g.decision_many([child0, child1], layers)        # [[lay, idx], [lay, idx]]

A subproblem is described exactly as the reference passes it to ``GraphChoice.decision`` (graph_score.py:21): bounds
with the leading ``unsqueeze(0)`` the caller adds (relu_conv_gnnkwthreshold.py:115-116, :228-229), duals, the LP
point, the per-layer primal lists and the {-1, 0, 1} BaB mask.  ``trace_line`` reproduces the trace format of :202 so
decision traces of a run can be diffed against the reference's ``gnn_dump_files``.
"""
from dataclasses import dataclass
from typing import List, Sequence

import numpy as np
import torch

from .graphnet.graph_score import GraphChoice
from .plnn.kw_score_conv import BabsrScorer


@dataclass
class Subproblem:
    """The per-domain arguments of ``GraphChoice.decision`` (graph_score.py:21), plus its property layer."""
    lower_bounds_all: Sequence[torch.Tensor]     # graph layers, each (1, *shape)
    upper_bounds_all: Sequence[torch.Tensor]
    dual_vars: Sequence[torch.Tensor]            # per ReLU layer (N, 3)
    primal_input: torch.Tensor                   # (1, C, H, W)
    primals: Sequence                            # per network layer: list / array / tensor of LP primal values
    mask: Sequence[torch.Tensor]                 # per ReLU layer, values in {-1, 0, 1}
    prop_layer: torch.nn.Module = None           # Linear(., 1); None -> layers['prop_layers'][0]


def _flat32(v):
    if torch.is_tensor(v):
        return v.detach().to(torch.float32).reshape(-1).cpu().numpy()
    return np.asarray(v, dtype=np.float32).reshape(-1)


def collate(subs: List[Subproblem], layers):
    """Stack B subproblems into the argument tuple of ``GraphNet.forward`` (graph_conv.py:479).

    One numpy concatenation per tensor group (then one H2D copy each inside the engine) replaces the B x (14 tensors +
    8 lists) conversions of the per-call path.  Returns (args tuple, masks_1d)."""
    B = len(subs)
    if B == 0:
        raise ValueError("no subproblem to score")
    ng = len(subs[0].lower_bounds_all)
    lbs = [torch.from_numpy(np.concatenate([_flat32(s.lower_bounds_all[k]) for s in subs]))
           .reshape((B,) + tuple(subs[0].lower_bounds_all[k].shape[1:])) for k in range(ng)]
    ubs = [torch.from_numpy(np.concatenate([_flat32(s.upper_bounds_all[k]) for s in subs]))
           .reshape((B,) + tuple(subs[0].upper_bounds_all[k].shape[1:])) for k in range(ng)]
    duals = [torch.from_numpy(np.concatenate([_flat32(s.dual_vars[j]) for s in subs])).reshape(-1, 3)
             for j in range(len(subs[0].dual_vars))]
    primals = [torch.from_numpy(np.concatenate([_flat32(s.primals[m]) for s in subs])) for m in range(len(subs[0].primals))]
    x_lp = torch.from_numpy(np.concatenate([_flat32(s.primal_input) for s in subs])).reshape((B,) + tuple(subs[0].primal_input.shape[1:]))
    masks = torch.stack([torch.cat([(m == -1).float().reshape(-1) for m in s.mask]) for s in subs])      # graph_score.py:22-24
    props = [s.prop_layer if s.prop_layer is not None else layers["prop_layers"][0] for s in subs]
    blayers = {"fixed_layers": layers["fixed_layers"], "prop_layers": props}
    return (lbs, ubs, duals, primals, x_lp, blayers, masks), masks


class BatchedGraphChoice(GraphChoice):
    """``GraphChoice`` plus batched entry points; the single-subproblem ``decision`` surface is inherited unchanged."""

    def decision_many(self, subs: List[Subproblem], layers):
        """[dec_lay, dec_idx] for every subproblem, one batched forward, one device->host copy."""
        args, _ = collate(subs, layers)
        with torch.no_grad():
            if len(subs) <= 16:          # a branch's children / a small frontier: host pointers in, decisions out (gnnb_forward_host)
                dec = self.model.engine().forward_host(*args)[0].tolist()
            else:                        # a large frontier: per-tensor copies overlap better than one staged 36 MB block
                res = self.model.forward_device(*args).check()
                dec = res.decisions.cpu().tolist()
        for b, d in enumerate(dec):
            if d[0] < 0:
                raise RuntimeError(f"decision_many: subproblem {b} has no undecided ReLU in its mask")
        return [[int(d[0]), int(d[1])] for d in dec]

    def children_decisions(self, child0: Subproblem, child1: Subproblem, layers):
        """The two ``graph.decision`` calls of one branch (relu_conv_gnnkwthreshold.py:230, :239) as one B=2 call."""
        return self.decision_many([child0, child1], layers)


    def kw_decision_many(self, subs: List[Subproblem], layers, icp_scores, random_order, sparsest_layer,
                         decision_threshold=0.001):
        """The BaBSR fall-back decisions (``choose_node_conv`` call of relu_conv_gnnkwthreshold.py:157) of B subproblems
        in one launch on the engine the GNN already bound.  Only bounds and masks of a Subproblem are read.
        Returns ([[lay, idx]] * B, [icp_score] * B)."""
        B = len(subs)
        ng = len(subs[0].lower_bounds_all)
        lbs = [torch.cat([torch.as_tensor(s.lower_bounds_all[k]).float().reshape((1,) + tuple(subs[0].lower_bounds_all[k].shape[1:]))
                          for s in subs]) for k in range(ng)]
        ubs = [torch.cat([torch.as_tensor(s.upper_bounds_all[k]).float().reshape((1,) + tuple(subs[0].upper_bounds_all[k].shape[1:]))
                          for s in subs]) for k in range(ng)]
        masks = torch.stack([torch.cat([(m == -1).float().reshape(-1) for m in s.mask]) for s in subs])
        props = [s.prop_layer if s.prop_layer is not None else layers["prop_layers"][0] for s in subs]
        if getattr(self, "_babsr", None) is None:
            self._babsr = BabsrScorer(self.model.engine())
        with torch.no_grad():
            res = self._babsr.scores(lbs, ubs, {"fixed_layers": layers["fixed_layers"], "prop_layers": props}, masks)
        return self._babsr.decide_many(res, list(icp_scores), random_order, sparsest_layer, decision_threshold)


def resolve_branching(gnn_decision, gnn_improvement_value, kw_decision, kw_improvement_value, ineff_kw_dc):
    """Which decision a branch keeps once both pairs of children are bounded (relu_conv_gnnkwthreshold.py:176-192):
    a KW point that improves less than the GNN's AND less than 0.05 is counted as inefficient in ``ineff_kw_dc``
    (mutated, key 'lay-idx'); a KW point that improves more replaces the GNN decision.  Returns (decision, used_kw)."""
    if kw_improvement_value < gnn_improvement_value and kw_improvement_value < 0.05:
        key = f'{kw_decision[0]}-{kw_decision[1]}'
        ineff_kw_dc[key] = ineff_kw_dc.get(key, 0) + 1
        return gnn_decision, False
    if kw_improvement_value > gnn_improvement_value:
        return kw_decision, True
    return gnn_decision, False


def resolve_online(gnn_decision, gnn_improvement_value, kw_decision, kw_improvement_value, wrong_pts_dc, online_threshold=5):
    """The online-learning variant of the same choice (plnn/relu_conv_online.py:183-207): the KW decision is kept when it
    improves MORE than the GNN's; the GNN's decision is then counted as a wrong point in ``wrong_pts_dc`` (mutated, key
    'lay-idx'), and once that count reaches ``online_threshold`` the caller runs
    ``graph.online_learning(kw_decision, improve)`` with improve = 1 if the KW point is better by more than 0.1, else 0.
    Returns (decision, used_kw, learn, improve); kw_decision None / kw_improvement -1 when the KW branch was not bounded."""
    if kw_decision is None or not gnn_improvement_value < kw_improvement_value:
        return gnn_decision, False, False, 0
    key = f'{gnn_decision[0]}-{gnn_decision[1]}'
    wrong_pts_dc[key] = wrong_pts_dc.get(key, 0) + 1
    learn = wrong_pts_dc[key] >= online_threshold
    improve = 1 if kw_improvement_value - gnn_improvement_value > 0.1 else 0
    return kw_decision, True, learn, improve


def gnn_improvement(dom_lb, dom_lb1, lower_bound):
    """relu_conv_gnnkwthreshold.py:151."""
    return (min(dom_lb, 0) + min(dom_lb1, 0) - 2 * lower_bound) / (-2 * lower_bound)


def trace_line(nb_visited_states, branching_decision, gnn_improvement_value, gnn_decision, kw_improvement=-1, kw_decision=None):
    """The per-branch trace line of relu_conv_gnnkwthreshold.py:202 (written to ./gnn_dump_files/<trace_name>)."""
    return (f'branch {nb_visited_states} decision {branching_decision} gnn: improvement {gnn_improvement_value} '
            f'decision {gnn_decision} kw: improvement {kw_improvement} decision {kw_decision}\n')
