"""MI355X build of the BaBSR ("KW") branching heuristic -- the fallback scorer of the BaB loop.

Mirror of reference plnn/kw_score_conv.py: ``choose_node_conv`` (:41-160) keeps its signature, argument meaning and
return value; the score computation (:56-113, the backward sweep of the scalar ratio through the verified network)
runs in libgnnb.so (``gnnb_babsr``, kernel k_babsr), the decision rule (:115-156) with its counter and random
fall-back stays on the host, as in the reference.  ``BabsrScorer`` is the batched form used by
``bab_caller.BatchedGraphChoice`` style callers (one launch for B subproblems).
"""
import torch
from torch import nn

from ..engine import ScorerEngine


class BabsrScorer:
    """Scores B subproblems in one launch.  ``engine``: an existing ScorerEngine (e.g. ``GraphNet.engine()``) to
    share its bound network; by default a GNN-free engine is created on first use."""

    def __init__(self, engine=None):
        self._engine = engine

    def engine(self):
        if self._engine is None:
            self._engine = ScorerEngine(None)
        return self._engine

    def scores(self, lower_bounds_all, upper_bounds_all, layers, bab_masks):
        """lower/upper_bounds_all: the per-graph-layer (B, ...) bounds handed to GraphNet.forward; layers:
        {'fixed_layers', 'prop_layers'}; bab_masks: per ReLU layer (B, N_k) BaB masks (-1 = undecided), or the
        (B, R) 0/1 matrix GraphNet.forward takes.  Returns engine.BabsrResult (device tensors)."""
        if torch.is_tensor(bab_masks):
            mask = bab_masks
        else:
            mask = torch.cat([(torch.as_tensor(m) == -1).float().reshape(m.shape[0], -1) for m in bab_masks], 1)
        return self.engine().babsr(lower_bounds_all, upper_bounds_all, layers, mask)

    def decide_many(self, res, icp_score_counters, random_order, sparsest_layer, decision_threshold=0.001):
        """The host decision rule for every subproblem of a BabsrResult: ONE device->host copy of the three (B, R)
        matrices, then ``decide`` per row.  Returns ([[lay, idx]] * B, [icp_score_counter] * B)."""
        host = torch.stack([res.scores, res.intercepts, res.masks]).cpu()
        decisions, counters = [], []
        for b in range(host.shape[1]):
            score, icp, mask = (list(torch.split(host[i, b], res.relu_sizes)) for i in range(3))
            d, c = decide(score, icp, mask, icp_score_counters[b], random_order, sparsest_layer, decision_threshold)
            decisions.append(d)
            counters.append(c)
        return decisions, counters


def decide(score, intercept_tb, mask, icp_score_counter, random_order, sparsest_layer, decision_threshold=0.001):
    """Decision rule of kw_score_conv.py:115-156 for ONE subproblem from per-layer 1-D tensors (any device)."""
    random_choice = random_order.copy()
    max_info = []
    for s in score:                                   # torch.max(i, 0): first maximum
        v, i = torch.max(s, 0)
        max_info.append((v.item(), i.item()))
    decision_layer = max_info.index(max(max_info))
    decision_index = max_info[decision_layer][1]
    if decision_layer != sparsest_layer and max_info[decision_layer][0] > decision_threshold:
        return [decision_layer, decision_index], icp_score_counter
    min_info = []
    for i, t in enumerate(intercept_tb):
        v, j = torch.min(t, 0)
        if v.item() < -1e-4:
            min_info.append((i, j.item()))
    if len(min_info) != 0 and icp_score_counter < 2:
        intercept_layer, intercept_index = min_info[-1]
        icp_score_counter += 1
        if intercept_layer != 0:
            icp_score_counter = 0
        print('\tusing intercept score')
        return [intercept_layer, intercept_index], icp_score_counter
    print('\t using a random choice')
    while True:
        preferred_layer = random_choice.pop(-1)
        nz = mask[preferred_layer].nonzero()
        if len(nz) != 0:
            return [preferred_layer, nz[0].item()], 0


_default = BabsrScorer()


def choose_node_conv(lower_bounds, upper_bounds, orig_mask, layers, pre_relu_indices, icp_score_counter, random_order,
                     sparsest_layer, decision_threshold=0.001, gt=False):
    """Drop-in for reference kw_score_conv.py:41.  lower/upper_bounds: per-network-layer lists (no batch dimension),
    read at ``pre_relu_indices``; orig_mask: per ReLU layer BaB mask; layers: net.layers with the folded property
    layer last.  Returns ``decision, icp_score_counter`` (and the per-layer score list if ``gt``)."""
    fixed, prop = list(layers[:-1]), layers[-1]
    if type(prop) is not nn.Linear or prop.weight.shape[0] != 1:
        raise NotImplementedError("the last layer must be the folded property layer Linear(., 1)")
    first = fixed[0]
    lb_all, ub_all = [], []
    # graph layer 0 (the input) is not read by the heuristic: a correctly sized placeholder keeps the shared binding
    if type(first) is nn.Conv2d:
        y = lower_bounds[pre_relu_indices[0]]
        h_in = (y.shape[-2] - 1) * first.stride[0] - 2 * first.padding[0] + first.kernel_size[0]
        w_in = (y.shape[-1] - 1) * first.stride[1] - 2 * first.padding[1] + first.kernel_size[1]
        x0 = lower_bounds[0] if lower_bounds[0] is not None else torch.zeros(first.in_channels, h_in, w_in)
    else:
        x0 = lower_bounds[0] if lower_bounds[0] is not None else torch.zeros(first.in_features)
    lb_all.append(torch.as_tensor(x0).unsqueeze(0))
    ub_all.append(torch.as_tensor(x0).unsqueeze(0))
    for i in pre_relu_indices:
        lb_all.append(lower_bounds[i].unsqueeze(0))
        ub_all.append(upper_bounds[i].unsqueeze(0))
    lb_all.append(torch.zeros(1, 1))
    ub_all.append(torch.zeros(1, 1))
    res = _default.scores(lb_all, ub_all, {"fixed_layers": fixed, "prop_layers": [prop]},
                          [torch.as_tensor(m).reshape(1, -1) for m in orig_mask])
    score, intercept_tb, mask = res.per_layer(0)
    decision, icp_score_counter = decide(score, intercept_tb, mask, icp_score_counter, random_order, sparsest_layer,
                                         decision_threshold)
    if gt is False:
        return decision, icp_score_counter
    return decision, icp_score_counter, score
