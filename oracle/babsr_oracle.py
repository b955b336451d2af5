"""CPU ORACLE for the BaBSR ("KW") branching heuristic.  TEST INFRASTRUCTURE ONLY (same rules as gnn_oracle.py).

Batch-vectorised torch-CPU restatement of reference plnn/kw_score_conv.py: ``compute_ratio`` :23-37 and the score
computation of ``choose_node_conv`` :41-113 (the backward sweep of the scalar ``ratio`` through the verified
network, the bias and intercept candidates per ReLU), plus the decision rule :115-156 (``decide``).
Pinned by tests/golden/*_babsr.npz, produced by oracle/make_golden_babsr.py from the imported reference.
"""
import torch
from torch import nn
from torch.nn import functional as F


def compute_ratio(lb, ub):
    """kw_score_conv.py:23-37 -> (slope_ratio, intercept)"""
    lower_temp = lb - F.relu(lb)
    upper_temp = F.relu(ub)
    slope = upper_temp / (upper_temp - lower_temp)
    return slope, -1 * lower_temp * slope


def babsr_scores(lbs, ubs, masks, fixed_layers, prop_w):
    """Scores and intercept terms of every ReLU for a batch.

    lbs/ubs: pre-activation bounds of the ReLU layers, each (B, *shape); masks: per ReLU layer (B, N) with 1 where the
    BaB mask is -1; fixed_layers: net.layers[:-1]; prop_w: (B, N_L) weights of the folded property layer.
    Returns (score list, intercept list), each entry (B, N_k) -- `score` / `intercept_tb` of :103 / :89."""
    B = lbs[0].shape[0]
    ratio = prop_w.clone()                                   # Linear(., 1): W^T @ ones(1)   (:73-77)
    relu_positions = [i for i, l in enumerate(fixed_layers) if isinstance(l, nn.ReLU)]
    score, icp = [None] * len(relu_positions), [None] * len(relu_positions)
    k = len(relu_positions) - 1
    for idx in range(len(fixed_layers) - 1, -1, -1):
        layer = fixed_layers[idx]
        if isinstance(layer, nn.Linear):
            ratio = ratio.reshape(B, -1) @ layer.weight                                     # :74-77
        elif isinstance(layer, nn.ReLU):
            lb, ub = lbs[k], ubs[k]
            ratio = ratio.reshape(lb.shape)
            r0, r1 = compute_ratio(lb, ub)                                                  # :82
            intercept_candidate = torch.clamp(ratio, max=0) * r1                            # :84-85
            icp[k] = intercept_candidate.reshape(B, -1) * masks[k]                          # :86
            b = fixed_layers[idx - 1].bias.detach()
            if isinstance(fixed_layers[idx - 1], nn.Conv2d):
                b = b.unsqueeze(-1).unsqueeze(-1)                                           # :90-91
            bias_1 = b * (ratio * (r0 - 1))                                                 # :92-93
            ratio = ratio * r0                                                              # :94
            bias_2 = b * ratio                                                              # :95
            score_candidate = torch.max(bias_1, bias_2) + intercept_candidate               # :96-101
            score[k] = score_candidate.abs().reshape(B, -1) * masks[k]                      # :103
            k -= 1
            if k < 0:
                break                                                                       # nothing reads ratio below the first ReLU
        elif isinstance(layer, nn.Conv2d):
            ratio = F.conv_transpose2d(ratio, layer.weight, stride=layer.stride, padding=layer.padding)   # :109-111
        elif type(layer).__name__ == "Flatten":
            pass                                                                            # reshape happens at the ReLU (:115)
        else:
            raise NotImplementedError(type(layer))
    return score, icp


def decide(score, icp, mask, icp_score_counter, random_order, sparsest_layer, decision_threshold=0.001):
    """Decision rule for ONE subproblem (kw_score_conv.py:117-152).  score/icp/mask: per-layer 1-D tensors.
    Returns (decision, icp_score_counter)."""
    random_choice = list(random_order)
    max_info = [torch.max(s, 0) for s in score]
    decision_layer = max_info.index(max(max_info))
    decision_index = max_info[decision_layer][1].item()
    if decision_layer != sparsest_layer and max_info[decision_layer][0].item() > decision_threshold:
        return [decision_layer, decision_index], icp_score_counter
    min_info = [[i, torch.min(icp[i], 0)] for i in range(len(icp)) if torch.min(icp[i]) < -1e-4]
    if len(min_info) != 0 and icp_score_counter < 2:
        layer = min_info[-1][0]
        index = min_info[-1][1][1].item()
        icp_score_counter += 1
        if layer != 0:
            icp_score_counter = 0
        return [layer, index], icp_score_counter
    while True:
        preferred = random_choice.pop(-1)
        nz = mask[preferred].nonzero()
        if len(nz) != 0:
            return [preferred, nz[0].item()], 0
