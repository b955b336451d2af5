"""the BaBSR fallback scorer (reference plnn/kw_score_conv.py choose_node_conv).
CPU: the oracle against the reference's golden vectors.  GPU: the HIP kernel against the same vectors."""
import os

import numpy as np
import pytest
import torch

from tests.common import GOLDEN, GOLDEN_CASES, load_golden


def babsr_golden(case):
    return dict(np.load(os.path.join(GOLDEN, case + "_babsr.npz")))


def relu_inputs(batch):
    L = len(batch.lower_bounds_all) - 2
    lbs = [batch.lower_bounds_all[k] for k in range(1, L + 1)]
    ubs = [batch.upper_bounds_all[k] for k in range(1, L + 1)]
    masks = [(m == -1).float() for m in batch.bab_masks]
    prop_w = torch.stack([p.weight[0] for p in batch.layers["prop_layers"]])
    return lbs, ubs, masks, prop_w


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_oracle_matches_reference(case):
    from oracle import babsr_oracle
    g, batch = load_golden(case)
    gb = babsr_golden(case)
    lbs, ubs, masks, prop_w = relu_inputs(batch)
    with torch.no_grad():
        score, icp = babsr_oracle.babsr_scores(lbs, ubs, masks, batch.layers["fixed_layers"], prop_w)
    L = len(score)
    for b in range(batch.batch_size):
        got = torch.cat([s[b] for s in score]).numpy()
        np.testing.assert_allclose(got, gb[f"score_{b}"], rtol=2e-5, atol=1e-6)
        for si, (sp, cnt, thr) in enumerate(gb["settings"]):
            dec, c = babsr_oracle.decide([s[b] for s in score], [i[b] for i in icp], [m[b] for m in masks],
                                         int(cnt), list(range(L)), int(sp), float(thr))
            assert dec + [c] == gb[f"dec_{b}_{si}"].tolist(), (b, si)


@pytest.mark.gpu
@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_hip_babsr_matches_reference(case):
    from gnn_branching_amd.plnn import kw_score_conv as kw
    g, batch = load_golden(case)
    gb = babsr_golden(case)
    L = len(batch.bab_masks)
    scorer = kw.BabsrScorer()
    res = scorer.scores(batch.lower_bounds_all, batch.upper_bounds_all, batch.layers, batch.bab_masks)
    R = batch.masks.shape[1]
    assert res.scores.shape == (batch.batch_size, R)
    for b in range(batch.batch_size):
        want = gb[f"score_{b}"]
        np.testing.assert_allclose(res.scores[b].cpu().numpy(), want, rtol=1e-4, atol=1e-6)
    # the reference's own call surface (B = 1, per-layer bounds list indexed by pre_relu_indices)
    fixed = batch.layers["fixed_layers"]
    pre_relu = [i for i, l in enumerate(fixed) if isinstance(l, torch.nn.ReLU)]
    for b in range(batch.batch_size):
        one = batch.slice(b, b + 1)
        nlay = len(fixed) + 1
        lbs, ubs = [None] * (nlay + 1), [None] * (nlay + 1)
        for k, i in enumerate(pre_relu):
            lbs[i], ubs[i] = one.lower_bounds_all[k + 1][0], one.upper_bounds_all[k + 1][0]
        layers = list(fixed) + [one.layers["prop_layers"][0]]
        for si, (sp, cnt, thr) in enumerate(gb["settings"]):
            dec, c = kw.choose_node_conv(lbs, ubs, [m[0] for m in one.bab_masks], layers, pre_relu, int(cnt),
                                         list(range(L)), int(sp), decision_threshold=float(thr))
            assert dec + [c] == gb[f"dec_{b}_{si}"].tolist(), (b, si)
