"""The CPU oracle against the reference's own outputs (tests/golden, made by oracle/make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import gnn_oracle
from tests.common import FAMILIES, GOLDEN_CASES, STAGES, load_golden, relu_sizes, state_of


@pytest.mark.parametrize("case", GOLDEN_CASES)
@pytest.mark.parametrize("fam", FAMILIES)
def test_oracle_matches_reference(case, fam):
    g, batch = load_golden(case)
    stages = {}
    with torch.no_grad():
        scores = gnn_oracle.oracle_forward(state_of(fam), *batch.forward_args(), stages=stages)
    got = gnn_oracle.padded_scores(scores, batch.masks).numpy()
    want = g[f"{fam}_scores"]
    assert np.array_equal(np.isinf(got), np.isinf(want))
    fin = np.isfinite(want)
    # same aten ops in a different batching: the reference's own batched-vs-single noise is ~4e-6
    assert np.abs(got[fin] - want[fin]).max() <= 2e-5
    stride = int(g["sample_stride"])
    for st in STAGES:
        for k, m in enumerate(stages[st]):
            rows = g[f"{fam}_{st}_mu{k}_rows"]
            np.testing.assert_allclose(m[:, ::stride, :].numpy(), rows, atol=2e-5, rtol=1e-5)
            s, a = g[f"{fam}_{st}_mu{k}_sum"]
            assert abs(m.double().abs().sum().item() - a) <= 1e-5 * max(a, 1.0)
    sizes = relu_sizes(batch)
    for b in range(batch.batch_size):
        dec = gnn_oracle.decision_from_scores(scores[b], batch.masks[b], sizes)
        assert dec == g[f"{fam}_decisions"][b].tolist()


def test_oracle_fp64_is_close_to_fp32():
    """fp64 evaluation of the same math: bounds the fp32 noise floor the 1e-4 budget sits on."""
    g, batch = load_golden("cifar_base_kw_B3")
    with torch.no_grad():
        s64 = gnn_oracle.oracle_forward(state_of("shipped"), *batch.forward_args(), dtype=torch.float64)
    got = gnn_oracle.padded_scores(s64, batch.masks).numpy()
    fin = np.isfinite(g["shipped_scores"])
    assert np.abs(got[fin] - g["shipped_scores"][fin]).max() < 1e-4


def test_compute_ratio_cases():
    lb = torch.tensor([-1.0, 0.5, -2.0, 0.0])
    ub = torch.tensor([3.0, 2.0, -1.0, 1.0])
    r0, r1, beta, amb = gnn_oracle.compute_ratio(lb, ub)
    assert r0.tolist() == [0.75, 1.0, 0.0, 1.0]
    assert r1.tolist() == [0.25, 1.0, 0.0, 1.0]
    assert amb.tolist() == [1.0, 0.0, 0.0, 0.0]
    assert beta[0].item() == 0.75
