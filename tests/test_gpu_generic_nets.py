"""Networks other than the three CIFAR models: the reference dispatches on layer types (graph_conv.py:110-192), so the
scorer has to work for any conv / linear stack.  These take the engine's other kernels -- Linear first layer (dense
edges everywhere, k_embed + k_input_update instead of the fused input kernels), 3x3 stride-1 convolutions, a last ReLU
layer too wide for k_top -- and are checked against the oracle (itself pinned to the reference by tests/golden)."""
import numpy as np
import pytest
import torch

from gnn_branching_amd import nets, synth
from gnn_branching_amd.graphnet.graph_conv import GraphNet
from tests.common import SCORE_ATOL, score_tol

pytestmark = pytest.mark.gpu

ARCHS = {
    # all-dense network: Flatten first, Linear edges only, top layer through k_top
    "toy_mlp": [("flatten",), ("linear", 3 * 32 * 32, 128), ("relu",), ("linear", 128, 64), ("relu",), ("linear", 64, 10)],
    # 3x3 stride-1 first conv (8192-node layer), stride-2 second conv, narrow Linear head
    "toy_conv3": [("conv", 3, 8, 3, 1, 1), ("relu",), ("conv", 8, 8, 4, 2, 1), ("relu",), ("flatten",), ("linear", 8 * 16 * 16, 48),
                  ("relu",), ("linear", 48, 10)],
    # last ReLU layer with 200 nodes: too wide for k_top, separate dense / update / property kernels
    "toy_widehead": [("conv", 3, 8, 4, 2, 1), ("relu",), ("flatten",), ("linear", 8 * 16 * 16, 200), ("relu",), ("linear", 200, 10)],
    # channel counts the VALU fallback kernels are not compiled for (12, 6): MFMA gather tables only
    "toy_oddch": [("conv", 3, 12, 3, 1, 1), ("relu",), ("conv", 12, 6, 4, 2, 1), ("relu",), ("flatten",), ("linear", 6 * 16 * 16, 32),
                  ("relu",), ("linear", 32, 10)],
    # kernel sizes / strides without a compile-time stencil in the bias-sum pass: 5x5 stride 1 pad 2, 2x2 stride 2 pad 0
    "toy_k5": [("conv", 3, 8, 5, 1, 2), ("relu",), ("conv", 8, 8, 2, 2, 0), ("relu",), ("flatten",), ("linear", 8 * 16 * 16, 40),
               ("relu",), ("linear", 40, 10)],
    # a 32768-node layer under the Linear head: too long for k_top's live-row list (LDS), so its forward edge walks every row
    "toy_longk": [("conv", 3, 32, 3, 1, 1), ("relu",), ("flatten",), ("linear", 32 * 32 * 32, 72), ("relu",), ("linear", 72, 10)],
    # a single ReLU layer (L = 1)
    "toy_single": [("conv", 3, 8, 4, 2, 1), ("relu",), ("flatten",), ("linear", 8 * 16 * 16, 10)],
}


@pytest.fixture(scope="module", autouse=True)
def _register():
    for i, (name, spec) in enumerate(ARCHS.items()):
        nets.register_arch(name, spec, seed=100 + i)


@pytest.mark.parametrize("name", list(ARCHS))
def test_scores_match_oracle(name):
    from oracle import gnn_oracle
    batch = synth.make_batch(name, 3, seed=5, props=[(3, 5), (1, 7), (0, 2)])
    state = gnn_oracle.random_gnn_state(20240917)
    model = GraphNet(2, 64)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in state.items()})
    with torch.no_grad():
        ragged = gnn_oracle.oracle_forward(state, *batch.forward_args())
        res = model.forward_device(*batch.forward_args()).check()
    want = gnn_oracle.padded_scores(ragged, batch.masks).numpy()
    got = res.scores.cpu().numpy()
    assert np.array_equal(np.isinf(got), np.isinf(want))
    fin = np.isfinite(want)
    assert fin.any()
    err = np.abs(got[fin] - want[fin]).max()
    tol = score_tol("random", want[fin])
    print(f"{name}: max|score - oracle| = {err:.3e} over {int(fin.sum())} scores (bar {tol:.1e})")
    assert err <= tol
    # nothing may depend on scratch the call did not write itself (rows of dead nodes are only written where a consumer
    # reads them): poison the workspace and run again
    model.engine().workspace(batch.batch_size).view(torch.float32).fill_(float("nan"))
    with torch.no_grad():
        again = model.forward_device(*batch.forward_args()).check()
    assert torch.equal(again.scores, res.scores)
    shapes, _ = nets.graph_layout(batch.layers["fixed_layers"] + [batch.layers["prop_layers"][0]])
    relu_sizes = [int(np.prod(sh)) for sh in shapes[1:-1]]
    dec = [gnn_oracle.decision_from_scores(ragged[b], batch.masks[b], relu_sizes) for b in range(batch.batch_size)]
    assert res.decisions.cpu().tolist() == dec


@pytest.mark.parametrize("T", [1, 3])
def test_other_round_counts_match_oracle(T):
    """GraphNet(T, 64) with T != 2 (the reference hard-codes T = 2 in graph_score.py:9 but GraphNet takes it as an argument):
    T = 1 has no live input-layer update at all, T = 3 feeds the second one into a third forward sweep."""
    from oracle import gnn_oracle
    batch = synth.make_batch("cifar_base_kw", 2, seed=9)
    state = gnn_oracle.random_gnn_state(4242)
    model = GraphNet(T, 64)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in state.items()})
    with torch.no_grad():
        ragged = gnn_oracle.oracle_forward(state, *batch.forward_args(), T=T)
        res = model.forward_device(*batch.forward_args()).check()
    want = gnn_oracle.padded_scores(ragged, batch.masks).numpy()
    got = res.scores.cpu().numpy()
    fin = np.isfinite(want)
    assert np.array_equal(np.isinf(got), ~fin)
    assert np.abs(got[fin] - want[fin]).max() <= score_tol("random", want[fin])


@pytest.mark.parametrize("B", [1, 5, 67])
def test_odd_batch_sizes_match_per_sample_runs(B):
    """batch sizes that are not multiples of anything: every sample's row equals its own B = 1 run bit for bit"""
    from gnn_branching_amd.graphnet.graph_conv import GraphNet as G
    from oracle import gnn_oracle
    state = gnn_oracle.random_gnn_state(77)
    model = G(2, 64)
    model.load_state_dict({k: torch.as_tensor(v) for k, v in state.items()})
    batch = synth.make_batch("cifar_deep_kw", B, seed=31)
    with torch.no_grad():
        full = model.forward_device(*batch.forward_args()).check().scores.cpu()
        for b in sorted({0, B // 2, B - 1}):
            one = model.forward_device(*batch.slice(b, b + 1).forward_args()).check().scores.cpu()
            assert torch.equal(one[0], full[b]), b
