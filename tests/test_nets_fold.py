"""CPU: SURVEY 8 row A0 -- the verified network's layer graph and the property fold.

Reference: add_single_prop (exp_utils/model_utils.py:187-208) appends Linear(10, 1) with +1 at `gt`, -1 at `cls`, zero bias;
simplify_network (plnn/model.py:597-622) merges it with the preceding Linear (W = W2 @ W1, b = b2 + W2 @ b1);
relu_gnn derives `bounds_indices = [0] + pre_relu_indices + [len(net.layers)]` (plnn/relu_conv_gnnkwthreshold.py:110).
The fixtures in tests/golden were made by feeding nets.fold_property's output to the reference AND to this build, so a
wrong fold would be invisible there: here the fold is checked against the UNFOLDED evaluation.
"""
import numpy as np
import pytest
import torch
from torch import nn

from gnn_branching_amd import nets

# SURVEY.md section 8, table of nets: (N per graph layer, pre_relu_indices, R)
EXPECTED = {
    "cifar_base_kw": ([3072, 2048, 1024, 100, 1], [1, 3, 6], 3172),
    "cifar_wide_kw": ([3072, 4096, 2048, 100, 1], [1, 3, 6], 6244),
    "cifar_deep_kw": ([3072, 2048, 2048, 2048, 512, 100, 1], [1, 3, 5, 7, 10], 6756),
}


def _run(layers, x):
    for l in layers:
        x = l(x)
    return x


@pytest.mark.parametrize("net", nets.NET_NAMES)
@pytest.mark.parametrize("gt,cls", [(3, 5), (0, 9)])
def test_fold_equals_unfolded_network_then_property_row(net, gt, cls):
    base = nets.build_net(net)                      # ... Linear(100, 10)
    folded = nets.fold_property(base, gt, cls)      # ... Linear(100, 1)
    assert isinstance(base[-1], nn.Linear) and base[-1].out_features == 10
    assert isinstance(folded[-1], nn.Linear) and folded[-1].out_features == 1 and len(folded) == len(base)
    assert all(a is b for a, b in zip(base[:-1], folded[:-1]))        # the fixed layers are shared, not copied
    rng = np.random.RandomState(11)
    x = torch.from_numpy(rng.standard_normal((5,) + nets.INPUT_SHAPE).astype(np.float32))
    with torch.no_grad():
        logits = _run(base, x)                      # (5, 10): the network as trained
        want = logits[:, gt] - logits[:, cls]       # the +1 / -1 property row applied to its output, zero bias
        got = _run(folded, x)[:, 0]
        # in fp64 the fold is an identity up to fp32 rounding of the folded weights
        w64 = base[-1].weight.double()
        c = torch.zeros(1, 10, dtype=torch.float64)
        c[0, gt], c[0, cls] = 1.0, -1.0
        assert torch.allclose(folded[-1].weight.double(), c @ w64, atol=1e-7)
        assert torch.allclose(folded[-1].bias.double(), c @ base[-1].bias.double(), atol=1e-7)
    scale = float(logits.abs().max())
    assert float((got - want).abs().max()) <= 1e-5 * max(scale, 1.0)
    assert nets.load_verified_net(net, gt, cls)[-1].weight.shape == (1, 100)


@pytest.mark.parametrize("net", nets.NET_NAMES)
def test_graph_layout_reproduces_the_survey_table(net):
    sizes, pre, R = EXPECTED[net]
    layers = nets.load_verified_net(net)
    shapes, pre_relu = nets.graph_layout(layers)
    assert [int(np.prod(s)) for s in shapes] == sizes
    assert pre_relu == pre
    assert sum(sizes[1:-1]) == R
    # bounds_indices = [0] + pre_relu_indices + [len(net.layers)] selects exactly one bounds tensor per graph layer
    assert len([0] + pre_relu + [len(layers)]) == len(sizes)
    # fixed_layers = net.layers[:-1] must end in a ReLU; the property layer is the folded Linear(., 1)
    assert isinstance(layers[-2], nn.ReLU) and layers[-1].out_features == 1


def test_synthetic_bounds_follow_the_folded_network():
    """synth.make_batch propagates its interval bounds through the fixed layers AND the folded property layer: the last
    graph layer's bounds must contain the folded network's output at the LP point it reports as primals[-1]."""
    from gnn_branching_amd import synth
    b = synth.make_batch("cifar_base_kw", 3, seed=5)
    out = b.primals[-1].reshape(3)
    lo, hi = b.lower_bounds_all[-1].reshape(3), b.upper_bounds_all[-1].reshape(3)
    assert bool(((lo - 1e-5 <= out) & (out <= hi + 1e-5)).all())
    with torch.no_grad():
        y = _run(list(b.layers["fixed_layers"]) + [b.layers["prop_layers"][0]], b.primal_inputs)[:, 0]
    assert torch.allclose(y, out, atol=1e-5)
