"""Verified-network zoo for the branching scorer's layer graph.

The GNN runs over the layer graph of the network being verified.  The
reference builds those networks in exp_utils/model_utils.py (cifar_model_m2
:155-166 = "base", cifar_model :120-131 = "wide", cifar_model_deep :136-151 =
"deep"), loads ``['state_dict'][0]`` of models/cifar_*_kw.pth (:214-225) and
folds the 1-vs-1 property into the last linear layer (add_single_prop
:187-208, simplify_network plnn/model.py:597-622).  That module cannot be
imported without gurobipy/torchvision, so the three architectures and the
fold are restated here from the text of those lines.

Only what the scorer needs: ``load_verified_net(name, gt, cls)`` returns the
list the BaB driver passes as ``layers`` (reference
plnn/relu_conv_gnnkwthreshold.py:110-113): ``net.layers[:-1]`` are the fixed
layers, ``net.layers[-1]`` is the folded Linear(.,1) property layer.
"""
import os

import numpy as np
import torch
from torch import nn

from .plnn.modules import Flatten

ASSETS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets")

# (kind, args...) per nn.Sequential slot; index in the list == state-dict index
_ARCH = {
    "cifar_base_kw": [("conv", 3, 8, 4, 2, 1), ("relu",), ("conv", 8, 16, 4, 2, 1), ("relu",),
                      ("flatten",), ("linear", 16 * 8 * 8, 100), ("relu",), ("linear", 100, 10)],
    "cifar_wide_kw": [("conv", 3, 16, 4, 2, 1), ("relu",), ("conv", 16, 32, 4, 2, 1), ("relu",),
                      ("flatten",), ("linear", 32 * 8 * 8, 100), ("relu",), ("linear", 100, 10)],
    "cifar_deep_kw": [("conv", 3, 8, 4, 2, 1), ("relu",), ("conv", 8, 8, 3, 1, 1), ("relu",),
                      ("conv", 8, 8, 3, 1, 1), ("relu",), ("conv", 8, 8, 4, 2, 1), ("relu",),
                      ("flatten",), ("linear", 8 * 8 * 8, 100), ("relu",), ("linear", 100, 10)],
}
NET_NAMES = tuple(_ARCH)
INPUT_SHAPE = (3, 32, 32)
_EXTRA_WEIGHTS = {}


def register_arch(name, spec, seed=0):
    """Register another architecture (same spec tuples as above) with seeded random weights: the scorer is not tied to
    the three CIFAR networks (the reference dispatches on layer types, graph_conv.py:110-192), and the tests exercise
    networks that take the engine's other kernels (first layer Linear, 3x3 stride-1 convolutions, ...)."""
    rng = np.random.RandomState(seed)
    layers = [_make(s) for s in spec]
    weights = {}
    for i, l in enumerate(layers):
        if isinstance(l, (nn.Conv2d, nn.Linear)):
            fan_in = int(np.prod(l.weight.shape[1:]))
            weights[f"{i}.weight"] = (rng.standard_normal(tuple(l.weight.shape)) / np.sqrt(fan_in)).astype(np.float32)
            weights[f"{i}.bias"] = (0.1 * rng.standard_normal(tuple(l.bias.shape))).astype(np.float32)
    _ARCH[name] = list(spec)
    _EXTRA_WEIGHTS[name] = weights
    return name


def _make(spec):
    kind = spec[0]
    if kind == "conv":
        _, ci, co, k, s, p = spec
        return nn.Conv2d(ci, co, k, stride=s, padding=p)
    if kind == "linear":
        return nn.Linear(spec[1], spec[2])
    if kind == "relu":
        return nn.ReLU()
    if kind == "flatten":
        return Flatten()
    raise ValueError(kind)


def build_net(name, weights=None):
    """nn.Module list of the un-folded network with trained weights loaded."""
    if name not in _ARCH:
        raise NotImplementedError(name)
    if weights is None:
        weights = _EXTRA_WEIGHTS[name] if name in _EXTRA_WEIGHTS else np.load(os.path.join(ASSETS, name + ".npz"))
    layers = [_make(s) for s in _ARCH[name]]
    with torch.no_grad():
        for i, l in enumerate(layers):
            if isinstance(l, (nn.Conv2d, nn.Linear)):
                l.weight.copy_(torch.from_numpy(np.asarray(weights[f"{i}.weight"])))
                l.bias.copy_(torch.from_numpy(np.asarray(weights[f"{i}.bias"])))
    for l in layers:
        for q in l.parameters():
            q.requires_grad = False
    return layers


def fold_property(layers, gt, cls):
    """Fold ``logit[gt] - logit[cls]`` into the last Linear: W = c.W_last, b = c.b_last.

    Restates add_single_prop (model_utils.py:187-208): a Linear(10,1) with
    +1 at ``gt``, -1 at ``cls`` and zero bias is merged with the preceding
    Linear by simplify_network (plnn/model.py:597-622: W = W2 @ W1,
    b = b2 + W2 @ b1).
    """
    last = layers[-1]
    c = torch.zeros(1, last.out_features)
    c[0, cls] = -1
    c[0, gt] = 1
    prop = nn.Linear(last.in_features, 1)
    with torch.no_grad():
        prop.weight.copy_(c @ last.weight)
        prop.bias.copy_(c @ last.bias)
    for q in prop.parameters():
        q.requires_grad = False
    return list(layers[:-1]) + [prop]


def load_verified_net(name, gt=3, cls=5):
    """Folded layer list: ``[:-1]`` fixed layers, ``[-1]`` the Linear(.,1) property layer."""
    return fold_property(build_net(name), gt, cls)


def graph_layout(layers, input_shape=INPUT_SHAPE):
    """Shapes of the GNN's graph layers for a folded layer list.

    Returns (shapes, pre_relu_indices): ``shapes[k]`` is the per-sample tensor
    shape of graph layer k (input, each pre-ReLU activation, property output);
    ``pre_relu_indices`` are the indices into the per-layer bounds list used by
    the BaB driver (bounds_indices = [0] + pre_relu_indices + [len(layers)],
    reference plnn/relu_conv_gnnkwthreshold.py:110).
    """
    shapes = [tuple(input_shape)]
    pre = []
    cur = tuple(input_shape)
    for i, l in enumerate(layers):
        if isinstance(l, nn.Conv2d):
            c, h, w = cur
            ho = (h + 2 * l.padding[0] - l.kernel_size[0]) // l.stride[0] + 1
            wo = (w + 2 * l.padding[1] - l.kernel_size[1]) // l.stride[1] + 1
            cur = (l.out_channels, ho, wo)
        elif isinstance(l, nn.Linear):
            cur = (l.out_features,)
        elif isinstance(l, Flatten):
            cur = (int(np.prod(cur)),)
        elif isinstance(l, nn.ReLU):
            pre.append(i)  # bounds list index i == output of layer i-1 == pre-activation
            shapes.append(cur)
        else:
            raise NotImplementedError(type(l))
    shapes.append(cur)  # property output
    return shapes, pre
