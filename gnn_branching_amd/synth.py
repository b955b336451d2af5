"""Synthetic branch-and-bound subproblems of the cifar_*_kw shapes.

The real inputs of the branching scorer come from a Gurobi LP per subproblem
(reference plnn/conv_kwinter_gen.py:179-555, out of scope); for parity tests
and the benchmark a self-contained generator produces tensors with the same
contract as ``GraphNet.forward`` (reference graphnet/graph_conv.py:479):

  lower/upper_bounds_all[k]  (B, *shape_k)   graph layers: input, pre-ReLU..., property
  dual_vars[j]               (B*N_j, 3)      per ReLU layer, batch-major
  primals[m]                 (B*n_m,)        one per entry of net.layers (LP-optimal activations)
  primal_inputs              (B, 3, 32, 32)
  layers                     {'fixed_layers': [...], 'prop_layers': [Linear]*B}
  masks                      (B, R)          1.0 where the BaB mask is -1 (undecided)

Recipe: image x ~ N(0,1); eps-ball bounds; deeper
bounds by interval arithmetic with the W+/W- split (as
conv_kwinter_gen.py:214-241 does for its first bounds); x_LP uniform in the
ball; primals = activations of x_LP; duals sparse uniform.
"""
from dataclasses import dataclass, field
from typing import List

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from .nets import INPUT_SHAPE, fold_property, build_net, graph_layout
from .plnn.modules import Flatten


@dataclass
class SubproblemBatch:
    lower_bounds_all: List[torch.Tensor]
    upper_bounds_all: List[torch.Tensor]
    dual_vars: List[torch.Tensor]
    primals: List[torch.Tensor]
    primal_inputs: torch.Tensor
    layers: dict
    masks: torch.Tensor                      # (B, R) float 0/1
    bab_masks: List[torch.Tensor] = field(default_factory=list)   # per ReLU layer (B, N) in {-1,0,1}

    @property
    def batch_size(self):
        return int(self.lower_bounds_all[0].shape[0])

    def forward_args(self):
        """Positional args of GraphNet.forward (note: primals BEFORE primal_inputs)."""
        return (self.lower_bounds_all, self.upper_bounds_all, self.dual_vars, self.primals,
                self.primal_inputs, self.layers, self.masks)

    def n_ambiguous(self):
        return self.masks.sum(1).to(torch.int64)

    def slice(self, lo, hi):
        """Subproblems [lo, hi) as their own batch (shards for data-parallel scoring)."""
        B = self.batch_size
        def rows(t):   # batch-major flat (B*n, ...) tensors
            n = t.shape[0] // B
            return t[lo * n:hi * n]
        return SubproblemBatch(
            [t[lo:hi] for t in self.lower_bounds_all], [t[lo:hi] for t in self.upper_bounds_all],
            [rows(t) for t in self.dual_vars], [rows(t) for t in self.primals],
            self.primal_inputs[lo:hi],
            {"fixed_layers": self.layers["fixed_layers"], "prop_layers": self.layers["prop_layers"][lo:hi]},
            self.masks[lo:hi], [m[lo:hi] for m in self.bab_masks])


def _interval(layer, lb, ub):
    if isinstance(layer, nn.Conv2d):
        wp, wn = layer.weight.clamp(min=0), layer.weight.clamp(max=0)
        kw = dict(stride=layer.stride, padding=layer.padding)
        return (F.conv2d(lb, wp, layer.bias, **kw) + F.conv2d(ub, wn, None, **kw),
                F.conv2d(ub, wp, layer.bias, **kw) + F.conv2d(lb, wn, None, **kw))
    if isinstance(layer, nn.Linear):
        wp, wn = layer.weight.clamp(min=0), layer.weight.clamp(max=0)
        return (F.linear(lb, wp, layer.bias) + F.linear(ub, wn), F.linear(ub, wp, layer.bias) + F.linear(lb, wn))
    if isinstance(layer, nn.ReLU):
        return lb.clamp(min=0), ub.clamp(min=0)
    if isinstance(layer, Flatten):
        return lb.flatten(1), ub.flatten(1)
    raise NotImplementedError(type(layer))


def make_batch(net_name, B, seed=0, eps=0.02, props=None, dual_density=0.3):
    """Seeded synthetic batch of B subproblems on the named verified network.

    ``props``: list of (gt, cls) per sample (or None -> (3, 5) for all): a
    batch may mix properties (reference graph_conv.py:196-197 indexes
    ``layers['prop_layers'][i]`` by batch element).
    """
    rng = np.random.RandomState(seed)
    base = build_net(net_name)
    if props is None:
        props = [(3, 5)] * B
    assert len(props) == B
    cache = {}
    prop_layers = []
    for pr in props:
        if pr not in cache:
            cache[pr] = fold_property(base, *pr)[-1]
        prop_layers.append(cache[pr])
    fixed = base[:-1]
    shapes, _ = graph_layout(fixed + [prop_layers[0]])

    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))

    x = t(rng.standard_normal((B,) + INPUT_SHAPE))
    x_lp = x + eps * t(rng.uniform(-1, 1, (B,) + INPUT_SHAPE))
    with torch.no_grad():
        lb, ub = x - eps, x + eps
        lbs, ubs = [lb], [ub]
        act = x_lp
        primals = []
        for l in fixed:
            if isinstance(l, nn.ReLU):       # pre-activation bounds feed a graph layer
                lbs.append(lb)
                ubs.append(ub)
            lb, ub = _interval(l, lb, ub)
            act = l(act)
            primals.append(act.reshape(-1).clone())
        # property layer, per sample
        pw = torch.stack([p.weight[0] for p in prop_layers])          # (B, n_last)
        pb = torch.stack([p.bias[0] for p in prop_layers])            # (B,)
        wp, wn = pw.clamp(min=0), pw.clamp(max=0)
        lbs.append(((lb * wp).sum(1) + (ub * wn).sum(1) + pb).unsqueeze(1))
        ubs.append(((ub * wp).sum(1) + (lb * wn).sum(1) + pb).unsqueeze(1))
        primals.append(((act * pw).sum(1) + pb).reshape(-1))
    duals, bab = [], []
    for k in range(1, len(shapes) - 1):
        n = int(np.prod(shapes[k]))
        d = rng.uniform(0, 1, (B * n, 3)) * (rng.uniform(0, 1, (B * n, 3)) < dual_density)
        duals.append(t(d))
        l2, u2 = lbs[k].reshape(B, n), ubs[k].reshape(B, n)
        m = torch.zeros(B, n, dtype=torch.int64)
        m[(l2 < 0) & (u2 > 0)] = -1
        m[l2 >= 0] = 1
        bab.append(m)
    masks = torch.cat([(m == -1).float() for m in bab], 1)
    return SubproblemBatch(lbs, ubs, duals, primals, x_lp,
                           {"fixed_layers": fixed, "prop_layers": prop_layers}, masks, bab)
