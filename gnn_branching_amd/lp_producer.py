"""Input producer without Gurobi: the LP relaxation of a BaB subproblem on scipy's HiGHS.

The reference builds its subproblems with Gurobi (plnn/conv_kwinter_gen.py build_the_model :179-555 and
update_the_model :558-795): intermediate bounds, the triangle relaxation of every undecided ReLU as three constraints
(``v >= 0``, ``v >= pre``, ``v <= slope * pre + bias``, :419-421 / :456-458), minimise the folded property output, then
read the primal point, the per-layer primal values and the constraint duals ``Pi`` (:535-552) -- exactly the arguments
``GraphChoice.decision`` takes.  Gurobi is not available here, so this module restates that producer on
``scipy.optimize.linprog(method="highs")``:

* intermediate bounds: Wong-Kolter bounds (``kw_bounds``: the dual-network backward pass of
  convex_adversarial/dual_network.py:15-121 for every pre-ReLU layer, as init_kw_bounds / update_kw_bounds use it,
  dual_network_linear_approximation.py:205-451) intersected with interval arithmetic and, for a child domain, with its
  parent's bounds; below the split layer the parent's bounds are kept (incremental update), honouring the split mask (a
  node forced passing gets ``pre >= 0``, forced blocking ``pre <= 0``).  ``bounds="interval"`` selects plain interval
  arithmetic (the looser relaxation round 1 used);
* the LP: one variable block per network layer, affine layers as equalities with their scipy.sparse matrix, decided
  ReLUs as equalities / fixed bounds, undecided ones as the two inequality rows (``v >= 0`` is the variable bound);
* duals in Gurobi's sign convention: ``Pi(v >= pre) = -marginal(pre - v <= 0)``, ``Pi(v <= slope pre + bias) =
  marginal(v - slope pre <= bias)``; the column of ``v >= 0`` stays 0 (graph_conv.py reads columns 1 and 2 only).

PARITY UNPINNED: neither the LP (no Gurobi to compare with; LP duals are not unique anyway) nor the Wong-Kolter bounds (the
reference module imports ``plnn.model`` -> ``gurobipy`` and cannot be imported here) can be pinned against the reference's
numbers.  ``tests/test_lp_producer.py`` checks what can be checked without it -- soundness of every bound on sampled
points, KW within interval, LP bound with KW >= LP bound with interval, primal feasibility, complementary slackness of the
reported duals, monotonicity under branching, incremental == inherited below the split.
Host-side CPU code, like the reference's; the GPU only scores.
"""
from dataclasses import dataclass
from typing import List

import numpy as np
import scipy.sparse as sp
import torch
from scipy.optimize import linprog
from torch import nn
from torch.nn import functional as F

from .plnn.modules import Flatten


def _is_flatten(layer):
    return isinstance(layer, Flatten) or type(layer).__name__ == "Flatten"


def conv_matrix(layer, in_shape):
    """scipy.sparse CSR matrix of a Conv2d on a (C, H, W) input, C-order flattened both sides."""
    c_in, h_in, w_in = in_shape
    w = layer.weight.detach().double().numpy()
    c_out, _, kh, kw = w.shape
    sy, sx = layer.stride
    py, px = layer.padding
    h_out = (h_in + 2 * py - kh) // sy + 1
    w_out = (w_in + 2 * px - kw) // sx + 1
    co, oy, ox, ci, ky, kx = np.meshgrid(np.arange(c_out), np.arange(h_out), np.arange(w_out), np.arange(c_in), np.arange(kh),
                                         np.arange(kw), indexing="ij")
    iy, ix = oy * sy - py + ky, ox * sx - px + kx
    ok = (iy >= 0) & (iy < h_in) & (ix >= 0) & (ix < w_in)
    rows = ((co * h_out + oy) * w_out + ox)[ok]
    cols = ((ci * h_in + iy) * w_in + ix)[ok]
    vals = w[co[ok], ci[ok], ky[ok], kx[ok]]
    n_out, n_in = c_out * h_out * w_out, c_in * h_in * w_in
    return sp.csr_matrix((vals, (rows, cols)), shape=(n_out, n_in)), (c_out, h_out, w_out)


@dataclass
class Subproblem:
    """What one LP solve yields: the arguments of ``GraphChoice.decision`` plus the bounds of the domain."""
    lb: float                        # LP optimum: lower bound on the property output over the domain
    ub: float                        # network output at the LP's input point: upper bound on the minimum
    ub_point: torch.Tensor           # (1, C, H, W)
    lower_all: List[torch.Tensor]    # per network layer (input first), layer-shaped
    upper_all: List[torch.Tensor]
    dual_vars: List[torch.Tensor]    # per ReLU layer (N, 3)
    primals: List[list]              # per network layer after the input
    mask: List[torch.Tensor]         # per ReLU layer, {-1, 0, 1}
    bounds64: tuple = None           # (lbs, ubs) in float64 as computed: what a child's bounds are intersected with

    def graph_bounds(self, pre_relu_indices, n_layers):
        idx = [0] + list(pre_relu_indices) + [n_layers]
        return [self.lower_all[i].unsqueeze(0) for i in idx], [self.upper_all[i].unsqueeze(0) for i in idx]


class LayerGraphLP:
    """LP relaxation of ``layers`` (net.layers with the folded Linear(., 1) property layer last) over an input box."""

    def __init__(self, layers, input_lb, input_ub, bounds="kw", lp_method="highs-ipm"):
        self.retry_stats = {"infeasible_first": 0, "flipped_to_feasible": 0}      # see solve(): the widened re-solve of an 'infeasible' IPM answer
        self.lp_method = lp_method
        if bounds not in ("kw", "interval"):
            raise ValueError(bounds)
        self.bound_mode = bounds
        self.layers = list(layers)
        self.input_lb, self.input_ub = input_lb.detach().double(), input_ub.detach().double()
        self.shapes = [tuple(input_lb.shape)]
        self.mats = []                       # per layer: ("affine", A, b) / ("relu",) / ("flatten",)
        shape = tuple(input_lb.shape)
        for l in self.layers:
            if type(l) is nn.Conv2d:
                A, shape = conv_matrix(l, shape)
                b = np.repeat(l.bias.detach().double().numpy(), shape[1] * shape[2])
                self.mats.append(("affine", A, b))
            elif type(l) is nn.Linear:
                A = sp.csr_matrix(l.weight.detach().double().numpy())
                shape = (l.out_features,)
                self.mats.append(("affine", A, l.bias.detach().double().numpy()))
            elif type(l) is nn.ReLU:
                self.mats.append(("relu",))
            elif _is_flatten(l):
                shape = (int(np.prod(shape)),)
                self.mats.append(("flatten",))
            else:
                raise NotImplementedError(type(l))
            self.shapes.append(shape)
        if self.shapes[-1] != (1,):
            raise ValueError("the last layer must be the folded property layer Linear(., 1)")
        self.pre_relu_indices = [i for i, l in enumerate(self.layers) if type(l) is nn.ReLU]   # bounds-list index of the pre-activation
        self.offsets = np.cumsum([0] + [int(np.prod(s)) for s in self.shapes])

    # ---- intermediate bounds ----------------------------------------------------------------
    def interval_bounds(self, mask):
        """Interval arithmetic through the layers, honouring the split mask (list per ReLU layer, {-1, 0, 1})."""
        lbs, ubs = [self.input_lb.clone()], [self.input_ub.clone()]
        r = 0
        for l in self.layers:
            lo, up = lbs[-1], ubs[-1]
            if type(l) is nn.Conv2d:
                wp, wn = l.weight.double().clamp(min=0), l.weight.double().clamp(max=0)
                nl = F.conv2d(lo[None], wp, l.bias.double(), l.stride, l.padding) + F.conv2d(up[None], wn, None, l.stride, l.padding)
                nu = F.conv2d(up[None], wp, l.bias.double(), l.stride, l.padding) + F.conv2d(lo[None], wn, None, l.stride, l.padding)
                nl, nu = nl[0], nu[0]
            elif type(l) is nn.Linear:
                wp, wn = l.weight.double().clamp(min=0), l.weight.double().clamp(max=0)
                nl = wp @ lo + wn @ up + l.bias.double()
                nu = wp @ up + wn @ lo + l.bias.double()
            elif type(l) is nn.ReLU:
                m = mask[r].reshape(lo.shape)
                # a split tightens the PRE-activation bounds of the node (update_the_model, conv_kwinter_gen.py:573-585)
                lo = torch.where(m == 1, lo.clamp(min=0), lo)
                up = torch.where(m == 0, up.clamp(max=0), up)
                lbs[-1], ubs[-1] = lo, up
                nl, nu = lo.clamp(min=0), up.clamp(min=0)
                r += 1
            else:
                nl, nu = lo.reshape(-1), up.reshape(-1)
            lbs.append(nl)
            ubs.append(nu)
        return lbs, ubs

    def _interval_step(self, l, lo, up):
        """Interval image of [lo, up] under one affine layer."""
        if type(l) is nn.Conv2d:
            wp, wn = l.weight.double().clamp(min=0), l.weight.double().clamp(max=0)
            nl = F.conv2d(lo[None], wp, l.bias.double(), l.stride, l.padding) + F.conv2d(up[None], wn, None, l.stride, l.padding)
            nu = F.conv2d(up[None], wp, l.bias.double(), l.stride, l.padding) + F.conv2d(lo[None], wn, None, l.stride, l.padding)
            return nl[0], nu[0]
        wp, wn = l.weight.double().clamp(min=0), l.weight.double().clamp(max=0)
        return wp @ lo + wn @ up + l.bias.double(), wp @ up + wn @ lo + l.bias.double()

    def _kw_layer(self, q, lbs, ubs):
        """Wong-Kolter bounds on the output of the affine layer ``self.layers[q]`` given pre-activation bounds ``lbs`` /
        ``ubs`` (bounds-list indexing: entry i = output of layer i - 1) of every ReLU below it.

        One backward pass of the dual network for all N_q output coordinates at once (the identity as a batch of N_q
        directions, convex_adversarial/dual_network.py:51-102 builds the same pass layer by layer): through an affine layer
        nu_hat = W^T nu and the constant gains nu.b; through a ReLU with pre-activation bounds (l, u) nu = d * nu_hat with
        d = 1 (l >= 0), 0 (u <= 0), u / (u - l) (ambiguous, the set I), and the lower / upper bound gain
        sum_I (-d l) min(nu_hat, 0) / sum_I (-d l) max(nu_hat, 0) (DualReLU.objective, dual_layers.py: the l [nu]_+ term); at the
        input the box [x_lo, x_hi] contributes nu_hat^+ x_lo + nu_hat^- x_hi (lower) and nu_hat^+ x_hi + nu_hat^- x_lo (upper) (InfBallBounded)."""
        n = int(np.prod(self.shapes[q + 1]))
        nu = torch.eye(n, dtype=torch.float64).reshape((n,) + tuple(self.shapes[q + 1]))
        lo = torch.zeros(n, dtype=torch.float64)
        up = torch.zeros(n, dtype=torch.float64)
        for i in range(q, -1, -1):
            l = self.layers[i]
            if type(l) is nn.Conv2d:
                bias = l.bias.double()
                c = (nu.sum((2, 3)) * bias[None]).sum(1)
                lo, up = lo + c, up + c
                hin, win = self.shapes[i][1], self.shapes[i][2]
                hout, wout = self.shapes[i + 1][1], self.shapes[i + 1][2]
                opad = (hin - ((hout - 1) * l.stride[0] - 2 * l.padding[0] + l.kernel_size[0]),
                        win - ((wout - 1) * l.stride[1] - 2 * l.padding[1] + l.kernel_size[1]))
                nu = F.conv_transpose2d(nu, l.weight.double(), None, l.stride, l.padding, output_padding=opad)
            elif type(l) is nn.Linear:
                c = nu @ l.bias.double()
                lo, up = lo + c, up + c
                nu = nu @ l.weight.double()
            elif type(l) is nn.ReLU:
                pl, pu = lbs[i].reshape(-1), ubs[i].reshape(-1)
                amb = (pl < 0) & (pu > 0)
                d = torch.where(pl >= 0, torch.ones_like(pl), torch.zeros_like(pl))
                d = torch.where(amb, pu / (pu - pl).clamp(min=1e-300), d)
                flat = nu.reshape(n, -1)
                gain = torch.where(amb, -d * pl, torch.zeros_like(pl))          # >= 0
                lo = lo + (flat.clamp(max=0) * gain[None]).sum(1)
                up = up + (flat.clamp(min=0) * gain[None]).sum(1)
                nu = (flat * d[None]).reshape(nu.shape)
            else:                                                               # Flatten
                nu = nu.reshape((n,) + tuple(self.shapes[i]))
        flat = nu.reshape(n, -1)
        xl, xu = self.input_lb.reshape(-1), self.input_ub.reshape(-1)
        lo = lo + flat.clamp(min=0) @ xl + flat.clamp(max=0) @ xu
        up = up + flat.clamp(min=0) @ xu + flat.clamp(max=0) @ xl
        return lo.reshape(self.shapes[q + 1]), up.reshape(self.shapes[q + 1])

    def kw_bounds(self, mask, parent=None, split_layer=None):
        """Intermediate bounds of the domain ``mask``: for every affine layer the Wong-Kolter bounds (``_kw_layer``, built on the
        tightened bounds of the layers below, as the reference's DualNetwork does with zl / zu) intersected with interval
        arithmetic, then the split mask applied to the pre-activation bounds.

        ``parent`` = (lbs, ubs) of the parent domain and ``split_layer`` = index of the ReLU layer the child was split on: the
        incremental form of update_kw_bounds (dual_network_linear_approximation.py:296-451) -- every bound up to and including
        that layer's pre-activation is the parent's (only the split node is clamped, :311-318), later layers are recomputed and
        intersected with the parent's bounds (:398-399).  Returns (lbs, ubs) like ``interval_bounds``."""
        lbs, ubs = [self.input_lb.clone()], [self.input_ub.clone()]
        keep_upto = self.pre_relu_indices[split_layer] if (parent is not None and split_layer is not None) else -1
        r = 0
        first_affine = True
        for q, l in enumerate(self.layers):
            lo, up = lbs[-1], ubs[-1]
            if type(l) in (nn.Conv2d, nn.Linear):
                if q + 1 <= keep_upto:
                    nl, nu = parent[0][q + 1].double().clone(), parent[1][q + 1].double().clone()
                else:
                    nl, nu = self._interval_step(l, lo, up)
                    if not first_affine:                       # the first affine layer's interval image of the box is exact
                        kl, ku = self._kw_layer(q, lbs + [None], ubs + [None])
                        nl, nu = torch.maximum(nl, kl), torch.minimum(nu, ku)
                    if parent is not None:
                        nl, nu = torch.maximum(nl, parent[0][q + 1].double()), torch.minimum(nu, parent[1][q + 1].double())
                first_affine = False
            elif type(l) is nn.ReLU:
                m = mask[r].reshape(lo.shape)
                lo = torch.where(m == 1, lo.clamp(min=0), lo)
                up = torch.where(m == 0, up.clamp(max=0), up)
                lbs[-1], ubs[-1] = lo, up
                nl, nu = lo.clamp(min=0), up.clamp(min=0)
                r += 1
            else:
                nl, nu = lo.reshape(-1), up.reshape(-1)
            lbs.append(nl)
            ubs.append(nu)
        return lbs, ubs

    def bounds(self, mask, parent=None, split_layer=None):
        if self.bound_mode == "interval":
            return self.interval_bounds(mask)
        return self.kw_bounds(mask, parent, split_layer)

    # ---- the LP -----------------------------------------------------------------------------
    def solve(self, mask, parent=None, split_layer=None):
        """Bounds + LP for the domain described by ``mask``; returns a Subproblem, or None when the domain is infeasible.
        ``parent`` (a Subproblem) and ``split_layer`` (the ReLU layer of the split that made this child): bounds below the
        split are inherited, bounds above it recomputed and intersected with the parent's (update_kw_bounds)."""
        mask = [m.clone() for m in mask]
        pb = None if parent is None else parent.bounds64
        lbs, ubs = self.bounds(mask, pb, split_layer)
        for lo, up in zip(lbs, ubs):
            if bool((lo > up + 1e-9).any()):
                return None
        off = self.offsets
        nvar = int(off[-1])
        lo_v = np.concatenate([t.reshape(-1).numpy() for t in lbs])
        up_v = np.concatenate([t.reshape(-1).numpy() for t in ubs])
        eq_rows, eq_rhs, ub_rows, ub_rhs = [], [], [], []
        amb_index = []                      # per ReLU layer: node ids of the undecided nodes, in constraint order
        r = 0
        for li, kind in enumerate(self.mats):
            src, dst = slice(off[li], off[li + 1]), slice(off[li + 1], off[li + 2])
            n_dst = dst.stop - dst.start
            eye = sp.identity(n_dst, format="csr")

            def row(block_src, block_dst, nrows):
                left = sp.csr_matrix((nrows, src.start))
                mid = sp.csr_matrix((nrows, dst.start - src.stop))
                right = sp.csr_matrix((nrows, nvar - dst.stop))
                return sp.hstack([left, block_src, mid, block_dst, right], format="csr")
            if kind[0] == "affine":
                _, A, b = kind
                eq_rows.append(row(-A, eye, n_dst))
                eq_rhs.append(b)
            elif kind[0] == "flatten":
                eq_rows.append(row(-eye, eye, n_dst))
                eq_rhs.append(np.zeros(n_dst))
            else:
                pre_lo, pre_up = lo_v[src], up_v[src]
                m = mask[r].reshape(-1).numpy().copy()
                # bounds decide what the split mask left open (build_the_model :404-421)
                m[(m == -1) & (pre_lo >= 0)] = 1
                m[(m == -1) & (pre_up <= 0)] = 0
                mask[r] = torch.from_numpy(m).to(mask[r].dtype)
                passing, blocked, amb = np.nonzero(m == 1)[0], np.nonzero(m == 0)[0], np.nonzero(m == -1)[0]
                if len(passing):
                    S = sp.csr_matrix((np.ones(len(passing)), (np.arange(len(passing)), passing)), shape=(len(passing), n_dst))
                    eq_rows.append(row(-S, S, len(passing)))
                    eq_rhs.append(np.zeros(len(passing)))
                post_up = np.maximum(pre_up, 0.0)
                post_up[blocked] = 0.0
                up_v[dst.start:dst.stop] = post_up
                lo_v[dst.start:dst.stop] = 0.0
                if len(amb):
                    S = sp.csr_matrix((np.ones(len(amb)), (np.arange(len(amb)), amb)), shape=(len(amb), n_dst))
                    slope = pre_up[amb] / (pre_up[amb] - pre_lo[amb])
                    bias = -pre_lo[amb] * slope
                    ub_rows.append(row(S, -S, len(amb)))                                   # pre - v <= 0
                    ub_rhs.append(np.zeros(len(amb)))
                    ub_rows.append(row(-sp.diags(slope) @ S, S, len(amb)))                 # v - slope pre <= bias
                    ub_rhs.append(bias)
                amb_index.append(amb)
                r += 1
        c = np.zeros(nvar)
        c[-1] = 1.0
        # variables whose box is narrower than 1e-6 without being fixed (pre-activations that do not depend on the input: their
        # two bounds differ by float64 rounding) get 1e-7 of slack: HiGHS' clean-up simplex after the crossover otherwise reports
        # a feasible cifar_deep_kw root as infeasible (259 such variables there, primal residual 2e-8 before the push phase)
        width = up_v - lo_v
        tiny = (width > 0) & (width < 1e-6)
        lo_b, up_b = np.where(tiny, lo_v - 1e-7, lo_v), np.where(tiny, up_v + 1e-7, up_v)
        args = dict(A_ub=sp.vstack(ub_rows, format="csr") if ub_rows else None, b_ub=np.concatenate(ub_rhs) if ub_rhs else None,
                    A_eq=sp.vstack(eq_rows, format="csr"), b_eq=np.concatenate(eq_rhs), bounds=np.stack([lo_b, up_b], 1))
        # interior point + crossover: on the CIFAR networks (10^4 variables, 3*10^5 non-zeros, almost every row an equality)
        # HiGHS' dual simplex needs 1.8*10^5 degenerate pivots = 180 s per LP, its IPM 17 iterations = 0.5 s; the crossover
        # ends in a basic solution, so the marginals are vertex duals like the simplex's (and Gurobi's)
        res = linprog(c, method=self.lp_method, **args)
        if res.status == 2 and self.lp_method != "highs":
            # "infeasible" from the clean-up simplex can still be rounding: once more with 1e-6 of slack on every free box (a relaxation: a
            # child it turns feasible is kept and bounded, which is sound; one that stays infeasible is pruned).  `retry_stats` counts how
            # often that happens and how often it flips the verdict (tests/test_gpu_bab_trace.py prints it per run)
            self.retry_stats["infeasible_first"] += 1
            free = width > 0
            args["bounds"] = np.stack([np.where(free, lo_v - 1e-6, lo_v), np.where(free, up_v + 1e-6, up_v)], 1)
            res = linprog(c, method=self.lp_method, **args)
            if res.status == 0:
                self.retry_stats["flipped_to_feasible"] += 1
        if res.status not in (0, 2):         # HiGHS' presolve sometimes ends without a model status on an infeasible child
            res = linprog(c, method="highs", options={"presolve": False}, **args)
        if res.status == 2:
            return None                      # infeasible domain
        if res.status != 0:
            raise RuntimeError(f"HiGHS: {res.message}")
        z = res.x
        marg = res.ineqlin.marginals if ub_rows else np.zeros(0)
        duals, pos = [], 0
        for ridx, amb in enumerate(amb_index):
            n = int(np.prod(self.shapes[self.pre_relu_indices[ridx] + 1]))
            d = np.zeros((n, 3))
            if len(amb):
                d[amb, 1] = -marg[pos:pos + len(amb)]                 # Pi of  v >= pre
                d[amb, 2] = marg[pos + len(amb):pos + 2 * len(amb)]   # Pi of  v <= slope pre + bias
                pos += 2 * len(amb)
            duals.append(torch.from_numpy(d).float())
        x0 = torch.from_numpy(z[off[0]:off[1]].copy()).float().reshape(self.shapes[0])
        with torch.no_grad():
            act = x0[None]
            for l in self.layers:
                act = l(act)
        primals = [z[off[i + 1]:off[i + 2]].tolist() for i in range(len(self.layers))]
        lower_all = [t.float() for t in lbs]
        upper_all = [t.float() for t in ubs]
        lower_all[-1] = torch.tensor([float(res.fun)])
        return Subproblem(float(res.fun), float(act.reshape(-1)[0]), x0[None], lower_all, upper_all, duals, primals, mask, (lbs, ubs))


def branch_and_bound(lp, scorer, layers, eps=1e-4, max_nodes=200, decision_bound=None, log=print, dump=None):
    """The BaB loop of plnn/relu_conv_gnnkwthreshold.py:120-262 in its plain form: pick the domain with the lowest bound,
    split the ReLU the scorer names, bound both children, keep those that can still improve the answer.

    ``scorer(sub, layers_dict) -> [layer, idx]`` (gnn_scorer / babsr_scorer below).  ``decision_bound``: stop as soon as the
    sign of (minimum - decision_bound) is known (the reference verifies with decision_bound = 0, :257-262); None: minimise
    to ``eps``.  ``dump``: a callable that receives the run's trace in the format the reference writes to
    ``./gnn_dump_files/<trace_name>`` (relu_conv_gnnkwthreshold.py:75-79): per branch the line of :201-202 (``kw: improvement -1
    decision None``: this plain loop never bounds the KW fall-back's children) and the global lower bound of :256-257 -- the file a
    Gurobi owner can diff against a ``--bab_gnn`` run's dump.  Returns (global_lb, global_ub, visited LP solves)."""
    from .bab_caller import gnn_improvement, trace_line
    fixed = {"fixed_layers": list(layers[:-1]), "prop_layers": [layers[-1]]}
    root_mask = [torch.full((int(np.prod(lp.shapes[i + 1])),), -1, dtype=torch.long) for i in lp.pre_relu_indices]
    root = lp.solve(root_mask)
    if root is None:
        raise RuntimeError("infeasible root domain")
    global_lb, global_ub, domains, visited = root.lb, root.ub, [root], 0
    closed_lb = float("inf")                # lowest bound among the leaves that were closed (proved >= decision_bound or optimal)
    while domains and global_ub - global_lb > eps and visited < max_nodes:
        if decision_bound is not None and (global_lb >= decision_bound or global_ub < decision_bound):
            break
        domains.sort(key=lambda d: d.lb)
        dom = domains.pop(0)
        if not any(bool((m == -1).any()) for m in dom.mask):
            closed_lb = min(closed_lb, dom.lb)                          # fully decided: its LP is exact
            global_lb = min([d.lb for d in domains] + [closed_lb])
            continue
        decision = scorer(dom, fixed)
        children = []
        for choice in (0, 1):
            m = [t.clone() for t in dom.mask]
            m[decision[0]][decision[1]] = choice
            child = lp.solve(m, parent=dom, split_layer=decision[0])
            visited += 1
            if child is None:
                continue
            global_ub = min(global_ub, child.ub)
            children.append(child)
        log(f"branch {visited} decision {decision} parent lb {dom.lb:.5f} children lb {[round(c.lb, 5) for c in children]}")
        for c in children:
            if c.lb < global_ub - eps and (decision_bound is None or c.lb < decision_bound):
                domains.append(c)
            else:
                closed_lb = min(closed_lb, c.lb)
        global_lb = min([d.lb for d in domains] + [closed_lb, global_ub])
        if dump is not None:
            lbs = [c.lb for c in children] + [float("inf")] * (2 - len(children))      # an infeasible child cannot contain a counter-example
            imp = gnn_improvement(lbs[0], lbs[1], dom.lb) if dom.lb < 0 else 0.0
            dump(trace_line(visited, decision, imp, decision))
            dump(f"{global_lb}\n")
    return global_lb, global_ub, visited


def branch_and_bound_threshold(lp, scorer, kw_scorer, layers, eps=1e-4, max_branches=50, decision_bound=None, branching_threshold=0.2,
                               kwbd_threshold=10, sparsest_layer=0, log=print, dump=None):
    """The BaB loop of plnn/relu_conv_gnnkwthreshold.py:126-262 WITH its control flow (the loop `bab_mip.py --bab_gnn` runs): branch on the
    GNN's decision and bound its two children (:143-146); when the GNN's improvement of the bound (:151) is below ``branching_threshold``
    (:155) ask the BaBSR heuristic (``choose_node_conv``, :157), skip a KW point that was inefficient ``kwbd_threshold`` times (:160-167),
    else bound ITS two children too (:168-173) and keep the better pair: a KW point that improves less than the GNN's and less than 0.05
    is counted as inefficient, one that improves more replaces the GNN's decision (:176-192, ``bab_caller.resolve_branching``).

    ``scorer(sub, layers_dict) -> [layer, idx]`` (the GNN); ``kw_scorer(sub, icp_score, random_order, sparsest_layer) -> (decision,
    icp_score)`` (BaBSR; the counter is the loop's state as at :119, :157).  ``log`` receives the per-branch line of :204 (= the line
    of :201-202 in the dump), ``dump`` the dump file's lines (:201-202, :256-257).  A domain's GNN decision is computed when the domain
    is picked (the reference computes it when the domain is created, :230 / :239: the same function of the same domain).
    With a ``decision_bound`` the dumped global lower bound follows the reference's rule (:251-255): the lowest bound among the open domains,
    or ``global_ub - eps`` once none is left, so a dump compares line for line.  Remaining deviations, none of which changes a decision: the
    loop stops after ``max_branches`` branches (the reference runs until its caller's timeout); a domain whose bound is already >= 0 would get
    GNN improvement 1.0 instead of the reference's division by ``-2 * lower_bound`` (with ``decision_bound`` = 0 such a domain is never
    added, :226 / :235); infeasible children count as lower bound +inf (Gurobi reports them infeasible too; the reference has no such branch
    because its children inherit feasible parents' bounds); without a ``decision_bound`` leaves closed at optimality keep the minimum honest.
    Returns (global_lb, global_ub, LP solves, branches, branches that bounded a KW decision, branches that used it)."""
    from .bab_caller import gnn_improvement, resolve_branching, trace_line
    fixed = {"fixed_layers": list(layers[:-1]), "prop_layers": [layers[-1]]}
    n_relu = len(lp.pre_relu_indices)
    root_mask = [torch.full((int(np.prod(lp.shapes[i + 1])),), -1, dtype=torch.long) for i in lp.pre_relu_indices]
    root = lp.solve(root_mask)
    if root is None:
        raise RuntimeError("infeasible root domain")
    random_order = [l for l in range(n_relu) if l != sparsest_layer]
    random_order = ([sparsest_layer] if 0 <= sparsest_layer < n_relu else []) + random_order      # :97-103
    global_lb, global_ub, domains = root.lb, root.ub, [root]
    solves, nb_states, icp, n_kw, n_kw_used = 0, 0, 0, 0, 0
    ineff_kw_dc, closed_lb = {}, float("inf")

    def bound_children(dom, decision):
        out = []
        for choice in (0, 1):
            m = [t.clone() for t in dom.mask]
            m[decision[0]][decision[1]] = choice
            out.append(lp.solve(m, parent=dom, split_layer=decision[0]))
        return out

    def child_lb(c):                          # an infeasible child cannot contain a counter-example
        return float("inf") if c is None else c.lb

    while domains and global_ub - global_lb > eps and nb_states < 2 * max_branches:
        if decision_bound is not None and (global_lb >= decision_bound or global_ub < decision_bound):
            break
        domains.sort(key=lambda d: d.lb)
        dom = domains.pop(0)
        if not any(bool((m == -1).any()) for m in dom.mask):
            closed_lb = min(closed_lb, dom.lb)
            global_lb = min([d.lb for d in domains] + [closed_lb])
            continue
        gnn_decision = scorer(dom, fixed)                                                                  # :117 / :230 / :239
        nb_states += 2                                                                                     # :138
        children = bound_children(dom, gnn_decision)                                                       # :143-146
        solves += 2
        gnn_imp = gnn_improvement(child_lb(children[0]), child_lb(children[1]), dom.lb) if dom.lb < 0 else 1.0     # :151
        decision, kw_decision, kw_imp = gnn_decision, None, -1
        if gnn_imp < branching_threshold:                                                                  # :155
            kw_decision, icp = kw_scorer(dom, icp, random_order, sparsest_layer)                           # :157
            if ineff_kw_dc.get(f"{kw_decision[0]}-{kw_decision[1]}", 0) < kwbd_threshold:                  # :160-167
                kw_children = bound_children(dom, kw_decision)                                             # :168-171
                solves += 2
                n_kw += 1
                kw_imp = gnn_improvement(child_lb(kw_children[0]), child_lb(kw_children[1]), dom.lb)       # :173
                decision, used_kw = resolve_branching(gnn_decision, gnn_imp, kw_decision, kw_imp, ineff_kw_dc)     # :176-192
                if used_kw:
                    children = kw_children
                    n_kw_used += 1
        line = trace_line(nb_states, decision, gnn_imp, gnn_decision, kw_imp, kw_decision)                 # :201-204
        log(line.rstrip())
        for c in children:
            if c is not None:
                global_ub = min(global_ub, c.ub)                                                           # :209-214
        for c in children:
            if c is None:
                continue
            if c.lb < global_ub - eps and (decision_bound is None or c.lb < decision_bound):               # :226, :235
                domains.append(c)
            else:
                closed_lb = min(closed_lb, c.lb)
        if decision_bound is not None:                                                                     # :251-255
            global_lb = min(d.lb for d in domains) if domains else global_ub - eps
        else:
            global_lb = min([d.lb for d in domains] + [closed_lb, global_ub])
        if dump is not None:
            dump(line)
            dump(f"{global_lb}\n")                                                                         # :256-257
    return global_lb, global_ub, solves, nb_states // 2, n_kw, n_kw_used


def branch_and_bound_online(lp, graph, layers, eps=1e-4, max_nodes=200, decision_bound=None, branching_threshold=0.2,
                            online_threshold=5, sparsest_layer=0, log=print):
    """The BaB loop of plnn/relu_conv_online.py:126-276: branch on the GNN's decision; when its improvement of the bound
    is below ``branching_threshold`` also bound the KW (BaBSR) decision's children and keep the better pair; a GNN decision
    that lost ``online_threshold`` times triggers ``graph.online_learning(kw_decision, improve)`` (:196-205).

    ``graph``: graphnet.graph_score_online.GraphChoice.  Returns (global_lb, global_ub, visited LP solves, online steps)."""
    from .bab_caller import gnn_improvement, resolve_online, trace_line
    from .plnn.kw_score_conv import choose_node_conv
    fixed = {"fixed_layers": list(layers[:-1]), "prop_layers": [layers[-1]]}
    n_relu = len(lp.pre_relu_indices)
    root_mask = [torch.full((int(np.prod(lp.shapes[i + 1])),), -1, dtype=torch.long) for i in lp.pre_relu_indices]
    root = lp.solve(root_mask)
    if root is None:
        raise RuntimeError("infeasible root domain")
    random_order = [l for l in range(n_relu) if l != sparsest_layer]
    random_order = ([sparsest_layer] if 0 <= sparsest_layer < n_relu else []) + random_order      # :104-109
    global_lb, global_ub, domains, visited, icp, steps = root.lb, root.ub, [root], 0, 0, 0
    wrong_pts_dc, closed_lb = {}, float("inf")

    def bound_children(dom, decision):
        out = []
        for choice in (0, 1):
            m = [t.clone() for t in dom.mask]
            m[decision[0]][decision[1]] = choice
            out.append(lp.solve(m, parent=dom, split_layer=decision[0]))
        return out

    def child_lb(c):                          # an infeasible child cannot contain a counter-example
        return float("inf") if c is None else c.lb

    while domains and global_ub - global_lb > eps and visited < max_nodes:
        if decision_bound is not None and (global_lb >= decision_bound or global_ub < decision_bound):
            break
        domains.sort(key=lambda d: d.lb)
        dom = domains.pop(0)
        if not any(bool((m == -1).any()) for m in dom.mask):
            closed_lb = min(closed_lb, dom.lb)
            global_lb = min([d.lb for d in domains] + [closed_lb])
            continue
        lbg, ubg = dom.graph_bounds(lp.pre_relu_indices, len(lp.layers))
        gnn_decision = graph.decision(lbg, ubg, dom.dual_vars, dom.ub_point, dom.primals, fixed, dom.mask)       # :146
        children = bound_children(dom, gnn_decision)
        visited += 2
        gnn_imp = gnn_improvement(child_lb(children[0]), child_lb(children[1]), dom.lb) if dom.lb < 0 else 1.0     # :156
        kw_decision, kw_imp, kw_children = None, -1, None
        if gnn_imp < branching_threshold:                                                                         # :158-166
            kw_decision, icp = choose_node_conv(dom.lower_all, dom.upper_all, dom.mask, lp.layers, lp.pre_relu_indices, icp,
                                                random_order, sparsest_layer)
            kw_children = bound_children(dom, kw_decision)
            visited += 2
            kw_imp = gnn_improvement(child_lb(kw_children[0]), child_lb(kw_children[1]), dom.lb)
        decision, used_kw, learn, improve = resolve_online(gnn_decision, gnn_imp, kw_decision, kw_imp, wrong_pts_dc, online_threshold)
        if used_kw:
            children = kw_children
        if learn:
            graph.online_learning(kw_decision, improve)                                                           # :205
            steps += 1
        graph.del_score()                                                                                         # :209
        log(trace_line(visited, decision, gnn_imp, gnn_decision, kw_imp, kw_decision).rstrip())
        for c in children:
            if c is None:
                continue
            global_ub = min(global_ub, c.ub)
        for c in children:
            if c is None:
                continue
            if c.lb < global_ub - eps and (decision_bound is None or c.lb < decision_bound):
                domains.append(c)
            else:
                closed_lb = min(closed_lb, c.lb)
        global_lb = min([d.lb for d in domains] + [closed_lb, global_ub])
    return global_lb, global_ub, visited, steps


# ---- scorers for branch_and_bound -------------------------------------------------------------
def gnn_scorer(choice, lp):
    """The GNN decision on a Subproblem (``choice``: graph_score.GraphChoice or bab_caller.BatchedGraphChoice)."""
    def score(sub, layers):
        lbg, ubg = sub.graph_bounds(lp.pre_relu_indices, len(lp.layers))
        return choice.decision(lbg, ubg, sub.dual_vars, sub.ub_point, sub.primals, layers, sub.mask)
    return score


def babsr_scorer(lp):
    """The BaBSR heuristic on a Subproblem (plnn/kw_score_conv.choose_node_conv; relu_conv_gnnkwthreshold.py:157)."""
    from .plnn.kw_score_conv import choose_node_conv
    state = {"icp": 0}

    def score(sub, layers):
        order = list(range(len(lp.pre_relu_indices)))
        decision, state["icp"] = choose_node_conv(sub.lower_all, sub.upper_all, sub.mask, lp.layers, lp.pre_relu_indices, state["icp"], order, -1)
        return decision
    return score
