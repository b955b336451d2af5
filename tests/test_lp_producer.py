"""the Gurobi-free input producer (LP relaxation on scipy-HiGHS).  Parity with the reference's Gurobi numbers
is unpinned (no Gurobi here), so these check what is checkable: soundness, feasibility, dual signs and complementary
slackness, monotonicity under branching -- and, on the GPU, that the producer's output drives the scorer through a BaB run."""
import numpy as np
import pytest
import torch
from torch import nn

from gnn_branching_amd import lp_producer, nets

SPEC = [("conv", 3, 8, 4, 2, 1), ("relu",), ("flatten",), ("linear", 8 * 16 * 16, 32), ("relu",), ("linear", 32, 10)]


@pytest.fixture(scope="module")
def problem():
    nets.register_arch("toy_lp", SPEC, seed=321)
    layers = nets.load_verified_net("toy_lp", 3, 5)
    rng = np.random.RandomState(4)
    x = torch.from_numpy(rng.standard_normal((3, 32, 32)).astype(np.float32))
    eps = 0.03
    lp = lp_producer.LayerGraphLP(layers, x - eps, x + eps)
    mask = [torch.full((int(np.prod(lp.shapes[i + 1])),), -1, dtype=torch.long) for i in lp.pre_relu_indices]
    return layers, x, eps, lp, mask, lp.solve(mask)


def amax(t):
    return float(t.abs().max()) if t.numel() else 0.0


def run(layers, x):
    acts = [x]
    with torch.no_grad():
        a = x[None]
        for l in layers:
            a = l(a)
            acts.append(a[0])
    return acts


def test_bound_is_sound_and_point_is_feasible(problem):
    layers, x, eps, lp, mask, sub = problem
    rng = np.random.RandomState(0)
    outs = [run(layers, x + eps * torch.from_numpy(rng.uniform(-1, 1, x.shape).astype(np.float32)))[-1].item() for _ in range(300)]
    assert sub.lb <= min(outs) + 1e-6                       # a lower bound on everything in the box
    assert sub.lb <= sub.ub + 1e-6
    assert float((sub.ub_point[0] - x).abs().max()) <= eps + 1e-6
    assert abs(run(layers, sub.ub_point[0])[-1].item() - sub.ub) < 1e-5
    # interval bounds contain every concrete activation
    acts = run(layers, x + eps * torch.from_numpy(rng.uniform(-1, 1, x.shape).astype(np.float32)))
    for a, lo, up in zip(acts[:-1], sub.lower_all[:-1], sub.upper_all[:-1]):
        assert bool((a >= lo - 1e-4).all()) and bool((a <= up + 1e-4).all())


def test_primals_satisfy_the_relaxation(problem):
    layers, x, eps, lp, mask, sub = problem
    prev = sub.ub_point[0].double()
    r = 0
    for li, (l, vals) in enumerate(zip(layers, sub.primals)):
        v = torch.tensor(vals, dtype=torch.float64)
        if type(l) is nn.Conv2d or type(l) is nn.Linear:
            with torch.no_grad():
                w, b = l.weight.double(), l.bias.double()
                want = (torch.nn.functional.conv2d(prev.reshape(lp.shapes[li])[None], w, b, l.stride, l.padding)[0] if type(l) is nn.Conv2d
                        else prev.reshape(-1) @ w.t() + b).reshape(-1)
            assert float((v - want).abs().max()) < 1e-5           # affine layers hold with equality
        elif type(l) is nn.ReLU:
            m = sub.mask[r].reshape(-1)
            pre = prev.reshape(-1)
            plo, pup = sub.lower_all[li].reshape(-1).double(), sub.upper_all[li].reshape(-1).double()
            assert amax(v[m == 1] - pre[m == 1]) < 1e-6 and amax(v[m == 0]) < 1e-9
            a = m == -1
            slope = pup[a] / (pup[a] - plo[a])
            assert bool((v[a] >= -1e-7).all()) and bool((v[a] >= pre[a] - 1e-7).all())
            assert bool((v[a] <= slope * (pre[a] - plo[a]) + 1e-6).all())  # the upper face of the triangle
            # duals: Gurobi's signs (minimisation), and complementary slackness
            d = sub.dual_vars[r].double()
            assert bool((d[:, 1] >= -1e-7).all()) and bool((d[:, 2] <= 1e-7).all()) and amax(d[~a]) == 0.0
            slack1 = v[a] - pre[a]
            slack2 = slope * (pre[a] - plo[a]) - v[a]
            assert amax(d[a, 1] * slack1) < 1e-5 and amax(d[a, 2] * slack2) < 1e-5
            r += 1
        prev = v
    assert abs(sub.primals[-1][0] - sub.lb) < 1e-7


def test_branching_is_monotone_and_masks_are_resolved(problem):
    layers, x, eps, lp, mask, sub = problem
    assert all(int((m == -1).sum()) > 0 for m in sub.mask)
    lay = int(np.argmax([int((m == -1).sum()) for m in sub.mask]))
    idx = int((sub.mask[lay] == -1).nonzero()[0])
    kids = []
    for choice in (0, 1):
        m = [t.clone() for t in sub.mask]
        m[lay][idx] = choice
        kids.append(lp.solve(m))
    assert all(k is not None for k in kids)
    assert min(k.lb for k in kids) >= sub.lb - 1e-7          # splitting never loosens the bound
    assert all(int(k.mask[lay][idx]) in (0, 1) for k in kids)


@pytest.mark.gpu
def test_bab_with_the_gpu_scorers(problem):
    """The producer's output is exactly what the scorer's call surface takes: a short BaB run with the GNN decisions and
    one with the BaBSR heuristic, both must tighten the root bound."""
    import os
    from gnn_branching_amd.graphnet.graph_score import GraphChoice
    layers, x, eps, lp, mask, sub = problem
    ckpt = os.path.join(os.path.dirname(__file__), "..", "models", "cifar_trained_gnn", "best_snapshot_None_0_val_acc_0.826_loss_val_0.1036_epoch_57.pt")
    choice = GraphChoice(sub.mask, ckpt)
    choice.verbose = False
    for scorer in (lp_producer.gnn_scorer(choice, lp), lp_producer.babsr_scorer(lp)):
        lines = []
        glb, gub, visited = lp_producer.branch_and_bound(lp, scorer, layers, max_nodes=12, log=lines.append)
        assert visited >= 2 and glb >= sub.lb - 1e-7 and gub <= sub.ub + 1e-7
        assert any("decision" in l for l in lines)


@pytest.mark.gpu
def test_online_bab_learns_on_the_device(problem):
    """The loop of plnn/relu_conv_online.py: GNN decision, KW decision when the GNN's improves too little, online learning
    once a GNN decision lost `online_threshold` times -- with the threshold at 1 and the branching threshold above any
    improvement, every lost comparison is a device-side Adam step; the model must change and the bound must still tighten."""
    import os
    from gnn_branching_amd.graphnet.graph_score_online import GraphChoice
    layers, x, eps, lp, mask, sub = problem
    ckpt = os.path.join(os.path.dirname(__file__), "..", "models", "cifar_trained_gnn", "best_snapshot_None_0_val_acc_0.826_loss_val_0.1036_epoch_57.pt")
    graph = GraphChoice(sub.mask, ckpt)
    graph.verbose = False
    before = np.concatenate([v.numpy().reshape(-1) for v in graph.model.state_dict().values()])
    lines = []
    glb, gub, visited, steps = lp_producer.branch_and_bound_online(lp, graph, layers, max_nodes=24, branching_threshold=2.0,
                                                                   online_threshold=1, sparsest_layer=0, log=lines.append)
    assert visited >= 4 and glb >= sub.lb - 1e-7 and gub <= sub.ub + 1e-7
    assert all(l.startswith("branch ") and " kw: improvement " in l for l in lines)
    after = np.concatenate([v.numpy().reshape(-1) for v in graph.model.state_dict().values()])
    # branches that ended on the KW decision (it may also coincide with the GNN's): threshold 1 makes every KW win a step
    kw_lines = sum(1 for l in lines if l.split(" decision ")[1].split(" gnn:")[0] == l.rsplit(" decision ", 1)[1])
    assert 1 <= steps <= kw_lines
    assert np.abs(after - before).max() > 0 and np.abs(after - before).max() <= 1.05e-4 * steps + 1e-7


# ---- Wong-Kolter intermediate bounds (reference: plnn/dual_network_linear_approximation.py:205-451) -------------------------
KW_SPEC = [("conv", 3, 4, 4, 2, 1), ("relu",), ("conv", 4, 4, 4, 2, 1), ("relu",), ("flatten",), ("linear", 4 * 8 * 8, 24), ("relu",),
           ("linear", 24, 10)]


@pytest.fixture(scope="module")
def kw_problem():
    nets.register_arch("toy_kw", KW_SPEC, seed=77)
    layers = nets.load_verified_net("toy_kw", 2, 6)
    rng = np.random.RandomState(9)
    x = torch.from_numpy(rng.standard_normal((3, 32, 32)).astype(np.float32))
    eps = 0.04
    lp = lp_producer.LayerGraphLP(layers, x - eps, x + eps)
    mask = [torch.full((int(np.prod(lp.shapes[i + 1])),), -1, dtype=torch.long) for i in lp.pre_relu_indices]
    return layers, x, eps, lp, mask


def test_kw_bounds_are_sound_and_within_interval(kw_problem):
    """Every activation of 10^4 points sampled in the box lies inside the KW bounds; the bounds lie inside the interval
    bounds and are strictly tighter on the layers behind the second affine map."""
    layers, x, eps, lp, mask = kw_problem
    kl, ku = lp.kw_bounds(mask)
    il, iu = lp.interval_bounds(mask)
    rng = np.random.RandomState(1)
    pts = x[None] + eps * torch.from_numpy(rng.uniform(-1, 1, (10000,) + tuple(x.shape)).astype(np.float32))
    pts[:64] = x[None] + eps * torch.from_numpy(np.sign(rng.standard_normal((64,) + tuple(x.shape))).astype(np.float32))   # box corners
    a = pts.double()
    with torch.no_grad():
        for i, l in enumerate(layers):
            a = l.double()(a) if isinstance(l, (nn.Conv2d, nn.Linear)) else l(a)
            lo, up = kl[i + 1].reshape(a.shape[1:]), ku[i + 1].reshape(a.shape[1:])
            assert bool((a >= lo[None] - 1e-6).all()) and bool((a <= up[None] + 1e-6).all()), i
    for l in layers:
        l.float()
    tighter = 0.0
    for i in range(len(kl)):
        assert bool((kl[i] >= il[i] - 1e-9).all()) and bool((ku[i] <= iu[i] + 1e-9).all())
        assert bool((kl[i] <= ku[i] + 1e-9).all())
        if i >= 3:
            tighter += float(((iu[i] - il[i]) - (ku[i] - kl[i])).sum())
    assert tighter > 0
    # the raw dual-network pass alone (no interval intersection) is sound too, and it is what tightens the last layers
    q = max(i for i, l in enumerate(layers[:-1]) if isinstance(l, nn.Linear))
    rl, ru = lp._kw_layer(q, kl, ku)
    with torch.no_grad():
        a = pts[:2000].double()
        for i, l in enumerate(layers[:q + 1]):
            a = l.double()(a) if isinstance(l, (nn.Conv2d, nn.Linear)) else l(a)
    for l in layers:
        l.float()
    assert bool((a >= rl[None] - 1e-6).all()) and bool((a <= ru[None] + 1e-6).all())
    assert float((ru - rl).sum()) < float((iu[q + 1] - il[q + 1]).sum())


def test_kw_lp_bound_is_at_least_the_interval_lp_bound(kw_problem):
    layers, x, eps, lp, mask = kw_problem
    sub_kw = lp.solve(mask)
    lp_int = lp_producer.LayerGraphLP(layers, x - eps, x + eps, bounds="interval")
    sub_int = lp_int.solve(mask)
    assert sub_kw.lb >= sub_int.lb - 1e-7                   # a tighter relaxation cannot give a lower bound
    outs = [run(layers, x + eps * torch.from_numpy(np.random.RandomState(s).uniform(-1, 1, x.shape).astype(np.float32)))[-1].item() for s in range(200)]
    assert sub_kw.lb <= min(outs) + 1e-6


def test_kw_bounds_are_monotone_and_incremental_under_branching(kw_problem):
    """A child's bounds lie inside its parent's; below the split layer they ARE the parent's (only the split node clamped);
    the incremental form lies inside the from-scratch bounds of the same mask."""
    layers, x, eps, lp, mask = kw_problem
    root = lp.solve(mask)
    rl = 1                                                   # split on the second ReLU layer
    amb = torch.nonzero(root.mask[rl] == -1).reshape(-1)
    node = int(amb[len(amb) // 2])
    for choice in (0, 1):
        m = [t.clone() for t in root.mask]
        m[rl][node] = choice
        child = lp.solve(m, parent=root, split_layer=rl)
        assert child is not None
        cl, cu = child.bounds64
        pl, pu = root.bounds64
        sl, su = lp.kw_bounds(m)                             # from scratch, no parent
        cut = lp.pre_relu_indices[rl]
        for i in range(len(cl) - 1):
            assert bool((cl[i] >= pl[i] - 1e-9).all()) and bool((cu[i] <= pu[i] + 1e-9).all())
            assert bool((cl[i] >= sl[i] - 1e-9).all()) and bool((cu[i] <= su[i] + 1e-9).all())
            if i < cut:
                assert torch.equal(cl[i], pl[i]) and torch.equal(cu[i], pu[i])
        same = torch.ones_like(cl[cut].reshape(-1), dtype=torch.bool)
        same[node] = False
        assert torch.equal(cl[cut].reshape(-1)[same], pl[cut].reshape(-1)[same])
        assert (cu[cut].reshape(-1)[node] <= 0) if choice == 0 else (cl[cut].reshape(-1)[node] >= 0)
        assert child.lb >= root.lb - 1e-7
