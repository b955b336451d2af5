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
