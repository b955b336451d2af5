"""-m gpu: decision-trace parity on LP-PRODUCED inputs -- the stand-in for BASELINE config 5 (bab_mip.py --bab_gnn, "same
branching decisions"), which needs Gurobi + CIFAR-10 and cannot run here.

Every other parity test feeds `synth.py` batches (duals U(0,1).Bernoulli(0.3) >= 0, interval bounds, nothing decided).  Here
the scorer sees what the BaB loop of plnn/relu_conv_gnnkwthreshold.py:126-243 really hands it: Wong-Kolter bounds, LP duals of
either sign in Gurobi's convention (conv_kwinter_gen.py:529-554), LP primals, masks with decided 0 / 1 nodes that grow along
the run, bounds clamped by earlier splits.  `lp_producer.branch_and_bound` (HiGHS producer) runs TWICE on the
same verification problem:

  run A: the HIP scorer through `GraphChoice.decision`, called with the reference driver's argument form (device-resident
         layer modules, CPU bounds / duals, python-list primals);
  run B: `oracle.gnn_oracle` (the CPU restatement pinned by the reference's own outputs) as the scorer -- test side only.

Asserted: identical `[layer, idx]` traces -- a mismatch is tolerated only at a node where the oracle's top-2 score gap is below
1e-3 (a near tie the 1e-4 parity budget cannot order; it is logged, and the comparison of decisions ends there because the two
runs explore different trees afterwards) -- and max |HIP score - oracle score| <= 1e-4 at EVERY node run A visited.
The LP solves are cached by (parent mask, child mask), so run B costs only its oracle calls while the traces agree.
"""
import copy
import os
import time

import numpy as np
import pytest
import torch

from gnn_branching_amd import lp_producer, nets

pytestmark = pytest.mark.gpu

CKPT = os.path.join(os.path.dirname(__file__), "..", "models", "cifar_trained_gnn",
                    "best_snapshot_None_0_val_acc_0.826_loss_val_0.1036_epoch_57.pt")
MIN_NODES = 40


class CachedLP:
    """lp_producer.LayerGraphLP whose `solve` remembers its results: both runs pose the same LPs while their traces agree."""

    def __init__(self, lp):
        self._lp, self._cache, self.hits, self.solves, self.solve_s = lp, {}, 0, 0, 0.0

    def __getattr__(self, name):
        return getattr(self._lp, name)

    @staticmethod
    def _key(mask):
        return b"".join(m.numpy().astype(np.int8).tobytes() for m in mask)

    def solve(self, mask, parent=None, split_layer=None):
        key = (self._key(mask), None if parent is None else self._key(parent.mask), split_layer)
        if key in self._cache:
            self.hits += 1
            return self._cache[key]
        self.solves += 1
        t0 = time.perf_counter()
        sub = self._lp.solve(mask, parent=parent, split_layer=split_layer)
        self.solve_s += time.perf_counter() - t0
        self._cache[key] = sub
        return sub


def mask_1d(sub):
    return torch.cat([(m == -1).float().reshape(-1) for m in sub.mask]).unsqueeze(0)


def run_trace(name, gt, cls, seed, eps, max_nodes):
    from gnn_branching_amd.graphnet.graph_score import GraphChoice
    from oracle import gnn_oracle
    from tests.common import shipped_state
    layers = nets.load_verified_net(name, gt, cls)
    rng = np.random.RandomState(seed)
    x = torch.from_numpy(rng.standard_normal((3, 32, 32)).astype(np.float32))
    lp = CachedLP(lp_producer.LayerGraphLP(layers, x - eps, x + eps))
    sizes = [int(np.prod(lp.shapes[i + 1])) for i in lp.pre_relu_indices]
    state = shipped_state()
    # the reference driver's `layers` dict: deep copies moved to the GPU (relu_conv_gnnkwthreshold.py:111-113)
    dev_layers = {"fixed_layers": [copy.deepcopy(l).cuda() for l in layers[:-1]], "prop_layers": [copy.deepcopy(layers[-1]).cuda()]}
    host_layers = {"fixed_layers": list(layers[:-1]), "prop_layers": [layers[-1]]}

    oracle_cache = {}

    def oracle_scores(sub):
        if id(sub) not in oracle_cache:
            lbg, ubg = sub.graph_bounds(lp.pre_relu_indices, len(lp.layers))
            with torch.no_grad():
                s = gnn_oracle.oracle_forward(state, lbg, ubg, sub.dual_vars, sub.primals, sub.ub_point, host_layers, mask_1d(sub))[0]
            oracle_cache[id(sub)] = (sub, s)                      # (keeps `sub` alive: ids stay unique)
        return oracle_cache[id(sub)][1]

    # ---- run A: the HIP scorer, the reference's call form
    root_mask = [torch.full((n,), -1, dtype=torch.long) for n in sizes]
    graph = GraphChoice(root_mask, CKPT)
    graph.verbose = False
    eng = graph.model.engine()
    trace_a, nodes_a = [], []

    def hip_scorer(sub, _layers):
        lbg, ubg = sub.graph_bounds(lp.pre_relu_indices, len(lp.layers))
        dec = graph.decision(lbg, ubg, sub.dual_vars, sub.ub_point, sub.primals, dev_layers, sub.mask)
        d2, scores = eng.forward_host(lbg, ubg, sub.dual_vars, sub.primals, sub.ub_point, dev_layers, mask_1d(sub), want_scores=True)
        assert d2[0].tolist() == dec
        trace_a.append(dec)
        nodes_a.append((sub, scores[0].copy()))
        return dec
    lines_a, dump_a = [], []
    res_a = lp_producer.branch_and_bound(lp, hip_scorer, layers, max_nodes=max_nodes, log=lines_a.append, dump=dump_a.append)

    # ---- run B: the oracle as the scorer
    trace_b, gaps_b = [], []

    def oracle_scorer(sub, _layers):
        s = oracle_scores(sub)
        top = torch.sort(s, descending=True)[0]
        gaps_b.append(float(top[0] - top[1]) if len(top) > 1 else float("inf"))
        dec = gnn_oracle.decision_from_scores(s, mask_1d(sub)[0], sizes)
        trace_b.append(dec)
        return dec
    lines_b = []
    res_b = lp_producer.branch_and_bound(lp, oracle_scorer, layers, max_nodes=max_nodes, log=lines_b.append)

    # ---- scores at every node of run A against the oracle on the same LP output
    worst, worst_node, stats = 0.0, -1, []
    for i, (sub, got) in enumerate(nodes_a):
        want = oracle_scores(sub)
        m = mask_1d(sub)[0].numpy() != 0
        assert np.isinf(got[~m]).all() and np.isfinite(got[m]).all()
        err = float(np.abs(got[m] - want.numpy()).max())
        if err > worst:
            worst, worst_node = err, i
        d = torch.cat([t[:, 1:].reshape(-1) for t in sub.dual_vars])
        stats.append((int(m.sum()), sum(int((t == 0).sum()) + int((t == 1).sum()) for t in sub.mask), float(d.min()), float(d.max())))
    return dict(trace_a=trace_a, trace_b=trace_b, gaps_b=gaps_b, worst=worst, worst_node=worst_node, stats=stats, res_a=res_a, res_b=res_b,
                lines_a=lines_a, lines_b=lines_b, lp=lp, dump_a=dump_a)


# (network, property gt vs cls, input seed, eps, branching decisions asked for): round 4 adds cifar_wide_kw (fewer nodes: its LPs are 4x
# base's) and a second property / input on base
# Default `-m gpu` run (the driver's step has a time limit; the LP solves on the host are what these tests cost): base 16 / deep 8 / wide 3
# decisions.  The round-4 lengths (40 / 40 / 24 / 12 decisions, 330 s) carry the `slow` marker: GNNB_RUN_SLOW=1 python -m pytest -m "gpu and slow".
SLOW = pytest.mark.slow
@pytest.mark.parametrize("name,gt,cls,seed,eps,min_nodes", [
    ("cifar_base_kw", 3, 5, 4, 0.02, 16), ("cifar_deep_kw", 3, 5, 4, 0.02, 8), ("cifar_wide_kw", 3, 5, 4, 0.02, 3),
    pytest.param("cifar_base_kw", 3, 5, 4, 0.02, MIN_NODES, marks=SLOW), pytest.param("cifar_deep_kw", 3, 5, 4, 0.02, MIN_NODES, marks=SLOW),
    pytest.param("cifar_base_kw", 7, 2, 11, 0.02, 24, marks=SLOW), pytest.param("cifar_wide_kw", 3, 5, 4, 0.02, 12, marks=SLOW)])
def test_decision_trace_on_lp_inputs(name, gt, cls, seed, eps, min_nodes):
    r = run_trace(name, gt, cls, seed, eps, max_nodes=2 * min_nodes + 4)
    # the run's trace in the reference's dump format (relu_conv_gnnkwthreshold.py:75-79, :201-202, :256-257), for a Gurobi owner to diff
    out_dir = os.path.join(os.path.dirname(__file__), "..", "gpurun_out", "bab_traces")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"{name}_gt{gt}_cls{cls}_seed{seed}_eps{eps}_n{min_nodes}.trace"), "w") as f:
        f.writelines(r["dump_a"])
    ta, tb = r["trace_a"], r["trace_b"]
    n = min(len(ta), len(tb))
    first_diff = next((i for i in range(n) if ta[i] != tb[i]), None)
    amb = [s[0] for s in r["stats"]]
    decided = [s[1] for s in r["stats"]]
    dmin, dmax = min(s[2] for s in r["stats"]), max(s[3] for s in r["stats"])
    print(f"\n{name}: {len(ta)} branching decisions on LP inputs (HIP run: bounds {r['res_a'][0]:.5f} / {r['res_a'][1]:.5f}, {r['res_a'][2]} LPs; "
          f"LP solves {r['lp'].solves}, cache hits {r['lp'].hits}); undecided ReLUs per node {min(amb)}..{max(amb)}, decided (0/1) {min(decided)}..{max(decided)}; "
          f"duals in [{dmin:.4g}, {dmax:.4g}]; max |HIP score - oracle| over all nodes = {r['worst']:.3e} (node {r['worst_node']}); "
          f"smallest oracle top-2 gap {min(r['gaps_b']):.3e}; LP 'infeasible' answers re-solved with slack: {r['lp'].retry_stats}; traces " + ("identical" if first_diff is None and len(ta) == len(tb) else f"differ at node {first_diff}"))
    assert len(ta) >= min_nodes, f"the run ended after {len(ta)} decisions: pick a harder property"
    assert dmin < 0 < dmax, "LP duals of both signs are the point of this test"
    assert max(decided) > min(decided), "masks must gain decided nodes along the run"
    assert r["worst"] <= 1e-4, (r["worst"], r["worst_node"])
    if first_diff is not None:
        gap = r["gaps_b"][first_diff]
        print(f"{name}: decision {first_diff} differs (HIP {ta[first_diff]}, oracle {tb[first_diff]}), oracle top-2 gap {gap:.3e}")
        assert gap < 1e-3, f"decision {first_diff}: HIP {ta[first_diff]} vs oracle {tb[first_diff]} with a top-2 gap of {gap:.3e}"
    else:
        assert len(ta) == len(tb) and r["lines_a"] == r["lines_b"]           # same tree, same bounds, line by line
        assert r["res_a"] == r["res_b"]


# ---- the reference loop's OWN control flow (round 5): GNN decision -> improvement below the branching threshold -> BaBSR decision -> two more
# LPs -> keep the better pair (plnn/relu_conv_gnnkwthreshold.py:150-199), with both scorers on the device against an oracle twin ------------
def record_wallclock(key, r, extra=None):
    """BASELINE config 5's other half ("wall-clock vs reference CPU GNN path"), for the stand-in: what the loop's scorer calls cost on the
    MI355X path (GraphChoice.decision: host tensors in, decision out, synchronous -- the reference's own call form) and on the CPU twin of
    the reference path (oracle, ONE thread as scripts/bab_mip.sh deploys it), over the same LP inputs; and what the LPs cost (HiGHS here,
    Gurobi in the reference).  Collected by tests/margins.py into profiles/r06_bab_loop_wallclock.json."""
    from tests import margins
    c, res = r["clock"], r["res_a"]
    rec = {"branches": int(res[3]), "lp_solves_posed": int(res[2]), "branches_that_bounded_a_kw_decision": int(res[4]), "branches_that_kept_it": int(res[5]),
           "gnn_decisions": int(c["hip_gnn_calls"]), "gnn_ms_total_hip": round(1e3 * c["hip_gnn_s"], 3), "gnn_ms_total_cpu_1_thread": round(1e3 * c["cpu_gnn_s"], 3),
           # the first call of a run builds the handle, binds the network and allocates (the reference pays its model load + .cuda() there,
           # graph_score.py:9-13): reported apart from the steady calls
           "gnn_ms_first_decision_hip": round(1e3 * c["hip_gnn_each"][0], 3) if c["hip_gnn_each"] else None,
           "gnn_ms_per_decision_hip_median_after_the_first": round(1e3 * float(np.median(c["hip_gnn_each"][1:])), 4) if len(c["hip_gnn_each"]) > 1 else None,
           "gnn_ms_per_decision_cpu_1_thread_median": round(1e3 * float(np.median(c["cpu_gnn_each"])), 3) if c["cpu_gnn_each"] else None,
           "kw_decisions": int(c["hip_kw_calls"]), "kw_ms_total_hip": round(1e3 * c["hip_kw_s"], 3), "kw_ms_total_cpu_1_thread": round(1e3 * c["cpu_kw_s"], 3),
           "lp_ms_total_highs": round(1e3 * r["lp"].solve_s, 1), "lp_solves_run": int(r["lp"].solves),
           "trace_lines_equal_hip_vs_cpu_twin": bool(r["lines_a"] == r["lines_b"]), "final_bounds": [float(res[0]), float(res[1])]}
    rec.update(extra or {})
    margins.record("bab_loop_wallclock", key, **rec)
    return rec


def run_threshold_trace(name, gt, cls, seed, eps, max_branches, branching_threshold=0.2):
    from gnn_branching_amd.graphnet.graph_score import GraphChoice
    from gnn_branching_amd.plnn import kw_score_conv as kw
    from oracle import babsr_oracle, gnn_oracle
    from tests.common import shipped_state
    layers = nets.load_verified_net(name, gt, cls)
    rng = np.random.RandomState(seed)
    x = torch.from_numpy(rng.standard_normal((3, 32, 32)).astype(np.float32))
    lp = CachedLP(lp_producer.LayerGraphLP(layers, x - eps, x + eps))
    sizes = [int(np.prod(lp.shapes[i + 1])) for i in lp.pre_relu_indices]
    state = shipped_state()
    dev_layers = {"fixed_layers": [copy.deepcopy(l).cuda() for l in layers[:-1]], "prop_layers": [copy.deepcopy(layers[-1]).cuda()]}
    host_layers = {"fixed_layers": list(layers[:-1]), "prop_layers": [layers[-1]]}
    graph = GraphChoice([torch.full((n,), -1, dtype=torch.long) for n in sizes], CKPT)
    graph.verbose = False

    clock = {"hip_gnn_s": 0.0, "hip_gnn_calls": 0, "hip_kw_s": 0.0, "hip_kw_calls": 0, "cpu_gnn_s": 0.0, "cpu_gnn_calls": 0, "cpu_kw_s": 0.0, "cpu_kw_calls": 0,
             "hip_gnn_each": [], "cpu_gnn_each": []}

    # ---- run A: both scorers on the MI355X, called the way the reference driver calls them
    def hip_gnn(sub, _layers):
        lbg, ubg = sub.graph_bounds(lp.pre_relu_indices, len(lp.layers))
        t0 = time.perf_counter()
        dec = graph.decision(lbg, ubg, sub.dual_vars, sub.ub_point, sub.primals, dev_layers, sub.mask)      # (synchronous: host tensors in, two ints out)
        clock["hip_gnn_each"].append(time.perf_counter() - t0)
        clock["hip_gnn_s"] += clock["hip_gnn_each"][-1]
        clock["hip_gnn_calls"] += 1
        return dec

    def hip_kw(sub, icp, random_order, sparsest_layer):
        t0 = time.perf_counter()
        out = kw.choose_node_conv(sub.lower_all, sub.upper_all, sub.mask, lp.layers, lp.pre_relu_indices, icp, random_order, sparsest_layer)
        clock["hip_kw_s"] += time.perf_counter() - t0
        clock["hip_kw_calls"] += 1
        return out
    lines_a, dump_a = [], []
    res_a = lp_producer.branch_and_bound_threshold(lp, hip_gnn, hip_kw, layers, max_branches=max_branches, branching_threshold=branching_threshold,
                                                   decision_bound=0.0, log=lines_a.append, dump=dump_a.append)

    # ---- run B: the oracle twin (gnn_oracle + babsr_oracle, both pinned by the reference's own outputs) -- test side only
    gaps = []

    def oracle_gnn(sub, _layers):
        lbg, ubg = sub.graph_bounds(lp.pre_relu_indices, len(lp.layers))
        nthr = torch.get_num_threads()
        torch.set_num_threads(1)                  # the reference deploys its BaB on ONE core (scripts/bab_mip.sh:3-5: taskset -c <core>)
        t0 = time.perf_counter()
        with torch.no_grad():
            s = gnn_oracle.oracle_forward(state, lbg, ubg, sub.dual_vars, sub.primals, sub.ub_point, host_layers, mask_1d(sub))[0]
        clock["cpu_gnn_each"].append(time.perf_counter() - t0)
        clock["cpu_gnn_s"] += clock["cpu_gnn_each"][-1]
        clock["cpu_gnn_calls"] += 1
        torch.set_num_threads(nthr)
        top = torch.sort(s, descending=True)[0]
        gaps.append(float(top[0] - top[1]) if len(top) > 1 else float("inf"))
        return gnn_oracle.decision_from_scores(s, mask_1d(sub)[0], sizes)

    def oracle_kw(sub, icp, random_order, sparsest_layer):
        lbs = [sub.lower_all[i].unsqueeze(0) for i in lp.pre_relu_indices]
        ubs = [sub.upper_all[i].unsqueeze(0) for i in lp.pre_relu_indices]
        masks = [(m == -1).float().reshape(1, -1) for m in sub.mask]
        nthr = torch.get_num_threads()
        torch.set_num_threads(1)
        t0 = time.perf_counter()
        with torch.no_grad():
            score, icpt = babsr_oracle.babsr_scores(lbs, ubs, masks, list(layers[:-1]), layers[-1].weight.detach().reshape(1, -1))
        clock["cpu_kw_s"] += time.perf_counter() - t0
        clock["cpu_kw_calls"] += 1
        torch.set_num_threads(nthr)
        return babsr_oracle.decide([t[0] for t in score], [t[0] for t in icpt], [m[0] for m in masks], icp, random_order, sparsest_layer)
    lines_b = []
    res_b = lp_producer.branch_and_bound_threshold(lp, oracle_gnn, oracle_kw, layers, max_branches=max_branches, branching_threshold=branching_threshold,
                                                   decision_bound=0.0, log=lines_b.append)
    return dict(lines_a=lines_a, lines_b=lines_b, dump_a=dump_a, res_a=res_a, res_b=res_b, gaps=gaps, lp=lp, clock=clock)


# eps chosen so that the ROOT is undecided (lower bound < 0 < upper bound: at eps = 0.02 these properties hold at the root and the reference loop would
# not branch at all): base 0.09 (root -0.19 / 0.05), deep 0.05 (root -0.035 / 0.007); decision_bound = 0 as the reference verifies (:257-262)
@pytest.mark.parametrize("name,gt,cls,seed,eps,branches", [("cifar_base_kw", 3, 5, 4, 0.09, 12), ("cifar_deep_kw", 3, 5, 4, 0.05, 6),
                                                            pytest.param("cifar_base_kw", 3, 5, 4, 0.09, 30, marks=pytest.mark.slow),
                                                            pytest.param("cifar_deep_kw", 3, 5, 4, 0.05, 16, marks=pytest.mark.slow)])
def test_threshold_loop_trace_with_the_kw_fallback(name, gt, cls, seed, eps, branches):
    """BASELINE config 5's stand-in with the reference loop's control flow: `gnn_improvement < branching_threshold` -> `choose_node_conv` on
    the device (gnnb_babsr) -> two more LPs -> keep the better pair (relu_conv_gnnkwthreshold.py:150-199).  The HIP run's trace LINES
    (branch count, decision kept, GNN improvement and decision, KW improvement and decision: the line of :201-202) equal the oracle
    twin's, and at least one branch bounded a KW decision."""
    r = run_threshold_trace(name, gt, cls, seed, eps, max_branches=branches)
    out_dir = os.path.join(os.path.dirname(__file__), "..", "gpurun_out", "bab_traces")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, f"{name}_gt{gt}_cls{cls}_seed{seed}_eps{eps}_threshold0.2_n{branches}.trace"), "w") as f:
        f.writelines(r["dump_a"])
    la, lb = r["lines_a"], r["lines_b"]
    n_kw = sum("kw: improvement -1 decision None" not in l for l in la)
    print("wall-clock of the loop's scorer calls:", record_wallclock(f"{name}_gt{gt}_cls{cls}_seed{seed}_eps{eps}_branches{branches}", r, {"input": "seeded N(0,1) image", "eps": eps}))
    print(f"\n{name}: {len(la)} branches with the KW fall-back (threshold 0.2): {r['res_a'][4]} bounded a KW decision, {r['res_a'][5]} kept it; "
          f"{r['res_a'][2]} LPs posed (LP solves {r['lp'].solves}, cache hits {r['lp'].hits}); bounds {r['res_a'][0]:.5f} / {r['res_a'][1]:.5f}; "
          f"smallest oracle top-2 GNN gap {min(r['gaps']):.3e}")
    for l in la:
        print("   ", l)
    assert len(la) >= min(branches, 4), f"the run ended after {len(la)} branches"
    assert n_kw >= 1 and r["res_a"][4] >= 1, "no branch fell below the branching threshold: nothing exercised the KW fall-back"
    first_diff = next((i for i in range(min(len(la), len(lb))) if la[i] != lb[i]), None)
    if first_diff is not None:
        # a different line is only acceptable where the oracle's GNN scores were a near tie the 1e-4 parity budget cannot order
        print(f"{name}: line {first_diff} differs\n  HIP    {la[first_diff]}\n  oracle {lb[first_diff]}")
        assert min(r["gaps"]) < 1e-3, (la[first_diff], lb[first_diff])
    else:
        assert la == lb and r["res_a"] == r["res_b"]            # same tree, same bounds, same counters, line by line


def _predicted_class(name, x):
    with torch.no_grad():
        z = x.unsqueeze(0)
        for l in nets.build_net(name):
            z = l(z)
    return int(z.argmax())


@pytest.mark.parametrize("row", [0, 1])
def test_threshold_loop_on_base_easy_rows_with_wallclock(row):
    """(Eps, prop) of rows of the reference's experiment table cifar_exp/base_easy.pkl (fixture tests/golden/base_easy_props.npz, made by
    oracle/make_golden_props.py) -- the properties `bab_mip.py --bab_gnn` runs (bab_mip.py:91-120) -- on a seeded stand-in image (CIFAR-10 is
    not available offline; ground truth := the class the network predicts for the stand-in): the reference loop's control flow, HIP scorers
    against the CPU twin, trace lines equal, and the wall-clock of the GNN calls on both paths on record beside what the reference's own run
    of that table row took (BBran_gnnkwT branches, BTime_gnnkwT seconds, with Gurobi and the real image: context, not a comparison)."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "base_easy_props.npz"))
    table_eps, prop = float(g["Eps"][row]), int(g["prop"][row])
    seed = 100 + row
    x = torch.from_numpy(np.random.RandomState(seed).standard_normal((3, 32, 32)).astype(np.float32))
    gt = _predicted_class("cifar_base_kw", x)
    cls = prop if prop != gt else (prop + 1) % 10
    # On the stand-in image the table's eps is already violated at the root (upper bound < 0: nothing to branch on), so eps is shrunk in
    # fixed steps until the root is undecided (lower bound < 0 < upper bound) -- the state the reference loop starts branching from
    layers = nets.load_verified_net("cifar_base_kw", gt, cls)
    eps = None
    for f in (1.0, 0.75, 0.55, 0.4, 0.3, 0.2):
        lp0 = lp_producer.LayerGraphLP(layers, x - f * table_eps, x + f * table_eps)
        root = lp0.solve([torch.full((int(np.prod(lp0.shapes[i + 1])),), -1, dtype=torch.long) for i in lp0.pre_relu_indices])
        if root is not None and root.lb < 0 < root.ub:
            eps = round(f * table_eps, 6)
            break
    assert eps is not None, "no eps step leaves the root undecided"
    branches = 5
    r = run_threshold_trace("cifar_base_kw", gt, cls, seed, eps, max_branches=branches)
    rec = record_wallclock(f"base_easy_row{row}_prop{prop}", r, {
        "input": "seeded N(0,1) stand-in for CIFAR-10 image %d" % int(g["Idx"][row]), "table_eps": table_eps, "eps": eps, "target_class": cls, "ground_truth_class": gt,
        "reference_recorded_for_this_row": {"BBran_gnnkwT": float(g["BBran_gnnkwT"][row]), "BTime_gnnkwT_s": float(g["BTime_gnnkwT"][row]), "BSAT_gnnkwT": str(g["BSAT_gnnkwT"][row])}})
    print(f"\nbase_easy row {row} (eps {eps}, prop {prop}):", rec)
    la, lb = r["lines_a"], r["lines_b"]
    assert len(la) >= 1
    first_diff = next((i for i in range(min(len(la), len(lb))) if la[i] != lb[i]), None)
    if first_diff is not None:
        assert min(r["gaps"]) < 1e-3, (la[first_diff], lb[first_diff])
    else:
        assert la == lb and r["res_a"] == r["res_b"]
