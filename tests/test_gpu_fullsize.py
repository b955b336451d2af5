"""-m gpu: properties at BASELINE.json's full sizes: batched == re-scored subsets bit for bit, permutation equivariance, decisions ==
first argmax of the scores, and 64+ samples of every configuration against the CPU oracle within the 1e-4 budget (graph_conv.py:442-470 is
per sample, so the oracle scores any subset of a batch on its own: ~20 ms per sample) -- the margins go on record (tests/margins.py)."""
import numpy as np
import pytest
import torch

from tests import margins
from tests.common import SCORE_ATOL, shipped_state

pytestmark = pytest.mark.gpu


def model_for(state):
    from gnn_branching_amd.graphnet.graph_conv import GraphNet
    m = GraphNet(2, 64)
    m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in state.items()})
    return m.eval()


def first_argmax_decisions(scores, sizes):
    cum = np.cumsum(sizes)
    out = []
    for row in scores:
        if not np.isfinite(row).any():
            out.append([-1, -1])
            continue
        j = int(np.argmax(row))                    # numpy: first maximal index
        lay = int(np.searchsorted(cum, j, side="right"))
        out.append([lay, j - (int(cum[lay - 1]) if lay else 0)])
    return out


# BASELINE.json configs 2, 3, 4: deep at its per-rank shard (128) and at the config's whole batch (1024) on one GPU
@pytest.mark.parametrize("net,B", [("cifar_base_kw", 256), ("cifar_wide_kw", 256), ("cifar_deep_kw", 128), ("cifar_deep_kw", 1024)])
def test_full_size_batch_properties(net, B):
    from gnn_branching_amd import synth
    from oracle import gnn_oracle
    state = shipped_state()
    model = model_for(state)
    batch = synth.make_batch(net, B, seed=4321, props=[(3, 5) if i % 3 else (1, 7) for i in range(B)])
    sizes = [int(np.prod(t.shape[1:])) for t in batch.lower_bounds_all[1:-1]]
    with torch.no_grad():
        res = model.forward_device(*batch.forward_args()).check()
        scores = res.scores.cpu()
        dec = res.decisions.cpu().tolist()
        # 1. decisions are the first argmax of the padded scores
        assert dec == first_argmax_decisions(scores.numpy(), sizes)
        # 2. the mask decides where scores exist
        assert torch.equal(torch.isfinite(scores), batch.masks != 0)
        # 3. a re-scored contiguous shard reproduces its rows bit for bit (what data-parallel sharding relies on)
        lo, hi = B // 3, B // 3 + 16
        part = model.forward_device(*batch.slice(lo, hi).forward_args()).check()
        assert torch.equal(part.scores.cpu(), scores[lo:hi])
        assert part.decisions.cpu().tolist() == dec[lo:hi]
        # 4. permuting the batch permutes the result
        perm = torch.randperm(B, generator=torch.Generator().manual_seed(0))
        pb = synth.SubproblemBatch(
            [t[perm] for t in batch.lower_bounds_all], [t[perm] for t in batch.upper_bounds_all],
            [t.view(B, -1, 3)[perm].reshape(-1, 3) for t in batch.dual_vars],
            [t.view(B, -1)[perm].reshape(-1) for t in batch.primals], batch.primal_inputs[perm],
            {"fixed_layers": batch.layers["fixed_layers"], "prop_layers": [batch.layers["prop_layers"][i] for i in perm.tolist()]},
            batch.masks[perm])
        pres = model.forward_device(*pb.forward_args()).check()
        assert torch.equal(pres.scores.cpu(), scores[perm])
        # 5. 64+ samples against the CPU oracle: a seeded spread, both ends, and the samples with the largest |score|, the most scored
        #    nodes and the closest decision (smallest top-1 / top-2 gap) -- where an error would show or matter first
        sc = scores.numpy()
        fin = np.isfinite(sc)
        absmax = np.where(fin, np.abs(sc), 0.0).max(1)
        n_scored = fin.sum(1)
        top2 = np.sort(np.where(fin, sc, -np.inf), 1)[:, -2:]
        gap = np.where(n_scored >= 2, top2[:, 1] - top2[:, 0], np.inf)
        pick = {0, B // 2, B - 1, int(absmax.argmax()), int(n_scored.argmax()), int(gap.argmin())}
        pick |= set(int(i) for i in np.random.RandomState(6).choice(B, size=min(B, 64), replace=False))
        pick = sorted(pick)
        worst, worst_b = 0.0, -1
        excused = []                                 # nodes where the REFERENCE arithmetic itself is further than the bar from its fp64 evaluation
        torch.set_num_threads(min(16, torch.get_num_threads()))
        for c in range(0, len(pick), 16):
            idx = pick[c:c + 16]
            it = torch.tensor(idx)
            sub = synth.SubproblemBatch(
                [t[it] for t in batch.lower_bounds_all], [t[it] for t in batch.upper_bounds_all],
                [t.view(B, -1, 3)[it].reshape(-1, 3) for t in batch.dual_vars],
                [t.view(B, -1)[it].reshape(-1) for t in batch.primals], batch.primal_inputs[it],
                {"fixed_layers": batch.layers["fixed_layers"], "prop_layers": [batch.layers["prop_layers"][i] for i in idx]},
                batch.masks[it])
            want = gnn_oracle.oracle_forward(state, *sub.forward_args())
            for b, w in zip(idx, want):
                got = scores[b][batch.masks[b] != 0]
                e32 = (got - w).abs()
                if w.numel() and e32.max().item() > SCORE_ATOL:
                    # Past the bar against the fp32 oracle.  Before calling it a failure, ask what the reference's own fp32 arithmetic is worth
                    # at those nodes: the same formulas in fp64 (oracle dtype=float64).  A node is excused only if the fp32 REFERENCE is itself
                    # further from the fp64 value than the HIP score is, and the HIP score is within the bar of the fp64 value -- i.e. the
                    # disagreement is the reference's rounding at an ill-conditioned node, not ours.  Everything else fails.
                    w64 = gnn_oracle.oracle_forward(state, *batch.slice(b, b + 1).forward_args(), dtype=torch.float64)[0]
                    e_hip64, e_ref64 = (got.double() - w64).abs(), (w.double() - w64).abs()
                    bad = e32 > SCORE_ATOL
                    ok = bad & (e_hip64 <= SCORE_ATOL) & (e_ref64 > e_hip64)
                    for i in torch.nonzero(bad).reshape(-1).tolist():
                        excused.append({"sample": int(b), "node": int(i), "score": float(w64[i]), "hip_minus_ref_fp32": float(e32[i]), "hip_minus_fp64": float(e_hip64[i]),
                                        "ref_fp32_minus_fp64": float(e_ref64[i]), "excused": bool(ok[i])})
                    assert bool((ok == bad).all()), (net, B, b, excused[-3:])
                    e32 = torch.where(bad, torch.zeros_like(e32), e32)
                err = e32.max().item() if w.numel() else 0.0
                if err > worst:
                    worst, worst_b = err, b
        n_nodes = int(sum(int((batch.masks[b] != 0).sum()) for b in pick))
        margins.record("full_size_vs_oracle", f"{net}_B{B}", n_oracle_samples=len(pick), n_scored_nodes_checked=n_nodes, worst_abs_err=worst, worst_sample=worst_b, bar=SCORE_ATOL,
                       max_abs_score=float(absmax.max()), min_top2_gap=float(gap.min()), min_top2_gap_among_checked=float(gap[pick].min()),
                       max_scored_nodes=int(n_scored.max()), weights="shipped", nodes_where_the_fp32_reference_is_off_its_fp64_value=excused)
        print(f"{net} B={B}: {len(pick)} samples ({n_nodes} scored nodes) vs oracle, worst |score - oracle| {worst:.3e} (sample {worst_b}; bar {SCORE_ATOL:g}), max |score| {absmax.max():.4g}, "
              f"min top-2 gap {gap.min():.3e}; nodes past the bar where the fp32 reference itself is off its fp64 value: {excused}")
        assert worst <= SCORE_ATOL, (net, B, worst, worst_b)
        assert len(excused) <= max(1, n_nodes // 5000), excused      # an exception, not a class: at most one node in 5000


@pytest.mark.parametrize("net,B", [("cifar_base_kw", 256), ("cifar_wide_kw", 64), ("cifar_deep_kw", 128), ("cifar_base_kw", 3)])
def test_fused_halfpass_kernel_at_size(monkeypatch, net, B):
    """The default (k_gather_update_q wherever it exists) against GNNB_FUSE=0 (two kernels) at batch sizes where every workgroup
    runs many rounds of tiles and the LDS row queue wraps hundreds of times: identical scores and decisions, bit for bit."""
    from gnn_branching_amd import synth
    from oracle.gnn_oracle import random_gnn_state
    state = random_gnn_state(20240917)
    batch = synth.make_batch(net, B, seed=99)
    out = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("GNNB_FUSE", fuse)
        model = model_for(state)
        with torch.no_grad():
            res = model.forward_device(*batch.forward_args()).check()
            out[fuse] = (res.scores.cpu(), res.decisions.cpu().tolist())
    assert torch.equal(out["0"][0], out["1"][0])
    assert out["0"][1] == out["1"][1]


@pytest.mark.parametrize("net,B", [("cifar_base_kw", 256), ("cifar_wide_kw", 256), ("cifar_deep_kw", 128)])
@pytest.mark.parametrize("weights", ["shipped", "random"])
def test_top_kernel_at_size(monkeypatch, net, B, weights):
    """k_top (the last Linear edge both ways on the bf16x3 matrix rate, live-row lists in LDS, both node updates and the property
    node, one workgroup per sample) against GNNB_NO_TOP=1 (separate kernels, fp32 MFMA edges, every row) at the bench batch
    sizes: two independent GPU implementations of the same sums must agree inside the parity bar on every score, and pick the
    same branching decision wherever the best score is not a near-tie."""
    from gnn_branching_amd import synth
    from oracle.gnn_oracle import random_gnn_state
    from tests.common import RANDOM_ATOL, shipped_state
    state = shipped_state() if weights == "shipped" else random_gnn_state(20240917)
    batch = synth.make_batch(net, B, seed=7)
    out = {}
    for no_top in ("1", "0"):
        monkeypatch.setenv("GNNB_NO_TOP", no_top)
        model = model_for(state)
        with torch.no_grad():
            res = model.forward_device(*batch.forward_args()).check()
            out[no_top] = (res.scores.cpu(), res.decisions.cpu())
    a, b = out["0"][0], out["1"][0]
    assert torch.equal(torch.isinf(a), torch.isinf(b))
    fin = torch.isfinite(a)
    tol = SCORE_ATOL if weights == "shipped" else RANDOM_ATOL
    err = (a[fin] - b[fin]).abs().max().item()
    print(f"{net} B={B} {weights}: max|k_top - separate kernels| = {err:.3e} (bar {tol:.1e})")
    assert err <= tol
    differ = (out["0"][1] != out["1"][1]).any(dim=1)
    for s in torch.nonzero(differ).flatten().tolist():        # a different argmax is only acceptable between scores closer than the bar
        row_a, row_b = a[s], b[s]
        assert abs(row_a[torch.isfinite(row_a)].max().item() - row_b[torch.isfinite(row_b)].max().item()) <= tol


# The three-piece bf16 blocks against the exact-fp32 MFMA path AT BENCH SIZES (round 3 quoted these figures in DESIGN.md; round 4 asserts
# them).  Both are fp32-grade evaluations that add in different orders; the bound is what two such evaluations differ by on the shipped
# checkpoint: scores of magnitude 5..50 on base / wide (measured 1.7-1.9e-5), up to 70 on deep (measured 5.1e-5) -- always inside the
# 1e-4 budget, with the same decisions wherever the exact path's top-2 gap exceeds the delta.
@pytest.mark.parametrize("net,B,bound", [("cifar_base_kw", 256, 3e-5), ("cifar_wide_kw", 256, 3e-5), ("cifar_deep_kw", 128, 7e-5)])
def test_bf16x3_delta_at_bench_size(monkeypatch, net, B, bound):
    from gnn_branching_amd import synth
    state = shipped_state()
    batch = synth.make_batch(net, B, seed=1234)
    out = {}
    for bf3 in ("1", "0"):
        monkeypatch.setenv("GNNB_BF3", bf3)
        model = model_for(state)                     # a new engine: the knob is read by gnnb_create
        with torch.no_grad():
            res = model.forward_device(*batch.forward_args()).check()
        out[bf3] = (res.scores.cpu().numpy(), res.decisions.cpu().numpy())
    a, b = out["1"][0], out["0"][0]
    fin = np.isfinite(b)
    assert np.array_equal(fin, np.isfinite(a))
    delta = float(np.abs(a[fin] - b[fin]).max())
    top2 = np.sort(np.where(fin, b, -np.inf), axis=1)[:, -2:]
    differ = np.nonzero((out["1"][1] != out["0"][1]).any(axis=1))[0]
    print(f"{net} B={B}: max |bf16x3 - exact fp32| = {delta:.3e} (bound {bound:.0e}, max |score| {np.abs(b[fin]).max():.1f}); "
          f"{len(differ)} of {B} decisions differ")
    assert delta <= bound
    for i in differ:                                 # a decision may only flip at a near tie of the exact path
        assert top2[i, 1] - top2[i, 0] <= 2 * delta, (i, top2[i])
