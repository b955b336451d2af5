"""-m gpu: the HIP path (through the C-ABI) against the golden vectors of the reference and the CPU oracle."""
import numpy as np
import pytest
import torch

from tests.common import FAMILIES, GOLDEN_CASES, SCORE_ATOL, STAGES, load_golden, relu_sizes, score_tol, state_of

pytestmark = pytest.mark.gpu


def make_model(fam):
    from gnn_branching_amd.graphnet.graph_conv import GraphNet
    m = GraphNet(2, 64)
    m.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in state_of(fam).items()})
    return m.eval()


@pytest.mark.parametrize("case", GOLDEN_CASES)
@pytest.mark.parametrize("fam", FAMILIES)
def test_scores_and_decisions_match_reference(case, fam):
    g, batch = load_golden(case)
    model = make_model(fam)
    with torch.no_grad():
        res = model.forward_device(*batch.forward_args()).check()
    got = res.scores.cpu().numpy()
    want = g[f"{fam}_scores"]
    assert np.array_equal(np.isinf(got), np.isinf(want))
    fin = np.isfinite(want)
    err = np.abs(got[fin] - want[fin]).max()
    tol = score_tol(fam, want[fin])
    print(f"{case} {fam}: max|score - reference| = {err:.3e} (bar {tol:.1e}, score range [{want[fin].min():.3g}, {want[fin].max():.3g}])")
    from tests import margins
    margins.record("golden_reference_scores", f"{case}_{fam}", worst_abs_err=float(err), bar=float(tol), max_abs_score=float(np.abs(want[fin]).max()),
                   decisions_equal=bool(res.decisions.cpu().tolist() == g[f"{fam}_decisions"].tolist()))
    assert err <= tol
    assert res.decisions.cpu().tolist() == g[f"{fam}_decisions"].tolist()
    # ragged list, as the reference returns it
    with torch.no_grad():
        rag = model(*batch.forward_args())
    for b, s in enumerate(rag):
        np.testing.assert_allclose(s.cpu().numpy(), want[b][fin[b]], atol=tol, rtol=0)


@pytest.mark.parametrize("case", GOLDEN_CASES)
@pytest.mark.parametrize("fam", FAMILIES)
def test_embeddings_after_every_halfpass(case, fam):
    """mu after r0_fwd, r0_bwd, r1_fwd, r1_bwd against rows sampled from the reference's embeddings."""
    g, batch = load_golden(case)
    model = make_model(fam)
    eng = model.engine()
    stride = int(g["sample_stride"])
    B = batch.batch_size
    try:
        for n, st in enumerate(STAGES, 1):
            eng.set_halfpass_limit(n)
            with torch.no_grad():
                model.forward_device(*batch.forward_args()).check()
            for k in range(len(batch.lower_bounds_all)):
                rows = g[f"{fam}_{st}_mu{k}_rows"]
                got = eng.mu(B, k)[:, ::stride, :].cpu().numpy()
                scale = max(1.0, float(np.abs(rows).max()))
                err = np.abs(got - rows).max()
                assert err <= 2e-5 * scale, (st, k, err)
                want_abs = g[f"{fam}_{st}_mu{k}_sum"][1]
                got_abs = eng.mu(B, k).double().abs().sum().item()
                assert abs(got_abs - want_abs) <= 1e-4 * max(want_abs, 1.0), (st, k, got_abs, want_abs)
    finally:
        eng.set_halfpass_limit(0)


@pytest.mark.parametrize("case", GOLDEN_CASES)
@pytest.mark.parametrize("fam", FAMILIES)
def test_default_path_embeddings_after_the_last_sweep(case, fam):
    """The DEFAULT path (fused top of the network, embedding inside the first gather, restricted last step -- no half-pass
    limit, no knob): after a full forward the rows the score head consumed are still in the workspace -- mu[2..L] as the last
    backward sweep left them, mu[1] at the scored nodes -- and must match the reference's embeddings after r1_bwd.  Rows hold
    the embedding before its producer's last Linear; the projection is applied on the host in float64 (numpy)."""
    g, batch = load_golden(case)
    model = make_model(fam)
    eng = model.engine()
    stride = int(g["sample_stride"])
    B = batch.batch_size
    with torch.no_grad():
        model.forward_device(*batch.forward_args()).check()           # (binds the network, allocates the workspace)
        eng.workspace(B).view(torch.float32).fill_(float("nan"))      # what the forward does not write stays NaN
        model.forward_device(*batch.forward_args()).check()
    L = len(batch.lower_bounds_all) - 2
    sizes = relu_sizes(batch)
    masks = batch.masks.numpy()
    off, checked = 0, 0
    for k in range(1, L + 1):
        want = g[f"{fam}_r1_bwd_mu{k}_rows"]                          # (B, ceil(N_k / stride), 64) rows of the reference
        rows, lid = eng.mu_rows(B, k)
        assert lid >= 0, "the default path defers every producer's last Linear"
        rows = rows[:, ::stride, :].cpu().numpy().astype(np.float64)
        lb = batch.lower_bounds_all[k].reshape(B, -1)[:, ::stride].numpy()
        ub = batch.upper_bounds_all[k].reshape(B, -1)[:, ::stride].numpy()
        live = ~((np.maximum(ub, 0) == 0) & ((lb - np.maximum(lb, 0)) != 0))       # [r0 != 0] without the division
        scored = masks[:, off:off + sizes[k - 1]][:, ::stride] != 0
        off += sizes[k - 1]
        sel = scored if k == 1 else np.ones_like(live)                # layer 1: the last step only wrote the scored nodes
        W, b = eng.linear_host(lid)
        got = (np.where(live[..., None], rows, 0.0) @ W.T + b) * live[..., None]
        assert np.isfinite(got[sel & live]).all(), (k, "a live row the score head needs was not written")
        scale = max(1.0, float(np.abs(want).max()))
        err = np.abs(got - want)[sel].max() if sel.any() else 0.0
        assert err <= 2e-5 * scale, (k, err)
        checked += int(sel.sum())
    assert checked > 0


def test_graphchoice_decision_surface():
    """The reference's B=1 call surface: python-list primals, {-1,0,1} masks, CPU tensors in."""
    import os
    from gnn_branching_amd.graphnet.graph_score import GraphChoice
    g, batch = load_golden("cifar_base_kw_B3")
    ckpt = os.path.join(os.path.dirname(__file__), "..", "models", "cifar_trained_gnn",
                        "best_snapshot_None_0_val_acc_0.826_loss_val_0.1036_epoch_57.pt")
    for b in range(batch.batch_size):
        one = batch.slice(b, b + 1)
        init_mask = [m[0] for m in one.bab_masks]
        gc = GraphChoice(init_mask, ckpt)
        lb_before = [t.clone() for t in one.lower_bounds_all]
        dec = gc.decision(one.lower_bounds_all, one.upper_bounds_all, one.dual_vars, one.primal_inputs,
                          [p.tolist() for p in one.primals], one.layers, init_mask)
        assert dec == g["shipped_decisions"][b].tolist()
        assert all(isinstance(v, int) for v in dec)
        for t0, t1 in zip(lb_before, one.lower_bounds_all):      # caller's tensors untouched
            assert torch.equal(t0, t1)


def test_batched_equals_single_and_permutation():
    """Subproblems are independent: batched == per-sample, and permuting the batch permutes the result."""
    from gnn_branching_amd import synth
    model = make_model("random")
    batch = synth.make_batch("cifar_base_kw", 5, seed=11, props=[(3, 5), (1, 2), (3, 5), (0, 9), (4, 4 - 1)])
    with torch.no_grad():
        full = model.forward_device(*batch.forward_args()).check()
        s_full = full.scores.cpu()
        for b in range(5):
            one = model.forward_device(*batch.slice(b, b + 1).forward_args()).check()
            assert torch.equal(one.scores.cpu()[0], s_full[b])        # same kernels, same order: bit-exact
            assert one.decisions.cpu()[0].tolist() == full.decisions.cpu()[b].tolist()


def test_masks_all_and_none():
    """Edge cases: nothing undecided -> all -inf and decision [-1,-1]; everything marked undecided."""
    from gnn_branching_amd import synth
    from oracle import gnn_oracle
    model = make_model("random")
    batch = synth.make_batch("cifar_base_kw", 2, seed=3)
    args = list(batch.forward_args())
    args[6] = torch.zeros_like(batch.masks)
    with torch.no_grad():
        res = model.forward_device(*args).check()
    assert torch.isinf(res.scores).all() and res.decisions.cpu().tolist() == [[-1, -1], [-1, -1]]
    assert [int(s.numel()) for s in res.ragged()] == [0, 0]
    args[6] = torch.ones_like(batch.masks)
    with torch.no_grad():
        res = model.forward_device(*args).check()
        want = gnn_oracle.oracle_forward(state_of("random"), *args)
    got = res.scores.cpu()
    for b in range(2):
        np.testing.assert_allclose(got[b].numpy(), want[b].numpy(), atol=score_tol("random", want[b].numpy()), rtol=0)


def test_nan_is_reported():
    """lb = ub = 0 makes compute_ratio 0/0 (graph_conv.py:502); the reference drops into pdb, we raise."""
    from gnn_branching_amd import synth
    model = make_model("random")
    batch = synth.make_batch("cifar_base_kw", 1, seed=5)
    batch.lower_bounds_all[1].view(-1)[7] = 0.0
    batch.upper_bounds_all[1].view(-1)[7] = 0.0
    with torch.no_grad(), pytest.raises(FloatingPointError):
        model(*batch.forward_args())


def test_bad_inputs_raise():
    from gnn_branching_amd import synth
    model = make_model("random")
    batch = synth.make_batch("cifar_base_kw", 2, seed=5)
    args = list(batch.forward_args())
    args[2] = args[2][:-1]                      # one dual tensor short
    with pytest.raises((ValueError, RuntimeError)):
        model.forward_device(*args)
    args = list(batch.forward_args())
    args[6] = args[6][:, :-1]                   # ragged mask
    with pytest.raises(ValueError):
        model.forward_device(*args)


def test_unfused_valu_gather_path_matches(monkeypatch):
    """GNNB_NO_GATHER=1 selects the unfused VALU conv gathers + flat node update: same scores."""
    monkeypatch.setenv("GNNB_NO_GATHER", "1")
    g, batch = load_golden("cifar_deep_kw_B2")
    model = make_model("random")
    with torch.no_grad():
        res = model.forward_device(*batch.forward_args()).check()
    want = g["random_scores"]
    fin = np.isfinite(want)
    assert np.abs(res.scores.cpu().numpy()[fin] - want[fin]).max() <= score_tol("random", want[fin])
    assert res.decisions.cpu().tolist() == g["random_decisions"].tolist()


@pytest.mark.parametrize("case", GOLDEN_CASES)
@pytest.mark.parametrize("fam", FAMILIES)
def test_fused_halfpass_kernel_matches(monkeypatch, case, fam):
    """Default (GNNB_FUSE=1): every conv half-pass that has a fused kernel runs as ONE kernel, k_gather_update_q (gather waves,
    chain waves and an LDS row queue between them: the aggregate never reaches HBM); GNNB_FUSE=0: always k_gather +
    k_node_update.  Both compute the same arithmetic per node, so the scores must be IDENTICAL (which form runs is a pure
    scheduling choice) -- and inside the parity bar, with the reference's decisions."""
    g, batch = load_golden(case)
    want = g[f"{fam}_scores"]
    fin = np.isfinite(want)
    out = {}
    for fuse in ("0", "1"):
        monkeypatch.setenv("GNNB_FUSE", fuse)
        model = make_model(fam)                      # a new engine: the knob is read by gnnb_create
        with torch.no_grad():
            res = model.forward_device(*batch.forward_args()).check()
        out[fuse] = res.scores.cpu().numpy()
        assert np.abs(out[fuse][fin] - want[fin]).max() <= score_tol(fam, want[fin])
        assert res.decisions.cpu().tolist() == g[f"{fam}_decisions"].tolist()
        if fuse == "1":                              # nothing may depend on scratch the call did not write itself
            model.engine().workspace(batch.batch_size).view(torch.float32).fill_(float("nan"))
            again = model.forward_device(*batch.forward_args()).check().scores.cpu().numpy()
            assert np.array_equal(again, out[fuse], equal_nan=True)
    assert np.array_equal(out["0"], out["1"])


@pytest.mark.parametrize("case", ["cifar_base_kw_B3", "cifar_deep_kw_B2"])
@pytest.mark.parametrize("fuse", ["0", "1"])
def test_round0_rows_from_the_embedding_kernel_match(monkeypatch, case, fuse):
    """GNNB_NO_EMBED_FUSE=1: round 0's input-layer rows are written by k_embed and read back by the first aggregate, instead of being
    computed inside it -- the form bench.py's aggregate-only leg runs (with GNNB_FUSE=0) so that every launch of the stand-alone
    aggregation class is a pure aggregate.  Same scores within the parity bar, the reference's decisions."""
    monkeypatch.setenv("GNNB_NO_EMBED_FUSE", "1")
    monkeypatch.setenv("GNNB_FUSE", fuse)
    g, batch = load_golden(case)
    for fam in FAMILIES:
        model = make_model(fam)
        with torch.no_grad():
            res = model.forward_device(*batch.forward_args()).check()
        want = g[f"{fam}_scores"]
        fin = np.isfinite(want)
        assert np.abs(res.scores.cpu().numpy()[fin] - want[fin]).max() <= score_tol(fam, want[fin])
        assert res.decisions.cpu().tolist() == g[f"{fam}_decisions"].tolist()
        assert model.engine().describe()["embed_fused"] == 0


def test_per_tile_dense_kernel_path_matches(monkeypatch):
    """GNNB_NO_DENSE_LDS=1 selects the per-tile dense edge kernels (the fallback for Linear layers whose source does
    not fit the LDS-staged kernels): same scores."""
    monkeypatch.setenv("GNNB_NO_DENSE_LDS", "1")
    g, batch = load_golden("cifar_wide_kw_B2")
    model = make_model("random")
    with torch.no_grad():
        res = model.forward_device(*batch.forward_args()).check()
    want = g["random_scores"]
    fin = np.isfinite(want)
    assert np.abs(res.scores.cpu().numpy()[fin] - want[fin]).max() <= score_tol("random", want[fin])
    assert res.decisions.cpu().tolist() == g["random_decisions"].tolist()


@pytest.mark.parametrize("case", ["cifar_base_kw_B3", "cifar_deep_kw_B2"])
def test_separate_top_kernels_path_matches(monkeypatch, case):
    """GNNB_NO_TOP=1 runs the top of the network (last Linear edge, last ReLU layer, property node) as its separate
    kernels instead of the fused k_top: same scores (the embeddings-after-every-half-pass test also takes this path)."""
    monkeypatch.setenv("GNNB_NO_TOP", "1")
    g, batch = load_golden(case)
    model = make_model("random")
    with torch.no_grad():
        res = model.forward_device(*batch.forward_args()).check()
    want = g["random_scores"]
    fin = np.isfinite(want)
    assert np.abs(res.scores.cpu().numpy()[fin] - want[fin]).max() <= score_tol("random", want[fin])
    assert res.decisions.cpu().tolist() == g["random_decisions"].tolist()


# What the three-piece bf16 blocks may differ from the exact-fp32 MFMA path by (absolute).  Two exact-fp32 evaluations of this
# network that merely add in a different order -- the reference's aten kernels and GNNB_BF3=0 -- already differ by 2-3e-5 on the
# shipped checkpoint (scores of magnitude 5..50; the reference's own fp32-vs-fp64 noise is 1-3e-5), so the bar
# between the two paths is 3e-5 there (measured 1.9e-5 on base) and 1e-6 on the seeded random weight set (scores in [-1, 0.05]).
# The sharper statement is against float64: the bf16x3 path may not be further from the fp64 truth than 1.25x the worse of the two
# exact-fp32 evaluations (+ 2e-6).
BF3_DELTA_ATOL = {"shipped": 3e-5, "random": 1e-6}


@pytest.mark.parametrize("case", GOLDEN_CASES)
@pytest.mark.parametrize("fam", ["shipped", "random"])
def test_bf16x3_blocks_match_the_fp32_mfma(monkeypatch, case, fam):
    """The 64x64 blocks of the node update, of the input update and of k_top's edges run on the bf16 matrix rate with three-piece
    operands (default); GNNB_BF3=0 keeps every block on the exact-fp32 MFMA.  Both must sit inside the parity bar against the
    reference and take its decisions; their difference is printed and bounded (BF3_DELTA_ATOL); and against the float64
    evaluation of the same network (the oracle in double precision) the bf16x3 path is an fp32-grade evaluation like the other
    two: its error is bounded by theirs."""
    from oracle import gnn_oracle
    g, batch = load_golden(case)
    want = g[f"{fam}_scores"]
    fin = np.isfinite(want)
    out = {}
    for bf3 in ("1", "0"):
        monkeypatch.setenv("GNNB_BF3", bf3)
        model = make_model(fam)                      # a new engine: the knob is read by gnnb_create
        with torch.no_grad():
            res = model.forward_device(*batch.forward_args()).check()
        out[bf3] = res.scores.cpu().numpy()
        assert np.abs(out[bf3][fin] - want[fin]).max() <= score_tol(fam, want[fin])
        assert res.decisions.cpu().tolist() == g[f"{fam}_decisions"].tolist()
    with torch.no_grad():
        a64 = [[t.double() for t in grp] if isinstance(grp, (list, tuple)) else grp for grp in batch.forward_args()]
        a64[4] = batch.primal_inputs.double()
        layers64 = {k: [__import__("copy").deepcopy(l).double() for l in v] for k, v in batch.layers.items()}
        a64[5] = layers64
        truth = gnn_oracle.padded_scores(gnn_oracle.oracle_forward(state_of(fam), *a64, dtype=torch.float64), batch.masks).numpy()
    delta = float(np.abs(out["1"][fin] - out["0"][fin]).max())
    e_bf3, e_f32, e_ref = (float(np.abs(x[fin] - truth[fin]).max()) for x in (out["1"], out["0"], want))
    print(f"{case} {fam}: max|bf16x3 - fp32 MFMA| = {delta:.3e} (bar {BF3_DELTA_ATOL[fam]:.0e}); against float64: bf16x3 {e_bf3:.3e}, "
          f"fp32 MFMA {e_f32:.3e}, the reference's fp32 {e_ref:.3e}; max|score| {np.abs(want[fin]).max():.3g}")
    assert delta <= BF3_DELTA_ATOL[fam]
    assert e_bf3 <= 1.25 * max(e_f32, e_ref) + 2e-6


@pytest.mark.parametrize("case", ["cifar_base_kw_B3", "cifar_wide_kw_B2", "cifar_deep_kw_B2"])
def test_nothing_reads_unwritten_workspace(case):
    """The rows of dead nodes are not written where every consumer walks live rows only, and several regions of the scratch
    are written for some nodes only: with the whole workspace poisoned with NaN before the call the scores, decisions and
    status must not change (twice: the second call sees the first call's leftovers plus the poison in between)."""
    g, batch = load_golden(case)
    model = make_model("random")
    eng = model.engine()
    want = g["random_scores"]
    fin = np.isfinite(want)
    with torch.no_grad():
        ref = model.forward_device(*batch.forward_args()).check().scores.cpu().numpy()
        for _ in range(2):
            ws = eng.workspace(batch.batch_size)
            ws.view(torch.float32).fill_(float("nan"))
            res = model.forward_device(*batch.forward_args()).check()
            got = res.scores.cpu().numpy()
            assert np.array_equal(got[fin], ref[fin])
            assert np.abs(got[fin] - want[fin]).max() <= score_tol("random", want[fin])
            assert res.decisions.cpu().tolist() == g["random_decisions"].tolist()


@pytest.mark.parametrize("case", ["cifar_base_kw_B3", "cifar_deep_kw_B2"])
def test_host_entry_point_matches(case):
    """gnnb_forward_host (CPU inputs, one pinned transfer, results back on the host -- what GraphChoice.decision uses): same
    scores as the device-pointer entry point bit for bit, decisions of the reference; python-list primals and B = 1 slices too."""
    g, batch = load_golden(case)
    model = make_model("random")
    eng = model.engine()
    with torch.no_grad():
        ref = model.forward_device(*batch.forward_args()).check()
    dec, scores = eng.forward_host(*batch.forward_args(), want_scores=True)
    assert np.array_equal(scores, ref.scores.cpu().numpy())
    assert dec.tolist() == g["random_decisions"].tolist()
    for b in range(batch.batch_size):
        one = batch.slice(b, b + 1)
        args = list(one.forward_args())
        args[3] = [p.tolist() for p in one.primals]                     # LP primals as python lists (graph_score.py:30)
        d1, s1 = eng.forward_host(*args, want_scores=True)
        assert d1[0].tolist() == g["random_decisions"][b].tolist()
        assert np.array_equal(s1[0], scores[b])
    with pytest.raises(ValueError):
        eng.forward_host(batch.lower_bounds_all[:-1], *batch.forward_args()[1:])


def test_two_stream_batch_pipelining_is_bit_identical():
    """engine.n_streams = 2 cuts a large batch into two chunks on two HIP streams: identical bytes out."""
    from gnn_branching_amd import synth
    model = make_model("shipped")
    batch = synth.make_batch("cifar_base_kw", 160, seed=77)
    eng = model.engine()
    with torch.no_grad():
        one = eng.forward(*batch.forward_args()).check()
        eng.n_streams = 2
        try:
            two = eng.forward(*batch.forward_args()).check()
        finally:
            eng.n_streams = 1
    assert torch.equal(one.scores, two.scores) and torch.equal(one.decisions, two.decisions)


@pytest.mark.parametrize("net,B", [("cifar_base_kw", 1), ("cifar_base_kw", 3), ("cifar_deep_kw", 2), ("cifar_wide_kw", 2), ("cifar_base_kw", 40)])
@pytest.mark.parametrize("fam", ["shipped", "random"])
def test_top_workgroup_split_is_bit_identical(monkeypatch, net, B, fam):
    """k_top spreads one sample over S = 2 / 4 workgroups (by output tile of its two Linear edges, partial sums and rows exchanged
    through global memory) while B x S workgroups fit the chip; GNNB_TOP_SPLIT caps S.  Every sum keeps its order, so S = 1, 2, 4
    must give IDENTICAL scores -- which is also what makes a large batch (S = 1) agree with its samples scored alone (S = 4)."""
    from gnn_branching_amd import synth
    from oracle import gnn_oracle
    batch = synth.make_batch(net, B, seed=21 + B)
    out = {}
    for S in ("1", "2", "4"):
        monkeypatch.setenv("GNNB_TOP_SPLIT", S)
        model = make_model(fam)                      # a new engine: the knob is read by gnnb_create
        with torch.no_grad():
            res = model.forward_device(*batch.forward_args()).check()
            eng = model.engine()
            eng.workspace(B).view(torch.float32).fill_(float("nan"))          # exchange buffers and counters included
            again = model.forward_device(*batch.forward_args()).check()
        out[S] = res.scores.cpu().numpy()
        assert np.array_equal(again.scores.cpu().numpy(), out[S], equal_nan=True)
        plan = eng.describe()
        assert any("k_top" in u["kernel"] for u in plan["updates"])
    assert np.array_equal(out["1"], out["2"]) and np.array_equal(out["1"], out["4"])
    with torch.no_grad():
        want = gnn_oracle.padded_scores(gnn_oracle.oracle_forward(state_of(fam), *batch.forward_args()), batch.masks).numpy()
    fin = np.isfinite(want)
    assert np.abs(out["4"][fin] - want[fin]).max() <= score_tol(fam, want[fin])


@pytest.mark.parametrize("net,B", [("cifar_base_kw", 1), ("cifar_base_kw", 5), ("cifar_wide_kw", 3), ("cifar_deep_kw", 2), ("cifar_base_kw", 70)])
@pytest.mark.parametrize("fam", ["shipped", "random"])
def test_top_kernel_with_the_update_below_inside_is_bit_identical(monkeypatch, net, B, fam):
    """k_top runs the backward node update of layer L-1 on its transposed edge's row tiles (computed transposed, so that a tile's
    accumulators are the update's input fragment): the aggregate rows never reach memory and the k_node_update launch behind k_top
    is gone.  GNNB_TOP_FUSE_UPD=0 keeps the launch.  Every sum keeps its terms and order: identical scores, decisions and rows of
    layer L-1, for one workgroup per sample (B = 70) and for four (small B)."""
    from gnn_branching_amd import synth
    batch = synth.make_batch(net, B, seed=31 + B)
    out = {}
    for knob in ("0", "1"):
        monkeypatch.setenv("GNNB_TOP_FUSE_UPD", knob)
        model = make_model(fam)
        with torch.no_grad():
            eng = model.engine()
            res = model.forward_device(*batch.forward_args()).check()
            eng.workspace(B).view(torch.float32).fill_(float("nan"))
            again = model.forward_device(*batch.forward_args()).check()
        out[knob] = (res.scores.cpu().numpy(), res.decisions.cpu().numpy())
        assert np.array_equal(again.scores.cpu().numpy(), out[knob][0], equal_nan=True)
        launches = [u["kernel"] for u in eng.describe()["updates"]]
        assert any("k_top" in k for k in launches)
        n_upd = sum("k_node_update" in k for k in launches)
        out[knob + "n"] = n_upd
    assert out["1n"] < out["0n"], (out["0n"], out["1n"])          # the launches really went away
    assert np.array_equal(out["0"][0], out["1"][0], equal_nan=True) and np.array_equal(out["0"][1], out["1"][1])


def test_large_batch_equals_its_samples_scored_alone():
    """B = 160 runs k_top with one workgroup per sample, a single subproblem with four: same bits."""
    from gnn_branching_amd import synth
    model = make_model("shipped")
    batch = synth.make_batch("cifar_base_kw", 160, seed=9)
    with torch.no_grad():
        full = model.forward_device(*batch.forward_args()).check().scores.cpu()
        for b in (0, 77, 159):
            one = model.forward_device(*batch.slice(b, b + 1).forward_args()).check().scores.cpu()
            assert torch.equal(one[0], full[b])


@pytest.mark.parametrize("net,B", [("cifar_base_kw", 1), ("cifar_base_kw", 2), ("cifar_base_kw", 8), ("cifar_base_kw", 37), ("cifar_base_kw", 256),
                                   ("cifar_deep_kw", 1), ("cifar_deep_kw", 5), ("cifar_deep_kw", 128), ("cifar_wide_kw", 1), ("cifar_wide_kw", 3),
                                   ("cifar_wide_kw", 200)])
@pytest.mark.parametrize("fam", ["shipped", "random"])
def test_small_batch_tail_kernel_is_bit_identical(monkeypatch, net, B, fam):
    """A forward ends in ONE launch, k_scored_tail (the restricted last step's scored gather + node update of layer 1 + the
    score head + the decision), instead of three; GNNB_TAIL_MAX_B=0 keeps the three kernels.  Same arithmetic per node: identical
    scores and decisions -- also with some dead nodes marked undecided and with a sample that has nothing to score.  Round 4: every
    batch size (segments of 16 .. 48 scored nodes, one and several rounds per workgroup), cifar_wide_kw's 128-slot window (two nodes per
    gather wave instead of four)."""
    from gnn_branching_amd import synth
    batch = synth.make_batch(net, B, seed=40 + B)
    args = list(batch.forward_args())
    rng = np.random.RandomState(B)
    extra = torch.from_numpy((rng.uniform(size=tuple(batch.masks.shape)) < 0.02).astype(np.float32))
    args[6] = torch.clamp(batch.masks + extra, max=1.0)                  # a few decided / dead nodes scored as well
    if B > 1:
        args[6][B - 1] = 0
    out, dec, launches = {}, {}, {}
    for knob in ("0", "8"):
        monkeypatch.setenv("GNNB_TAIL_MAX_B", "0" if knob == "0" else "1000000")
        model = make_model(fam)
        eng = model.engine()
        with torch.no_grad():
            model.forward_device(*args).check()
            eng.workspace(B).view(torch.float32).fill_(float("nan"))
            eng.profile_enable(True)
            eng.profile_read(reset=True)
            res = model.forward_device(*args).check()
            prof = eng.profile_read(reset=True)
            eng.profile_enable(False)
        out[knob], dec[knob] = res.scores.cpu().numpy(), res.decisions.cpu().tolist()
        launches[knob] = sum(v[1] for v in prof.values())
    assert np.array_equal(out["0"], out["8"], equal_nan=True) and dec["0"] == dec["8"]
    assert launches["8"] == launches["0"] - 2, launches
    if B > 1:
        assert dec["8"][B - 1] == [-1, -1]
    print(f"{net} B={B} {fam}: {launches['8']} launches per forward (three-kernel tail: {launches['0']})")


@pytest.mark.parametrize("net,B", [("cifar_base_kw", 1), ("cifar_base_kw", 8), ("cifar_deep_kw", 3), ("cifar_wide_kw", 2)])
@pytest.mark.parametrize("fam", ["shipped", "random"])
def test_small_batch_classify_pre_kernel_is_bit_identical(monkeypatch, net, B, fam):
    """Batches up to 8 classify their nodes and run the hoisted feature chains in ONE launch (k_classify_pre: a block handles the
    ambiguous nodes it found itself); GNNB_CLSPRE_MAX_B=0 keeps k_classify + k_pre.  P' rows are addressed by node id and every
    node's chain is its own column of an MFMA tile, so the scores must be identical."""
    from gnn_branching_amd import synth
    batch = synth.make_batch(net, B, seed=60 + B)
    out, launches = {}, {}
    for knob in ("0", "8"):                          # (default: 1 -- only a single subproblem takes the merged kernel)
        monkeypatch.setenv("GNNB_CLSPRE_MAX_B", knob)
        model = make_model(fam)
        eng = model.engine()
        with torch.no_grad():
            model.forward_device(*batch.forward_args()).check()
            eng.workspace(B).view(torch.float32).fill_(float("nan"))
            eng.profile_enable(True)
            eng.profile_read(reset=True)
            res = model.forward_device(*batch.forward_args()).check()
            prof = eng.profile_read(reset=True)
            eng.profile_enable(False)
        out[knob] = (res.scores.cpu().numpy(), res.decisions.cpu().tolist())
        launches[knob] = sum(v[1] for v in prof.values())
    assert np.array_equal(out["0"][0], out["8"][0], equal_nan=True) and out["0"][1] == out["8"][1]
    assert launches["8"] == launches["0"] - 1, launches
    print(f"{net} B={B} {fam}: {launches['8']} launches per forward")


@pytest.mark.gpu
def test_profile_trace_lists_the_launches_of_a_forward_in_order():
    """gnnb_profile_trace: class and duration of every launch gnnb_profile_read resolved, in launch order (bench.py prices single launches
    of the stand-alone aggregation class with it).  One forward of the default path: 11 launches at B = 40 (13 before round 4's k_scored_tail
    served every batch size), 10 for a single subproblem,
    the classification first and the score head last, the per-class sums equal to what profile_read returned."""
    from gnn_branching_amd import synth
    model = make_model("shipped")
    eng = model.engine()
    for B, want in ((40, 11), (1, 10)):
        batch = synth.make_batch("cifar_base_kw", B, seed=5)
        with torch.no_grad():
            model.forward_device(*batch.forward_args()).check()
            eng.profile_enable(True)
            eng.profile_read(reset=True)
            eng.profile_trace()
            model.forward_device(*batch.forward_args()).check()
            prof = eng.profile_read(reset=True)
            trace = eng.profile_trace()
            eng.profile_enable(False)
        assert len(trace) == want == sum(v[1] for v in prof.values()), (B, [t[0] for t in trace])
        assert trace[0][0] in ("k_classify", "k_pre") and all(ms > 0 for _, ms in trace)
        for name, (ms, n) in prof.items():
            mine = [m for c, m in trace if c == name]
            assert len(mine) == n and abs(sum(mine) - ms) <= 1e-3 * max(ms, 1e-3)
        assert eng.profile_trace() == []                      # read once
