"""the online-learning step (reference graphnet/graph_score_online.py:62-77).
CPU: the autograd oracle against gradients / parameters produced by the reference's own GraphChoice.online_learning.
GPU: the hand-written backward pass + Adam kernel (libgnnb.so gnnb_online_step) against the same vectors and the oracle."""
import os

import numpy as np
import pytest
import torch

from tests.common import GOLDEN, FAMILIES, load_golden, state_of, relu_sizes

CASE = "cifar_base_kw"
LR = 1e-4


def online_golden():
    return dict(np.load(os.path.join(GOLDEN, CASE + "_online.npz")))


def one_subproblem(go):
    _, batch = load_golden(CASE + "_B3")
    s = int(go["sample"])
    return batch.slice(s, s + 1)


def flat_index(one, kw):
    return int(sum(relu_sizes(one)[:int(kw[0])]) + int(kw[1]))


def blob_to_state(blob, like):
    out, off = {}, 0
    for k, v in like.items():
        v = np.asarray(v)
        out[k] = torch.from_numpy(np.asarray(blob[off:off + v.size], np.float32).reshape(v.shape).copy())
        off += v.size
    return out


def grad_close(got, want, rel=2e-4):
    """Gradients agree to `rel` of the largest entry of their tensor-wide scale (fp32 sums in a different order)."""
    scale = float(np.abs(want).max())
    assert scale > 0
    err = float(np.abs(got - want).max())
    assert err <= rel * scale, f"gradient differs by {err:.3e} (scale {scale:.3e})"


def params_close(got, want, grad, prev, step):
    """Adam turns a gradient into a step of about lr * g / |g|: where the gradient is far from zero the parameters must agree
    tightly; where it is at rounding-noise level the SIGN of the step is noise, so only its size (<= step * lr, a little
    more for the moment estimates) is checked."""
    scale = float(np.abs(grad).max())
    firm = np.abs(grad) > 1e-3 * scale
    assert firm.sum() > 1000
    np.testing.assert_allclose(got[firm], want[firm], rtol=0, atol=0.02 * LR)
    assert float(np.abs(got - prev).max()) <= 1.05 * LR * (step + 1) + 1e-7
    assert float(np.abs(got - want).max()) <= 2.1 * LR * (step + 1)


@pytest.mark.parametrize("fam", FAMILIES)
def test_oracle_matches_reference(fam):
    from oracle.online_oracle import OnlineOracle
    go = online_golden()
    one = one_subproblem(go)
    o = OnlineOracle(state_of(fam), lr=float(go["lr"]), wd=float(go["wd"]))
    prev = o.blob()
    for step, imp in enumerate(go["improvements"]):
        kw = go[f"{fam}_s{step}_kw"]
        loss, scores = o.step(one.forward_args(), [flat_index(one, kw)], [imp])
        np.testing.assert_allclose(loss[0], go[f"{fam}_s{step}_loss"], rtol=1e-4, atol=1e-5)
        grad_close(o.grad_blob(), go[f"{fam}_s{step}_grad"])
        params_close(o.blob(), go[f"{fam}_s{step}_params"], go[f"{fam}_s{step}_grad"], prev, step)
        prev = go[f"{fam}_s{step}_params"]


@pytest.mark.gpu
@pytest.mark.parametrize("fam", FAMILIES)
def test_hip_gradient_and_adam_match_reference(fam):
    from gnn_branching_amd.engine import ScorerEngine
    go = online_golden()
    one = one_subproblem(go)
    eng = ScorerEngine(state_of(fam))
    eng.online_create(float(go["lr"]), float(go["wd"]))
    w0 = eng.get_weights()
    fused = eng.forward(*one.forward_args()).check()
    # gradient only: nothing may move
    kw0 = flat_index(one, go[f"{fam}_s0_kw"])
    loss, scores = eng.online_step(one.forward_args(), [kw0], [go["improvements"][0]], apply=False, want_scores=True)
    np.testing.assert_array_equal(eng.get_weights(), w0)
    grad_close(eng.online_grad(), go[f"{fam}_s0_grad"])
    # the training-form forward computes the same scores as the fused scorer
    m = one.masks[0] != 0
    np.testing.assert_allclose(scores[0].cpu().numpy()[m], fused.scores[0].cpu().numpy()[m], rtol=0, atol=1e-4)
    assert torch.isinf(scores[0].cpu()[~m]).all()
    prev = w0
    for step, imp in enumerate(go["improvements"]):
        dec = eng.forward(*one.forward_args()).check().decisions[0].tolist()
        assert dec == go[f"{fam}_s{step}_decision"].tolist()
        kw = go[f"{fam}_s{step}_kw"]
        loss, _ = eng.online_step(one.forward_args(), [flat_index(one, kw)], [imp])
        np.testing.assert_allclose(loss[0], go[f"{fam}_s{step}_loss"], rtol=1e-4, atol=1e-4)
        grad_close(eng.online_grad(), go[f"{fam}_s{step}_grad"])
        params_close(eng.get_weights(), go[f"{fam}_s{step}_params"], go[f"{fam}_s{step}_grad"], prev, step)
        prev = go[f"{fam}_s{step}_params"]


@pytest.mark.gpu
def test_hip_batched_step_matches_oracle():
    """B = 3 subproblems with different properties in one step (the reference takes one): sum of the three losses."""
    from gnn_branching_amd.engine import ScorerEngine
    from oracle.online_oracle import OnlineOracle
    _, batch = load_golden(CASE + "_B3")
    state = state_of("random")
    rs = relu_sizes(batch)
    kws = []
    for b in range(batch.batch_size):
        idx = batch.masks[b].nonzero().view(-1)
        kws.append(int(idx[len(idx) // 2]))
    imps = [0.05, 0.2, 0.0]
    o = OnlineOracle(state)
    loss_o, _ = o.step(batch.forward_args(), kws, imps)
    eng = ScorerEngine(state)
    eng.online_create()
    loss, _ = eng.online_step(batch.forward_args(), kws, imps)
    np.testing.assert_allclose(loss, loss_o, rtol=1e-4, atol=1e-5)
    grad_close(eng.online_grad(), o.grad_blob())
    params_close(eng.get_weights(), o.blob(), o.grad_blob(), np.concatenate([np.asarray(v).reshape(-1) for v in state.values()]), 0)
    # the fused scorer now runs with the updated parameters: same scores as the oracle forward with them
    from oracle.gnn_oracle import oracle_forward, padded_scores
    with torch.no_grad():
        ref = padded_scores(oracle_forward(blob_to_state(eng.get_weights(), state), *batch.forward_args()), batch.masks)
    got = eng.forward(*batch.forward_args()).check().scores.cpu()
    m = batch.masks != 0
    np.testing.assert_allclose(got[m].numpy(), ref[m].numpy(), rtol=0, atol=1e-4)


@pytest.mark.gpu
def test_online_graph_choice_surface():
    """The reference's call pattern (relu_conv_online.py:109-117, :208): decision, online_learning, del_score."""
    from gnn_branching_amd import nets
    from gnn_branching_amd.graphnet.graph_score_online import GraphChoice
    go = online_golden()
    one = one_subproblem(go)
    init_mask = [m[0] for m in one.bab_masks]
    g = GraphChoice(init_mask, os.path.join(nets.ASSETS, "cifar_trained_gnn.npz"), lr=float(go["lr"]), wd=float(go["wd"]))
    g.verbose = False
    prev = np.concatenate([v.numpy().reshape(-1) for v in g.model.state_dict().values()])
    for step, imp in enumerate(go["improvements"]):
        d = g.decision(one.lower_bounds_all, one.upper_bounds_all, one.dual_vars, one.primal_inputs,
                       [p.tolist() for p in one.primals], one.layers, init_mask)
        assert d == go[f"shipped_s{step}_decision"].tolist()
        g.online_learning(go[f"shipped_s{step}_kw"].tolist(), float(imp))
        np.testing.assert_allclose(g.last_loss, go[f"shipped_s{step}_loss"], rtol=1e-4, atol=1e-4)
        now = np.concatenate([v.numpy().reshape(-1) for v in g.model.state_dict().values()])
        params_close(now, go[f"shipped_s{step}_params"], go[f"shipped_s{step}_grad"], prev, step)
        prev = go[f"shipped_s{step}_params"]
        g.del_score()
    with pytest.raises(RuntimeError):
        g.online_learning([0, 0], 0.1)


@pytest.mark.gpu
@pytest.mark.parametrize("name,T", [("toy_mlp", 2), ("toy_conv3", 2), ("toy_oddch", 2), ("toy_single", 2), ("toy_k5", 3), ("cifar_deep_kw", 1)])
def test_hip_step_on_other_networks(name, T):
    """Linear first layer (dense edges and their adjoints everywhere), 3x3 stride-1 / 5x5 / 2x2 convolutions, odd channel
    counts, a single ReLU layer, T = 1 and 3: gradient, loss and Adam step against the autograd oracle."""
    from gnn_branching_amd import nets, synth
    from gnn_branching_amd.engine import ScorerEngine
    from oracle.online_oracle import OnlineOracle
    from tests.test_gpu_generic_nets import ARCHS
    for i, (n, spec) in enumerate(ARCHS.items()):
        nets.register_arch(n, spec, seed=100 + i)
    batch = synth.make_batch(name, 2, seed=11, props=[(3, 5), (1, 7)])
    state = state_of("random")
    kws = [int(batch.masks[b].nonzero().view(-1)[-2]) for b in range(2)]
    imps = [0.3, 0.01]
    o = OnlineOracle(state, T=T)
    loss_o, _ = o.step(batch.forward_args(), kws, imps)
    eng = ScorerEngine(state, T=T)
    eng.online_create()
    loss, _ = eng.online_step(batch.forward_args(), kws, imps)
    np.testing.assert_allclose(loss, loss_o, rtol=1e-4, atol=1e-5)
    grad_close(eng.online_grad(), o.grad_blob())
    w0 = np.concatenate([np.asarray(v).reshape(-1) for v in state.values()])
    params_close(eng.get_weights(), o.blob(), o.grad_blob(), w0, 0)
