"""Two independent batches in flight (engine.BatchPipeline: one handle and one stream per slot, batches dealt in turn): the scores are the
bits of ``ScorerEngine.forward`` one batch at a time, whatever overlaps on the chip, and the status words stay clean -- including small
batches, where a single forward would split a sample's k_top over several workgroups (the pipeline's handles do not)."""
import numpy as np
import pytest
import torch

from tests.test_gpu_parity import make_model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("net,B", [("cifar_base_kw", 40), ("cifar_deep_kw", 6), ("cifar_wide_kw", 3)])
def test_two_batches_in_flight_are_bit_identical(net, B):
    from gnn_branching_amd import engine as E, synth
    model = make_model("shipped")
    eng = model.engine()
    dev = eng.device
    batches = [synth.make_batch(net, B, seed=300 + i) for i in range(5)]
    dargs, want = [], []
    with torch.no_grad():
        for b in batches:
            args = b.forward_args()
            d = [[t.to(dev) for t in g] if isinstance(g, list) else g for g in args]
            d[4], d[6] = args[4].to(dev), args[6].to(dev)
            dargs.append(d)
            r = eng.forward(*d).check()
            want.append((r.scores.cpu().numpy(), r.decisions.cpu().numpy()))
    pipe = E.BatchPipeline(model.state_dict(), depth=2)
    assert len(pipe.engines) == 2 and pipe.engines[0].h.value != pipe.engines[1].h.value
    torch.cuda.synchronize()
    results = []
    with torch.no_grad():
        for rep in range(3):
            for d in dargs:
                results.append(pipe.submit(*d))             # no synchronisation between submissions
    pipe.synchronize()
    for i, r in enumerate(results):
        r.check()
        ws, wd = want[i % len(want)]
        assert np.array_equal(r.scores.cpu().numpy(), ws, equal_nan=True), (net, B, i)
        assert np.array_equal(r.decisions.cpu().numpy(), wd), (net, B, i)
    # wait(): the caller's stream is ordered behind a result without a host synchronisation
    r = pipe.submit(*dargs[0]).wait()
    s = r.scores.clone()
    torch.cuda.synchronize()
    assert np.array_equal(s.cpu().numpy(), want[0][0], equal_nan=True)


def test_inputs_may_be_dropped_right_after_submit():
    """The lifetime contract of BatchPipeline.submit: a caller that builds a fresh batch per step and drops it as
    soon as submit returns -- then allocates and fills new tensors of the same sizes on ITS stream, which the caching allocator
    would serve from the dropped inputs' memory -- still gets the bits of the plain forward: submit marks every device input as
    in use by the side stream."""
    from gnn_branching_amd import engine as E, synth
    model = make_model("shipped")
    eng = model.engine()
    dev = eng.device
    batches = [synth.make_batch("cifar_base_kw", 32, seed=500 + i) for i in range(4)]

    def to_dev(b):
        args = b.forward_args()
        d = [[t.to(dev) for t in g] if isinstance(g, list) else g for g in args]
        d[4], d[6] = args[4].to(dev), args[6].to(dev)
        return d
    want = []
    with torch.no_grad():
        for b in batches:
            r = eng.forward(*to_dev(b)).check()
            want.append(r.scores.cpu().numpy())
    pipe = E.BatchPipeline(model.state_dict(), depth=2)
    torch.cuda.synchronize()
    results = []
    with torch.no_grad():
        for rep in range(3):
            for b in batches:
                d = to_dev(b)
                shapes = [t.shape for g in d[:4] for t in g] + [d[4].shape, d[6].shape]
                results.append(pipe.submit(*d))
                del d                                         # dropped at once ...
                junk = [torch.full(s, float("nan"), device=dev) for s in shapes]      # ... and the same sizes asked for again, poisoned
                del junk
    pipe.synchronize()
    for i, r in enumerate(results):
        r.check()
        assert np.array_equal(r.scores.cpu().numpy(), want[i % len(want)], equal_nan=True), i
