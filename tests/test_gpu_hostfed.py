"""Host-fed batches through the cross-batch double buffer (engine.HostFedPipeline): the copies of batch i + 1 run on a copy stream under
the forward of batch i; results are the bits of the device-resident forward.  (reference: the H2D copies of graph_score.py:26-30.)"""
import numpy as np
import pytest
import torch

from tests.test_gpu_parity import make_model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("pinned", [False, True])
def test_host_fed_pipeline_is_bit_identical_and_reuses_its_buffers(pinned):
    from gnn_branching_amd import engine as E, synth
    model = make_model("shipped")
    eng = model.engine()
    dev = eng.device
    batches = [synth.make_batch("cifar_base_kw", 24, seed=100 + i) for i in range(5)]      # five different batches through two buffer sets
    want = []
    with torch.no_grad():
        for b in batches:
            args = b.forward_args()
            d = [[t.to(dev) for t in g] if isinstance(g, list) else g for g in args]
            d[4], d[6] = args[4].to(dev), args[6].to(dev)
            r = eng.forward(*d).check()
            want.append((r.scores.cpu().numpy(), r.decisions.cpu().numpy(), [t.cpu().numpy() for t in r.ragged()]))
    pipe = E.HostFedPipeline(eng)
    results = []
    with torch.no_grad():
        for b in batches:
            args = list(b.forward_args())
            if pinned:
                args = [[t.pin_memory() for t in g] if isinstance(g, list) else g for g in args]
                args[4], args[6] = b.forward_args()[4].pin_memory(), b.forward_args()[6].pin_memory()
            results.append(pipe.submit(*args))              # no synchronisation between submissions
    for r, (ws, wd, wr) in zip(results, want):
        r.check()
        assert np.array_equal(r.scores.cpu().numpy(), ws, equal_nan=True)
        assert np.array_equal(r.decisions.cpu().numpy(), wd)
        # results held across more than `depth` submits keep their OWN mask (it used to be a view of the slot's buffer, which the
        # submit two calls later overwrites): the ragged score lists are cut with it
        got = [t.cpu().numpy() for t in r.ragged()]
        assert len(got) == len(wr) and all(np.array_equal(g, w) for g, w in zip(got, wr))
