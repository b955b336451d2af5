"""Config 4 (cifar_deep_kw, 128 subproblems per rank, one RCCL all-gather per step overlapped with the next forward) safe by construction:
a forward whose handle was created for the multi-GPU path (parallel.DIST_ENGINE_OPTIONS: k_top's workgroup split off) needs NO two
workgroups resident at the same time, so it completes -- status 0, the bits of an undisturbed forward -- while another stream holds most
of the chip's CUs.  SURVEY 8(e): "N-rank output must equal 1-rank output bit-for-bit"; reference path: graph_conv.py:77-470."""
import ctypes as C

import pytest
import torch

from tests.test_gpu_parity import make_model

pytestmark = pytest.mark.gpu


def _occupy(eng, n_wg, ms, stream):
    from gnn_branching_amd import _lib
    # 512-thread workgroups that take a whole CU's LDS: nothing else fits beside one of them
    _lib.check(eng.lib.gnnb_debug_occupy(n_wg, 512, 160 * 1024 - 256, float(ms), C.c_void_p(stream.cuda_stream)), "gnnb_debug_occupy")


@pytest.mark.parametrize("net,B", [("cifar_deep_kw", 128), ("cifar_base_kw", 64)])
def test_forward_beside_a_cu_hog_is_clean_and_bit_identical(net, B):
    from gnn_branching_amd import parallel, synth
    model = make_model("shipped")
    model.engine_options = dict(parallel.DIST_ENGINE_OPTIONS)
    eng = model.engine()
    assert eng.get_option("top_split") == 1
    dev = eng.device
    batch = synth.make_batch(net, B, seed=4242)
    args = batch.forward_args()
    d = [[t.to(dev) for t in g] if isinstance(g, list) else g for g in args]
    d[4], d[6] = args[4].to(dev), args[6].to(dev)
    n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
    side = torch.cuda.Stream(device=dev)
    with torch.no_grad():
        calm = eng.forward(*d).check()
        # the split form (default handle) computes the same bits when nothing disturbs it: the rule costs time, never results
        default = make_model("shipped")
        assert default.engine().get_option("top_split") == 4
        assert torch.equal(default.engine().forward(*d).check().scores, calm.scores)
        torch.cuda.synchronize()
        # three quarters of the CUs are taken for 60 ms; five forwards (~1 ms each undisturbed) start and finish inside that window
        _occupy(eng, (3 * n_cu) // 4, 60.0, side)
        t0 = torch.cuda.Event(enable_timing=True)
        t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        results = [eng.forward(*d) for _ in range(5)]
        t1.record()
        t1.synchronize()
        busy_ms = t0.elapsed_time(t1)
        hog_still_running = not side.query()
        torch.cuda.synchronize()
    for r in results:
        assert int(r.status.cpu()[0]) == 0
        r.check()
        assert torch.equal(r.scores, calm.scores) and torch.equal(r.decisions, calm.decisions)
    # the forwards really ran beside the hog, not after it
    assert hog_still_running and busy_ms < 60.0, (hog_still_running, busy_ms)


def test_options_travel_through_the_abi():
    """gnnb_set_option / gnnb_get_option on a live handle: values round-trip, bind-shaping options are refused on a bound handle,
    out-of-range values and unknown names are refused; BatchPipeline's handles have the k_top split off whatever the caller passes."""
    from gnn_branching_amd import _lib, engine as E, synth
    from tests.common import shipped_state
    sd = {k: torch.as_tensor(v) for k, v in shipped_state().items()}
    eng = E.ScorerEngine(sd, options={"fuse": 0, "tail_max_b": 7})
    assert eng.get_option("fuse") == 0 and eng.get_option("tail_max_b") == 7 and eng.get_option("bf3") == 1
    eng.set_option("top_split", 2)
    assert eng.get_option("top_split") == 2
    with pytest.raises(RuntimeError, match="outside"):
        eng.set_option("top_split", 9)
    with pytest.raises(RuntimeError, match="unknown option"):
        eng.set_option("sweep", 1)
    batch = synth.make_batch("cifar_base_kw", 2, seed=1)
    eng.forward(*batch.forward_args()).check()                    # binds
    with pytest.raises(RuntimeError, match="before gnnb_bind_network"):
        eng.set_option("gather", 0)
    eng.set_option("fuse", 1)                                     # forward-time options may change between forwards
    pipe = E.BatchPipeline(sd, depth=2, options={"top_split": 4, "fuse": 0})
    assert [e.get_option("top_split") for e in pipe.engines] == [1, 1] and pipe.engines[0].get_option("fuse") == 0
    assert sorted(_lib.OPTIONS) == sorted(eng.lib.gnnb_option_name(i).decode() for i in range(eng.lib.gnnb_option_count()))
