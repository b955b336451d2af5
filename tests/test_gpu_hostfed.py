"""Host-fed batches through the cross-batch double buffer (engine.HostFedPipeline): the copies of batch i + 1 run on a copy stream under
the forward of batch i; results are the bits of the device-resident forward.  (reference: the H2D copies of graph_score.py:26-30.)"""
import numpy as np
import pytest
import torch

from tests.test_gpu_parity import make_model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("compact", [True, False])
@pytest.mark.parametrize("pinned", [False, True])
def test_host_fed_pipeline_is_bit_identical_and_reuses_its_buffers(pinned, compact):
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
    pipe = E.HostFedPipeline(eng, compact=compact)       # compact: dual_vars / primals cross the link as records of the ambiguous nodes only
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


@pytest.mark.parametrize("net,B", [("cifar_base_kw", 40), ("cifar_deep_kw", 9)])
def test_amb_record_image_and_scatter(net, B):
    """gnnb_pack_amb_records / gnnb_scatter_amb_records (include/gnnb.h): the image holds exactly the nodes with lb < 0 < ub of every ReLU
    layer, in index order, with dual[:, 1], dual[:, 2] and the two primals of the node -- and primals[-1]; scattered into poisoned
    full-size arrays, those entries (and only those) carry the batch's values.  LP-like signed duals and decided nodes included."""
    import ctypes as C
    from gnn_branching_amd import _lib, synth
    model = make_model("shipped")
    eng = model.engine()
    batch = synth.make_batch(net, B, seed=21)
    rng = np.random.RandomState(3)
    args = list(batch.forward_args())
    args[2] = [torch.from_numpy(rng.standard_normal(tuple(t.shape)).astype(np.float32)) for t in args[2]]      # duals of either sign
    lbs, ubs, duals, prims, x, layers, mask = args
    eng.bind(layers["fixed_layers"], tuple(lbs[0].shape[1:]))
    nb, nd, npr = len(lbs), len(duals), len(prims)
    host = [t.float().contiguous() for t in list(lbs) + list(ubs) + list(duals) + list(prims)]
    tabs = [(C.c_void_p * n)(*[t.data_ptr() for t in g]) for n, g in ((nb, host[:nb]), (nb, host[nb:2 * nb]), (nd, host[2 * nb:2 * nb + nd]), (npr, host[2 * nb + nd:]))]
    hb = _lib.Batch(tabs[0], tabs[1], tabs[2], tabs[3], None, None, None, None, nb, nd, npr)
    cap = int(eng.lib.gnnb_amb_records_bytes(eng.h, B))
    img = torch.zeros(cap // 4, dtype=torch.int32)
    used = C.c_size_t(0)
    _lib.check(eng.lib.gnnb_pack_amb_records(eng.h, C.byref(hb), B, img.data_ptr(), cap, C.byref(used)), "gnnb_pack_amb_records")
    words = img.numpy()
    L = nd
    nrec = int(words[2])
    assert words[1] == L and words[3] == B and used.value == 4 * (16 + 6 * nrec + ((B + 3) & ~3)) <= cap
    rec = words[16:16 + 6 * nrec].reshape(nrec, 6)
    relu_q = [i for i, l in enumerate(layers["fixed_layers"]) if isinstance(l, torch.nn.ReLU)]
    total = 0
    for k in range(L):
        lb, ub = lbs[k + 1].reshape(-1).numpy(), ubs[k + 1].reshape(-1).numpy()
        want = np.flatnonzero((lb < 0) & (ub > 0))
        mine = rec[rec[:, 0] == k]
        mine = mine[np.argsort(mine[:, 1], kind="stable")]        # (the records come in whatever order the packer's threads finished)
        assert np.array_equal(mine[:, 1], want)
        vals = mine[:, 2:].copy().view(np.float32)
        d = duals[k].numpy().reshape(-1, 3)
        assert np.array_equal(vals[:, 0], d[want, 1]) and np.array_equal(vals[:, 1], d[want, 2])
        assert np.array_equal(vals[:, 2], prims[relu_q[k] - 1].numpy()[want]) and np.array_equal(vals[:, 3], prims[relu_q[k]].numpy()[want])
        total += len(want)
    assert total == nrec
    assert np.array_equal(words[16 + 6 * nrec:16 + 6 * nrec + B].view(np.float32), prims[-1].numpy())
    # ---- scatter into NaN-poisoned arrays on the device
    dev = eng.device
    d_img = img[:used.value // 4].to(dev)
    d_dual = [torch.full(tuple(t.shape), float("nan"), device=dev) for t in duals]
    d_prim = [torch.full(tuple(t.shape), float("nan"), device=dev) for t in prims]
    dptr = (C.c_void_p * nd)(*[t.data_ptr() for t in d_dual])
    pptr = (C.c_void_p * npr)(*[t.data_ptr() for t in d_prim])
    st = torch.zeros(1, dtype=torch.int32, device=dev)
    cur = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(eng.lib.gnnb_scatter_amb_records(eng.h, d_img.data_ptr(), B, dptr, nd, pptr, npr, st.data_ptr(), cur), "gnnb_scatter_amb_records")
    torch.cuda.synchronize()
    assert int(st.cpu()[0]) == 0
    for k in range(L):
        lb, ub = lbs[k + 1].reshape(-1), ubs[k + 1].reshape(-1)
        amb = (lb < 0) & (ub > 0)
        got = d_dual[k].cpu().reshape(-1, 3)
        assert torch.equal(got[amb][:, 1:], duals[k].reshape(-1, 3)[amb][:, 1:])
        assert torch.isnan(got[~amb]).all() and torch.isnan(got[:, 0]).all()
        for m in (relu_q[k] - 1, relu_q[k]):
            gp = d_prim[m].cpu().reshape(-1)
            assert torch.equal(gp[amb], prims[m].reshape(-1)[amb]) and torch.isnan(gp[~amb]).all()
    assert torch.equal(d_prim[-1].cpu().reshape(-1), prims[-1].reshape(-1))
    # ---- a foreign or corrupt image is refused, not scattered (status bit 2): wrong batch size in the header, a record beyond its array
    for how in ("batch", "index", "layer"):
        bad = img[:used.value // 4].clone()
        if how == "batch":
            bad[3] = B + 1
        elif how == "index":
            bad[16 + 1] = B * eng.sizes[1 + int(bad[16])] + 5          # record 0: flat node index past the end of its layer
        else:
            bad[16] = L + 3
        d_bad = bad.to(dev)
        p_dual = [torch.full(tuple(t.shape), float("nan"), device=dev) for t in duals]
        p_prim = [torch.full(tuple(t.shape), float("nan"), device=dev) for t in prims]
        dptr = (C.c_void_p * nd)(*[t.data_ptr() for t in p_dual])
        pptr = (C.c_void_p * npr)(*[t.data_ptr() for t in p_prim])
        st.zero_()
        _lib.check(eng.lib.gnnb_scatter_amb_records(eng.h, d_bad.data_ptr(), B, dptr, nd, pptr, npr, st.data_ptr(), cur), "gnnb_scatter_amb_records")
        torch.cuda.synchronize()
        assert int(st.cpu()[0]) == 4, how
        if how == "batch":                                       # header mismatch: nothing at all is written
            assert all(torch.isnan(t).all() for t in p_dual + p_prim)


def test_two_networks_alternate_through_one_engine_and_pipeline():
    """ADVICE (round 5): the compact path packs with the sizes of whatever network the HANDLE is bound to.  Two networks of equal depth
    (cifar_base_kw, cifar_wide_kw) alternating through one engine -- through the pipeline and through eng.forward between submits -- must
    each be packed under their own binding: scores equal the device-resident forward's; tensors of the wrong size raise ValueError
    before the C packer sees a pointer."""
    from gnn_branching_amd import engine as E, synth
    model = make_model("shipped")
    eng = model.engine()
    dev = eng.device
    batches = [synth.make_batch(net, 6, seed=300 + i) for i, net in enumerate(["cifar_base_kw", "cifar_wide_kw", "cifar_base_kw", "cifar_wide_kw", "cifar_wide_kw"])]
    want = []
    with torch.no_grad():
        for b in batches:
            args = b.forward_args()
            d = [[t.to(dev) for t in g] if isinstance(g, list) else g for g in args]
            d[4], d[6] = args[4].to(dev), args[6].to(dev)
            want.append(eng.forward(*d).check().scores.cpu().numpy())
    pipe = E.HostFedPipeline(eng, compact=True)
    other = synth.make_batch("cifar_deep_kw", 2, seed=9)
    got = []
    with torch.no_grad():
        for i, b in enumerate(batches):
            got.append(pipe.submit(*b.forward_args()))
            if i == 1:                                            # someone else rebinds the shared engine between two submits
                eng.forward(*other.forward_args()).check()
    for r, w in zip(got, want):
        r.check()
        assert np.array_equal(r.scores.cpu().numpy(), w, equal_nan=True)
    bad = list(batches[0].forward_args())
    bad[2] = [t[:-3] for t in bad[2]]                             # dual_vars three rows short
    with pytest.raises(ValueError):
        pipe.submit(*bad)
