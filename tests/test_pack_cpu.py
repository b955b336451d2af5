"""Host logic on the CPU: the operand-order weight packs of gnnb_pack.h, checked by emulating the
gfx950 MFMA lane maps (v_mfma_f32_32x32x2_f32) in numpy against plain matmuls."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests.common import random_state

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gnn_branching_amd", "csrc")
LANES = np.arange(64)
J, H = LANES & 31, LANES >> 5


@pytest.fixture(scope="module")
def packlib(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("pack") / "libgnnb_packtest.so")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-o", so,
                           os.path.join(CSRC, "gnnb_pack_test.cpp")])
    lib = C.CDLL(so)
    lib.gnnb_pt_blob_floats.restype = C.c_size_t
    lib.gnnb_pt_pack.restype = C.c_size_t
    lib.gnnb_pt_pack.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    return lib


@pytest.fixture(scope="module")
def packs(packlib):
    sd = random_state()
    blob = np.concatenate([np.asarray(v, np.float32).reshape(-1) for v in sd.values()])
    assert blob.size == packlib.gnnb_pt_blob_floats() == 117825
    out = {}
    for which, name in enumerate(["embed", "pre_fwd", "pre_bwd", "pre_inp", "prop", "upd_fwd_e", "upd_fwd_i", "upd_fwd_f", "upd_bwd",
                                  "upd_bwd_b", "upd_inp", "post_inp", "score_b", "score_f"]):
        n = packlib.gnnb_pt_pack(blob.ctypes.data, which, None, 0)
        buf = np.zeros(n, np.float32)
        assert packlib.gnnb_pt_pack(blob.ctypes.data, which, buf.ctypes.data, n) == n
        out[name] = buf
    return sd, out


# ---- numpy model of the wave-level data layout used by the kernels ----
def feat(R, h):
    return 8 * (R >> 2) + 4 * h + (R & 3)


def frag_from_rows(X):
    """(32 nodes, 64 features) -> frag[lane, R]"""
    f = np.zeros((64, 32), np.float64)
    for R in range(32):
        f[:, R] = X[J, feat(R, H)]
    return f


def rows_from_frag(f):
    X = np.zeros((32, 64))
    for R in range(32):
        X[J, feat(R, H)] = f[:, R]
    return X


def mfma(a, b, acc):
    """acc[lane, r] (16 regs) += A(32x2) B(2x32) with the gfx950 operand maps."""
    A = np.zeros((32, 2)); Bm = np.zeros((2, 32))
    A[J, H] = a
    Bm[H, J] = b
    D = A @ Bm
    for r in range(16):
        acc[:, r] += D[(r & 3) + 8 * (r >> 2) + 4 * H, J]


def frag_bias(bl):
    f = np.zeros((64, 32))
    for R in range(32):
        f[:, R] = bl[H * 32 + R]
    return f


def gemm_w64(wl, ksteps, acc, getB):
    for s in range(ksteps):
        for it in range(2):
            a = wl[(((s >> 2) * 2 + it) * 64 + LANES) * 4 + (s & 3)]
            mfma(a, getB(s), acc[:, 16 * it:16 * it + 16])


def bf16_pieces(x):
    """x (float32) -> three bf16-valued float32 arrays, round to nearest even each time (the kernel's v_cvt_pk_bf16_f32 + subtract)."""
    out, r = [], np.asarray(x, np.float32).copy()
    for _ in range(3):
        u = r.view(np.uint32).astype(np.uint64)
        b = (((u + 0x7fff + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)
        out.append(b)
        r = (r - b).astype(np.float32)
    return out


def gemm_w64_bf3(wl, nfrag, acc, getB):
    """numpy model of gemm_w64_bf3 (gnnb.hip): v_mfma_f32_32x32x16_bf16 operand maps, operands in three bf16 pieces, the six
    products of total order <= 4.  wl: the pack as float32 (reinterpreted as 8 bf16 per lane and entry)."""
    w16 = np.ascontiguousarray(wl[:6144 * nfrag]).view(np.uint16).reshape(nfrag * 4, 2, 3, 64, 8)
    wf = (w16.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    for fk in range(4 * nfrag):
        xs = [np.stack([pc for pc in bf16_pieces(np.stack([getB(8 * fk + j) for j in range(8)], 1).astype(np.float32))][i], 0)
              for i in range(3)]                                  # xs[piece][lane, j]
        for ot in range(2):
            D = np.zeros((32, 32))
            for pw, px in ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)):
                A = np.zeros((32, 16)); Bm = np.zeros((16, 32))
                for j in range(8):
                    A[J, 8 * H + j] = wf[fk, ot, pw, LANES, j]
                    Bm[8 * H + j, J] = xs[px][LANES, j]
                D += A @ Bm
            for r in range(16):
                acc[:, 16 * ot + r] += D[(r & 3) + 8 * (r >> 2) + 4 * H, J]


def gemm_small(wl, ksteps, acc, x):
    for s in range(ksteps):
        for it in range(2):
            mfma(wl[(s * 2 + it) * 64 + LANES], x[s], acc[:, 16 * it:16 * it + 16])


def lin(sd, name, x):
    return x @ np.asarray(sd[name + ".weight"], np.float64).T + np.asarray(sd[name + ".bias"], np.float64)


E = "EmbedUpdates.update."


UPD = dict(WA=0, WAS=8192, BA=12288, WCB=12352, BCB=16448, BCBROW=16512, VAW=16576, WAS3=16704, WCB3=16704 + 6144,
           FLOATS3=16704 + 2 * 6144)
# the second half of Wa alone, bf16 x 3: general nodes go through WAS.(r0 x) + Wa[:, 64:].((r1 - r0) x) (PackUpdL3)
UPD.update(WA1S3=UPD["FLOATS3"], FLOATS=UPD["FLOATS3"] + 6144)


@pytest.mark.parametrize("pack,chain,proj", [
    ("upd_fwd_e", ("fc3", "fc3_2", "fc4", "fc4_2"), "inp_f_1"), ("upd_fwd_i", ("fc3", "fc3_2", "fc4", "fc4_2"), "inp_b2_2"),
    ("upd_fwd_f", ("fc3", "fc3_2", "fc4", "fc4_2"), "fc4_2"), ("upd_bwd_b", ("bc3", "bc3_1", "bc4", "bc4_1"), "bc4_1"),
    ("upd_bwd", ("bc3", "bc3_1", "bc4", "bc4_1"), None)])
def test_node_update_chain(packs, pack, chain, proj):
    """k_node_update's folded chain on one tile.  Reference: mu = d(relu(c([relax, b(relu(a([r0 nb, r1 nb])))]))).

    Folds (gnnb_pack.h PackUpd): b feeds c linearly, so Wcb = c[:, 64:].b.W and the cached term is
    P' = c[:, :64].relax + c.b + c[:, 64:].b.b; nodes with r0 == r1 use the summed halves of a; the last layer d is NOT
    applied (the kernel stores E with mu = d(E), deferred into the consumers); and when the aggregate was built from
    rows with a deferred projection p (nb = p.W.G + s.p.b), p.W is folded into a and the s-term is one small k-step."""
    sd, pk = packs
    la, lb_, lc, ld = chain
    rng = np.random.RandomState(7)
    G = rng.standard_normal((32, 64)); relax = rng.standard_normal((32, 64)); sw = rng.standard_normal(32)
    r0 = rng.uniform(0, 1, 32); r1 = 1 - r0
    if proj is None:
        nb = G
    else:
        wp, bp = np.asarray(sd[E + proj + ".weight"], np.float64), np.asarray(sd[E + proj + ".bias"], np.float64)
        nb = G @ wp.T + sw[:, None] * bp[None, :]
    w4, b4 = np.asarray(sd[E + lc + ".weight"], np.float64), np.asarray(sd[E + lc + ".bias"], np.float64)
    bcb = b4 + w4[:, 64:] @ np.asarray(sd[E + lb_ + ".bias"], np.float64)
    Pp = relax @ w4[:, :64].T + bcb                      # what k_pre caches for ambiguous nodes
    p = pk[pack]
    assert p.size == UPD["FLOATS"]
    WA, WAS, BA, WCB, BCB, BCBROW, VAW = (UPD[k] for k in ("WA", "WAS", "BA", "WCB", "BCB", "BCBROW", "VAW"))
    np.testing.assert_allclose(p[BCBROW:BCBROW + 64], bcb, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(rows_from_frag(frag_bias(p[BCB:BCB + 64]))[0], bcb, rtol=1e-6, atol=1e-7)
    if proj is None:
        assert not p[VAW:VAW + 128].any()

    def reference(nb_, r0_, r1_, relax_):
        """the rows the reference would hold BEFORE its last Linear d"""
        e = lin(sd, E + lb_, np.maximum(lin(sd, E + la, np.concatenate([nb_ * r0_[:, None], nb_ * r1_[:, None]], 1)), 0))
        return np.maximum(lin(sd, E + lc, np.concatenate([relax_, e], 1)), 0)

    def tail(Hf, H2):
        Hf = np.maximum(Hf, 0)
        gemm_w64(p[WCB:], 32, H2, lambda s: Hf[:, s])
        return rows_from_frag(np.maximum(H2, 0))
    X = frag_from_rows(G)
    # general nodes: lane half 0 feeds r0.s, half 1 feeds r1.s into the small k-step
    Hf = frag_bias(p[BA:BA + 64])
    gemm_small(p[VAW:], 1, Hf, [np.where(H == 0, r0[J], r1[J]) * sw[J]])
    gemm_w64(p[WA:], 64, Hf, lambda s: X[:, s & 31] * (r0[J] if s < 32 else r1[J]))
    np.testing.assert_allclose(tail(Hf, frag_from_rows(Pp)), reference(nb, r0, r1, relax), atol=2e-5)
    # r0 == r1 nodes without relaxation term: summed halves, P' = bcb, both lane halves feed r0.s
    Hf = frag_bias(p[BA:BA + 64])
    gemm_small(p[VAW:], 1, Hf, [r0[J] * sw[J]])
    gemm_w64(p[WAS:], 32, Hf, lambda s: X[:, s] * r0[J])
    np.testing.assert_allclose(tail(Hf, frag_bias(p[BCB:BCB + 64])), reference(nb, r0, r0, np.zeros_like(relax)), atol=2e-5)
    # the short chain on the bf16 matrix rate (three-piece operands): WAS3 / WCB3 hold the same matrices
    Hf = frag_bias(p[BA:BA + 64])
    gemm_small(p[VAW:], 1, Hf, [r0[J] * sw[J]])
    gemm_w64_bf3(p[UPD["WAS3"]:], 1, Hf, lambda s: X[:, s] * r0[J])
    Hf = np.maximum(Hf, 0)
    H2 = frag_bias(p[BCB:BCB + 64])
    gemm_w64_bf3(p[UPD["WCB3"]:], 1, H2, lambda s: Hf[:, s])
    np.testing.assert_allclose(rows_from_frag(np.maximum(H2, 0)), reference(nb, r0, r0, np.zeros_like(relax)), atol=2e-5)
    # the bf16 x 3 node update (k_node_update, k_gather_update_q): EVERY node -- ambiguous or not -- goes through
    # H = WAS.(r0 x) + Wa[:, 64:].((r1 - r0) x); the second block adds exact zeros for r0 == r1, so a node's result does not depend on
    # whether its tile ran it
    amb = np.arange(32) % 3 == 0                      # a mixed tile: every third node ambiguous (r1 != r0, cached P' row)
    r1m = np.where(amb, r1, r0)
    relaxm = np.where(amb[:, None], relax, 0.0)
    Ppm = relaxm @ w4[:, :64].T + bcb
    Hf = frag_bias(p[BA:BA + 64])
    gemm_small(p[VAW:], 1, Hf, [np.where(H == 0, r0[J], r1m[J]) * sw[J]])
    gemm_w64_bf3(p[UPD["WAS3"]:], 1, Hf, lambda s: X[:, s] * r0[J])
    before = Hf.copy()
    gemm_w64_bf3(p[UPD["WA1S3"]:], 1, Hf, lambda s: X[:, s] * (r1m[J] - r0[J]))
    assert np.array_equal(Hf[~amb[J]], before[~amb[J]])          # exact zeros for the nodes with r0 == r1
    Hf = np.maximum(Hf, 0)
    H2 = frag_from_rows(Ppm)
    gemm_w64_bf3(p[UPD["WCB3"]:], 1, H2, lambda s: Hf[:, s])
    np.testing.assert_allclose(rows_from_frag(np.maximum(H2, 0)), reference(nb, r0, r1m, relaxm), atol=2e-5)


def test_input_update_packs(packs):
    """E_0 = relu(Q + inp_b2[:, 64:].nb) with nb = bc4_1.W.G + s.bc4_1.b.  The 64x64 map is applied on the producer side
    (PackPostInp: rows F = (inp_b2[:, 64:].bc4_1.W).E of layer 1), the input kernels add the aggregate of F and the bias
    small k-step (PackUpdInp VC)."""
    sd, pk = packs
    rng = np.random.RandomState(11)
    Erows = rng.standard_normal((32, 64)); Q = rng.standard_normal((32, 64)); sw = rng.standard_normal(32)
    wp, bp = np.asarray(sd[E + "bc4_1.weight"], np.float64), np.asarray(sd[E + "bc4_1.bias"], np.float64)
    w2 = np.asarray(sd[E + "inp_b2.weight"], np.float64)
    wc = w2[:, 64:] @ wp
    # producer side, natural row order: the flat input update adds the row-major aggregate
    p = pk["post_inp"]
    assert p.size == 8192 + 2 * 6144
    X = frag_from_rows(Erows)
    F = np.zeros((64, 32)); gemm_w64(p[0:], 32, F, lambda s: X[:, s])
    np.testing.assert_allclose(rows_from_frag(F), Erows @ wc.T, atol=2e-5)
    F3 = np.zeros((64, 32)); gemm_w64_bf3(p[8192:], 1, F3, lambda s: X[:, s])          # bf16 x 3 forms of the same two maps
    np.testing.assert_allclose(rows_from_frag(F3), Erows @ wc.T, atol=2e-5)
    Fg3 = np.zeros((64, 32)); gemm_w64_bf3(p[8192 + 6144:], 1, Fg3, lambda s: X[:, s])
    # producer side, gather-permuted rows: stored channel gather_feature(R, h) holds feature frag_feature(R, h)
    Fg = np.zeros((64, 32)); gemm_w64(p[4096:], 32, Fg, lambda s: X[:, s])
    stored = rows_from_frag(Fg)                          # what the producer writes, row-major
    np.testing.assert_allclose(rows_from_frag(Fg3), stored, atol=2e-5)
    want = Erows @ wc.T
    for hh in range(2):
        for R in range(32):
            it, r = R >> 4, R & 15
            g = 2 * ((r & 3) + 8 * (r >> 2) + 4 * hh) + it
            np.testing.assert_allclose(stored[:, g], want[:, feat(R, hh)], atol=2e-5)
    # consumer side: with every node's row equal to its own aggregate (identity edge), H = Q + VC.s + F
    pu = pk["upd_inp"]
    assert pu.size == 128
    Hf = frag_from_rows(Q)
    gemm_small(pu[0:], 1, Hf, [np.where(H == 0, sw[J], 0.0)])
    Hf += frag_from_rows(want)
    nb = Erows @ wp.T + sw[:, None] * bp[None, :]
    np.testing.assert_allclose(rows_from_frag(np.maximum(Hf, 0)), np.maximum(Q + nb @ w2[:, 64:].T, 0), atol=2e-5)


def test_pre_fwd_chain(packs):
    """k_pre, forward: fc1 on the 7 scalar features, then fc1_1 folded into fc4[:, :64] (one 64x64 map), bias = folded bcb."""
    sd, pk = packs
    rng = np.random.RandomState(12)
    f7 = rng.standard_normal((32, 7))
    p = pk["pre_fwd"]
    assert p.size == 512 + 64 + 4096 + 64 + 6144          # + W2 in three bf16 pieces
    f8 = np.concatenate([f7, np.zeros((32, 1))], 1)
    H1 = frag_bias(p[512:576]); gemm_small(p[0:], 4, H1, [f8[J, 2 * s + H] for s in range(4)]); H1 = np.maximum(H1, 0)
    Pf = frag_bias(p[4672:4736]); gemm_w64(p[576:], 32, Pf, lambda s: H1[:, s])
    relax = lin(sd, E + "fc1_1", np.maximum(lin(sd, E + "fc1", f7), 0))
    w4, b4 = np.asarray(sd[E + "fc4.weight"], np.float64), np.asarray(sd[E + "fc4.bias"], np.float64)
    bcb = b4 + w4[:, 64:] @ np.asarray(sd[E + "fc3_2.bias"], np.float64)
    np.testing.assert_allclose(rows_from_frag(Pf), relax @ w4[:, :64].T + bcb, atol=1e-5)
    Pf3 = frag_bias(p[4672:4736]); gemm_w64_bf3(p[4736:], 1, Pf3, lambda s: H1[:, s])          # the same map as three-piece bf16 block
    np.testing.assert_allclose(rows_from_frag(Pf3), relax @ w4[:, :64].T + bcb, atol=1e-5)


def test_pre_bwd_chain(packs):
    """k_pre_bwd: bc1 (7 scalar features, zero-padded k-steps) .. bc2 on [s, -d2 s, d1 s] .. bc4[:, :64]."""
    sd, pk = packs
    rng = np.random.RandomState(2)
    f7 = rng.standard_normal((32, 7)); d1 = rng.uniform(0, 1, 32); d2 = rng.uniform(0, 1, 32)
    p = pk["pre_bwd"]
    W1, B1, W2, B2, W3, B3, W4, B4, W5, B5 = 0, 512, 576, 4672, 4736, 8832, 8896, 21184, 21248, 25344
    W23, W33, W53, W43 = 25408, 25408 + 6144, 25408 + 2 * 6144, 25408 + 3 * 6144      # W2, W3, W5 and the 192-wide W4 in three bf16 pieces
    assert p.size == 25408 + 3 * 6144 + 18432
    f8 = np.concatenate([f7, np.zeros((32, 1))], 1)
    x = [f8[J, 2 * s + H] for s in range(4)]
    H1 = frag_bias(p[B1:B1 + 64]); gemm_small(p[W1:], 4, H1, x); H1 = np.maximum(H1, 0)
    H2 = frag_bias(p[B2:B2 + 64]); gemm_w64(p[W2:], 32, H2, lambda s: H1[:, s]); H2 = np.maximum(H2, 0)
    S = frag_bias(p[B3:B3 + 64]); gemm_w64(p[W3:], 32, S, lambda s: H2[:, s])
    H4 = frag_bias(p[B4:B4 + 64])
    gemm_w64(p[W4:], 96, H4, lambda s: S[:, s & 31] * (1.0 if s < 32 else (-d2[J] if s < 64 else d1[J])))
    H4 = np.maximum(H4, 0)
    Pb = frag_bias(p[B5:B5 + 64]); gemm_w64(p[W5:], 32, Pb, lambda s: H4[:, s])     # bc2_1 folded into bc4[:, :64]
    got = rows_from_frag(Pb)
    s_ = lin(sd, E + "bc1_2", np.maximum(lin(sd, E + "bc1_1", np.maximum(lin(sd, E + "bc1", f7), 0)), 0))
    relax = lin(sd, E + "bc2_1", np.maximum(lin(sd, E + "bc2", np.concatenate([s_, s_ * -d2[:, None], s_ * d1[:, None]], 1)), 0))
    w4, b4 = np.asarray(sd[E + "bc4.weight"], np.float64), np.asarray(sd[E + "bc4.bias"], np.float64)
    bcb = b4 + w4[:, 64:] @ np.asarray(sd[E + "bc3_1.bias"], np.float64)      # folded bias (PackUpd)
    np.testing.assert_allclose(got, relax @ w4[:, :64].T + bcb, atol=1e-5)
    # the same chain on the bf16 matrix rate: the 64x64 blocks and the 192-wide W4 as three input fragments
    H2 = frag_bias(p[B2:B2 + 64]); gemm_w64_bf3(p[W23:], 1, H2, lambda s: H1[:, s]); H2 = np.maximum(H2, 0)
    S = frag_bias(p[B3:B3 + 64]); gemm_w64_bf3(p[W33:], 1, S, lambda s: H2[:, s])
    H4 = frag_bias(p[B4:B4 + 64])
    gemm_w64_bf3(p[W43:], 3, H4, lambda s: S[:, s & 31] * (1.0 if s < 32 else (-d2[J] if s < 64 else d1[J])))
    H4 = np.maximum(H4, 0)
    Pb = frag_bias(p[B5:B5 + 64]); gemm_w64_bf3(p[W53:], 1, Pb, lambda s: H4[:, s])
    np.testing.assert_allclose(rows_from_frag(Pb), relax @ w4[:, :64].T + bcb, atol=1e-5)


def test_embed_and_score_packs(packs):
    sd, pk = packs
    rng = np.random.RandomState(3)
    f3 = rng.standard_normal((32, 3))
    p = pk["embed"]
    # inp_f only, row-major for the VALU kernel: inp_f_1 is deferred into the forward update of ReLU layer 1
    assert p.size == 256
    np.testing.assert_array_equal(p[:192].reshape(64, 3), np.asarray(sd[E + "inp_f.weight"]))
    np.testing.assert_array_equal(p[192:256], np.asarray(sd[E + "inp_f.bias"]))
    # score head: per-lane partial dot over the lane's 32 features + the other half
    mu = rng.standard_normal((32, 64))
    live = (rng.uniform(0, 1, 32) > 0.3).astype(np.float64)
    for name, proj in (("score_b", "bc4_1"), ("score_f", "fc4_2")):
        # the rows hold E with mu = (Wp.E + bp).live
        wp, bp = np.asarray(sd[E + proj + ".weight"], np.float64), np.asarray(sd[E + proj + ".bias"], np.float64)
        Erows = mu * live[:, None]
        mu_true = (Erows @ wp.T + bp) * live[:, None]
        p = pk[name]
        X = frag_from_rows(Erows)
        Hs = frag_bias(p[4096:4160])
        gemm_small(p[4228:], 1, Hs, [np.where(H == 0, live[J], 0.0)])
        gemm_w64(p[0:], 32, Hs, lambda s: X[:, s]); Hs = np.maximum(Hs, 0)
        ws = p[4160:4224]
        part = np.array([sum(Hs[l, R] * ws[H[l] * 32 + R] for R in range(32)) for l in range(64)])
        score = part[:32] + part[32:] + p[4224]
        want = lin(sd, "ComputeFinalScore.fscore", np.maximum(lin(sd, "ComputeFinalScore.fnode", mu_true), 0))[:, 0]
        np.testing.assert_allclose(score, want, atol=2e-5)


def test_prop_pack_is_transposed(packs):
    sd, pk = packs
    p = pk["prop"]
    w2 = np.asarray(sd[E + "out2.weight"])
    np.testing.assert_array_equal(p[320:320 + 64 * 64].reshape(64, 64), w2[:, :64].T)
    # second half: folded with fc4_2 (deferred in the rows of the top ReLU layer), bias vector V2
    wp, bp = np.asarray(sd[E + "fc4_2.weight"], np.float64), np.asarray(sd[E + "fc4_2.bias"], np.float64)
    np.testing.assert_allclose(p[320 + 64 * 64:320 + 128 * 64].reshape(64, 64), (w2[:, 64:].astype(np.float64) @ wp).T, atol=1e-6)
    v2 = p[320 + 128 * 64 + 64 + 64 * 64 + 64:][:64]
    np.testing.assert_allclose(v2, w2[:, 64:].astype(np.float64) @ bp, atol=1e-6)
    np.testing.assert_array_equal(p[0:256].reshape(4, 64), np.asarray(sd[E + "out1.weight"]).T)


# ---- MFMA gather tables (conv / conv-transpose message passing as dense local blocks) ----
GEOM = ["N", "C", "H", "W", "CT", "PY", "PX", "ay", "ax", "NBY", "NBX", "NCG", "TPS", "K2", "Hs", "Ws", "Ns",
        "ystep", "ybase", "xstep", "xbase", "WY", "WX", "normalise", "n_cmat", "n_koff", "lanes"]


def build_gather(packlib, w, h_in, w_in, stride, pad, direction, normalise, allow16=0):
    c_out, c_in, kh, kw = w.shape
    packlib.gnnb_pt_gather.restype = C.c_long
    packlib.gnnb_pt_gather.argtypes = [C.c_void_p] + [C.c_int] * 11 + [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    w = np.ascontiguousarray(w, np.float32)
    geom = np.zeros(27, np.int32)
    cost = packlib.gnnb_pt_gather(w.ctypes.data, c_in, h_in, w_in, c_out, kh, kw, stride, pad, direction, normalise, allow16,
                                  geom.ctypes.data, None, 0, None, 0)
    assert cost > 0
    g = dict(zip(GEOM, geom.tolist()))
    cmat = np.zeros(g["n_cmat"], np.float32)
    koff = np.zeros(g["n_koff"], np.int32)
    packlib.gnnb_pt_gather(w.ctypes.data, c_in, h_in, w_in, c_out, kh, kw, stride, pad, direction, normalise, allow16,
                           geom.ctypes.data, cmat.ctypes.data, cmat.size, koff.ctypes.data, koff.size)
    return g, cmat.reshape(g["NCG"], g["K2"], 64), koff.reshape(-1, 2), cost


def emulate_gather(g, cmat, koff, mu_src):
    """nb (N_dst, p) for one sample from the tables, the way k_gather_update walks them."""
    p = mu_src.shape[1]
    nb = np.zeros((g["N"], p))
    seen = np.zeros(g["N"], np.int32)
    for t in range(g["TPS"]):
        cg, rem = divmod(t, g["NBY"] * g["NBX"])
        by, bx = divmod(rem, g["NBX"])
        y0, x0 = by * g["PY"] + g["ay"], bx * g["PX"] + g["ax"]
        wy0, wx0 = by * g["ystep"] + g["ybase"], bx * g["xstep"] + g["xbase"]
        origin = wy0 * g["Ws"] + wx0
        lanes = g["lanes"]
        spk = 64 // lanes                      # window slots per k-step: 2 (32x32x2 MFMA) or 4 (16x16x4)
        acc = np.zeros((lanes, p))
        assert len(koff) == spk * g["K2"] + 8 * spk and all(p & 0xffff == 0x7fff for _, p in koff[spk * g["K2"]:])
        for k in range(spk * g["K2"]):
            off, packed = koff[k]
            wy, wx = wy0 + (packed & 0xffff), wx0 + (packed >> 16)
            if not (0 <= wy < g["Hs"] and 0 <= wx < g["Ws"]):
                continue
            row = mu_src[origin + off]
            acc += np.outer(cmat[cg, k // spk, (k % spk) * lanes:(k % spk) * lanes + lanes], row)
        for j in range(g["CT"] * g["PY"] * g["PX"]):
            cl, rem = divmod(j, g["PY"] * g["PX"])
            py, px = divmod(rem, g["PX"])
            y, x = y0 + py, x0 + px
            if 0 <= y < g["H"] and 0 <= x < g["W"]:
                n = ((cg * g["CT"] + cl) * g["H"] + y) * g["W"] + x
                nb[n] = acc[j]
                seen[n] += 1
    assert (seen == 1).all()          # every dst node belongs to exactly one (tile, lane)
    return nb


CONVS = [  # (c_in, c_out, k, stride, pad, h_in)  -- every conv of cifar_{base,wide,deep}_kw
    (3, 8, 4, 2, 1, 32), (8, 16, 4, 2, 1, 16), (3, 16, 4, 2, 1, 32), (16, 32, 4, 2, 1, 16),
    (8, 8, 3, 1, 1, 16), (8, 8, 4, 2, 1, 16),
]


@pytest.mark.parametrize("cfg", CONVS)
@pytest.mark.parametrize("direction,allow16", [(0, 0), (1, 0), (0, 1)])
def test_gather_tables_match_torch_conv(packlib, cfg, direction, allow16):
    import torch
    import torch.nn.functional as F
    c_in, c_out, k, s, pad, h_in = cfg
    rng = np.random.RandomState(c_in * 100 + c_out + direction)
    w = rng.standard_normal((c_out, c_in, k, k)).astype(np.float32)
    g, cmat, koff, cost = build_gather(packlib, w, h_in, h_in, s, pad, direction, normalise=direction, allow16=allow16)
    assert g["lanes"] in ((16, 32) if allow16 else (32,))
    if allow16:                                      # never worse than the 32-node tiling, usually a third cheaper
        assert cost <= build_gather(packlib, w, h_in, h_in, s, pad, direction, normalise=direction)[3]
    h_out = (h_in + 2 * pad - k) // s + 1
    p = 4
    if direction == 0:
        mu = rng.standard_normal((c_in * h_in * h_in, p))
        x = torch.from_numpy(mu.T.reshape(p, c_in, h_in, h_in))
        want = F.conv2d(x, torch.from_numpy(w).double(), None, s, pad).reshape(p, -1).T.numpy()
    else:
        mu = rng.standard_normal((c_out * h_out * h_out, p))
        x = torch.from_numpy(mu.T.reshape(p, c_out, h_out, h_out))
        y = F.conv_transpose2d(x, torch.from_numpy(w).double(), None, s, pad)
        assert y.shape[-1] == h_in
        want = y.reshape(p, -1).T.numpy()            # the tap-count division is applied by the kernel, not the tables
    got = emulate_gather(g, cmat, koff, mu)
    np.testing.assert_allclose(got, want, atol=1e-5)
    dense = cost * 2 * 32 * 32 * 2 / 2          # MACs issued per sample (each MFMA: 32x32x2)
    useful = w.size * (h_out * h_out)           # MACs of the sparse map per channel
    print(f"conv {cfg} dir {direction} lanes {g['lanes']}: tile {g['CT']}x{g['PY']}x{g['PX']} align ({g['ay']},{g['ax']}) window {g['WY']}x{g['WX']} "
          f"K2={g['K2']} tiles/sample={g['TPS']} mfma/sample={cost} density={useful / (dense / 64 * 32):.2f}")
