// CPU-only test shim for gnnb_pack.h (built with g++ by tests/test_pack_cpu.py): exposes the
// operand-order packs so a numpy emulation of the MFMA lane maps can be checked against a plain matmul.
#include "gnnb_pack.h"

extern "C" {
size_t gnnb_pt_blob_floats() { return gnnb::blob_floats(); }
// which: 0 embed, 1 pre_fwd, 2 pre_bwd, 3 pre_inp, 4 prop, 5 upd_fwd_e, 6 upd_fwd_i, 7 upd_fwd_f, 8 upd_bwd, 9 upd_bwd_b,
// 10 upd_inp, 11 post_inp, 12 score_b, 13 score_f
size_t gnnb_pt_pack(const float* blob, int which, float* out, size_t cap) {
  gnnb::Packs pk;
  gnnb::build_packs(blob, pk);
  const std::vector<float>* v[14] = {&pk.embed, &pk.pre_fwd, &pk.pre_bwd, &pk.pre_inp, &pk.prop, &pk.upd_fwd_e, &pk.upd_fwd_i,
                                     &pk.upd_fwd_f, &pk.upd_bwd, &pk.upd_bwd_b, &pk.upd_inp, &pk.post_inp, &pk.score_b, &pk.score_f};
  if (which < 0 || which > 13) return 0;
  if (out && cap >= v[which]->size()) std::memcpy(out, v[which]->data(), v[which]->size() * sizeof(float));
  return v[which]->size();
}

// gather tables for one conv edge.  geom receives 27 ints (see tests/test_pack_cpu.py); returns the
// number of MFMAs per sample, or -1.  cmat / koff are filled when large enough.
long gnnb_pt_gather(const float* w, int c_in, int h_in, int w_in, int c_out, int kh, int kw, int stride, int pad,
                    int dir, int normalise, int allow16, int* geom, float* cmat, size_t cmat_cap, int* koff, size_t koff_cap) {
  gnnb::Edge e;
  e.kind = 0; e.c_in = c_in; e.h_in = h_in; e.w_in = w_in; e.c_out = c_out; e.kh = kh; e.kw = kw; e.stride = stride; e.pad = pad;
  e.h_out = (h_in + 2 * pad - kh) / stride + 1; e.w_out = (w_in + 2 * pad - kw) / stride + 1;
  e.n_in = c_in * h_in * w_in; e.n_out = c_out * e.h_out * e.w_out;
  e.w.assign(w, w + (size_t)c_out * c_in * kh * kw);
  gnnb::GatherHost g;
  if (!gnnb::build_gather(e, dir, normalise != 0, g, 0, allow16 != 0)) return -1;
  const gnnb::GatherGeom& q = g.g;
  const int v[27] = {q.tm.N, q.tm.C, q.tm.H, q.tm.W, q.tm.CT, q.tm.PY, q.tm.PX, q.tm.ay, q.tm.ax, q.tm.NBY, q.tm.NBX, q.tm.NCG,
                     q.tm.TPS, q.K2, q.Hs, q.Ws, q.Ns, q.ystep, q.ybase, q.xstep, q.xbase, q.WY, q.WX, q.normalise,
                     (int)g.cmat.size(), (int)g.koff.size(), q.lanes};
  for (int i = 0; i < 27; ++i) geom[i] = v[i];
  if (cmat && cmat_cap >= g.cmat.size()) std::memcpy(cmat, g.cmat.data(), g.cmat.size() * sizeof(float));
  if (koff && koff_cap >= g.koff.size()) std::memcpy(koff, g.koff.data(), g.koff.size() * sizeof(int));
  return g.mfma_per_sample;
}
}
