// CPU-only test shim for gnnb_pack.h (built with g++ by tests/test_pack_cpu.py): exposes the
// operand-order packs so a numpy emulation of the MFMA lane maps can be checked against a plain matmul.
#include "gnnb_pack.h"

extern "C" {
size_t gnnb_pt_blob_floats() { return gnnb::blob_floats(); }
// which: 0 embed, 1 pre_fwd, 2 upd_fwd, 3 pre_bwd, 4 upd_bwd, 5 pre_inp, 6 upd_inp, 7 score, 8 prop
size_t gnnb_pt_pack(const float* blob, int which, float* out, size_t cap) {
  gnnb::Packs pk;
  gnnb::build_packs(blob, pk);
  const std::vector<float>* v[9] = {&pk.embed, &pk.pre_fwd, &pk.upd_fwd, &pk.pre_bwd, &pk.upd_bwd,
                                    &pk.pre_inp, &pk.upd_inp, &pk.score, &pk.prop};
  if (which < 0 || which > 8) return 0;
  if (out && cap >= v[which]->size()) std::memcpy(out, v[which]->data(), v[which]->size() * sizeof(float));
  return v[which]->size();
}
}
