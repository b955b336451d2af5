// gnnb_pack.h -- host-side packing of the GNN's Linear layers into MFMA operand order.
//
// Pure C++ (no HIP): included by gnnb.hip and by the CPU-only pack test library, so the
// layout algebra below is checked on the CPU against a plain matmul (tests/test_pack_cpu.py).
//
// Register/lane conventions on gfx950 for v_mfma_f32_32x32x2_f32 (D = A*B + C, wave64):
//   A operand : lane l holds A[i = l&31][k = l>>5]
//   B operand : lane l holds B[k = l>>5][j = l&31]
//   C/D       : lane l, register r (0..15) holds D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
//
// The node MLPs run TRANSPOSED: out^T (64 features x 32 nodes) = W (64 x K) * in^T (K x 32 nodes).
// W is the A operand, activations are the B operand, so the node index sits on the lane (j) and a
// 64-feature activation vector of 32 nodes is a "fragment" of 32 registers per lane:
//   register R (0..31), lane half h = l>>5   <->   feature  f(R,h) = 8*(R>>2) + 4*h + (R&3)
// which is exactly how two stacked 32x32 D tiles (R = 16*it + r) come out of the MFMA, so the
// accumulators of one layer are the B operands of the next with no data movement: k-step s of the
// next layer takes register R = s of the fragment; half h of the wave then contributes input feature
// f(s,h), and the A operand of that k-step must hold W[out][f(s,h)] -- the permutation baked in here.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <vector>

namespace gnnb {

constexpr int P = 64;  // embedding width (reference graph_score.py:9: GraphNet(2, 64))

// feature index held by fragment register R in lane half h
inline int frag_feature(int R, int h) { return 8 * (R >> 2) + 4 * h + (R & 3); }

// ---- the 26 Linear layers in state-dict order ----
enum LayerId {
  L_INP_F, L_INP_F_1, L_INP_B, L_INP_B_1, L_INP_B2, L_INP_B2_2, L_FC1, L_FC1_1, L_FC3, L_FC3_2,
  L_FC4, L_FC4_2, L_OUT1, L_OUT2, L_OUT3, L_BC1, L_BC1_1, L_BC1_2, L_BC2, L_BC2_1, L_BC3, L_BC3_1,
  L_BC4, L_BC4_1, L_FNODE, L_FSCORE, L_COUNT
};
struct LinDef { int out, in; };
static const LinDef kLin[L_COUNT] = {
    {64, 3},  {64, 64}, {64, 2},  {64, 64},  {64, 128}, {64, 64}, {64, 7},   {64, 64}, {64, 128},
    {64, 64}, {64, 128}, {64, 64}, {64, 4},  {64, 128}, {64, 64}, {64, 7},   {64, 64}, {64, 64},
    {64, 192}, {64, 64}, {64, 128}, {64, 64}, {64, 128}, {64, 64}, {64, 64}, {1, 64}};

inline size_t blob_floats() {
  size_t n = 0;
  for (int i = 0; i < L_COUNT; ++i) n += (size_t)kLin[i].out * kLin[i].in + kLin[i].out;
  return n;  // 117825
}
inline size_t weight_offset(int id) {
  size_t n = 0;
  for (int i = 0; i < id; ++i) n += (size_t)kLin[i].out * kLin[i].in + kLin[i].out;
  return n;
}
inline size_t bias_offset(int id) { return weight_offset(id) + (size_t)kLin[id].out * kLin[id].in; }

// ---- operand-order packing ----
// 64 x (64*nfrag) block of W (row stride ldw, starting at column col0), A-operand order for
// ds_read_b128: float index = (((s>>2)*2 + it)*64 + lane)*4 + (s&3), k-step s in [0, 32*nfrag),
// value = W[32*it + (lane&31)][col0 + 64*(s>>5) + f(s&31, lane>>5)].
inline void pack_w64(float* dst, const float* W, int ldw, int col0, int nfrag) {
  for (int s = 0; s < 32 * nfrag; ++s)
    for (int it = 0; it < 2; ++it)
      for (int lane = 0; lane < 64; ++lane) {
        int out = 32 * it + (lane & 31);
        int in = col0 + 64 * (s >> 5) + frag_feature(s & 31, lane >> 5);
        dst[(((size_t)(s >> 2) * 2 + it) * 64 + lane) * 4 + (s & 3)] = W[(size_t)out * ldw + in];
      }
}
inline size_t w64_floats(int nfrag) { return (size_t)4096 * nfrag; }

// First layers on scalar node features (K = 2,3,4,7): natural order, k-step s holds input feature
// 2*s + h (zero-padded to 2*ksteps); float index = (s*2 + it)*64 + lane  (ds_read_b32).
inline void pack_wsmall(float* dst, const float* W, int K, int ksteps) {
  for (int s = 0; s < ksteps; ++s)
    for (int it = 0; it < 2; ++it)
      for (int lane = 0; lane < 64; ++lane) {
        int out = 32 * it + (lane & 31);
        int in = 2 * s + (lane >> 5);
        dst[((size_t)s * 2 + it) * 64 + lane] = in < K ? W[(size_t)out * K + in] : 0.0f;
      }
}
inline size_t wsmall_floats(int ksteps) { return (size_t)ksteps * 128; }

// 64-vector (bias, or the 1x64 score weight) in fragment order: index h*32 + R = v[f(R,h)].
inline void pack_vec64(float* dst, const float* v) {
  for (int h = 0; h < 2; ++h)
    for (int R = 0; R < 32; ++R) dst[h * 32 + R] = v[frag_feature(R, h)];
}

// W (out x in, row-major) -> W^T (in x out): lane = out index reads consecutive floats.
inline void pack_transposed(float* dst, const float* W, int out, int in) {
  for (int o = 0; o < out; ++o)
    for (int i = 0; i < in; ++i) dst[(size_t)i * out + o] = W[(size_t)o * in + i];
}

// ---- per-kernel weight packs: float offsets inside each pack (LDS image == global image) ----
// k_embed: mu0 = inp_f_1(relu(inp_f([l0, x, u0])))                       (graph_conv.py:90-95)
struct PackEmbed { enum { W1 = 0, B1 = W1 + 256, W2 = B1 + 64, B2 = W2 + 4096, FLOATS = B2 + 64 }; };
// k_pre_fwd: P = fc4[:, :64] (fc1_1(relu(fc1 feat7)) * amb) + fc4.bias    (:153-161, :176-177)
struct PackPreFwd { enum { W1 = 0, B1 = W1 + 512, W2 = B1 + 64, B2 = W2 + 4096, W3 = B2 + 64, B3 = W3 + 4096, FLOATS = B3 + 64 }; };
// k_node_update (forward: fc3, fc3_2, fc4[:, 64:], fc4_2; backward: bc3, bc3_1, bc4[:, 64:], bc4_1)
struct PackUpd { enum { WA = 0, BA = WA + 8192, WB = BA + 64, BB = WB + 4096, WC = BB + 64, WD = WC + 4096, BD = WD + 4096, FLOATS = BD + 64 }; };
// k_pre_bwd: P = bc4[:, :64] (bc2_1(relu(bc2([s, -d2 s, d1 s]))) * amb) + bc4.bias,
//            s = bc1_2(relu(bc1_1(relu(bc1 feat7'))))                      (:273-293, :344-345)
struct PackPreBwd {
  enum { W1 = 0, B1 = W1 + 512, W2 = B1 + 64, B2 = W2 + 4096, W3 = B2 + 64, B3 = W3 + 4096, W4 = B3 + 64,
         B4 = W4 + 12288, W5 = B4 + 64, B5 = W5 + 4096, W6 = B5 + 64, B6 = W6 + 4096, FLOATS = B6 + 64 };
};
// k_pre_inp: Q = inp_b2[:, :64] inp_b_1(relu(inp_b([l0,u0]))) + inp_b2.bias   (:380-384)
struct PackPreInp { enum { W1 = 0, B1 = W1 + 128, W2 = B1 + 64, B2 = W2 + 4096, W3 = B2 + 64, B3 = W3 + 4096, FLOATS = B3 + 64 }; };
// k_input_update: mu0 = inp_b2_2(relu(Q + inp_b2[:, 64:] nb))               (:383-385)
struct PackUpdInp { enum { WC = 0, WD = WC + 4096, BD = WD + 4096, FLOATS = BD + 64 }; };
// k_score: fscore(relu(fnode(mu)))                                           (:448-449)
struct PackScore { enum { W1 = 0, B1 = W1 + 4096, WS = B1 + 64, BS = WS + 64, FLOATS = BS + 4 }; };
// k_prop_fwd (VALU, one wave per sample): transposed row-major copies          (:196-210)
struct PackProp { enum { W1T = 0, B1 = W1T + 4 * 64, W2T = B1 + 64, B2 = W2T + 128 * 64, W3T = B2 + 64, B3 = W3T + 64 * 64, FLOATS = B3 + 64 }; };

struct Packs {
  std::vector<float> embed, pre_fwd, upd_fwd, pre_bwd, upd_bwd, pre_inp, upd_inp, score, prop;
};

inline void build_packs(const float* blob, Packs& pk) {
  auto W = [&](int id) { return blob + weight_offset(id); };
  auto Bv = [&](int id) { return blob + bias_offset(id); };
  pk.embed.assign(PackEmbed::FLOATS, 0.f);
  pack_wsmall(&pk.embed[PackEmbed::W1], W(L_INP_F), 3, 2);
  pack_vec64(&pk.embed[PackEmbed::B1], Bv(L_INP_F));
  pack_w64(&pk.embed[PackEmbed::W2], W(L_INP_F_1), 64, 0, 1);
  pack_vec64(&pk.embed[PackEmbed::B2], Bv(L_INP_F_1));

  pk.pre_fwd.assign(PackPreFwd::FLOATS, 0.f);
  pack_wsmall(&pk.pre_fwd[PackPreFwd::W1], W(L_FC1), 7, 4);
  pack_vec64(&pk.pre_fwd[PackPreFwd::B1], Bv(L_FC1));
  pack_w64(&pk.pre_fwd[PackPreFwd::W2], W(L_FC1_1), 64, 0, 1);
  pack_vec64(&pk.pre_fwd[PackPreFwd::B2], Bv(L_FC1_1));
  pack_w64(&pk.pre_fwd[PackPreFwd::W3], W(L_FC4), 128, 0, 1);
  pack_vec64(&pk.pre_fwd[PackPreFwd::B3], Bv(L_FC4));

  auto upd = [&](std::vector<float>& v, int a, int b, int c, int d) {
    v.assign(PackUpd::FLOATS, 0.f);
    pack_w64(&v[PackUpd::WA], W(a), 128, 0, 2);
    pack_vec64(&v[PackUpd::BA], Bv(a));
    pack_w64(&v[PackUpd::WB], W(b), 64, 0, 1);
    pack_vec64(&v[PackUpd::BB], Bv(b));
    pack_w64(&v[PackUpd::WC], W(c), 128, 64, 1);
    pack_w64(&v[PackUpd::WD], W(d), 64, 0, 1);
    pack_vec64(&v[PackUpd::BD], Bv(d));
  };
  upd(pk.upd_fwd, L_FC3, L_FC3_2, L_FC4, L_FC4_2);
  upd(pk.upd_bwd, L_BC3, L_BC3_1, L_BC4, L_BC4_1);

  pk.pre_bwd.assign(PackPreBwd::FLOATS, 0.f);
  pack_wsmall(&pk.pre_bwd[PackPreBwd::W1], W(L_BC1), 7, 4);
  pack_vec64(&pk.pre_bwd[PackPreBwd::B1], Bv(L_BC1));
  pack_w64(&pk.pre_bwd[PackPreBwd::W2], W(L_BC1_1), 64, 0, 1);
  pack_vec64(&pk.pre_bwd[PackPreBwd::B2], Bv(L_BC1_1));
  pack_w64(&pk.pre_bwd[PackPreBwd::W3], W(L_BC1_2), 64, 0, 1);
  pack_vec64(&pk.pre_bwd[PackPreBwd::B3], Bv(L_BC1_2));
  pack_w64(&pk.pre_bwd[PackPreBwd::W4], W(L_BC2), 192, 0, 3);
  pack_vec64(&pk.pre_bwd[PackPreBwd::B4], Bv(L_BC2));
  pack_w64(&pk.pre_bwd[PackPreBwd::W5], W(L_BC2_1), 64, 0, 1);
  pack_vec64(&pk.pre_bwd[PackPreBwd::B5], Bv(L_BC2_1));
  pack_w64(&pk.pre_bwd[PackPreBwd::W6], W(L_BC4), 128, 0, 1);
  pack_vec64(&pk.pre_bwd[PackPreBwd::B6], Bv(L_BC4));

  pk.pre_inp.assign(PackPreInp::FLOATS, 0.f);
  pack_wsmall(&pk.pre_inp[PackPreInp::W1], W(L_INP_B), 2, 1);
  pack_vec64(&pk.pre_inp[PackPreInp::B1], Bv(L_INP_B));
  pack_w64(&pk.pre_inp[PackPreInp::W2], W(L_INP_B_1), 64, 0, 1);
  pack_vec64(&pk.pre_inp[PackPreInp::B2], Bv(L_INP_B_1));
  pack_w64(&pk.pre_inp[PackPreInp::W3], W(L_INP_B2), 128, 0, 1);
  pack_vec64(&pk.pre_inp[PackPreInp::B3], Bv(L_INP_B2));

  pk.upd_inp.assign(PackUpdInp::FLOATS, 0.f);
  pack_w64(&pk.upd_inp[PackUpdInp::WC], W(L_INP_B2), 128, 64, 1);
  pack_w64(&pk.upd_inp[PackUpdInp::WD], W(L_INP_B2_2), 64, 0, 1);
  pack_vec64(&pk.upd_inp[PackUpdInp::BD], Bv(L_INP_B2_2));

  pk.score.assign(PackScore::FLOATS, 0.f);
  pack_w64(&pk.score[PackScore::W1], W(L_FNODE), 64, 0, 1);
  pack_vec64(&pk.score[PackScore::B1], Bv(L_FNODE));
  pack_vec64(&pk.score[PackScore::WS], W(L_FSCORE));
  pk.score[PackScore::BS] = Bv(L_FSCORE)[0];

  pk.prop.assign(PackProp::FLOATS, 0.f);
  pack_transposed(&pk.prop[PackProp::W1T], W(L_OUT1), 64, 4);
  std::memcpy(&pk.prop[PackProp::B1], Bv(L_OUT1), 64 * sizeof(float));
  pack_transposed(&pk.prop[PackProp::W2T], W(L_OUT2), 64, 128);
  std::memcpy(&pk.prop[PackProp::B2], Bv(L_OUT2), 64 * sizeof(float));
  pack_transposed(&pk.prop[PackProp::W3T], W(L_OUT3), 64, 64);
  std::memcpy(&pk.prop[PackProp::B3], Bv(L_OUT3), 64 * sizeof(float));
}

// ---- verified-network (layer graph) descriptors ----
struct Edge {          // linear map between graph layer k-1 and k
  int kind;            // 0 conv, 1 linear
  int c_in, h_in, w_in, c_out, h_out, w_out, kh, kw, stride, pad;
  int n_in, n_out;
  std::vector<float> w, b;   // torch layout
};

// conv weight [co][ci][ky][kx] -> [ci][ky][kx][co]  (forward gather: scalar loads of CO weights per tap)
inline void pack_conv_fwd(float* dst, const Edge& e) {
  for (int co = 0; co < e.c_out; ++co)
    for (int ci = 0; ci < e.c_in; ++ci)
      for (int ky = 0; ky < e.kh; ++ky)
        for (int kx = 0; kx < e.kw; ++kx)
          dst[(((size_t)ci * e.kh + ky) * e.kw + kx) * e.c_out + co] =
              e.w[(((size_t)co * e.c_in + ci) * e.kh + ky) * e.kw + kx];
}
// conv weight -> [co][ky][kx][ci]  (transposed gather: CI weights per tap)
inline void pack_conv_bwd(float* dst, const Edge& e) {
  for (int co = 0; co < e.c_out; ++co)
    for (int ci = 0; ci < e.c_in; ++ci)
      for (int ky = 0; ky < e.kh; ++ky)
        for (int kx = 0; kx < e.kw; ++kx)
          dst[(((size_t)co * e.kh + ky) * e.kw + kx) * e.c_in + ci] =
              e.w[(((size_t)co * e.c_in + ci) * e.kh + ky) * e.kw + kx];
}

}  // namespace gnnb
