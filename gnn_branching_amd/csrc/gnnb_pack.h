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
#include <cstdlib>
#include <cstring>
#include <vector>
#include <thread>
#include <system_error>

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

// ---- the same block for v_mfma_f32_32x32x16_bf16 with every weight split into three bf16 pieces, w = w1 + w2 + w3 (round
// to nearest each time: 24 mantissa bits together).  The kernel splits the activations the same way and sums the six
// products of total order <= 4 in the fp32 accumulator -- fp32 accuracy (tools/micro/bf16x3_chain.hip: same error against
// fp64 as the fp32 MFMA) at a quarter of the matrix-pipe time.  k-step ks of input fragment f covers fragment registers
// 8 ks .. 8 ks + 7, i.e. features 64 f + frag_feature(8 ks + j, h): entry ((f*4 + ks)*2 + ot)*3 + piece holds, for lane
// (r, h), the 8 bf16 of W[32 ot + r][col0 + 64 f + frag_feature(8 ks + j, h)], j = 0..7 (16 bytes: one ds_read_b128).
inline unsigned short bf16_rne(float f) {
  unsigned u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7f800000u) == 0x7f800000u) return (unsigned short)((u >> 16) | ((u & 0xffffu) ? 0x40u : 0u));   // inf / nan
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
inline float bf16_f32(unsigned short b) {
  const unsigned u = (unsigned)b << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
inline size_t w64_bf3_floats(int nfrag) { return (size_t)6144 * nfrag; }
// three bf16 pieces of 512 weights: o[p * 512 + i] = piece p of w[i] (bf16_rne's rounding, branch-free so that it vectorises)
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
__attribute__((target_clones("avx512f", "avx2", "default")))
#endif
inline void bf3_split_512(unsigned short* o, const float* w) {
  for (int i = 0; i < 512; ++i) {
    float x = w[i];
    for (int p = 0; p < 3; ++p) {
      unsigned u;
      std::memcpy(&u, &x, 4);
      const unsigned special = (u >> 16) | ((u & 0xffffu) ? 0x40u : 0u);                 // inf / nan
      const unsigned rounded = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
      const unsigned b = ((u & 0x7f800000u) == 0x7f800000u) ? special : rounded;
      o[p * 512 + i] = (unsigned short)b;
      const unsigned back = b << 16;
      float bf;
      std::memcpy(&bf, &back, 4);
      x -= bf;
    }
  }
}
inline void pack_w64_bf3(float* dst, const float* W, int ldw, int col0, int nfrag) {
  // entry e = ((ks * 2 + ot) * 64 + lane) * 8 + j of a fragment reads W[row[e]][col0 + 64 f + col[e]]; its pieces go to
  // ((f*4 + ks)*2 + ot)*3*512 + p*512 + lane*8 + j: 512 consecutive entries share (ks, ot), so a block of 512 is gathered through
  // the table and split in one vectorised pass (this runs 25 fragment-times after every online-learning step: 22 -> 6 us each)
  struct Tab { unsigned char row[4096], col[4096]; };
  static const Tab tab = [] {
    Tab t;
    for (int ks = 0; ks < 4; ++ks)
      for (int ot = 0; ot < 2; ++ot)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 8; ++j) {
            const int e = ((ks * 2 + ot) * 64 + lane) * 8 + j;
            t.row[e] = (unsigned char)(32 * ot + (lane & 31));
            t.col[e] = (unsigned char)frag_feature(8 * ks + j, lane >> 5);
          }
    return t;
  }();
  unsigned short* d = reinterpret_cast<unsigned short*>(dst);
  float tmp[512];
  for (int f = 0; f < nfrag; ++f)
    for (int c = 0; c < 8; ++c) {
      for (int i = 0; i < 512; ++i) tmp[i] = W[(size_t)tab.row[c * 512 + i] * ldw + col0 + 64 * f + tab.col[c * 512 + i]];
      bf3_split_512(d + (size_t)f * 12288 + (size_t)c * 1536, tmp);
    }
}

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

inline int gather_feature(int R, int h) {           // channel held by register R = 16*it + r after a gather
  const int it = R >> 4, r = R & 15;
  return 2 * ((r & 3) + 8 * (r >> 2) + 4 * h) + it;
}
// W (out x in, row-major) -> W^T (in x out): lane = out index reads consecutive floats.
inline void pack_transposed(float* dst, const float* W, int out, int in) {
  for (int o = 0; o < out; ++o)
    for (int i = 0; i < in; ++i) dst[(size_t)i * out + o] = W[(size_t)o * in + i];
}

// ---- per-kernel weight packs: float offsets inside each pack (LDS image == global image) ----
// Deferred projection (exact algebra).  Every producer ends in a plain Linear, mu = (Wp.E + bp).live
// (live = 1 on the input layer).  It stores E.live instead; the edge aggregate commutes with Wp,
//   sum_n A[n',n] mu_n = Wp.(sum_n A[n',n] E_n) + bp.s[n'],   s[n'] = sum_n A[n',n] live_n   (k_livesum, once per forward),
// so Wp is folded into the FIRST layer of the consumer: Wa.[r0 nb, r1 nb] = (Wa0.Wp).(r0 G) + (Wa1.Wp).(r1 G) +
// s.(r0.Wa0.bp + r1.Wa1.bp) with G the aggregate of the stored rows.  Every node update thereby loses its last 64x64
// GEMM; the consumers' GEMM count is unchanged (the s-term is one small k-step, 2 MFMAs).
// k_embed: E0 = relu(inp_f([l0, x, u0])); mu0 = inp_f_1(E0) is deferred       (graph_conv.py:90-95)
// (VALU kernel: plain row-major inp_f.weight (64 x 3) and bias)
struct PackEmbed { enum { W = 0, B = W + 192, FLOATS = B + 64 }; };
// k_pre, forward chain: P' = fc4[:, :64] (fc1_1(relu(fc1 feat7)) * amb) + bcb    (:153-161, :176-177)
// fc1_1 feeds fc4 linearly, so W2 = fc4[:, :64].fc1_1.W (one GEMM instead of two), B2 = fc4[:, :64].fc1_1.b + bcb
struct PackPreFwd { enum { W1 = 0, B1 = W1 + 512, W2 = B1 + 64, B2 = W2 + 4096, FLOATS = B2 + 64, W23 = FLOATS /* W2 as bf16 x 3 */,
                           FLOATS3 = W23 + 6144 }; };
struct PackPreFwdL3 { enum { W1 = 0, B1 = W1 + 512, B2 = B1 + 64, W23 = B2 + 64, FLOATS = W23 + 6144 }; };      // LDS image, bf16 x 3 form
// k_node_update (forward: fc3, fc3_2, fc4[:, 64:], fc4_2; backward: bc3, bc3_1, bc4[:, 64:], bc4_1)
// Folded form (exact algebra): e = Wb.h + bb enters the next layer linearly, so
//   relu(W4.[relax, e] + b4) = relu(P' + Wcb.h),  Wcb = W4[:, 64:].Wb,  P' = W4[:, :64].relax + b4 + W4[:, 64:].bb
// and for nodes with r0 == r1 (every live node that is not ambiguous)  Wa.[r0 x, r1 x] = WAS.(r0 x),  WAS = Wa[:, :64] + Wa[:, 64:].
//   WA  : Wa, 128 -> 64 (general nodes)      WAS : summed halves, 64 -> 64 (r0 == r1 nodes)      BA : bias of Wa
//   WCB : Wcb                                  BCB : b4 + W4[:, 64:].bb  (= P' of a node without relaxation term)
//   BCBROW : BCB again, row-major (read like a P' row)
//   VAW : the 64 x 2 matrix [Wa[:, :64].bp, Wa[:, 64:].bp] as one small k-step -- bias terms of the deferred projection of
//         the SOURCE rows: H += VAW.[r0 s, r1 s] costs 2 MFMAs and no registers (zero when the aggregate is final)
// The last layer (fc4_2 / bc4_1) is not here: it is deferred into whoever consumes the rows this kernel writes.
struct PackUpd { enum { WA = 0, WAS = WA + 8192, BA = WAS + 4096, WCB = BA + 64, BCB = WCB + 4096, BCBROW = BCB + 64,
                        VAW = BCBROW + 64, FLOATS = VAW + 128,
                        // bf16 x 3 forms of WAS and WCB (pack_w64_bf3) behind the fp32 image
                        WAS3 = FLOATS, WCB3 = WAS3 + 6144, FLOATS3 = WCB3 + 6144,
                        // the second half of Wa alone (bf16 x 3): see PackUpdL3
                        WA1S3 = FLOATS3, FLOATS_ALL = WA1S3 + 6144 }; };
// LDS image of the bf16 x 3 node update (k_node_update, k_gather_update_q): three 64x64 blocks in three bf16 pieces each -- WAS
// (nodes with r0 == r1), WCB, and the second half of Wa: a general node goes through
//   Wa.[r0 x, r1 x] = WAS.(r0 x) + Wa[:, 64:].((r1 - r0) x)
// (the 128-wide fp32 first layer it replaced cost 8192 matrix-pipe cycles per tile against 3072; and the same arithmetic per
// node whether its tile holds ambiguous nodes or not is what makes the fused and the two-kernel half-pass bit-identical)
struct PackUpdL3 { enum { BA = 0, BCB = BA + 64, BCBROW = BCB + 64, VAW = BCBROW + 64, WAS3 = VAW + 128, WCB3 = WAS3 + 6144, WA1S3 = WCB3 + 6144,
                          FLOATS = WA1S3 + 6144 }; };
// k_pre_bwd: P = bc4[:, :64] (bc2_1(relu(bc2([s, -d2 s, d1 s]))) * amb) + bc4.bias,
//            s = bc1_2(relu(bc1_1(relu(bc1 feat7'))))                      (:273-293, :344-345)
// likewise W5 = bc4[:, :64].bc2_1.W, B5 = bc4[:, :64].bc2_1.b + bcb
struct PackPreBwd {
  enum { W1 = 0, B1 = W1 + 512, W2 = B1 + 64, B2 = W2 + 4096, W3 = B2 + 64, B3 = W3 + 4096, W4 = B3 + 64,
         B4 = W4 + 12288, W5 = B4 + 64, B5 = W5 + 4096, FLOATS = B5 + 64,
         W23 = FLOATS, W33 = W23 + 6144, W53 = W33 + 6144, W43 = W53 + 6144, FLOATS3 = W43 + 18432 };   // W2, W3, W5 and the 192-wide W4 as bf16 x 3
};
// LDS image, bf16 x 3 form.  Its first PackPreFwdL3::FLOATS floats (W1, B1, B2, W23) share their place with the forward image:
// k_pre runs the forward tiles first, with everything behind HEAD already in place, then drops these four pieces in.
// (W4 as fp32 MFMAs was 12.3 k of a backward tile's 17 k matrix-pipe cycles; as three bf16 x 3 blocks it is 4.6 k, and the
// image no longer fits beside the forward one: 147 KB.)
struct PackPreBwdL3 {
  enum { W1 = 0, B1 = W1 + 512, B2 = B1 + 64, W23 = B2 + 64, HEAD = W23 + 6144, B3 = HEAD, B4 = B3 + 64, B5 = B4 + 64, W33 = B5 + 64,
         W43 = W33 + 6144, W53 = W43 + 18432, FLOATS = W53 + 6144 };
};
static_assert((int)PackPreBwdL3::HEAD == (int)PackPreFwdL3::FLOATS, "k_pre: the forward image takes the head of the backward image");
// k_pre_inp: Q = inp_b2[:, :64] inp_b_1(relu(inp_b([l0,u0]))) + inp_b2.bias   (:380-384)
// folded: Q = (inp_b2[:, :64].inp_b_1.W) relu(inp_b([l0,u0])) + (inp_b2[:, :64].inp_b_1.b + inp_b2.b)
struct PackPreInp { enum { W1 = 0, B1 = W1 + 128, W2 = B1 + 64, B2 = W2 + 4096, W23 = B2 + 64 /* W2 as bf16 x 3 */, FLOATS = W23 + 6144 }; };
// k_input_update: E0 = relu(Q + inp_b2[:, 64:] nb); mu0 = inp_b2_2(E0) is deferred   (:383-385)
// The 64x64 map inp_b2[:, 64:].bc4_1.W that the aggregate of layer 1 has to go through commutes with the aggregation, and
// layer 1 has ~1100 live producer nodes per sample against 3072 consumer nodes here: it is applied on the PRODUCER side
// (PackPostInp, the last backward update of layer 1 before an input update) and the input kernels just add the aggregate.
// VC = [inp_b2[:, 64:].bc4_1.b, 0] small k-step (bias term of the deferred bc4_1, times the bias-sum scalar)
struct PackUpdInp { enum { VC = 0, FLOATS = VC + 128 }; };
// WPN: inp_b2[:, 64:].bc4_1.W in pack_w64 order, rows natural (the flat input update reads the aggregate row-major);
// WPG: the same with output rows permuted so that, after the MFMA gather, register R of lane half h (gather channel
//      gather_feature(R, h)) holds feature frag_feature(R, h) -- the fragment layout of the chain it is added to.
struct PackPostInp { enum { WPN = 0, WPG = WPN + 4096, WPN3 = WPG + 4096, WPG3 = WPN3 + 6144, FLOATS = WPG3 + 6144 }; };   // ..3: bf16 x 3
// k_score: fscore(relu(fnode(mu)))                                           (:448-449)
// W1 = fnode.Wp, V1 = [fnode.bp, 0] small k-step fed with live (the scored rows have their last Linear Wp deferred)
struct PackScore { enum { W1 = 0, B1 = W1 + 4096, WS = B1 + 64, BS = WS + 64, V1 = BS + 4, FLOATS = V1 + 128 }; };
// k_prop_fwd (VALU, one wave per sample): transposed row-major copies          (:196-210)
// the second half of out2 is folded with fc4_2 (deferred in the rows of mu_L): W2T rows 64.. = (out2[:, 64:].fc4_2.W)^T,
// V2 = out2[:, 64:].fc4_2.b multiplies sum_n W_prop[n] live_n
struct PackProp { enum { W1T = 0, B1 = W1T + 4 * 64, W2T = B1 + 64, B2 = W2T + 128 * 64, W3T = B2 + 64, B3 = W3T + 64 * 64, V2 = B3 + 64, FLOATS = V2 + 64 }; };

// The folds run on the host after every online-learning step (gnnb_online_step -> load_weights): their double-precision inner
// loops are compiled for AVX-512 / AVX2 as well and picked at load time (function multiversioning; host pass only -- the device
// pass of hipcc sees plain functions).  Same sums in the same order whichever clone runs.
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
#define GNNB_HOST_SIMD __attribute__((target_clones("avx512f", "avx2", "default")))
#else
#define GNNB_HOST_SIMD
#endif
// Cd (64 x 64, double) = A (64 x 64, row stride lda, columns [acol0, acol0+64)) . B (64 x 64 row-major)
GNNB_HOST_SIMD inline void matmul64d(double* Cd, const float* A, int lda, int acol0, const float* B) {
  alignas(64) double Bd[64 * 64];          // B widened once (the conversions, not the fmas, bounded the plain loop)
  for (int q = 0; q < 64 * 64; ++q) Bd[q] = (double)B[q];
  for (int i = 0; i < 64; i += 2) {        // two rows share every load of B; per row: k ascending, one fma per term
    alignas(64) double acc0[64], acc1[64];
    for (int j = 0; j < 64; ++j) { acc0[j] = 0.0; acc1[j] = 0.0; }
    const float* A0 = A + (size_t)i * lda + acol0;
    const float* A1 = A0 + lda;
    for (int k = 0; k < 64; ++k) {
      const double a0 = (double)A0[k], a1 = (double)A1[k];
      const double* Bk = Bd + (size_t)k * 64;
      for (int j = 0; j < 64; ++j) {
        acc0[j] += a0 * Bk[j];
        acc1[j] += a1 * Bk[j];
      }
    }
    for (int j = 0; j < 64; ++j) { Cd[(size_t)i * 64 + j] = acc0[j]; Cd[(size_t)(i + 1) * 64 + j] = acc1[j]; }
  }
}
// C (64 x 64) = the same product rounded to float once
inline void matmul64(float* C, const float* A, int lda, int acol0, const float* B) {
  double t[64 * 64];
  matmul64d(t, A, lda, acol0, B);
  for (int q = 0; q < 64 * 64; ++q) C[q] = (float)t[q];
}
// y (64) = A[:, acol0:acol0+64] . x (64) + y0 (64)
inline void matvec64(float* y, const float* A, int lda, int acol0, const float* x, const float* y0) {
  for (int i = 0; i < 64; ++i) {
    double acc = y0 ? (double)y0[i] : 0.0;
    for (int k = 0; k < 64; ++k) acc += (double)A[(size_t)i * lda + acol0 + k] * (double)x[k];
    y[i] = (float)acc;
  }
}

struct Packs {
  std::vector<float> embed, pre_fwd, pre_bwd, pre_inp, prop;
  // node updates by the projection deferred in the rows they aggregate:
  std::vector<float> upd_fwd_e, upd_fwd_i, upd_fwd_f;   // forward: inp_f_1 (layer 1, round 0) / inp_b2_2 (layer 1, later) / fc4_2
  std::vector<float> upd_bwd, upd_bwd_b;                // backward: none (top layer: aggregate from the property node) / bc4_1
  std::vector<float> upd_inp, post_inp;                 // input layer: bias small k-step; the producer-side map (PackPostInp)
  std::vector<float> score_b, score_f;                  // score head on rows with bc4_1 / fc4_2 deferred
  // work space of build_packs, one per update pack (they may be built side by side), kept across calls: allocating and
  // releasing these 190 KB per pack on every rebuild cost more in page faults than the arithmetic
  struct Scratch { std::vector<float> was, wcb, wa; std::vector<double> t0, t1; };
  Scratch scratch[5];
};

inline void build_packs(const float* blob, Packs& pk) {
  auto W = [&](int id) { return blob + weight_offset(id); };
  auto Bv = [&](int id) { return blob + bias_offset(id); };
  pk.embed.assign(PackEmbed::FLOATS, 0.f);
  std::memcpy(&pk.embed[PackEmbed::W], W(L_INP_F), 192 * sizeof(float));
  std::memcpy(&pk.embed[PackEmbed::B], Bv(L_INP_F), 64 * sizeof(float));

  pk.pre_fwd.assign(PackPreFwd::FLOATS3, 0.f);
  pack_wsmall(&pk.pre_fwd[PackPreFwd::W1], W(L_FC1), 7, 4);
  pack_vec64(&pk.pre_fwd[PackPreFwd::B1], Bv(L_FC1));

  // folded bias of the update chain: bcb = b_c + W_c[:, 64:].b_b  (also added to the P' the feature chains cache)
  auto bcb_of = [&](int b, int c, float* out) { matvec64(out, W(c), 128, 64, Bv(b), Bv(c)); };
  // Wcb = W_c[:, 64:].W_b and bcb are the same for every update pack of a direction: folded once each, before the packs
  struct CbFold { int b, c; std::vector<float> w; float bias[64]; };
  CbFold cb_fold[2] = {{L_FC3_2, L_FC4, std::vector<float>(64 * 64), {}}, {L_BC3_1, L_BC4, std::vector<float>(64 * 64), {}}};
  for (CbFold& f : cb_fold) {
    matmul64(f.w.data(), W(f.c), 128, 64, W(f.b));
    bcb_of(f.b, f.c, f.bias);
  }
  auto fold_cb = [&](int b, int c, float* wcb, float* bcb) {
    const CbFold& f = cb_fold[(b == cb_fold[0].b && c == cb_fold[0].c) ? 0 : 1];
    std::memcpy(wcb, f.w.data(), sizeof(float) * 64 * 64);
    std::memcpy(bcb, f.bias, sizeof(float) * 64);
  };
  // proj >= 0: the aggregate this update reads is built from rows whose projection Linear `proj` is deferred
  auto upd = [&](std::vector<float>& v, Packs::Scratch& sc, int a, int b, int c, int d, int proj = -1) {
    v.assign(PackUpd::FLOATS_ALL, 0.f);
    std::vector<float>&was = sc.was, &wcb = sc.wcb, &wa = sc.wa;
    was.resize(64 * 64); wcb.resize(64 * 64); wa.resize(64 * 128);
    float bcb[64];
    std::memcpy(wa.data(), W(a), sizeof(float) * 64 * 128);
    if (proj >= 0) {
      std::vector<double>&t0 = sc.t0, &t1 = sc.t1;           // Wa[:, :64].Wp and Wa[:, 64:].Wp in double
      t0.resize(64 * 64); t1.resize(64 * 64);
      float vaw[128];
      matmul64d(t0.data(), W(a), 128, 0, W(proj));
      matmul64d(t1.data(), W(a), 128, 64, W(proj));
      const float* Wa = W(a);                 // (weight_offset walks the layer table: not inside the loops)
      const float* bp = Bv(proj);
      for (int i = 0; i < 64; ++i) {
        double s0 = 0.0, s1 = 0.0;
        for (int k = 0; k < 64; ++k) {
          s0 += (double)Wa[i * 128 + k] * (double)bp[k];
          s1 += (double)Wa[i * 128 + 64 + k] * (double)bp[k];
        }
        vaw[2 * i] = (float)s0; vaw[2 * i + 1] = (float)s1;
        for (int j = 0; j < 64; ++j) {
          was[i * 64 + j] = (float)(t0[i * 64 + j] + t1[i * 64 + j]);
          wa[i * 128 + j] = (float)t0[i * 64 + j];
          wa[i * 128 + 64 + j] = (float)t1[i * 64 + j];
        }
      }
      pack_wsmall(&v[PackUpd::VAW], vaw, 2, 1);
    } else {
      const float* Wa = W(a);
      for (int i = 0; i < 64; ++i)
        for (int k = 0; k < 64; ++k) was[i * 64 + k] = (float)((double)Wa[i * 128 + k] + (double)Wa[i * 128 + 64 + k]);
    }
    fold_cb(b, c, wcb.data(), bcb);
    pack_w64(&v[PackUpd::WA], wa.data(), 128, 0, 2);
    pack_w64(&v[PackUpd::WAS], was.data(), 64, 0, 1);
    pack_vec64(&v[PackUpd::BA], Bv(a));
    pack_w64(&v[PackUpd::WCB], wcb.data(), 64, 0, 1);
    pack_w64_bf3(&v[PackUpd::WAS3], was.data(), 64, 0, 1);
    pack_w64_bf3(&v[PackUpd::WCB3], wcb.data(), 64, 0, 1);
    pack_w64_bf3(&v[PackUpd::WA1S3], wa.data(), 128, 64, 1);
    pack_vec64(&v[PackUpd::BCB], bcb);
    (void)d;   // the last layer is folded into the consumers of the rows
    std::memcpy(&v[PackUpd::BCBROW], bcb, 64 * sizeof(float));
  };
  // The four update packs with a folded projection are two thirds of this function's work and independent of each other and of
  // the rest: two helper threads take two each while this one builds everything else (the rebuild tails every online-learning
  // step).  Without threads (creation refused) the same calls run here.
  auto upd_e_i = [&]() {
    upd(pk.upd_fwd_e, pk.scratch[0], L_FC3, L_FC3_2, L_FC4, L_FC4_2, L_INP_F_1);      // layer 1, round 0: mu0 comes from the embedding
    upd(pk.upd_fwd_i, pk.scratch[1], L_FC3, L_FC3_2, L_FC4, L_FC4_2, L_INP_B2_2);     // layer 1, later rounds: from the input-layer update
  };
  auto upd_f_b = [&]() {
    upd(pk.upd_fwd_f, pk.scratch[2], L_FC3, L_FC3_2, L_FC4, L_FC4_2, L_FC4_2);        // layers >= 2: from the forward update below
    upd(pk.upd_bwd_b, pk.scratch[3], L_BC3, L_BC3_1, L_BC4, L_BC4_1, L_BC4_1);        // below: from the backward update above
  };
  std::thread helper[2];
  bool threaded[2] = {false, false};
#ifndef GNNB_PACK_NO_THREADS
#ifdef GNNB_DEV
  static const bool use_threads = [] { const char* e = std::getenv("GNNB_PACK_THREADS"); return !e || std::atoi(e) != 0; }();
#else
  constexpr bool use_threads = true;
#endif
  if (use_threads) {
    try { helper[0] = std::thread(upd_e_i); threaded[0] = true; } catch (const std::system_error&) {}
    try { helper[1] = std::thread(upd_f_b); threaded[1] = true; } catch (const std::system_error&) {}
  }
#endif
  struct Join {
    std::thread* t; bool* on;
    ~Join() { for (int i = 0; i < 2; ++i) if (on[i]) t[i].join(); }
  } join{helper, threaded};
  if (!threaded[0]) upd_e_i();
  if (!threaded[1]) upd_f_b();
  upd(pk.upd_bwd, pk.scratch[4], L_BC3, L_BC3_1, L_BC4, L_BC4_1);                   // top layer: final aggregate from the property node
  {   // the feature chains cache P' = W4[:, :64].relax + bcb, relax = fc1_1(.) folded in
    float bcb[64], b2[64];
    std::vector<float> w2(64 * 64);
    bcb_of(L_FC3_2, L_FC4, bcb);
    matmul64(w2.data(), W(L_FC4), 128, 0, W(L_FC1_1));
    matvec64(b2, W(L_FC4), 128, 0, Bv(L_FC1_1), bcb);
    pack_w64(&pk.pre_fwd[PackPreFwd::W2], w2.data(), 64, 0, 1);
    pack_w64_bf3(&pk.pre_fwd[PackPreFwd::W23], w2.data(), 64, 0, 1);
    pack_vec64(&pk.pre_fwd[PackPreFwd::B2], b2);
  }

  pk.pre_bwd.assign(PackPreBwd::FLOATS3, 0.f);
  pack_wsmall(&pk.pre_bwd[PackPreBwd::W1], W(L_BC1), 7, 4);
  pack_vec64(&pk.pre_bwd[PackPreBwd::B1], Bv(L_BC1));
  pack_w64(&pk.pre_bwd[PackPreBwd::W2], W(L_BC1_1), 64, 0, 1);
  pack_w64_bf3(&pk.pre_bwd[PackPreBwd::W23], W(L_BC1_1), 64, 0, 1);
  pack_vec64(&pk.pre_bwd[PackPreBwd::B2], Bv(L_BC1_1));
  pack_w64(&pk.pre_bwd[PackPreBwd::W3], W(L_BC1_2), 64, 0, 1);
  pack_w64_bf3(&pk.pre_bwd[PackPreBwd::W33], W(L_BC1_2), 64, 0, 1);
  pack_vec64(&pk.pre_bwd[PackPreBwd::B3], Bv(L_BC1_2));
  pack_w64(&pk.pre_bwd[PackPreBwd::W4], W(L_BC2), 192, 0, 3);
  pack_w64_bf3(&pk.pre_bwd[PackPreBwd::W43], W(L_BC2), 192, 0, 3);
  pack_vec64(&pk.pre_bwd[PackPreBwd::B4], Bv(L_BC2));
  {
    float bcb[64], b5[64];
    std::vector<float> w5(64 * 64);
    bcb_of(L_BC3_1, L_BC4, bcb);
    matmul64(w5.data(), W(L_BC4), 128, 0, W(L_BC2_1));
    matvec64(b5, W(L_BC4), 128, 0, Bv(L_BC2_1), bcb);
    pack_w64(&pk.pre_bwd[PackPreBwd::W5], w5.data(), 64, 0, 1);
    pack_w64_bf3(&pk.pre_bwd[PackPreBwd::W53], w5.data(), 64, 0, 1);
    pack_vec64(&pk.pre_bwd[PackPreBwd::B5], b5);
  }

  pk.pre_inp.assign(PackPreInp::FLOATS, 0.f);
  pack_wsmall(&pk.pre_inp[PackPreInp::W1], W(L_INP_B), 2, 1);
  pack_vec64(&pk.pre_inp[PackPreInp::B1], Bv(L_INP_B));
  {
    std::vector<float> w2(64 * 64);
    float b2[64];
    matmul64(w2.data(), W(L_INP_B2), 128, 0, W(L_INP_B_1));
    matvec64(b2, W(L_INP_B2), 128, 0, Bv(L_INP_B_1), Bv(L_INP_B2));
    pack_w64(&pk.pre_inp[PackPreInp::W2], w2.data(), 64, 0, 1);
    pack_w64_bf3(&pk.pre_inp[PackPreInp::W23], w2.data(), 64, 0, 1);
    pack_vec64(&pk.pre_inp[PackPreInp::B2], b2);
  }

  pk.upd_inp.assign(PackUpdInp::FLOATS, 0.f);
  pk.post_inp.assign(PackPostInp::FLOATS, 0.f);
  {
    std::vector<float> wc(64 * 64), wcg(64 * 64);
    float vc[128], t[64];
    matmul64(wc.data(), W(L_INP_B2), 128, 64, W(L_BC4_1));
    matvec64(t, W(L_INP_B2), 128, 64, Bv(L_BC4_1), nullptr);
    for (int i = 0; i < 64; ++i) { vc[2 * i] = t[i]; vc[2 * i + 1] = 0.f; }
    pack_wsmall(&pk.upd_inp[PackUpdInp::VC], vc, 2, 1);
    pack_w64(&pk.post_inp[PackPostInp::WPN], wc.data(), 64, 0, 1);
    for (int h = 0; h < 2; ++h)
      for (int R = 0; R < 32; ++R) std::memcpy(&wcg[(size_t)gather_feature(R, h) * 64], &wc[(size_t)frag_feature(R, h) * 64], 64 * sizeof(float));
    pack_w64(&pk.post_inp[PackPostInp::WPG], wcg.data(), 64, 0, 1);
    pack_w64_bf3(&pk.post_inp[PackPostInp::WPN3], wc.data(), 64, 0, 1);
    pack_w64_bf3(&pk.post_inp[PackPostInp::WPG3], wcg.data(), 64, 0, 1);
  }

  auto score = [&](std::vector<float>& v, int proj) {
    v.assign(PackScore::FLOATS, 0.f);
    std::vector<float> w1(64 * 64);
    float t[64], v1[128];
    matmul64(w1.data(), W(L_FNODE), 64, 0, W(proj));
    matvec64(t, W(L_FNODE), 64, 0, Bv(proj), nullptr);
    for (int i = 0; i < 64; ++i) { v1[2 * i] = t[i]; v1[2 * i + 1] = 0.f; }
    pack_w64(&v[PackScore::W1], w1.data(), 64, 0, 1);
    pack_vec64(&v[PackScore::B1], Bv(L_FNODE));
    pack_vec64(&v[PackScore::WS], W(L_FSCORE));
    v[PackScore::BS] = Bv(L_FSCORE)[0];
    pack_wsmall(&v[PackScore::V1], v1, 2, 1);
  };
  score(pk.score_b, L_BC4_1);
  score(pk.score_f, L_FC4_2);      // only reached when a half-pass limit stops the forward after a forward sweep

  pk.prop.assign(PackProp::FLOATS, 0.f);
  pack_transposed(&pk.prop[PackProp::W1T], W(L_OUT1), 64, 4);
  std::memcpy(&pk.prop[PackProp::B1], Bv(L_OUT1), 64 * sizeof(float));
  {
    std::vector<float> w2(64 * 128), w2b(64 * 64);
    std::memcpy(w2.data(), W(L_OUT2), sizeof(float) * 64 * 128);
    matmul64(w2b.data(), W(L_OUT2), 128, 64, W(L_FC4_2));
    for (int i = 0; i < 64; ++i)
      for (int j = 0; j < 64; ++j) w2[i * 128 + 64 + j] = w2b[i * 64 + j];
    pack_transposed(&pk.prop[PackProp::W2T], w2.data(), 64, 128);
    matvec64(&pk.prop[PackProp::V2], W(L_OUT2), 128, 64, Bv(L_FC4_2), nullptr);
  }
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


// ------------------------------------------------------------------------------------------
// MFMA gather descriptors: conv / conv-transpose message passing as dense local blocks.
//
// A tile = 32 dst nodes of ONE sample = CT channels x (PY x PX) pixel block; all of them read the
// same window of src nodes (C_s x WY x WX).  nb^T (64 ch x 32 dst) = mu_src^T (64 ch x K) . Cmat (K x 32):
// the A operand is the source embedding row (lane i holds channels 2i, 2i+1 of src row 2s+h), the
// B operand is the translation-invariant tap matrix Cmat[k][j] (weight of src window node k for dst
// lane j, 0 where no tap connects them), held in LDS in operand order [cg][s][lane].
// The result lands in fragment layout with the channel map  (it, r, h) -> 2*((r&3)+8*(r>>2)+4*h) + it,
// which the first layer of the node MLP absorbs (the input update's producer-side map, PackPostInp::WPG, is packed for it).
// ------------------------------------------------------------------------------------------
struct TileMap {        // lane j of tile t <-> node of the dst layer
  int mode = 0;         // 0: flat, tile = 32 consecutive rows of the (B*N) layer; 1: block
  int N = 0;            // nodes per sample
  int C = 0, H = 0, W = 0;
  int CT = 0, PY = 0, PX = 0, ay = 0, ax = 0, NBY = 0, NBX = 0, NCG = 0, TPS = 0;
};

struct GatherGeom {     // everything the kernel needs besides the tables
  TileMap tm;
  int lanes = 32;       // dst nodes per tile: 32 (v_mfma_f32_32x32x2, 2 window slots per k-step) or 16 (16x16x4, 4 slots per
                        // k-step).  A k-step costs 128 MFMA cycles either way, so the cost of a tile is its window size;
                        // the 16-node tiles of a forward conv edge see a third less window per node.
  int K2 = 0;           // k-steps (padded to a multiple of 8 for 32 lanes, of 4 for 16 lanes)
  int Hs = 0, Ws = 0, Ns = 0;                 // src layer
  int ystep = 0, ybase = 0, xstep = 0, xbase = 0;   // window origin: wy0 = by*ystep + ybase
  int WY = 0, WX = 0;
  int normalise = 0, kh = 0, kw = 0, stride = 0, pad = 0;
};

struct GatherHost {
  GatherGeom g;
  std::vector<float> cmat;       // [NCG][K2][64]: column of slot k, dst lane j: (k / spk) * 64 + (k % spk) * lanes + j, spk = 64 / lanes
  std::vector<int32_t> koff;     // [spk*K2 + pad][2]: {row offset relative to the window origin, wy | wx << 16}
  std::vector<uint32_t> taps3;   // 32-node tiles: [NCG][2 K2 slots][32 dst][2]: the taps in three bf16 pieces {p1 | p2 << 16, p3} (the bf16 x 3 gathers)
  long mfma_per_sample = 0;      // in units of one 32x32x2 MFMA (64 cycles): 2 per k-step
};

inline int floor_div(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }
inline int ceil_div(int a, int b) { return -floor_div(-a, b); }

// dir 0: forward (src = conv input, dst = conv output); dir 1: transposed (src = conv output, dst = conv input)
inline bool build_gather_candidate(const Edge& e, int dir, int CT, int PY, int PX, int ay, int ax, GatherHost& out, int lanes = 32) {
  const int s = e.stride, p = e.pad;
  const int Cd = dir == 0 ? e.c_out : e.c_in, Hd = dir == 0 ? e.h_out : e.h_in, Wd = dir == 0 ? e.w_out : e.w_in;
  const int Cs = dir == 0 ? e.c_in : e.c_out, Hs = dir == 0 ? e.h_in : e.h_out, Ws = dir == 0 ? e.w_in : e.w_out;
  if (CT * PY * PX > lanes || Cd % CT) return false;
  if (dir == 1 && (PY % s || PX % s)) return false;
  GatherGeom& g = out.g;
  g.lanes = lanes;
  g.tm.mode = 1; g.tm.N = Cd * Hd * Wd; g.tm.C = Cd; g.tm.H = Hd; g.tm.W = Wd;
  g.tm.CT = CT; g.tm.PY = PY; g.tm.PX = PX; g.tm.ay = ay; g.tm.ax = ax;
  g.tm.NBY = ceil_div(Hd - ay, PY); g.tm.NBX = ceil_div(Wd - ax, PX); g.tm.NCG = Cd / CT;
  g.tm.TPS = g.tm.NCG * g.tm.NBY * g.tm.NBX;
  g.Hs = Hs; g.Ws = Ws; g.Ns = Cs * Hs * Ws;
  if (dir == 0) {
    g.ystep = PY * s; g.ybase = ay * s - p; g.WY = (PY - 1) * s + e.kh;
    g.xstep = PX * s; g.xbase = ax * s - p; g.WX = (PX - 1) * s + e.kw;
  } else {
    g.ystep = PY / s; g.ybase = ceil_div(ay + p - e.kh + 1, s); g.WY = floor_div(ay + PY - 1 + p, s) - g.ybase + 1;
    g.xstep = PX / s; g.xbase = ceil_div(ax + p - e.kw + 1, s); g.WX = floor_div(ax + PX - 1 + p, s) - g.xbase + 1;
  }
  if (g.WY < 1 || g.WX < 1 || g.WY > 255 || g.WX > 255) return false;
  const int K = Cs * g.WY * g.WX;
  g.K2 = lanes == 32 ? ((K + 15) / 16) * 8 : ((K + 15) / 16) * 4;
  g.normalise = 0; g.kh = e.kh; g.kw = e.kw; g.stride = s; g.pad = p;
  out.mfma_per_sample = (long)g.tm.TPS * g.K2 * 2;
  return true;
}

inline void fill_gather_tables(const Edge& e, int dir, GatherHost& out) {
  const GatherGeom& g = out.g;
  const int s = e.stride, p = e.pad;
  const int Cs = dir == 0 ? e.c_in : e.c_out;
  const int spk = 64 / g.lanes;                             // window slots per k-step
  const int K = Cs * g.WY * g.WX, Kpad = spk * g.K2 + 16 * (spk / 2);   // always-masked entries behind the table (kernel prefetch: one chunk)
  out.koff.assign((size_t)Kpad * 2, 0);
  for (int k = 0; k < Kpad; ++k) {
    if (k < K) {
      const int cs = k / (g.WY * g.WX), wy = (k / g.WX) % g.WY, wx = k % g.WX;
      out.koff[2 * k] = (cs * g.Hs + wy) * g.Ws + wx;
      out.koff[2 * k + 1] = wy | (wx << 16);
    } else {
      out.koff[2 * k] = 0;
      out.koff[2 * k + 1] = 0x7fff | (0x7fff << 16);      // always out of bounds -> contributes 0
    }
  }
  out.cmat.assign((size_t)g.tm.NCG * g.K2 * 64, 0.f);
  for (int cg = 0; cg < g.tm.NCG; ++cg)
    for (int k = 0; k < K; ++k) {
      const int cs = k / (g.WY * g.WX), wy = (k / g.WX) % g.WY, wx = k % g.WX;
      for (int j = 0; j < g.tm.CT * g.tm.PY * g.tm.PX; ++j) {
        const int cl = j / (g.tm.PY * g.tm.PX), py = (j / g.tm.PX) % g.tm.PY, px = j % g.tm.PX;
        const int cd = cg * g.tm.CT + cl;
        int ky, kx, co, ci;
        if (dir == 0) {        // iy = oy*s - p + ky, window row wy = iy - (y0*s - p) = py*s + ky
          ky = wy - py * s; kx = wx - px * s; co = cd; ci = cs;
        } else {               // y = oy*s - p + ky with oy = wy0 + wy, y = y0 + py, y0 = by*PY + ay, wy0 = by*PY/s + ybase
          ky = (g.tm.ay + py) + p - (g.ybase + wy) * s; kx = (g.tm.ax + px) + p - (g.xbase + wx) * s; co = cs; ci = cd;
        }
        if (ky < 0 || ky >= e.kh || kx < 0 || kx >= e.kw) continue;
        const float w = e.w[(((size_t)co * e.c_in + ci) * e.kh + ky) * e.kw + kx];
        const int sidx = k / spk, h = k % spk;
        out.cmat[((size_t)cg * g.K2 + sidx) * 64 + h * g.lanes + j] = w;
      }
    }
  // the same taps as three bf16 pieces (w = p1 + p2 + p3 to 24 bits, round to nearest each time: pack_w64_bf3's split), by window slot:
  // entry (cg, slot, dst j) sits where the sparse walk's table points with the offset it uses for cmat (slot * 32 + j)
  out.taps3.clear();
  if (g.lanes == 32) {
    out.taps3.assign((size_t)g.tm.NCG * g.K2 * 64 * 2, 0u);
    for (int cg = 0; cg < g.tm.NCG; ++cg)
      for (int sl = 0; sl < 2 * g.K2; ++sl)
        for (int j = 0; j < 32; ++j) {
          const float w = out.cmat[((size_t)cg * g.K2 + sl / 2) * 64 + (sl % 2) * 32 + j];
          const unsigned short p1 = bf16_rne(w);
          const float r1 = w - bf16_f32(p1);
          const unsigned short p2 = bf16_rne(r1);
          const unsigned short p3 = bf16_rne(r1 - bf16_f32(p2));
          const size_t o = (((size_t)cg * 2 * g.K2 + sl) * 32 + j) * 2;
          out.taps3[o] = (uint32_t)p1 | ((uint32_t)p2 << 16);
          out.taps3[o + 1] = (uint32_t)p3;
        }
  }
}

// pick the tile shape with the fewest MFMAs per sample
// `tile_mfma`: MFMAs of whatever runs per tile after the gather in the same kernel (0 for a stand-alone gather)
// `allow16`: also consider 16-node tiles (stand-alone forward gathers: k_gather has a 16x16x4 variant for those)
inline bool build_gather(const Edge& e, int dir, bool normalise, GatherHost& best, int tile_mfma = 0, bool allow16 = false) {
  const int Cd = dir == 0 ? e.c_out : e.c_in;
  bool found = false;
  long best_cost = 0;
  for (int lanes = 32; lanes >= (allow16 && dir == 0 && tile_mfma == 0 ? 16 : 32); lanes /= 2)
    for (int CT = 1; CT <= lanes && CT <= Cd; ++CT) {
      if (Cd % CT) continue;
      for (int PY = 1; PY <= 8; PY *= 2)
        for (int PX = 1; PX <= 8; PX *= 2)
          for (int ay = 0; ay > -PY; --ay)
            for (int ax = 0; ax > -PX; --ax) {
              GatherHost c;
              if (!build_gather_candidate(e, dir, CT, PY, PX, ay, ax, c, lanes)) continue;
              if ((size_t)c.g.tm.NCG * c.g.K2 * 64 * 4 > 40 * 1024) continue;   // tap matrix must fit beside the MLP weights in LDS
              const long cost = c.mfma_per_sample + (long)c.g.tm.TPS * tile_mfma;
              if (!found || cost < best_cost) { best = c; best_cost = cost; found = true; }
            }
    }
  if (!found) return false;
  best.g.normalise = normalise ? 1 : 0;
  fill_gather_tables(e, dir, best);
  return true;
}

}  // namespace gnnb
