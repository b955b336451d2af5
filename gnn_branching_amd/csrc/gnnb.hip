// gnnb.hip -- MI355X (gfx950 / CDNA4) GNN branching-score forward pass: HIP kernels + C-ABI.
//
// What runs here is the reference's graphnet/graph_conv.py (EmbedLayerUpdate.forward :77-388,
// ComputeFinalScore.forward :442-470), the argmax of graphnet/graph_score.py :41-47 and the BaBSR heuristic of
// plnn/kw_score_conv.py :41-113, for a batch of B subproblems, re-designed for CDNA4 (DESIGN.md sections 3-5):
//
//   * embeddings mu[k] live in HBM as (B, N_k, 64) fp32, one 256-B row per node;
//   * every node MLP is a chain of exact-fp32 MFMAs (v_mfma_f32_32x32x2_f32) run TRANSPOSED: weights are the A operand
//     (staged once per workgroup in LDS, pre-permuted on the host, gnnb_pack.h), the 32 nodes of a tile sit on the lanes,
//     and the accumulators of one layer are the B operands of the next -- no LDS round trip between layers;
//   * node-feature-only sub-chains do not depend on the embeddings: evaluated ONCE per forward (k_pre) and folded
//     into a cached 64-vector per ambiguous node; linear layers that meet are pre-multiplied on the host; every producer's
//     last Linear is deferred into its consumers ("deferred projection", gnnb_pack.h);
//   * node classes (live / ambiguous / scored) are compacted once per forward (k_classify) and the node MLPs run over
//     the lists only;
//   * conv / conv-transpose message passing is a dense local block per tile on the MFMA (k_gather, k_gather16,
//     k_gather_input_update), Linear edges and everything above the last conv layer run per sample out of LDS (k_top);
//   * provably dead work of the reference is not executed: the `ratio` chain (:214-216,228,243,356) and the last
//     round's input-layer update (:360-385), whose result nothing reads.
//
// gfx950 only.  No HIP call at load time.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gnnb.h"
#include "gnnb_pack.h"
#include "gnnb_train.h"

using namespace gnnb;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
struct Frag {
  f32x16 t[2];  // 64 features x 32 nodes; register R = 16*it + r <-> feature 8*(R>>2) + 4*h + (R&3)
};

#define FRAG_AT(x, R) ((x).t[(R) >> 4][(R)&15])

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// acc += W(64 x 2*KSTEPS, operand order in LDS) * in, where getB(s) yields the B operand of k-step s
template <int KSTEPS, class GetB>
__device__ __forceinline__ void gemm_w64(const float* wl, int lane, Frag& acc, GetB getB) {
  // A operands are prefetched one 4-k-step block ahead; the sched_barrier keeps hipcc from hoisting
  // every LDS read of the fully unrolled chain to the top (which spills the 256-VGPR budget).
  const f32x4* w4 = reinterpret_cast<const f32x4*>(wl) + lane;
  f32x4 a0 = w4[0], a1 = w4[64];
#pragma unroll
  for (int s4 = 0; s4 < KSTEPS / 4; ++s4) {
    f32x4 n0 = a0, n1 = a1;
    if (s4 + 1 < KSTEPS / 4) {
      n0 = w4[((s4 + 1) * 2 + 0) * 64];
      n1 = w4[((s4 + 1) * 2 + 1) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);      // reads of the next block issue BEFORE this block's MFMAs, not after them
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float b = getB(s4 * 4 + c);
      acc.t[0] = mfma32(a0[c], b, acc.t[0]);
      acc.t[1] = mfma32(a1[c], b, acc.t[1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    a0 = n0;
    a1 = n1;
  }
}

// first layers on scalar node features: x[s] = input feature 2*s + h of this lane's node
template <int KSTEPS>
__device__ __forceinline__ void gemm_small(const float* wl, int lane, Frag& acc, const float (&x)[KSTEPS]) {
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    const float a0 = wl[(s * 2 + 0) * 64 + lane];
    const float a1 = wl[(s * 2 + 1) * 64 + lane];
    acc.t[0] = mfma32(a0, x[s], acc.t[0]);
    acc.t[1] = mfma32(a1, x[s], acc.t[1]);
  }
}

// ---- the 64x64 block on v_mfma_f32_32x32x16_bf16 with both operands in three bf16 pieces (gnnb_pack.h pack_w64_bf3):
// acc += W.x with the six products w1x1 + w1x2 + w2x1 + w1x3 + w2x2 + w3x1, smallest first.  48 MFMAs of 32 cycles per
// 64 inputs instead of 64 of 64 cycles; the price is the VALU work of splitting the activations (cvt_pk + subtract per piece).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pk_bf16(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{a, b}, bf16x2v)); }
__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

template <int NFRAG, class GetB>
__device__ __forceinline__ void gemm_w64_bf3(const float* wl, int lane, Frag& acc, GetB getB) {
  const u32x4* w = reinterpret_cast<const u32x4*>(wl) + lane;
#pragma unroll
  for (int fk = 0; fk < 4 * NFRAG; ++fk) {         // fragment fk / 4, k-step fk % 4: registers 8 (fk % 4) .. + 7
    u32x4 p1, p2, p3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float a = getB(8 * fk + 2 * q), b = getB(8 * fk + 2 * q + 1);
      const unsigned u1 = pk_bf16(a, b);
      const float ra = a - __uint_as_float(u1 << 16), rb = b - __uint_as_float(u1 & 0xffff0000u);
      const unsigned u2 = pk_bf16(ra, rb);
      const float sa = ra - __uint_as_float(u2 << 16), sb = rb - __uint_as_float(u2 & 0xffff0000u);
      p1[q] = u1; p2[q] = u2; p3[q] = pk_bf16(sa, sb);
    }
    const bf16x8 x1 = __builtin_bit_cast(bf16x8, p1), x2 = __builtin_bit_cast(bf16x8, p2), x3 = __builtin_bit_cast(bf16x8, p3);
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
      const bf16x8 w1 = __builtin_bit_cast(bf16x8, w[((fk * 2 + ot) * 3 + 0) * 64]);
      const bf16x8 w2 = __builtin_bit_cast(bf16x8, w[((fk * 2 + ot) * 3 + 1) * 64]);
      const bf16x8 w3 = __builtin_bit_cast(bf16x8, w[((fk * 2 + ot) * 3 + 2) * 64]);
      acc.t[ot] = mfma_bf16(w3, x1, acc.t[ot]);
      acc.t[ot] = mfma_bf16(w2, x2, acc.t[ot]);
      acc.t[ot] = mfma_bf16(w1, x3, acc.t[ot]);
      acc.t[ot] = mfma_bf16(w2, x1, acc.t[ot]);
      acc.t[ot] = mfma_bf16(w1, x2, acc.t[ot]);
      acc.t[ot] = mfma_bf16(w1, x1, acc.t[ot]);
    }
    __builtin_amdgcn_sched_barrier(0);      // one k-step's pieces and weight fragments at a time (else hipcc hoists them all and spills)
  }
}

__device__ __forceinline__ void frag_bias(Frag& a, const float* bl, int h) {
  const f32x4* b4 = reinterpret_cast<const f32x4*>(bl + h * 32);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 v = b4[q];
#pragma unroll
    for (int c = 0; c < 4; ++c) FRAG_AT(a, 4 * q + c) = v[c];
  }
}

// torch's relu propagates NaN (fmaxf would swallow it and hide a 0/0 of compute_ratio)
__device__ __forceinline__ float relu_nan(float x) { return x < 0.0f ? 0.0f : x; }

__device__ __forceinline__ void frag_relu(Frag& a) {
#pragma unroll
  for (int R = 0; R < 32; ++R) FRAG_AT(a, R) = relu_nan(FRAG_AT(a, R));
}

__device__ __forceinline__ void frag_scale(Frag& a, float s) {
#pragma unroll
  for (int R = 0; R < 32; ++R) FRAG_AT(a, R) *= s;
}

// row-major (G, 64) <-> fragment: lane (j, h) owns features [8q+4h, 8q+4h+4) of row `row`, q = 0..7
__device__ __forceinline__ void frag_load_rows(Frag& x, const float* base, long row, int h) {
  const f32x4* p = reinterpret_cast<const f32x4*>(base + row * 64 + 4 * h);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 v = p[2 * q];
#pragma unroll
    for (int c = 0; c < 4; ++c) FRAG_AT(x, 4 * q + c) = v[c];
  }
}
__device__ __forceinline__ void frag_load_rowptr(Frag& x, const float* rowptr, int h) {
  const f32x4* p = reinterpret_cast<const f32x4*>(rowptr + 4 * h);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 v = p[2 * q];
#pragma unroll
    for (int c = 0; c < 4; ++c) FRAG_AT(x, 4 * q + c) = v[c];
  }
}
__device__ __forceinline__ void frag_store_rows(const Frag& x, float* base, long row, int h) {
  f32x4* p = reinterpret_cast<f32x4*>(base + row * 64 + 4 * h);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    f32x4 v;
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = FRAG_AT(x, 4 * q + c);
    p[2 * q] = v;
  }
}
// tile-major scratch layout for the cached P vectors: float4 index (tile*8 + q)*64 + lane (1 KiB per wave-instruction)
__device__ __forceinline__ void frag_load_tiled(Frag& x, const float* base, long tile, int lane) {
  const f32x4* p = reinterpret_cast<const f32x4*>(base) + tile * 512 + lane;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 v = p[q * 64];
#pragma unroll
    for (int c = 0; c < 4; ++c) FRAG_AT(x, 4 * q + c) = v[c];
  }
}
__device__ __forceinline__ void frag_store_tiled(const Frag& x, float* base, long tile, int lane) {
  f32x4* p = reinterpret_cast<f32x4*>(base) + tile * 512 + lane;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    f32x4 v;
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = FRAG_AT(x, 4 * q + c);
    p[q * 64] = v;
  }
}

__device__ __forceinline__ bool frag_has_nan(const Frag& x) {
  bool bad = false;
#pragma unroll
  for (int R = 0; R < 32; ++R) bad |= (FRAG_AT(x, R) != FRAG_AT(x, R));
  return bad;
}

// compute_ratio (graph_conv.py:499-514), op for op
struct Ratio { float r0, r1, beta, amb, live; };
__device__ __forceinline__ Ratio compute_ratio(float lb, float ub) {
  Ratio r;
  const float lower_temp = lb - relu_nan(lb);
  const float upper_temp = relu_nan(ub);
  r.r0 = upper_temp / (upper_temp - lower_temp);
  r.beta = -1.0f * lower_temp * r.r0;
  r.amb = r.beta > 0.0f ? 1.0f : 0.0f;
  r.r1 = (1.0f - 2.0f * (r.r0 * r.amb)) * r.amb + r.r0;
  r.live = (r.r0 != 0.0f) ? 1.0f : 0.0f;   // (ratio_0 != 0), :178 / :347 (NaN != 0 is true)
  return r;
}

// global -> LDS copy of a weight pack.  Loads are issued 8 at a time before their LDS stores: a plain copy loop keeps
// one 16-B load in flight per thread and serialises ~10 L2 round trips per workgroup at the start of every launch.
__device__ __forceinline__ void copy_to_lds(float* lds, const float* src, int nfloats) {
  const f32x4* g = reinterpret_cast<const f32x4*>(src);
  f32x4* l = reinterpret_cast<f32x4*>(lds);
  const int n4 = nfloats / 4, stride = blockDim.x;
  for (int i0 = threadIdx.x; i0 < n4; i0 += 8 * stride) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * stride;
      v[u] = g[i < n4 ? i : i0];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * stride;
      if (i < n4) l[i] = v[u];
    }
  }
}
__device__ __forceinline__ void stage_pack(float* lds, const float* pack, int nfloats) {
  copy_to_lds(lds, pack, nfloats);
  __syncthreads();
}

// ---- tile -> node mapping (gnnb_pack.h TileMap) ----
struct DTileMap { int mode, N, C, H, W, CT, PY, PX, ay, ax, NBY, NBX, NCG, TPS, lpy, lpx; };
struct TileCtx { long sample; int n, cg, by, bx, y, x; bool valid; };

// lane j of tile `tile`: which node of which sample.  mode 0: 32 consecutive rows of the flat (B*N) layer;
// mode 1: CT channels x (PY x PX) pixel block of one sample (the blocks an MFMA gather works on).
__device__ __forceinline__ TileCtx tile_decode(const DTileMap& tm, long tile, int j, long total_rows) {
  TileCtx c;
  if (tm.mode == 0) {
    const long g = tile * 32 + j;
    c.valid = g < total_rows;
    const long gc = c.valid ? g : total_rows - 1;
    c.sample = gc / tm.N;
    c.n = (int)(gc - c.sample * tm.N);
    c.cg = c.by = c.bx = c.y = c.x = 0;
  } else {
    c.sample = tile / tm.TPS;
    const int t = (int)(tile - c.sample * tm.TPS);
    const int nb = tm.NBY * tm.NBX;
    c.cg = t / nb;
    const int rem = t - c.cg * nb;
    c.by = rem / tm.NBX;
    c.bx = rem - c.by * tm.NBX;
    const int pp = tm.PY * tm.PX;
    const int cl = j / pp;
    const int r2 = j - cl * pp;
    const int py = r2 / tm.PX, px = r2 - py * tm.PX;
    c.y = c.by * tm.PY + tm.ay + py;
    c.x = c.bx * tm.PX + tm.ax + px;
    c.valid = cl < tm.CT && (unsigned)c.y < (unsigned)tm.H && (unsigned)c.x < (unsigned)tm.W;
    c.n = c.valid ? ((c.cg * tm.CT + cl) * tm.H + c.y) * tm.W + c.x : 0;
  }
  return c;
}

// block tiles without integer divisions: ttab[t] = cg | by << 8 | bx << 20 for tile t of a sample (built on the host),
// PY and PX are powers of two.
__device__ __forceinline__ TileCtx block_decode(const DTileMap& tm, const int* ttab, long sample, int t, int j) {
  TileCtx c;
  c.sample = sample;
  const int e = ttab[t];
  c.cg = e & 0xff;
  c.by = (e >> 8) & 0xfff;
  c.bx = (e >> 20) & 0xfff;
  const int cl = j >> (tm.lpy + tm.lpx);
  const int py = (j >> tm.lpx) & (tm.PY - 1), px = j & (tm.PX - 1);
  c.y = c.by * tm.PY + tm.ay + py;
  c.x = c.bx * tm.PX + tm.ax + px;
  c.valid = cl < tm.CT && (unsigned)c.y < (unsigned)tm.H && (unsigned)c.x < (unsigned)tm.W;
  c.n = c.valid ? ((c.cg * tm.CT + cl) * tm.H + c.y) * tm.W + c.x : 0;
  return c;
}

// persistent tile loop: workgroup -> contiguous chunk of tiles, chunks dealt so that the workgroups of one
// XCD (blockIdx % 8 labels the XCD group) own neighbouring chunks: the samples they gather from stay in that L2.
__device__ __forceinline__ void tile_range(long ntiles, int waves, long& begin, long& end) {
  int wg = blockIdx.x;
  const int nwg = gridDim.x;
  if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);
  // chunks differ by at most one tile (rounding every chunk up to whole rounds of `waves` tiles left up to a sixth of the
  // workgroups without work on the 81-tiles-per-sample edge)
  (void)waves;
  const long base = ntiles / nwg, rem = ntiles - base * nwg;
  begin = (long)wg * base + (wg < rem ? wg : rem);
  end = begin + base + (wg < rem ? 1 : 0);
}

#define WG_MLP 512       // 8 waves: 2 per SIMD share one LDS copy of the weights
#define WAVES_MLP 8

// ------------------------------------------------------------------------------------------
// MFMA node-MLP kernels.  One wave = one tile of 32 consecutive nodes of a (B*N_k) flat layer.
// ------------------------------------------------------------------------------------------
struct EmbedArgs { const float* w; const float* b; const float* lb; const float* x; const float* ub; float* mu; long G; };

// E0 = relu(inp_f([l0, x_LP, u0])); mu0 = inp_f_1(E0) is deferred into the forward update of ReLU layer 1
// (gnnb_pack.h "deferred projection")   graph_conv.py:90-95
// 3 -> 64 features per node: 192 FMAs against a 256-B row written, i.e. HBM-write-bound VALU work, not an MFMA job.
// A thread owns 4 consecutive features (its 12 weights + 4 biases stay in registers) and walks nodes; 16 threads
// write one 256-B row, one wave instruction writes 1 KiB contiguous.
#define EMBED_UNROLL 4
__global__ __launch_bounds__(256) void k_embed(EmbedArgs a) {
  const int q = threadIdx.x & 15;                       // feature quad
  float w[4][3], bias[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    bias[c] = a.b[4 * q + c];
#pragma unroll
    for (int i = 0; i < 3; ++i) w[c][i] = a.w[(4 * q + c) * 3 + i];
  }
  const long nodes_per_pass = (long)gridDim.x * 16;      // 16 nodes per workgroup and pass
  long g = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
  for (; g < a.G; g += nodes_per_pass * EMBED_UNROLL) {
    float l[EMBED_UNROLL], x[EMBED_UNROLL], u[EMBED_UNROLL];
#pragma unroll
    for (int r = 0; r < EMBED_UNROLL; ++r) {
      const long gg = g + r * nodes_per_pass;
      const long gc = gg < a.G ? gg : a.G - 1;
      l[r] = a.lb[gc]; x[r] = a.x[gc]; u[r] = a.ub[gc];
    }
#pragma unroll
    for (int r = 0; r < EMBED_UNROLL; ++r) {
      const long gg = g + r * nodes_per_pass;
      f32x4 o;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        // torch addmm order: bias + sum_k in_k w_k
        o[c] = relu_nan(fmaf(u[r], w[c][2], fmaf(x[r], w[c][1], fmaf(l[r], w[c][0], bias[c]))));
      }
      if (gg < a.G) *reinterpret_cast<f32x4*>(a.mu + gg * 64 + 4 * q) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------
// classification: what each ReLU node needs this forward (static over the T rounds)
//   live  = [r0 != 0]  (graph_conv.py:178/:347): only these rows of mu can be non-zero -> node MLP runs on them only
//   amb   = [beta > 0] (:504): only these have a non-zero relaxation term -> the hoisted feature chains run on them only
//   score = BaB mask == -1 (:447): only these are scored
// Each class is compacted into a list of flat node ids (wave-aggregated atomics; the order inside a list does not
// affect any result: every lane of an MLP tile computes its own column).  Dead rows of mu are zeroed here, once,
// and scores are preset to -inf.
// ------------------------------------------------------------------------------------------
#define MAXL 8            // ReLU layers handled by the merged per-layer kernels (bind rejects deeper networks for them)
struct ClassifyArgs {    // every ReLU layer of the network in one launch
  int L;
  const float* lb[MAXL]; const float* ub[MAXL];
  float* mu[MAXL];                 // (B*N_k, 64) rows of layer k
  float* mu2;                      // second row buffer of layer 1 (F1, PackPostInp) whose dead rows must read as zero too, or null
  int* live[MAXL]; int* amb[MAXL]; int* score[MAXL];
  float* livef[MAXL];              // (B*N_k) 1.0 / 0.0: [r0 != 0], read by k_livesum
  long G[MAXL];
  int N[MAXL], off[MAXL], blk0[MAXL + 1];   // first workgroup of each layer
  const float* mask;
  float* scores;                   // (B, R)
  int* cnt;                        // 4 ints per layer: plain (live, not ambiguous), ambiguous, scored, 0
  int R;
};

#define CLS_THREADS 1024
// one global atomic per list and workgroup (a single counter word only sustains ~90 atomics/us)
__global__ __launch_bounds__(CLS_THREADS) void k_classify(ClassifyArgs a) {
  __shared__ int wcnt[3][CLS_THREADS / 64];
  __shared__ int wbase[3][CLS_THREADS / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int k = 0;
  while (k + 1 < a.L && (int)blockIdx.x >= a.blk0[k + 1]) ++k;
  const long G = a.G[k];
  const int N = a.N[k];
  const long g = (long)(blockIdx.x - a.blk0[k]) * CLS_THREADS + threadIdx.x;
  const bool valid = g < G;
  const long gc = valid ? g : G - 1;
  const Ratio r = compute_ratio(a.lb[k][gc], a.ub[k][gc]);
  const long b = gc / N;
  const long sidx = b * a.R + a.off[k] + (gc - b * N);
  bool flag[3];
  const bool live = valid && r.live != 0.0f;
  flag[1] = valid && r.amb != 0.0f;                 // ambiguous (a subset of live)
  flag[0] = live && !flag[1];                       // live with r0 == r1: the cheap update path
  flag[2] = valid && a.mask[sidx] != 0.0f;
  if (valid) {
    a.scores[sidx] = -INFINITY;
    a.livef[k][g] = live ? 1.0f : 0.0f;
  }
  unsigned long long bal[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    bal[c] = __ballot(flag[c]);
    if (lane == 0) wcnt[c][wave] = __popcll(bal[c]);
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int c = threadIdx.x;
    int total = 0;
    for (int w = 0; w < CLS_THREADS / 64; ++w) { wbase[c][w] = total; total += wcnt[c][w]; }
    const int base = total ? atomicAdd(a.cnt + 4 * k + c, total) : 0;
    for (int w = 0; w < CLS_THREADS / 64; ++w) wbase[c][w] += base;
  }
  __syncthreads();
  int* lists[3] = {a.live[k], a.amb[k], a.score[k]};
#pragma unroll
  for (int c = 0; c < 3; ++c)
    if (flag[c]) lists[c][wbase[c][wave] + __popcll(bal[c] & ((1ull << lane) - 1ull))] = (int)gc;
  unsigned long long dead = __ballot(valid && !live);
  float* mu = a.mu[k];
  float* mu2 = k == 0 ? a.mu2 : nullptr;
  while (dead) {                       // the whole wave zeroes one dead row per iteration (coalesced 256 B)
    const int l = __ffsll((long long)dead) - 1;
    dead &= dead - 1;
    const long row = g - lane + l;
    mu[row * 64 + lane] = 0.0f;
    if (mu2) mu2[row * 64 + lane] = 0.0f;
  }
}

struct PreArgs {
  const float* pack;
  const float *lb, *ub, *dual, *z_pre, *z_post, *bias;   // per-node scalars (flat B*N), bias per channel
  float* P;                                               // out: tile-major (k_pre_inp) or rows by node id (k_pre_fwd/bwd)
  long G, ntiles;
  int N, hw;                                              // nodes per sample; nodes per bias entry (H*W or 1)
  DTileMap tm;                                            // k_pre_inp: which node sits on which (tile, lane)
  const int* list;                                        // k_pre_fwd/bwd: ambiguous nodes of the layer
  const int* cnt;
};

struct PreAllArgs {       // hoisted feature chains of every ReLU layer, forward and backward, in one launch
  const float* pack_f;   // PackPreFwd
  const float* pack_b;   // PackPreBwd
  int L, do_bwd;
  const float* lb[MAXL]; const float* ub[MAXL]; const float* dual[MAXL];
  const float* z_pre[MAXL]; const float* z_post[MAXL]; const float* bias[MAXL];
  float* Pf[MAXL]; float* Pb[MAXL];            // out: P' rows by node id
  const int* list[MAXL];                       // ambiguous nodes of layer k
  const int* cnt;                              // cnt[4k + 1] = number of ambiguous nodes of layer k
  int N[MAXL], hw[MAXL];
};

// tile space: (do_bwd) the backward tiles of all layers, ceil(c_k/32) each, then the forward tiles of all layers: a
// backward tile is 392 MFMAs, a forward tile 72, and there are only a few tiles per wave, so the strided dealing below
// hands every wave its share of the long ones first
//   forward  P'_f[g] = fc4[:, :64] . fc1_1(relu(fc1(feat7))) + bcb_f                           graph_conv.py:153-161,176-177
//   backward P'_b[g] = bc4[:, :64] . bc2_1(relu(bc2([s, -d2 s, d1 s]))) + bcb_b,
//            s = bc1_2(relu(bc1_1(relu(bc1(feat7')))))                                        graph_conv.py:273-293,344-345
// for the ambiguous nodes g (everywhere else the relaxation term is multiplied by amb = 0, :161 / :293)
// one tile (32 ambiguous nodes `list[32 t ..]` of layer k) of the hoisted chains; lds / lds_b: PackPreFwd / PackPreBwd in LDS
__device__ __forceinline__ void pre_tile(const PreAllArgs& a, const float* lds, const float* lds_b, int k, bool bwd, const int* list, int count,
                                         long t, int lane) {
  const int h = lane >> 5, j = lane & 31;
  {
    const long idx = t * 32 + j;
    const bool valid = idx < count;
    const long gc = list[valid ? idx : 0];
    const int n = (int)(gc % a.N[k]);
    const float lb = a.lb[k][gc], ub = a.ub[k][gc];
    const Ratio r = compute_ratio(lb, ub);
    const float d1 = a.dual[k][gc * 3 + 1], d2 = a.dual[k][gc * 3 + 2];
    const float c = a.bias[k][n / a.hw[k]];
    const float zpre = a.z_pre[k][gc], zpost = a.z_post[k][gc];
    float x[4];
    if (!bwd) {
      // feat7 = [beta, l, u, d1-d2, z_pre, z_post, c]: even features on half 0, odd on half 1
      x[0] = h ? lb : r.beta;
      x[1] = h ? (d1 - d2) : ub;
      x[2] = h ? zpost : zpre;
      x[3] = h ? 0.0f : c;
      Frag H;
      frag_bias(H, lds + PackPreFwd::B1, h);
      gemm_small<4>(lds + PackPreFwd::W1, lane, H, x);
      frag_relu(H);
      Frag Pf;                                   // fc1_1 and the first half of fc4 are one folded 64x64 map
      frag_bias(Pf, lds + PackPreFwd::B2, h);
      gemm_w64<32>(lds + PackPreFwd::W2, lane, Pf, [&](int s) { return FRAG_AT(H, s); });
      if (valid) frag_store_rows(Pf, a.Pf[k], gc, h);
    } else {
      // feat7' = [l, u, beta, -d2+d1, z_post, z_pre, c]
      x[0] = h ? ub : lb;
      x[1] = h ? (-d2 + d1) : r.beta;
      x[2] = h ? zpre : zpost;
      x[3] = h ? 0.0f : c;
      Frag H1;
      frag_bias(H1, lds_b + PackPreBwd::B1, h);
      gemm_small<4>(lds_b + PackPreBwd::W1, lane, H1, x);
      frag_relu(H1);
      Frag H2;
      frag_bias(H2, lds_b + PackPreBwd::B2, h);
      gemm_w64<32>(lds_b + PackPreBwd::W2, lane, H2, [&](int s) { return FRAG_AT(H1, s); });
      frag_relu(H2);
      Frag S;
      frag_bias(S, lds_b + PackPreBwd::B3, h);
      gemm_w64<32>(lds_b + PackPreBwd::W3, lane, S, [&](int s) { return FRAG_AT(H2, s); });
      // bc2 on [s, s*(-d2), s*d1]  (:287-291)
      const float nd2 = -d2;
      Frag H4;
      frag_bias(H4, lds_b + PackPreBwd::B4, h);
      gemm_w64<96>(lds_b + PackPreBwd::W4, lane, H4, [&](int s) {
        const float v = FRAG_AT(S, s & 31);
        return s < 32 ? v : (s < 64 ? v * nd2 : v * d1);
      });
      frag_relu(H4);
      Frag Pb;                                   // bc2_1 and the first half of bc4 are one folded 64x64 map
      frag_bias(Pb, lds_b + PackPreBwd::B5, h);
      gemm_w64<32>(lds_b + PackPreBwd::W5, lane, Pb, [&](int s) { return FRAG_AT(H4, s); });
      if (valid) frag_store_rows(Pb, a.Pb[k], gc, h);
    }
  }
}

__global__ __launch_bounds__(WG_MLP, 2) void k_pre(PreAllArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lds_b = lds + PackPreFwd::FLOATS;
  copy_to_lds(lds_b, a.pack_b, PackPreBwd::FLOATS);
  stage_pack(lds, a.pack_f, PackPreFwd::FLOATS);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long nhalf = 0;
  for (int k = 0; k < a.L; ++k) nhalf += (long)((a.cnt[4 * k + 1] + 31) / 32);
  const long ntiles = nhalf * (a.do_bwd ? 2 : 1);
  for (long tile = (long)wave * gridDim.x + blockIdx.x; tile < ntiles; tile += (long)gridDim.x * WAVES_MLP) {
    // which layer / direction (wave-uniform)
    const bool bwd = a.do_bwd && tile < nhalf;
    int k = 0, count = 0;
    long t = (a.do_bwd && !bwd) ? tile - nhalf : tile;
    for (; k < a.L; ++k) {
      count = a.cnt[4 * k + 1];
      const long tk = (count + 31) / 32;
      if (t < tk) break;
      t -= tk;
    }
    pre_tile(a, lds, lds_b, k, bwd, a.list[k], count, t, lane);
  }
}

// Q = inp_b2[:, :64] . inp_b_1(relu(inp_b([l0, u0]))) + inp_b2.bias       graph_conv.py:380-384
__global__ __launch_bounds__(WG_MLP, 2) void k_pre_inp(PreArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_pack(lds, a.pack, PackPreInp::FLOATS);
  const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31, wave = threadIdx.x >> 6;
  for (long tile = (long)blockIdx.x * WAVES_MLP + wave; tile < a.ntiles; tile += (long)gridDim.x * WAVES_MLP) {
    const TileCtx tc = tile_decode(a.tm, tile, j, a.G);
    const long gc = tc.sample * a.N + tc.n;
    float x[1];
    x[0] = h ? a.ub[gc] : a.lb[gc];
    Frag H;
    frag_bias(H, lds + PackPreInp::B1, h);
    gemm_small<1>(lds + PackPreInp::W1, lane, H, x);
    frag_relu(H);
    Frag Q;                                  // inp_b_1 and the first half of inp_b2 are folded into one 64x64 map
    frag_bias(Q, lds + PackPreInp::B2, h);
    gemm_w64<32>(lds + PackPreInp::W2, lane, Q, [&](int s) { return FRAG_AT(H, s); });
    frag_store_tiled(Q, a.P, tile, lane);
  }
}

struct UpdArgs {
  const float* pack;
  const float *lb, *ub;     // pre-activation bounds of this layer, flat (B*N)
  const float* nb;          // aggregated neighbour embeddings, rows by node id (B*N, 64)
  const float* P;           // cached P' of the ambiguous nodes, rows by node id
  float* mu;                // out: rows by node id
  int* status;
  const int *list0, *cnt0;  // nodes with r0 == r1 and no relaxation term (live, not ambiguous): short chain
  const int *list1, *cnt1;  // general nodes (ambiguous; or the scored nodes for the last backward step of layer 1)
  const float* sarr;        // DEFERRED: s[g] = sum over the edge of live_src (k_livesum), the bias term of the source rows' projection
  // POST (layer 1, backward, an input-layer update follows): the consumer's 64x64 map inp_b2[:, 64:].bc4_1.W is applied here,
  // on the ~3x fewer producer nodes: F = WP.E goes to `post` (rows by node id), and `mu` may be null (nothing else reads E)
  float* post;
  const float* wp;          // PackPostInp block (WPN or WPG), staged behind the update pack
};

// folded node update (gnnb_pack.h PackUpd):  E_g = relu(P'_g + Wcb.h) [r0 != 0],  h = relu(Wa.[r0 nb_g, r1 nb_g] + ba);
// the last layer, mu_g = (Wd.E_g + bd) [r0 != 0], is deferred into the consumers of the rows ("deferred projection").
//   kind 0 tiles (list0): r0 == r1, P' = bcb:  h = relu(WAS.(r0 nb_g) + ba)                      128 MFMAs per 32 nodes
//   kind 1 tiles (list1): general                                                              192 MFMAs per 32 nodes
// DEFERRED: nb is an aggregate G of rows whose own last layer Wp is deferred: Wa is pre-multiplied by Wp and the bias
// term s.(r0 Wa0.bp + r1 Wa1.bp) enters as one small k-step.
// forward:  fc3, fc3_2, fc4, fc4_2   graph_conv.py:169-181        backward: bc3, bc3_1, bc4, bc4_1   :331-349
// The tile loop of the node update: tiles `tile`, `tile + stride`, ... of the lists in `a` (c0 / c1 entries); the weight
// pack is staged into `lds` here (the first fetch overlaps it).
// BF3: the 64x64 blocks of the short chain (WAS, WCB) and of POST run on the bf16 matrix rate with three-piece operands
// (gemm_w64_bf3; LDS image PackUpdL3); the general chain's 128-wide first layer stays on the fp32 MFMA (6-11 % of the tiles).
template <bool DEFERRED, bool POST = false, bool BF3 = false>
__device__ __forceinline__ void node_update_loop(const UpdArgs& a, float* lds, int c0, int c1, long tile, long stride, int lane) {
  constexpr int O_WA = BF3 ? (int)PackUpdL3::WA : (int)PackUpd::WA, O_BA = BF3 ? (int)PackUpdL3::BA : (int)PackUpd::BA;
  constexpr int O_BCB = BF3 ? (int)PackUpdL3::BCB : (int)PackUpd::BCB, O_VAW = BF3 ? (int)PackUpdL3::VAW : (int)PackUpd::VAW;
  constexpr int O_END = BF3 ? (int)PackUpdL3::FLOATS : (int)PackUpd::FLOATS;
  const int h = lane >> 5, j = lane & 31;
  // the general tiles (1.5-3x the work of a short-chain tile) come FIRST in the tile order, so they are never a SIMD's tail
  const long n1 = (c1 + 31) / 32, n0 = (c0 + 31) / 32, ntiles = n0 + n1;
  const float* bias_row = a.pack + PackUpd::BCBROW;
  long gc = 0, gc_n = 0;
  bool valid = false, valid_n = false;
  float lb = 0.0f, ub = 0.0f, lb_n = 0.0f, ub_n = 0.0f, sw = 0.0f, sw_n = 0.0f;
  Frag X, Xn;
  constexpr bool deferred = DEFERRED;            // the aggregate is built from rows with a deferred projection (gnnb_pack.h)
  auto fetch = [&](long tl, long& g_, bool& v_, float& l_, float& u_, float& s_, Frag& x_) {
    const bool k0 = tl >= n1;
    const long idx = (k0 ? tl - n1 : tl) * 32 + j;
    v_ = idx < (k0 ? c0 : c1);
    g_ = (k0 ? a.list0 : a.list1)[v_ ? idx : 0];
    l_ = a.lb[g_];
    u_ = a.ub[g_];
    if (deferred) s_ = a.sarr[g_];
    frag_load_rows(x_, a.nb, g_, h);
  };
  if (tile < ntiles) fetch(tile, gc, valid, lb, ub, sw, X);
  constexpr bool post = POST;
  if (post) copy_to_lds(lds + O_END, a.wp, BF3 ? 6144 : 4096);
  if (BF3) {
    copy_to_lds(lds + PackUpdL3::WA, a.pack + PackUpd::WA, 8192);
    copy_to_lds(lds + PackUpdL3::BA, a.pack + PackUpd::BA, 64);
    copy_to_lds(lds + PackUpdL3::BCB, a.pack + PackUpd::BCB, 64 + 64 + 128);          // BCB, BCBROW, VAW
    stage_pack(lds + PackUpdL3::WAS3, a.pack + PackUpd::WAS3, 2 * 6144);               // WAS3, WCB3
  } else stage_pack(lds, a.pack, PackUpd::FLOATS);
  if (tile >= ntiles) return;
  for (;;) {
    const Ratio r = compute_ratio(lb, ub);
    const bool kind0 = tile >= n1;             // wave-uniform
    const long next = tile + stride;
    const bool has_next = next < ntiles;
    Frag H, H2;
    frag_bias(H, lds + O_BA, h);
    if (kind0) {
      if (has_next) fetch(next, gc_n, valid_n, lb_n, ub_n, sw_n, Xn);
      const float r0 = r.r0;
      if (deferred) {                        // + s.(r0 Wa0.bp + r1 Wa1.bp), r0 == r1: one small k-step
        const float x[1] = {r0 * sw};
        gemm_small<1>(lds + O_VAW, lane, H, x);
      }
      if (BF3) gemm_w64_bf3<1>(lds + PackUpdL3::WAS3, lane, H, [&](int s) { return FRAG_AT(X, s) * r0; });
      else gemm_w64<32>(lds + PackUpd::WAS, lane, H, [&](int s) { return FRAG_AT(X, s) * r0; });
      frag_bias(H2, lds + O_BCB, h);
    } else {
      // nodes without a relaxation term (amb = 0) read the bias row instead of their (never written) P' row
      frag_load_rowptr(H2, r.amb != 0.0f ? a.P + gc * 64 : bias_row, h);
      if (has_next) fetch(next, gc_n, valid_n, lb_n, ub_n, sw_n, Xn);
      const float r0 = r.r0, r1 = r.r1;
      if (deferred) {
        const float x[1] = {(h ? r1 : r0) * sw};
        gemm_small<1>(lds + O_VAW, lane, H, x);
      }
      gemm_w64<64>(lds + O_WA, lane, H, [&](int s) { return FRAG_AT(X, s & 31) * (s < 32 ? r0 : r1); });
    }
    frag_relu(H);
    if (BF3) gemm_w64_bf3<1>(lds + PackUpdL3::WCB3, lane, H2, [&](int s) { return FRAG_AT(H, s); });
    else gemm_w64<32>(lds + PackUpd::WCB, lane, H2, [&](int s) { return FRAG_AT(H, s); });
    frag_relu(H2);
    if (r.live == 0.0f) {                      // a dead node's row is zero whatever its (possibly never written) aggregate held
#pragma unroll
      for (int R = 0; R < 32; ++R) FRAG_AT(H2, R) = 0.0f;
    }
    if (valid) {
      if (frag_has_nan(H2)) atomicOr(a.status, 1);      // a NaN here is a NaN in mu = Wd.E + bd (:184-186, :339-341)
      if (a.mu) frag_store_rows(H2, a.mu, gc, h);
    }
    if (post) {
#pragma unroll
      for (int R = 0; R < 32; ++R) FRAG_AT(H, R) = 0.0f;
      if (BF3) gemm_w64_bf3<1>(lds + O_END, lane, H, [&](int s) { return FRAG_AT(H2, s); });
      else gemm_w64<32>(lds + O_END, lane, H, [&](int s) { return FRAG_AT(H2, s); });
      if (valid) frag_store_rows(H, a.post, gc, h);
    }
    if (!has_next) break;
    tile = next; gc = gc_n; valid = valid_n; lb = lb_n; ub = ub_n; sw = sw_n;
#pragma unroll
    for (int R = 0; R < 32; ++R) FRAG_AT(X, R) = FRAG_AT(Xn, R);
  }
}

template <int WAVES, bool DEFERRED, bool POST = false, bool BF3 = false>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k_node_update(UpdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // Only a few tiles per wave, so balance matters more than locality (rows stream): tiles are dealt round-robin over
  // the SIMDs of the whole grid (4 per workgroup), and the two waves that share a SIMD (w, w+4) take alternate rounds,
  // so every SIMD's MFMA pipe gets floor or ceil of the average.  The inputs of the next tile (list entry -> bounds ->
  // aggregate row) are fetched while this tile's MFMA chain runs; the first fetch overlaps the weight staging.
  static_assert(WAVES % 4 == 0, "tile dealing assumes whole waves per SIMD");
  const long stride = (long)gridDim.x * 4 * (WAVES / 4);
  const long tile = (long)(wave >> 2) * gridDim.x * 4 + (long)blockIdx.x * 4 + (wave & 3);
  node_update_loop<DEFERRED, POST, BF3>(a, lds, *a.cnt0, *a.cnt1, tile, stride, lane);
}

struct UpdInpArgs { const float* pack; const float* nb; const float* Q; const float* sarr; float* mu; long G, ntiles; };

// E_0 = relu(Q + inp_b2[:, 64:] . nb); mu_0 = inp_b2_2(E_0) is deferred (gnnb_pack.h)         graph_conv.py:383-385
__global__ __launch_bounds__(WG_MLP, 2) void k_input_update(UpdInpArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_pack(lds, a.pack, PackUpdInp::FLOATS);
  const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31, wave = threadIdx.x >> 6;
  for (long tile = (long)blockIdx.x * WAVES_MLP + wave; tile < a.ntiles; tile += (long)gridDim.x * WAVES_MLP) {
    const long g = tile * 32 + j;
    const bool valid = g < a.G;
    const long gc = valid ? g : a.G - 1;
    Frag X;
    frag_load_rows(X, a.nb, gc, h);
    Frag H;
    frag_load_tiled(H, a.Q, tile, lane);
    {                                          // bias term of the projection deferred in the rows of mu_1
      const float x[1] = {h ? 0.0f : a.sarr[gc]};
      gemm_small<1>(lds + PackUpdInp::VC, lane, H, x);
    }
#pragma unroll
    for (int R = 0; R < 32; ++R) FRAG_AT(H, R) += FRAG_AT(X, R);      // the aggregate already went through inp_b2[:, 64:].bc4_1.W (PackPostInp)
    frag_relu(H);
    if (valid) frag_store_rows(H, a.mu, g, h);
  }
}

// ------------------------------------------------------------------------------------------
// fused message passing + node update for conv edges: the neighbour aggregate never leaves registers.
//   nb^T (64 ch x 32 dst) = mu_src^T (64 ch x K window nodes) . Cmat (K x 32 dst)      on the MFMA,
// A operand = source embedding rows straight from HBM/L2 (one coalesced 256-B row per lane half and k-step:
// lane i holds channels 2i, 2i+1), B operand = the tap matrix of the tile shape, resident in LDS.
// Forward edges: graph_conv.py:110-127; transposed edges + tap-count division: :299-318; input layer :361-372.
// ------------------------------------------------------------------------------------------
struct DGather {
  const float* cmat;     // [NCG][K2][64]
  const int2* koff;      // [2*K2]: {row offset relative to the window origin, wy | wx << 16}
  const int* ttab;       // [TPS]: cg | by << 8 | bx << 20
  const float* zero;     // 64 zero floats
  int K2, ncg_k2, Hs, Ws, Ns, ystep, ybase, xstep, xbase, WY, WX, normalise, kh, kw, stride, pad;
  int lanes;             // dst nodes per tile: 32 (32x32x2 MFMA, 2 window slots per k-step) or 16 (16x16x4, 4 slots per k-step)
};

#define GATHER_CH 8   // k-steps per prefetch chunk
#define KOFF_PAD (2 * GATHER_CH)   // always-masked koff entries behind the table (one chunk is loaded past the end)
// window slots in the koff / kvo tables, padding included (gnnb_pack.h fill_gather_tables)
__host__ __device__ inline int gather_slots(int K2, int lanes) { return lanes == 32 ? 2 * K2 + KOFF_PAD : 4 * K2 + 2 * KOFF_PAD; }

// Source rows are read with buffer loads: the descriptor covers exactly this sample's source layer (wave-uniform base
// in SGPRs), the per-lane part is a 32-bit byte offset, and a masked window node / the k padding simply gets an
// offset beyond the descriptor's range -- the hardware returns 0 for it without touching memory.  So nothing (no
// select, no copy) is applied to a loaded value before its MFMA and there is no control flow around the loads, which is
// what lets hipcc keep the next chunk in flight behind counted vmcnt waits (the koff table carries 16 always-masked
// entries for the one chunk issued past the end).
// INTERIOR = the whole window lies inside the source layer (wave-uniform; ~3/4 of the tiles): no bounds arithmetic.
#define BUF_OOB 0x80000000u
__device__ __forceinline__ float2 buf_load2(__amdgpu_buffer_rsrc_t r, unsigned voff) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, 0, 0);
  return make_float2(__uint_as_float(v[0]), __uint_as_float(v[1]));
}

template <bool INTERIOR>
__device__ __forceinline__ void gather_tile(Frag& X, const float* cm, const int2* ko, const unsigned* kvo, int K2,
                                            __amdgpu_buffer_rsrc_t rsrc, int j, int wy0, int wx0, int Hs, int Ws, int lane) {
  const int h = lane >> 5;
#pragma unroll
  for (int R = 0; R < 32; ++R) FRAG_AT(X, R) = 0.0f;
  const int origin = wy0 * Ws + wx0;
  const unsigned lane_off = 8u * (unsigned)j;        // channels 2j, 2j+1 of the row
  const unsigned soff = (unsigned)origin * 256u;     // INTERIOR: wave-uniform window origin goes into the scalar offset
  float2 cur[GATHER_CH], nxt[GATHER_CH];
  auto load = [&](float2 (&dst)[GATHER_CH], int s0) {
#pragma unroll
    for (int u = 0; u < GATHER_CH; ++u) {
      if (INTERIOR) {
        // per k-step: one LDS read of the tile-invariant byte offset + one add; no bounds arithmetic.  k padding
        // reads row `origin` (in range for an interior tile; its tap-matrix column is zero)
        const unsigned vo = kvo[2 * (s0 + u) + h] + lane_off;
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, vo, soff, 0);
        dst[u] = make_float2(__uint_as_float(v[0]), __uint_as_float(v[1]));
      } else {
        // one 64-bit LDS read per entry: with two 32-bit halves hipcc branches around the second one
        const unsigned long long ev = reinterpret_cast<const unsigned long long*>(ko)[2 * (s0 + u) + h];
        const int ex = (int)(unsigned)ev, ey = (int)(unsigned)(ev >> 32);
        const int wy = wy0 + (ey & 0xffff), wx = wx0 + (ey >> 16);
        unsigned o = (unsigned)(origin + ex) * 256u + lane_off;
        o = ((unsigned)wy < (unsigned)Hs && (unsigned)wx < (unsigned)Ws) ? o : BUF_OOB;
        dst[u] = buf_load2(rsrc, o);
      }
    }
  };
  auto mma = [&](const float2 (&v)[GATHER_CH], int s0) {
#pragma unroll
    for (int u = 0; u < GATHER_CH; ++u) {
      const float b = cm[(s0 + u) * 64 + lane];
      X.t[0] = mfma32(v[u].x, b, X.t[0]);
      X.t[1] = mfma32(v[u].y, b, X.t[1]);
    }
  };
  // Two register buffers in ping-pong (a rotating copy would have to wait for the data it copies); the sched_barriers
  // pin "issue the next chunk's loads, THEN this chunk's MFMAs".
  load(cur, 0);
  const int npairs = K2 / (2 * GATHER_CH);
  int s0 = 0;
  for (int pr = 0; pr < npairs; ++pr, s0 += 2 * GATHER_CH) {
    load(nxt, s0 + GATHER_CH);
    __builtin_amdgcn_sched_barrier(0);
    mma(cur, s0);
    __builtin_amdgcn_sched_barrier(0);
    load(cur, s0 + 2 * GATHER_CH);
    __builtin_amdgcn_sched_barrier(0);
    mma(nxt, s0 + GATHER_CH);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (K2 & GATHER_CH) mma(cur, s0);
}

__device__ __forceinline__ bool node_is_live(float lb, float ub);

// Sparse variant (source = a ReLU layer): the rows of its dead nodes are exactly zero, so the tile first compacts the live,
// in-range slots of its window into a per-wave LDS table {byte offset of the row, tap-matrix row} and walks only those.
// `tab`: 2*K2 + 32 entries of this wave; slb / sub: bounds of the source layer of this sample.
#ifndef GATHER_CHS
#define GATHER_CHS 4    // k-steps per prefetch chunk of the sparse walk (8 live slots: 3 % faster than 16)
#endif
__device__ __forceinline__ void gather_tile_sparse(Frag& X, const float* cm, const int2* ko, uint2* tab, int K2, __amdgpu_buffer_rsrc_t rsrc,
                                                   const float* slb, const float* sub, int j, int wy0, int wx0, int Hs, int Ws, int lane) {
  const int h = lane >> 5;
#pragma unroll
  for (int R = 0; R < 32; ++R) FRAG_AT(X, R) = 0.0f;
  const int origin = wy0 * Ws + wx0;
  int n = 0;
  for (int base = 0; base < 2 * K2; base += 64) {
    const int sl = base + lane;
    const unsigned long long ev = reinterpret_cast<const unsigned long long*>(ko)[sl];
    const int ex = (int)(unsigned)ev, ey = (int)(unsigned)(ev >> 32);
    const int wy = wy0 + (ey & 0xffff), wx = wx0 + (ey >> 16);
    const bool inb = sl < 2 * K2 && (unsigned)wy < (unsigned)Hs && (unsigned)wx < (unsigned)Ws;      // (table padding: 0x7fff, never in range)
    const int row = inb ? origin + ex : 0;
    const bool live = inb && node_is_live(slb[row], sub[row]);
    const unsigned long long bal = __ballot(live);
    if (live) tab[n + __popcll(bal & ((1ull << lane) - 1ull))] = make_uint2((unsigned)row * 256u, (unsigned)sl * 32u);
    n += __popcll(bal);
  }
  constexpr int CS = 2 * GATHER_CHS;                     // slots per chunk
  const int npad = (n + CS - 1) / CS * CS;
  for (int q = n + lane; q < npad + CS; q += 64) tab[q] = make_uint2(BUF_OOB, 0u);       // out-of-range offset: the load returns 0
  const int K2e = npad / 2;
  const unsigned lane_off = 8u * (unsigned)j;
  struct Chunk { float2 v[GATHER_CHS]; unsigned cr[GATHER_CHS]; };
  Chunk cur, nxt;
  auto load = [&](Chunk& c, int s0) {
#pragma unroll
    for (int u = 0; u < GATHER_CHS; ++u) {
      const uint2 e = tab[2 * (s0 + u) + h];
      c.v[u] = buf_load2(rsrc, e.x == BUF_OOB ? BUF_OOB : e.x + lane_off);
      c.cr[u] = e.y;
    }
  };
  auto mma = [&](const Chunk& c) {
#pragma unroll
    for (int u = 0; u < GATHER_CHS; ++u) {
      const float b = cm[c.cr[u] + j];
      X.t[0] = mfma32(c.v[u].x, b, X.t[0]);
      X.t[1] = mfma32(c.v[u].y, b, X.t[1]);
    }
  };
  load(cur, 0);
  const int npairs = K2e / (2 * GATHER_CHS);
  int s0 = 0;
  for (int pr = 0; pr < npairs; ++pr, s0 += 2 * GATHER_CHS) {
    load(nxt, s0 + GATHER_CHS);
    __builtin_amdgcn_sched_barrier(0);
    mma(cur);
    __builtin_amdgcn_sched_barrier(0);
    load(cur, s0 + 2 * GATHER_CHS);
    __builtin_amdgcn_sched_barrier(0);
    mma(nxt);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (K2e & GATHER_CHS) mma(cur);
}

// `sbase` = first row of this sample's source layer; must be built from wave-uniform values
// tab != nullptr: sparse walk (slb / sub = bounds of the source layer of this sample)
__device__ __forceinline__ void gather_dispatch(Frag& X, const float* cm, const int2* ko, const unsigned* kvo, const DGather& g,
                                                const float* sbase, int j, int wy0, int wx0, int lane,
                                                uint2* tab = nullptr, const float* slb = nullptr, const float* sub = nullptr) {
  const int uy = __builtin_amdgcn_readfirstlane(wy0), ux = __builtin_amdgcn_readfirstlane(wx0);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)sbase, 0, g.Ns * 256, 0x00020000);
  if (tab) { gather_tile_sparse(X, cm, ko, tab, g.K2, rsrc, slb, sub, j, uy, ux, g.Hs, g.Ws, lane); return; }
  if (uy >= 0 && ux >= 0 && uy + g.WY <= g.Hs && ux + g.WX <= g.Ws)
    gather_tile<true>(X, cm, ko, kvo, g.K2, rsrc, j, uy, ux, g.Hs, g.Ws, lane);
  else
    gather_tile<false>(X, cm, ko, kvo, g.K2, rsrc, j, uy, ux, g.Hs, g.Ws, lane);
}

// Round 0, forward edge into ReLU layer 1: the source rows are the input embedding E0 = relu(inp_f([l0, x, u0]))
// (graph_conv.py:90-95), 6 FMAs per row and channel pair -- cheaper to recompute per k-step under the MFMAs than to write
// 201 MB of rows (k_embed) and read them back.  Same structure as gather_tile; the three scalars of a window node come
// from buffer loads (out-of-range -> masked explicitly, since relu(bias) of a zero input is not zero).
struct EmbedSrc { const float *lb, *x, *ub; const float* wb; };     // (B, Ns) scalars; inp_f weight (64 x 3) then bias (64)

template <bool INTERIOR>
__device__ __forceinline__ void gather_tile_embed(Frag& X, const float* cm, const int2* ko, const unsigned* kvo, int K2,
                                                  __amdgpu_buffer_rsrc_t rl, __amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t ru,
                                                  const float (&w)[2][3], const float (&bias)[2], int wy0, int wx0, int Hs, int Ws, int lane) {
  const int h = lane >> 5;
#pragma unroll
  for (int R = 0; R < 32; ++R) FRAG_AT(X, R) = 0.0f;
  const int origin = wy0 * Ws + wx0;
  const unsigned soff = (unsigned)origin * 4u;
  struct Chunk { float l[GATHER_CH], x[GATHER_CH], u[GATHER_CH]; unsigned o[GATHER_CH]; };
  Chunk cur, nxt;
  auto load = [&](Chunk& c, int s0) {
#pragma unroll
    for (int q = 0; q < GATHER_CH; ++q) {
      unsigned o;
      if (INTERIOR) {
        o = kvo[2 * (s0 + q) + h] >> 6;                 // byte offset of the window node in a float array
        c.l[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rl, o, soff, 0));
        c.x[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, o, soff, 0));
        c.u[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ru, o, soff, 0));
      } else {
        const unsigned long long ev = reinterpret_cast<const unsigned long long*>(ko)[2 * (s0 + q) + h];
        const int ex = (int)(unsigned)ev, ey = (int)(unsigned)(ev >> 32);
        const int wy = wy0 + (ey & 0xffff), wx = wx0 + (ey >> 16);
        o = (unsigned)(origin + ex) * 4u;
        o = ((unsigned)wy < (unsigned)Hs && (unsigned)wx < (unsigned)Ws) ? o : BUF_OOB;
        c.l[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rl, o, 0, 0));
        c.x[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, o, 0, 0));
        c.u[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ru, o, 0, 0));
      }
      c.o[q] = o;
    }
  };
  auto mma = [&](const Chunk& c, int s0) {
#pragma unroll
    for (int q = 0; q < GATHER_CH; ++q) {
      const float b = cm[(s0 + q) * 64 + lane];
      float e0 = relu_nan(fmaf(c.u[q], w[0][2], fmaf(c.x[q], w[0][1], fmaf(c.l[q], w[0][0], bias[0]))));
      float e1 = relu_nan(fmaf(c.u[q], w[1][2], fmaf(c.x[q], w[1][1], fmaf(c.l[q], w[1][0], bias[1]))));
      if (!INTERIOR) {
        const bool v = c.o[q] != BUF_OOB;
        e0 = v ? e0 : 0.0f;
        e1 = v ? e1 : 0.0f;
      }
      X.t[0] = mfma32(e0, b, X.t[0]);
      X.t[1] = mfma32(e1, b, X.t[1]);
    }
  };
  load(cur, 0);
  const int npairs = K2 / (2 * GATHER_CH);
  int s0 = 0;
  for (int pr = 0; pr < npairs; ++pr, s0 += 2 * GATHER_CH) {
    load(nxt, s0 + GATHER_CH);
    __builtin_amdgcn_sched_barrier(0);
    mma(cur, s0);
    __builtin_amdgcn_sched_barrier(0);
    load(cur, s0 + 2 * GATHER_CH);
    __builtin_amdgcn_sched_barrier(0);
    mma(nxt, s0 + GATHER_CH);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (K2 & GATHER_CH) mma(cur, s0);
}

// ---- 16-node tiles on v_mfma_f32_16x16x4_f32 (forward conv edges: a third less window per node than 32-node tiles) ----
// lane l = (i = l & 15, g = l >> 4).  k-step s covers window slots 4s .. 4s+3; lane (i, g) loads channels 4i .. 4i+3 of slot 4s+g
// (one b128; the 16 lanes of a group read one whole 256-B row) and feeds channel 4i+t to the MFMA of M-tile t; the B operand
// is the tap weight of (slot 4s+g, dst node i).  D of tile t: lane (j, g'), register r = channel 16g' + 4r + t of dst node j,
// so a lane ends up with 16 consecutive channels of its node.
typedef float f32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
#define GATHER_CH16 4   // k-steps per prefetch chunk (16 window slots, as in the 32-lane variant)
#ifndef EMBED_MFMA
#define EMBED_MFMA 1    // round 0: the input embedding inside the first gather on the matrix pipe (0: VALU form)
#endif

template <bool INTERIOR>
__device__ __forceinline__ void gather_tile16(f32x4 (&acc)[4], const float* cm, const int2* ko, const unsigned* kvo, int K2,
                                              __amdgpu_buffer_rsrc_t rsrc, int wy0, int wx0, int Hs, int Ws, int lane) {
  const int g = lane >> 4, i = lane & 15;
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int origin = wy0 * Ws + wx0;
  const unsigned lane_off = 16u * (unsigned)i;
  const unsigned soff = (unsigned)origin * 256u;
  f32x4 cur[GATHER_CH16], nxt[GATHER_CH16];
  auto load = [&](f32x4 (&dst)[GATHER_CH16], int s0) {
#pragma unroll
    for (int u = 0; u < GATHER_CH16; ++u) {
      if (INTERIOR) {
        const unsigned vo = kvo[4 * (s0 + u) + g] + lane_off;
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, soff, 0);
        dst[u] = f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
      } else {
        const unsigned long long ev = reinterpret_cast<const unsigned long long*>(ko)[4 * (s0 + u) + g];
        const int ex = (int)(unsigned)ev, ey = (int)(unsigned)(ev >> 32);
        const int wy = wy0 + (ey & 0xffff), wx = wx0 + (ey >> 16);
        unsigned o = (unsigned)(origin + ex) * 256u + lane_off;
        o = ((unsigned)wy < (unsigned)Hs && (unsigned)wx < (unsigned)Ws) ? o : BUF_OOB;
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0);
        dst[u] = f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
      }
    }
  };
  auto mma = [&](const f32x4 (&v)[GATHER_CH16], int s0) {
#pragma unroll
    for (int u = 0; u < GATHER_CH16; ++u) {
      const float b = cm[(s0 + u) * 64 + lane];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mfma16(v[u][t], b, acc[t]);
    }
  };
  load(cur, 0);
  const int npairs = K2 / (2 * GATHER_CH16);
  int s0 = 0;
  for (int pr = 0; pr < npairs; ++pr, s0 += 2 * GATHER_CH16) {
    load(nxt, s0 + GATHER_CH16);
    __builtin_amdgcn_sched_barrier(0);
    mma(cur, s0);
    __builtin_amdgcn_sched_barrier(0);
    load(cur, s0 + 2 * GATHER_CH16);
    __builtin_amdgcn_sched_barrier(0);
    mma(nxt, s0 + GATHER_CH16);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (K2 & GATHER_CH16) mma(cur, s0);
}

// Sparse variant: the rows of dead source nodes are exactly zero, so a tile first compacts the live, in-range slots of its
// window into a per-wave LDS table {byte offset of the row, tap-matrix row} and then walks only those (-35..45 % k-steps
// behind a ReLU layer).  `tab`: 4*K2 + 32 entries of this wave; slb / sub: bounds of the source layer of this sample.
#ifndef GATHER_CHS16
#define GATHER_CHS16 4
#endif
__device__ __forceinline__ void gather_tile16_sparse(f32x4 (&acc)[4], const float* cm, const int2* ko, uint2* tab, int K2,
                                                     __amdgpu_buffer_rsrc_t rsrc, const float* slb, const float* sub, int wy0, int wx0,
                                                     int Hs, int Ws, int lane) {
  const int g = lane >> 4, i = lane & 15;
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int origin = wy0 * Ws + wx0;
  int n = 0;
  for (int base = 0; base < 4 * K2; base += 64) {
    const int sl = base + lane;
    const unsigned long long ev = reinterpret_cast<const unsigned long long*>(ko)[sl];
    const int ex = (int)(unsigned)ev, ey = (int)(unsigned)(ev >> 32);
    const int wy = wy0 + (ey & 0xffff), wx = wx0 + (ey >> 16);
    const bool inb = sl < 4 * K2 && (unsigned)wy < (unsigned)Hs && (unsigned)wx < (unsigned)Ws;      // (table padding: 0x7fff, never in range)
    const int row = inb ? origin + ex : 0;
    const bool live = inb && node_is_live(slb[row], sub[row]);
    const unsigned long long bal = __ballot(live);
    if (live) tab[n + __popcll(bal & ((1ull << lane) - 1ull))] = make_uint2((unsigned)row * 256u, (unsigned)sl * 16u);
    n += __popcll(bal);
  }
  constexpr int CS = 4 * GATHER_CHS16;
  const int npad = (n + CS - 1) / CS * CS;
  for (int q = n + lane; q < npad + CS; q += 64) tab[q] = make_uint2(BUF_OOB, 0u);       // out-of-range offset: the load returns 0
  const int K2e = npad / 4;
  const unsigned lane_off = 16u * (unsigned)i;
  struct Chunk { f32x4 v[GATHER_CHS16]; unsigned cr[GATHER_CHS16]; };
  Chunk cur, nxt;
  auto load = [&](Chunk& c, int s0) {
#pragma unroll
    for (int u = 0; u < GATHER_CHS16; ++u) {
      const uint2 e = tab[4 * (s0 + u) + g];
      const unsigned o = e.x == BUF_OOB ? BUF_OOB : e.x + lane_off;
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0);
      c.v[u] = f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
      c.cr[u] = e.y;
    }
  };
  auto mma = [&](const Chunk& c, int) {
#pragma unroll
    for (int u = 0; u < GATHER_CHS16; ++u) {
      const float b = cm[c.cr[u] + i];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mfma16(c.v[u][t], b, acc[t]);
    }
  };
  load(cur, 0);
  const int npairs = K2e / (2 * GATHER_CHS16);
  int s0 = 0;
  for (int pr = 0; pr < npairs; ++pr, s0 += 2 * GATHER_CHS16) {
    load(nxt, s0 + GATHER_CHS16);
    __builtin_amdgcn_sched_barrier(0);
    mma(cur, s0);
    __builtin_amdgcn_sched_barrier(0);
    load(cur, s0 + 2 * GATHER_CHS16);
    __builtin_amdgcn_sched_barrier(0);
    mma(nxt, s0 + GATHER_CHS16);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (K2e & GATHER_CHS16) mma(cur, s0);
}

// the embedding variant (round 0, first edge): the four channels of a slot are computed from its three input scalars
template <bool INTERIOR>
__device__ __forceinline__ void gather_tile16_embed(f32x4 (&acc)[4], const float* cm, const int2* ko, const unsigned* kvo, int K2,
                                                    __amdgpu_buffer_rsrc_t rl, __amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t ru,
                                                    const float (&w)[4][3], const float (&bias)[4], int wy0, int wx0, int Hs, int Ws, int lane) {
  const int g = lane >> 4;
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int origin = wy0 * Ws + wx0;
  const unsigned soff = (unsigned)origin * 4u;
  struct Chunk { float l[GATHER_CH16], x[GATHER_CH16], u[GATHER_CH16]; unsigned o[GATHER_CH16]; };
  Chunk cur, nxt;
  auto load = [&](Chunk& c, int s0) {
#pragma unroll
    for (int q = 0; q < GATHER_CH16; ++q) {
      unsigned o;
      if (INTERIOR) {
        o = kvo[4 * (s0 + q) + g] >> 6;                 // byte offset of the window node in a float array
        c.l[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rl, o, soff, 0));
        c.x[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, o, soff, 0));
        c.u[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ru, o, soff, 0));
      } else {
        const unsigned long long ev = reinterpret_cast<const unsigned long long*>(ko)[4 * (s0 + q) + g];
        const int ex = (int)(unsigned)ev, ey = (int)(unsigned)(ev >> 32);
        const int wy = wy0 + (ey & 0xffff), wx = wx0 + (ey >> 16);
        o = (unsigned)(origin + ex) * 4u;
        o = ((unsigned)wy < (unsigned)Hs && (unsigned)wx < (unsigned)Ws) ? o : BUF_OOB;
        c.l[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rl, o, 0, 0));
        c.x[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, o, 0, 0));
        c.u[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ru, o, 0, 0));
      }
      c.o[q] = o;
    }
  };
  auto mma = [&](const Chunk& c, int s0) {
#pragma unroll
    for (int q = 0; q < GATHER_CH16; ++q) {
      const float b = cm[(s0 + q) * 64 + lane];
      const bool v = INTERIOR || c.o[q] != BUF_OOB;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float e = relu_nan(fmaf(c.u[q], w[t][2], fmaf(c.x[q], w[t][1], fmaf(c.l[q], w[t][0], bias[t]))));
        e = v ? e : 0.0f;
        acc[t] = mfma16(e, b, acc[t]);
      }
    }
  };
  load(cur, 0);
  const int npairs = K2 / (2 * GATHER_CH16);
  int s0 = 0;
  for (int pr = 0; pr < npairs; ++pr, s0 += 2 * GATHER_CH16) {
    load(nxt, s0 + GATHER_CH16);
    __builtin_amdgcn_sched_barrier(0);
    mma(cur, s0);
    __builtin_amdgcn_sched_barrier(0);
    load(cur, s0 + 2 * GATHER_CH16);
    __builtin_amdgcn_sched_barrier(0);
    mma(nxt, s0 + GATHER_CH16);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (K2 & GATHER_CH16) mma(cur, s0);
}

// number of kernel taps that touch dst position t along one axis (the reference's `freq`, graph_conv.py:306-311)
__device__ __forceinline__ int tap_count(int t, int w0, int WN, int Hs, int k, int stride, int pad) {
  int n = 0;
  for (int w = 0; w < WN; ++w) {
    const int o = w0 + w;
    const int kk = t + pad - o * stride;
    n += ((unsigned)o < (unsigned)Hs && kk >= 0 && kk < k) ? 1 : 0;
  }
  return n;
}

__device__ __forceinline__ void stage_gather(float* lds_cm, int2* lds_ko, int* lds_tt, unsigned* lds_kvo, const DGather& g, int TPS) {
  copy_to_lds(lds_cm, g.cmat, g.ncg_k2 * 64);
  for (int i = threadIdx.x; i < gather_slots(g.K2, g.lanes); i += blockDim.x) {
    const int2 e = g.koff[i];
    lds_ko[i] = e;
    lds_kvo[i] = (e.y & 0xffff) == 0x7fff ? 0u : (unsigned)e.x * 256u;      // byte offset of window node i from the window origin
  }
  for (int i = threadIdx.x; i < TPS; i += blockDim.x) lds_tt[i] = g.ttab[i];
}

// [r0 != 0] without the division: r0 = u+/(u+ - l-) is zero iff u+ == 0 and l- != 0 (0/0 is NaN, and NaN != 0)
__device__ __forceinline__ bool node_is_live(float lb, float ub) {
  const float lower_temp = lb - relu_nan(lb);
  const float upper_temp = relu_nan(ub);
  return !(upper_temp == 0.0f) || lower_temp == 0.0f;
}

// rows of the gathered fragment (gather channel map) -> row-major (.., 64): lane (j,h) owns channels [16q+8h, 16q+8h+8)
__device__ __forceinline__ void frag_store_rows_gathered(const Frag& x, float* base, long row, int h) {
  f32x4* p = reinterpret_cast<f32x4*>(base + row * 64 + 8 * h);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 v0, v1;
    v0[0] = x.t[0][4 * q + 0]; v0[1] = x.t[1][4 * q + 0]; v0[2] = x.t[0][4 * q + 1]; v0[3] = x.t[1][4 * q + 1];
    v1[0] = x.t[0][4 * q + 2]; v1[1] = x.t[1][4 * q + 2]; v1[2] = x.t[0][4 * q + 3]; v1[3] = x.t[1][4 * q + 3];
    p[4 * q] = v0;
    p[4 * q + 1] = v1;
  }
}

struct GArgs {
  const float *lb, *ub;     // bounds of the dst layer, flat (B*N)
  const float* mask;        // (B, R) BaB mask, used when `need_scored`
  const float* mu_src;      // (B, Ns, 64)
  float* nb;                // out: rows by node id (B*N, 64), written for the lanes that need it
  long ntiles;
  int need_scored, R, off;  // 0: every live node needs its aggregate; 1: only the scored nodes (last backward step)
  DTileMap tm;
  DGather g;
  EmbedSrc es;              // EMBED: the source rows are computed from the input scalars (mu_src unused)
  const float *src_lb, *src_ub;   // SPARSE: bounds of the source layer (B, Ns): the rows of its dead nodes are zero and skipped
};

// EMBED: inp_f rows of this lane's channels 2j, 2j+1
struct EmbedLane { float w[2][3], b[2]; };
template <bool EMBED>
__device__ __forceinline__ EmbedLane embed_lane(const GArgs& a, int j) {
  EmbedLane e{};
  if (EMBED) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      e.b[c] = a.es.wb[192 + 2 * j + c];
#pragma unroll
      for (int i = 0; i < 3; ++i) e.w[c][i] = a.es.wb[(2 * j + c) * 3 + i];
    }
  }
  return e;
}

// one tile of phase A: the aggregate rows of the tile's dst nodes that will be updated
template <bool EMBED, bool SPARSE>
__device__ __forceinline__ void gather_process_tile(const GArgs& a, const TileCtx& tc, int sample, const float* lds_cm, const int2* lds_ko,
                                                    const unsigned* lds_kvo, uint2* tab, const EmbedLane& el, int lane) {
  const int h = lane >> 5, j = lane & 31;
  const long gc = tc.sample * a.tm.N + tc.n;
  bool need;
  if (a.need_scored) need = tc.valid && a.mask[tc.sample * a.R + a.off + tc.n] != 0.0f;
  else need = tc.valid && node_is_live(a.lb[gc], a.ub[gc]);    // (one load of k_classify's live flag instead: measured 1.7 % slower)
  if (!__any(need)) return;
  const int wy0 = tc.by * a.g.ystep + a.g.ybase, wx0 = tc.bx * a.g.xstep + a.g.xbase;
  Frag X;
  if (EMBED) {
    const int uy = __builtin_amdgcn_readfirstlane(wy0), ux = __builtin_amdgcn_readfirstlane(wx0);
    const long sb = (long)sample * a.g.Ns;
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)(a.es.lb + sb), 0, a.g.Ns * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.es.x + sb), 0, a.g.Ns * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void*)(a.es.ub + sb), 0, a.g.Ns * 4, 0x00020000);
    const float* cmt = lds_cm + tc.cg * a.g.K2 * 64;
    if (uy >= 0 && ux >= 0 && uy + a.g.WY <= a.g.Hs && ux + a.g.WX <= a.g.Ws)
      gather_tile_embed<true>(X, cmt, lds_ko, lds_kvo, a.g.K2, rl, rx, ru, el.w, el.b, uy, ux, a.g.Hs, a.g.Ws, lane);
    else
      gather_tile_embed<false>(X, cmt, lds_ko, lds_kvo, a.g.K2, rl, rx, ru, el.w, el.b, uy, ux, a.g.Hs, a.g.Ws, lane);
  } else {
    if (SPARSE)
      gather_dispatch(X, lds_cm + tc.cg * a.g.K2 * 64, lds_ko, lds_kvo, a.g, a.mu_src + (long)sample * a.g.Ns * 64, j, wy0, wx0, lane, tab,
                      a.src_lb + (long)sample * a.g.Ns, a.src_ub + (long)sample * a.g.Ns);
    else
      gather_dispatch(X, lds_cm + tc.cg * a.g.K2 * 64, lds_ko, lds_kvo, a.g, a.mu_src + (long)sample * a.g.Ns * 64, j, wy0, wx0, lane);
  }
  if (a.g.normalise) {
    const int ny = tap_count(tc.y, wy0, a.g.WY, a.g.Hs, a.g.kh, a.g.stride, a.g.pad);
    const int nx = tap_count(tc.x, wx0, a.g.WX, a.g.Ws, a.g.kw, a.g.stride, a.g.pad);
    const int f = tc.valid ? ny * nx : 1;
    const float freq = (float)f;
    if (__all((f & (f - 1)) == 0)) {         // power of two: x * (1/f) is exactly x / f
      const float inv = 1.0f / freq;
#pragma unroll
      for (int R = 0; R < 32; ++R) FRAG_AT(X, R) = FRAG_AT(X, R) * inv;
    } else {
#pragma unroll
      for (int R = 0; R < 32; ++R) FRAG_AT(X, R) = FRAG_AT(X, R) / freq;
    }
  }
  if (need) frag_store_rows_gathered(X, a.nb, gc, h);
}

// The embedding on the matrix pipe: E0 = relu(inp_f [l, x, u] + b) is itself a K = 4 product ([l, x, u, 1] against [W | b]),
// and the result layout of v_mfma_f32_16x16x4_f32 (lane (i, g), register r = row 4g + r, column i) is the A-operand layout
// of the tap MFMA (lane (i, g) = channel of row i at the slot of k-index g) if the embedding MFMA's row 4g + r is the window
// slot that k-step 4 grp + r wants at k-index g, i.e. slot 16 grp + 4 r + g.  So per group of 4 k-steps: one scalar load per
// lane (lane group 0 / 1 / 2 reads l / x / u of its row's slot, group 3 supplies the 1 of the bias), 4 embedding MFMAs (one
// per 16-channel tile), 16 v_max, 16 tap MFMAs -- instead of 64 FMAs + 16 v_max + 12 loads per lane.  An out-of-range slot
// feeds zeros (including its "1"), so its embedding is relu(0) = 0.
__device__ __forceinline__ void gather_tile16_embed_mfma(f32x4 (&acc)[4], const float* cm, const int2* ko, int K2, __amdgpu_buffer_rsrc_t rl,
                                                         __amdgpu_buffer_rsrc_t rx, __amdgpu_buffer_rsrc_t ru, const float (&bw)[4],
                                                         int wy0, int wx0, int Hs, int Ws, int lane) {
  const int m = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int origin = wy0 * Ws + wx0;
  const int sl = 4 * (m & 3) + (m >> 2);            // slot of embedding row m inside a group of 16
  auto load = [&](int grp) -> float {
    const unsigned long long ev = reinterpret_cast<const unsigned long long*>(ko)[16 * grp + sl];
    const int ex = (int)(unsigned)ev, ey = (int)(unsigned)(ev >> 32);
    const int wy = wy0 + (ey & 0xffff), wx = wx0 + (ey >> 16);
    const bool inb = (unsigned)wy < (unsigned)Hs && (unsigned)wx < (unsigned)Ws;      // (table padding: 0x7fff, never in range)
    const unsigned o = inb ? (unsigned)(origin + ex) * 4u : BUF_OOB;
    const float vl = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rl, o, 0, 0));
    const float vx = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, o, 0, 0));
    const float vu = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ru, o, 0, 0));
    return kq == 0 ? vl : kq == 1 ? vx : kq == 2 ? vu : (inb ? 1.0f : 0.0f);
  };
  const int ngrp = K2 / 4;
  float ain = load(0);
  for (int grp = 0; grp < ngrp; ++grp) {
    const float cur = ain;
    if (grp + 1 < ngrp) ain = load(grp + 1);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 e[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      e[t] = mfma16(cur, bw[t], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int r = 0; r < 4; ++r) e[t][r] = relu_nan(e[t][r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float b = cm[(4 * grp + r) * 64 + lane];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mfma16(e[t][r], b, acc[t]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// one 16-node tile of phase A (forward edges only: no tap-count division)
template <bool EMBED, bool SPARSE>
__device__ __forceinline__ void gather_process_tile16(const GArgs& a, const TileCtx& tc, int sample, const float* lds_cm, const int2* lds_ko,
                                                      const unsigned* lds_kvo, uint2* tab, const float (&ew)[4][3], const float (&eb)[4], int lane) {
  const int gq = lane >> 4;
  const long gc = tc.sample * a.tm.N + tc.n;
  bool need;
  if (a.need_scored) need = tc.valid && a.mask[tc.sample * a.R + a.off + tc.n] != 0.0f;
  else need = tc.valid && node_is_live(a.lb[gc], a.ub[gc]);
  if (!__any(need)) return;
  const int wy0 = tc.by * a.g.ystep + a.g.ybase, wx0 = tc.bx * a.g.xstep + a.g.xbase;
  const int uy = __builtin_amdgcn_readfirstlane(wy0), ux = __builtin_amdgcn_readfirstlane(wx0);
  const bool interior = uy >= 0 && ux >= 0 && uy + a.g.WY <= a.g.Hs && ux + a.g.WX <= a.g.Ws;
  const float* cmt = lds_cm + tc.cg * a.g.K2 * 64;
  f32x4 acc[4];
  if (EMBED) {
    const long sb = (long)sample * a.g.Ns;
    const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void*)(a.es.lb + sb), 0, a.g.Ns * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(a.es.x + sb), 0, a.g.Ns * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void*)(a.es.ub + sb), 0, a.g.Ns * 4, 0x00020000);
    if (EMBED_MFMA) {
      // B operand of the embedding MFMA of channel tile t: lane (n, k) = inp_f weight k of channel 4n + t, k = 3: its bias
      const int n = lane & 15, k = lane >> 4;
      float bw[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) bw[t] = k < 3 ? a.es.wb[(4 * n + t) * 3 + k] : a.es.wb[192 + 4 * n + t];
      gather_tile16_embed_mfma(acc, cmt, lds_ko, a.g.K2, rl, rx, ru, bw, uy, ux, a.g.Hs, a.g.Ws, lane);
    } else if (interior) gather_tile16_embed<true>(acc, cmt, lds_ko, lds_kvo, a.g.K2, rl, rx, ru, ew, eb, uy, ux, a.g.Hs, a.g.Ws, lane);
    else gather_tile16_embed<false>(acc, cmt, lds_ko, lds_kvo, a.g.K2, rl, rx, ru, ew, eb, uy, ux, a.g.Hs, a.g.Ws, lane);
  } else {
    const float* sbase = a.mu_src + (long)sample * a.g.Ns * 64;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)sbase, 0, a.g.Ns * 256, 0x00020000);
    if (SPARSE) {
      const long sb = (long)sample * a.g.Ns;
      gather_tile16_sparse(acc, cmt, lds_ko, tab, a.g.K2, rsrc, a.src_lb + sb, a.src_ub + sb, uy, ux, a.g.Hs, a.g.Ws, lane);
    } else if (interior) gather_tile16<true>(acc, cmt, lds_ko, lds_kvo, a.g.K2, rsrc, uy, ux, a.g.Hs, a.g.Ws, lane);
    else gather_tile16<false>(acc, cmt, lds_ko, lds_kvo, a.g.K2, rsrc, uy, ux, a.g.Hs, a.g.Ws, lane);
  }
  if (need) {                                  // lane (j, g'): channels 16g' + 4r + t of its node
    f32x4* p = reinterpret_cast<f32x4*>(a.nb + gc * 64 + 16 * gq);
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r] = f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
  }
}

// LDS image of a gather's tables: tap matrix, window offsets (two forms), tile table
struct GatherLds { float* cm; int2* ko; int* tt; unsigned* kvo; };
__device__ __forceinline__ GatherLds gather_lds(float* base, const DGather& g, int TPS) {
  GatherLds l;
  l.cm = base;
  l.ko = reinterpret_cast<int2*>(l.cm + g.ncg_k2 * 64);
  l.tt = reinterpret_cast<int*>(l.ko + gather_slots(g.K2, g.lanes));
  l.kvo = reinterpret_cast<unsigned*>(l.tt + ((TPS + 3) & ~3));
  return l;
}

// phase A of a half-pass over a conv edge: nb[g] = sum over the window for the dst nodes that will be updated
template <bool EMBED, bool SPARSE = false>
__global__ __launch_bounds__(WG_MLP, 2) void k_gather(GArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const GatherLds gl = gather_lds(lds, a.g, a.tm.TPS);
  stage_gather(gl.cm, gl.ko, gl.tt, gl.kvo, a.g, a.tm.TPS);
  __syncthreads();
  const int lane = threadIdx.x & 63, j = lane & 31, wave = threadIdx.x >> 6;
  uint2* tab = reinterpret_cast<uint2*>(gl.kvo + ((gather_slots(a.g.K2, 32) + 1) & ~1)) + wave * (2 * a.g.K2 + 32);      // SPARSE
  const EmbedLane el = embed_lane<EMBED>(a, j);
  long t0, t1;
  tile_range(a.ntiles, WAVES_MLP, t0, t1);
  long tile = t0 + wave;
  if (tile >= t1) return;
  // tile, sample and t are wave-uniform by construction; make them provably so (scalar registers, scalar base address)
  int sample = __builtin_amdgcn_readfirstlane((int)(tile / a.tm.TPS));
  int t = __builtin_amdgcn_readfirstlane((int)(tile - (long)sample * a.tm.TPS));
  for (; tile < t1; tile += WAVES_MLP, t += WAVES_MLP) {
    while (t >= a.tm.TPS) { t -= a.tm.TPS; ++sample; }
    const TileCtx tc = block_decode(a.tm, gl.tt, sample, t, j);
    gather_process_tile<EMBED, SPARSE>(a, tc, sample, gl.cm, gl.ko, gl.kvo, tab, el, lane);
  }
}

// the 16-node-tile form of k_gather (forward conv edges)
// (4 waves per SIMD: the embedding variant sits right at 128 VGPRs, and at 130 it loses a quarter of its waves and 10 %)
template <bool EMBED, bool SPARSE = false>
__global__ __launch_bounds__(WG_MLP, 4) void k_gather16(GArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const GatherLds gl = gather_lds(lds, a.g, a.tm.TPS);
  stage_gather(gl.cm, gl.ko, gl.tt, gl.kvo, a.g, a.tm.TPS);
  __syncthreads();
  const int lane = threadIdx.x & 63, j = lane & 15, wave = threadIdx.x >> 6;
  // SPARSE: per-wave table of the live window slots, behind the shared tables
  uint2* tab = reinterpret_cast<uint2*>(gl.kvo + ((gather_slots(a.g.K2, 16) + 1) & ~1)) + wave * (4 * a.g.K2 + 32);
  float ew[4][3] = {}, eb[4] = {};               // EMBED: inp_f rows of this lane's channels 4i .. 4i+3
  if (EMBED) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      eb[c] = a.es.wb[192 + 4 * j + c];
#pragma unroll
      for (int q = 0; q < 3; ++q) ew[c][q] = a.es.wb[(4 * j + c) * 3 + q];
    }
  }
  // Rounds of 8 tiles (one per wave) are dealt ROUND-ROBIN over the workgroups, in the XCD-grouped order of tile_range:
  // at any moment the 64 workgroups of an XCD then work on 8 neighbouring rounds each side by side, i.e. on 8-16 samples
  // whose source rows (~4 MB) stay in that XCD's L2 -- with one contiguous chunk per workgroup they covered 32 samples,
  // 16 MB, and every window row was fetched from HBM 1.7 times.
  int wg = blockIdx.x;
  const int nwg = gridDim.x;
  if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);
  const long nrounds = (a.ntiles + WAVES_MLP - 1) / WAVES_MLP;
  for (long r = wg; r < nrounds; r += nwg) {
    const long tile = r * WAVES_MLP + wave;
    if (tile >= a.ntiles) break;
    const int sample = __builtin_amdgcn_readfirstlane((int)(tile / a.tm.TPS));
    const int t = __builtin_amdgcn_readfirstlane((int)(tile - (long)sample * a.tm.TPS));
    const TileCtx tc = block_decode(a.tm, gl.tt, sample, t, j);
    gather_process_tile16<EMBED, SPARSE>(a, tc, sample, gl.cm, gl.ko, gl.kvo, tab, ew, eb, lane);
  }
}

struct GIArgs {
  const float* pack_pre;    // PackPreInp
  const float* pack;        // PackUpdInp (gather variant)
  const float *lb, *ub;     // input bounds, flat (B*N0)
  const float* mu_src; const float* sarr; float* mu; long ntiles; DTileMap tm; DGather g;
  const float *src_lb, *src_ub;     // SPARSE: bounds of ReLU layer 1 (the rows of its dead nodes are zero and skipped)
};

// input layer: E_0 = relu(Q + inp_b2[:, 64:] . (A_1^T mu_1)),  Q = inp_b2[:, :64] . inp_b_1(relu(inp_b([l0,u0]))) + b;
// mu_0 = inp_b2_2(E_0) is deferred into the next round's forward update of ReLU layer 1 (gnnb_pack.h).
// graph_conv.py:361-385; the aggregate, the feature chain and the update stay in registers.
// one tile of the fused input-layer update; lds_upd / lds_pre: PackUpdInp / PackPreInp in LDS
template <bool SPARSE, bool BF3>
__device__ __forceinline__ void input_update_tile(const GIArgs& a, const TileCtx& tc, int sample, const float* lds_upd, const float* lds_pre,
                                                  const GatherLds& gl, uint2* tab, int lane) {
  const int h = lane >> 5, j = lane & 31;
  if (!__any(tc.valid)) return;
  const long gc = tc.sample * a.tm.N + tc.n;
  const int wy0 = tc.by * a.g.ystep + a.g.ybase, wx0 = tc.bx * a.g.xstep + a.g.xbase;
  Frag X;
  if (SPARSE)
    gather_dispatch(X, gl.cm + tc.cg * a.g.K2 * 64, gl.ko, gl.kvo, a.g, a.mu_src + (long)sample * a.g.Ns * 64, j, wy0, wx0, lane, tab,
                    a.src_lb + (long)sample * a.g.Ns, a.src_ub + (long)sample * a.g.Ns);
  else
    gather_dispatch(X, gl.cm + tc.cg * a.g.K2 * 64, gl.ko, gl.kvo, a.g, a.mu_src + (long)sample * a.g.Ns * 64, j, wy0, wx0, lane);
  float x[1];
  x[0] = h ? a.ub[gc] : a.lb[gc];
  Frag H0;
  frag_bias(H0, lds_pre + PackPreInp::B1, h);
  gemm_small<1>(lds_pre + PackPreInp::W1, lane, H0, x);
  frag_relu(H0);
  Frag H;                                  // inp_b_1 and the first half of inp_b2 are folded into one 64x64 map
  frag_bias(H, lds_pre + PackPreInp::B2, h);
  if (BF3) gemm_w64_bf3<1>(lds_pre + PackPreInp::W23, lane, H, [&](int s) { return FRAG_AT(H0, s); });
  else gemm_w64<32>(lds_pre + PackPreInp::W2, lane, H, [&](int s) { return FRAG_AT(H0, s); });
  {                                          // bias term of the projection deferred in the rows of mu_1
    const float xs[1] = {h ? 0.0f : a.sarr[gc]};
    gemm_small<1>(lds_upd + PackUpdInp::VC, lane, H, xs);
  }
  // the aggregate already went through inp_b2[:, 64:].bc4_1.W on the producer side, with its rows permuted to this
  // fragment layout (PackPostInp::WPG): register for register
#pragma unroll
  for (int R = 0; R < 32; ++R) FRAG_AT(H, R) += FRAG_AT(X, R);
  frag_relu(H);
  if (tc.valid) frag_store_rows(H, a.mu, gc, h);
}

template <bool SPARSE, bool BF3>
__global__ __launch_bounds__(WG_MLP, 2) void k_gather_input_update(GIArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lds_pre = lds + PackUpdInp::FLOATS;
  const GatherLds gl = gather_lds(lds_pre + PackPreInp::FLOATS, a.g, a.tm.TPS);
  stage_gather(gl.cm, gl.ko, gl.tt, gl.kvo, a.g, a.tm.TPS);
  copy_to_lds(lds_pre, a.pack_pre, PackPreInp::FLOATS);
  stage_pack(lds, a.pack, PackUpdInp::FLOATS);
  const int lane = threadIdx.x & 63, j = lane & 31, wave = threadIdx.x >> 6;
  uint2* tab = reinterpret_cast<uint2*>(gl.kvo + ((gather_slots(a.g.K2, 32) + 1) & ~1)) + wave * (2 * a.g.K2 + 32);      // SPARSE
  long t0, t1;
  tile_range(a.ntiles, WAVES_MLP, t0, t1);
  long tile = t0 + wave;
  if (tile >= t1) return;
  // tile, sample and t are wave-uniform by construction; make them provably so (scalar registers, scalar base address)
  int sample = __builtin_amdgcn_readfirstlane((int)(tile / a.tm.TPS));
  int t = __builtin_amdgcn_readfirstlane((int)(tile - (long)sample * a.tm.TPS));
  for (; tile < t1; tile += WAVES_MLP, t += WAVES_MLP) {
    while (t >= a.tm.TPS) { t -= a.tm.TPS; ++sample; }
    const TileCtx tc = block_decode(a.tm, gl.tt, sample, t, j);
    input_update_tile<SPARSE, BF3>(a, tc, sample, lds, lds_pre, gl, tab, lane);
  }
}

struct ScoreArgs {        // every ReLU layer in one launch
  const float* pack; float* scores;
  int L, R;
  const float* mu[MAXL]; const int* list[MAXL];
  const float* lb[MAXL]; const float* ub[MAXL];
  const int* cnt;         // cnt[4k + 2] = number of scored nodes of layer k
  int N[MAXL], off[MAXL]; // nodes per sample in layer k, offset of layer k in the flat ReLU index
};

// score = fscore(relu(fnode(mu_g))) for the nodes g whose BaB mask is -1 (the rest stays -inf)    graph_conv.py:445-450
// the rows hold E_g with mu_g = (Wp.E_g + bp).live: fnode is pre-multiplied by Wp, fnode.bp.live enters as a small k-step
// one tile (32 scored nodes `list[32 t ..]` of layer k); lds: PackScore
__device__ __forceinline__ void score_tile(const ScoreArgs& a, const float* lds, int k, const int* list, int count, long t, int lane) {
  const int h = lane >> 5, j = lane & 31;
  const float bs = lds[PackScore::BS];
  {
    const long idx = t * 32 + j;
    const bool valid = idx < count;
    const long gc = list[valid ? idx : 0];
    const int N = a.N[k];
    const long b = gc / N;
    Frag X;
    frag_load_rows(X, a.mu[k], gc, h);
    Frag H;
    frag_bias(H, lds + PackScore::B1, h);
    {
      const float live = node_is_live(a.lb[k][gc], a.ub[k][gc]) ? 1.0f : 0.0f;
      const float x[1] = {h ? 0.0f : live};
      gemm_small<1>(lds + PackScore::V1, lane, H, x);
    }
    gemm_w64<32>(lds + PackScore::W1, lane, H, [&](int s) { return FRAG_AT(X, s); });
    frag_relu(H);
    const f32x4* w4 = reinterpret_cast<const f32x4*>(lds + PackScore::WS + h * 32);
    float part = 0.0f;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 w = w4[q];
#pragma unroll
      for (int c = 0; c < 4; ++c) part = fmaf(FRAG_AT(H, 4 * q + c), w[c], part);
    }
    part += __shfl_xor(part, 32);
    if (valid && h == 0) a.scores[b * a.R + a.off[k] + (gc - b * N)] = part + bs;
  }
}

__global__ __launch_bounds__(WG_MLP, 2) void k_score(ScoreArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_pack(lds, a.pack, PackScore::FLOATS);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long ntiles = 0;
  for (int k = 0; k < a.L; ++k) ntiles += (a.cnt[4 * k + 2] + 31) / 32;
  for (long tile = (long)blockIdx.x * WAVES_MLP + wave; tile < ntiles; tile += (long)gridDim.x * WAVES_MLP) {
    int k = 0, count = 0;
    long t = tile;
    for (; k < a.L; ++k) {
      count = a.cnt[4 * k + 2];
      const long tk = (count + 31) / 32;
      if (t < tk) break;
      t -= tk;
    }
    score_tile(a, lds, k, a.list[k], count, t, lane);
  }
}

// ------------------------------------------------------------------------------------------
// message passing (edge aggregation)
// ------------------------------------------------------------------------------------------
struct ConvArgs {
  const float* src; float* dst; const float* w;
  int B, C_in, H_in, W_in, C_out, H_out, W_out, kh, kw, stride, pad, normalise;
};

// forward: nb[b,(co,oy,ox),:] = sum_{ci,ky,kx} W[co,ci,ky,kx] * mu_src[b,(ci,iy,ix),:]   graph_conv.py:110-121
// one wave per (b, oy, ox): lane = embedding channel, all C_out accumulators in registers,
// the 256-B source row is loaded once per tap and reused for C_out FMAs with scalar weights.
template <int CO>
__global__ __launch_bounds__(256) void k_conv_fwd(ConvArgs a) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (wv >= a.B * a.H_out * a.W_out) return;
  const int ox = wv % a.W_out, oy = (wv / a.W_out) % a.H_out, b = wv / (a.W_out * a.H_out);
  float acc[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) acc[co] = 0.0f;
  const float* src = a.src + (long)b * a.C_in * a.H_in * a.W_in * 64 + lane;
  for (int ci = 0; ci < a.C_in; ++ci)
    for (int ky = 0; ky < a.kh; ++ky) {
      const int iy = oy * a.stride - a.pad + ky;
      if ((unsigned)iy >= (unsigned)a.H_in) continue;
      for (int kx = 0; kx < a.kw; ++kx) {
        const int ix = ox * a.stride - a.pad + kx;
        if ((unsigned)ix >= (unsigned)a.W_in) continue;
        const float v = src[(long)((ci * a.H_in + iy) * a.W_in + ix) * 64];
        const float* w = a.w + ((ci * a.kh + ky) * a.kw + kx) * CO;
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[co] = fmaf(w[co], v, acc[co]);
      }
    }
  float* dst = a.dst + ((long)b * CO * a.H_out * a.W_out + (long)oy * a.W_out + ox) * 64 + lane;
#pragma unroll
  for (int co = 0; co < CO; ++co) dst[(long)co * a.H_out * a.W_out * 64] = acc[co];
}

// backward: nb[b,(ci,y,x),:] = sum_{co,ky,kx} W[co,ci,ky,kx] * mu_up[b,(co,oy,ox),:] with y = oy*s - p + ky,
// divided by the number of taps touching (y,x) when `normalise`         graph_conv.py:299-318 (and :361-372 without)
template <int CI>
__global__ __launch_bounds__(256) void k_convT_bwd(ConvArgs a) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  if (wv >= a.B * a.H_in * a.W_in) return;
  const int x = wv % a.W_in, y = (wv / a.W_in) % a.H_in, b = wv / (a.W_in * a.H_in);
  float acc[CI];
#pragma unroll
  for (int ci = 0; ci < CI; ++ci) acc[ci] = 0.0f;
  const float* src = a.src + (long)b * a.C_out * a.H_out * a.W_out * 64 + lane;
  int ny = 0, nx = 0;
  for (int ky = 0; ky < a.kh; ++ky) {
    const int t = y + a.pad - ky;
    if (t >= 0 && t % a.stride == 0 && t / a.stride < a.H_out) ++ny;
  }
  for (int kx = 0; kx < a.kw; ++kx) {
    const int t = x + a.pad - kx;
    if (t >= 0 && t % a.stride == 0 && t / a.stride < a.W_out) ++nx;
  }
  for (int co = 0; co < a.C_out; ++co)
    for (int ky = 0; ky < a.kh; ++ky) {
      const int ty = y + a.pad - ky;
      if (ty < 0 || ty % a.stride != 0 || ty / a.stride >= a.H_out) continue;
      const int oy = ty / a.stride;
      for (int kx = 0; kx < a.kw; ++kx) {
        const int tx = x + a.pad - kx;
        if (tx < 0 || tx % a.stride != 0 || tx / a.stride >= a.W_out) continue;
        const int ox = tx / a.stride;
        const float v = src[(long)((co * a.H_out + oy) * a.W_out + ox) * 64];
        const float* w = a.w + ((co * a.kh + ky) * a.kw + kx) * CI;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci) acc[ci] = fmaf(w[ci], v, acc[ci]);
      }
    }
  const float freq = a.normalise ? (float)(ny * nx) : 1.0f;
  float* dst = a.dst + ((long)b * CI * a.H_in * a.W_in + (long)y * a.W_in + x) * 64 + lane;
#pragma unroll
  for (int ci = 0; ci < CI; ++ci) dst[(long)ci * a.H_in * a.W_in * 64] = a.normalise ? acc[ci] / freq : acc[ci];
}

struct DenseArgs {
  const float* At;   // (8*ksq, ldA) zero-padded: At[k][i] = A[i][k]
  const float* X;    // (B, K, 64)
  float* out;        // (B, M, 64)
  const float* zero; // 64 zero floats
  int B, K, M, ldA, MT, ksq;
};

// dense edge: out[b, i, :] = sum_k A[i][k] X[b, k, :]   (graph_conv.py:131 forward, :321 backward)
// one workgroup per (b, 32-row tile of i); its 4 waves split K and are summed through LDS in a fixed order.
// D_it[i][j] on the MFMA for both channel tiles (lane j holds channels 2j, 2j+1 of the source row: one coalesced
// 256-B row per half-wave and k-step), A from L2; loads run one 8-k-step chunk ahead.
#define DENSE_CH 8
// SPLIT = true: the 4 waves of a workgroup share one (b, row tile) and split K (long K, few tiles: the forward edge);
// SPLIT = false: every wave owns its own (b, row tile) and walks all of K (short K: the transposed edge).
template <bool SPLIT>
__global__ __launch_bounds__(256) void k_dense_agg(DenseArgs a) {
  __shared__ float red[SPLIT ? 4 : 1][32][64];
  const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // the MT row tiles of one sample all stream the same source rows: keep them on one XCD (blockIdx % 8 labels the
  // XCD group), so that sample is fetched into one L2 instead of up to MT of them
  int bid = blockIdx.x;
  if ((gridDim.x & 7) == 0) bid = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);
  const int tile = SPLIT ? bid : bid * 4 + wave;
  if (tile >= a.B * a.MT) return;
  const int mt = tile % a.MT, b = tile / a.MT;
  // At is zero-padded on the host to MT*32 columns and enough rows, the k padding of X reads a zero row: no select
  // touches a loaded value
  const float* At = a.At + mt * 32 + j;
  const float* X = a.X + (long)b * a.K * 64 + 2 * j;
  const long zdelta = (a.zero + 2 * j) - X;
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
  const int nks = SPLIT ? a.ksq : 4 * a.ksq;            // k-steps this wave walks (a multiple of DENSE_CH)
  const int s_begin = SPLIT ? wave * a.ksq : 0;
  float av[DENSE_CH];
  float2 bv[DENSE_CH];
  // measured: with 16 resident waves per CU the other waves cover a chunk's load latency; keeping a second chunk in
  // flight per wave made this kernel slower
  for (int s0 = s_begin; s0 < s_begin + nks; s0 += DENSE_CH) {
#pragma unroll
    for (int u = 0; u < DENSE_CH; ++u) {
      const int k = 2 * (s0 + u) + h;
      av[u] = At[(long)k * a.ldA];
      const long o = k < a.K ? (long)k * 64 : zdelta;
      bv[u] = *reinterpret_cast<const float2*>(X + o);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < DENSE_CH; ++u) {
      acc0 = mfma32(av[u], bv[u].x, acc0);
      acc1 = mfma32(av[u], bv[u].y, acc1);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float* out = a.out + (long)b * a.M * 64 + 2 * j;
  if (SPLIT) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { red[wave][r][lane] = acc0[r]; red[wave][16 + r][lane] = acc1[r]; }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = wave * 4 + rr;
      const float v0 = ((red[0][r][lane] + red[SPLIT ? 1 : 0][r][lane]) + red[SPLIT ? 2 : 0][r][lane]) + red[SPLIT ? 3 : 0][r][lane];
      const float v1 = ((red[0][16 + r][lane] + red[SPLIT ? 1 : 0][16 + r][lane]) + red[SPLIT ? 2 : 0][16 + r][lane]) +
                       red[SPLIT ? 3 : 0][16 + r][lane];
      const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row < a.M) *reinterpret_cast<float2*>(out + (long)row * 64) = make_float2(v0, v1);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row < a.M) *reinterpret_cast<float2*>(out + (long)row * 64) = make_float2(acc0[r], acc1[r]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// dense edges, one workgroup per sample: the source rows of the sample are staged in LDS once and shared by all its
// row tiles (stand-alone tiles re-fetched them through L2 up to MT times: ~40 % L2 misses, 95 us per launch).
// B operand (source row, channels 2j, 2j+1) = one conflict-free ds_read_b64 per k-step; A operand (weights, shared by
// all samples) streams from L2 one chunk ahead.
// ------------------------------------------------------------------------------------------
struct DenseLArgs {
  const float* At;    // zero-padded (rows >= K_pad + 16, ldA columns): At[k][i] = A[i][k]
  const float* X;     // (B, K, 64)
  float* out;         // (B, M, 64)
  int B, K, M, ldA, MT, Kpad;
};

#define DL_CH 8
#define DENSE_FWD_LDS_FLOATS (2 * 2 * 32 * 64 + 4 * 32 * 64)     // xs + red = 64 KB
#define DENSE_BWD_ROWS (128 + 16)
// forward edge (long K) of sample b: 8 waves = 4 row tiles x 2 K-halves; X streams through LDS in double-buffered chunks
// of 2 x 32 rows (one slab per K-half), the two halves are summed through LDS in a fixed order.  Requires MT <= 4 and 512
// threads.  `scratch`: DENSE_FWD_LDS_FLOATS floats of LDS; store(row, channel pair index j, value pair).
// klist (LDS) != nullptr: only the K_eff source rows klist[0..K_eff) are walked (the caller dropped the all-zero rows of dead
// nodes); klist must be padded with a.Kpad (a zero row of At) up to round_up(K_eff, 64) + 32 entries.
template <class Store>
__device__ __forceinline__ void dense_fwd_sample(const DenseLArgs& a, int b, float* scratch, Store store, const int* klist = nullptr,
                                                 int K_eff = 0) {
  float (*xs)[2][32][64] = reinterpret_cast<float (*)[2][32][64]>(scratch);                 // [buffer][K-half][row][channel]  32 KB
  float (*red)[32][64] = reinterpret_cast<float (*)[32][64]>(scratch + 2 * 2 * 32 * 64);    // 32 KB
  const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int mt = wave & 3, kh = wave >> 2;
  const int Kw = klist ? K_eff : a.K;            // source rows walked
  const int khalf = (klist ? (K_eff + 63) / 64 * 64 : a.Kpad) / 2;      // rows per K-half, a multiple of 32
  const int nchunks = khalf / 32;
  const float* Xb = a.X + (long)b * a.K * 64;
  // cooperative stage of chunk c: 2 slabs x 32 rows x 256 B = 16 KB, 512 threads x 2 x 16 B.  The global loads are issued
  // BEFORE the MFMAs of the running chunk and written to LDS after them, so their latency is not exposed once per chunk
  // (as one load-then-store step this kernel ran at half its MFMA rate).
  f32x4 sv[2];
  auto gload = [&](int c) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int e = threadIdx.x + 512 * r;        // 16-B piece index, 0..1023
      const int slab = e >> 9, row = (e >> 4) & 31, piece = e & 15;
      const int k = slab * khalf + c * 32 + row;
      const int krow = k < Kw ? (klist ? klist[k] : k) : 0;
      const f32x4* src = reinterpret_cast<const f32x4*>(Xb + (long)krow * 64 + piece * 4);
      const f32x4 v = *src;
      sv[r] = k < Kw ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int e = threadIdx.x + 512 * r;
      const int slab = e >> 9, row = (e >> 4) & 31, piece = e & 15;
      *reinterpret_cast<f32x4*>(&xs[buf][slab][row][piece * 4]) = sv[r];
    }
  };
  const float* At = a.At + (klist ? 0 : (long)(kh * khalf) * a.ldA) + (mt < a.MT ? mt : 0) * 32 + j;   // waves beyond MT idle on tile 0
  const int* kl = klist ? klist + kh * khalf : nullptr;
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
  float av[16], nav[16];
  auto loadA = [&](float (&A)[16], int c) {
#pragma unroll
    for (int u = 0; u < 16; ++u) A[u] = At[(long)(kl ? kl[c * 32 + 2 * u + h] : c * 32 + 2 * u + h) * a.ldA];
  };
  auto mma = [&](const float (&A)[16], int buf) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const float2 bv = *reinterpret_cast<const float2*>(&xs[buf][kh][2 * u + h][2 * j]);
      acc0 = mfma32(A[u], bv.x, acc0);
      acc1 = mfma32(A[u], bv.y, acc1);
    }
  };
  gload(0);
  lstore(0);
  loadA(av, 0);
  __syncthreads();
  for (int c = 0; c < nchunks; c += 2) {
    const bool more1 = c + 1 < nchunks;
    if (more1) gload(c + 1);
    loadA(nav, c + 1);                      // At carries 32 extra zero rows: reading one chunk past the end is harmless
    __builtin_amdgcn_sched_barrier(0);
    mma(av, 0);
    __builtin_amdgcn_sched_barrier(0);
    if (more1) lstore(1);
    __syncthreads();
    if (!more1) break;
    const bool more2 = c + 2 < nchunks;
    if (more2) gload(c + 2);
    loadA(av, c + 2);
    __builtin_amdgcn_sched_barrier(0);
    mma(nav, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (more2) lstore(0);
    __syncthreads();
  }
  if (kh == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { red[mt][r][lane] = acc0[r]; red[mt][16 + r][lane] = acc1[r]; }
  }
  __syncthreads();
  if (kh == 0 && mt < a.MT) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row < a.M) store(row, j, make_float2(acc0[r] + red[mt][r][lane], acc1[r] + red[mt][16 + r][lane]));
    }
  }
}

__global__ __launch_bounds__(512, 2) void k_dense_fwd_lds(DenseLArgs a) {
  __shared__ __attribute__((aligned(16))) float scratch[DENSE_FWD_LDS_FLOATS];
  const int b = blockIdx.x;
  float* out = a.out + (long)b * a.M * 64;
  dense_fwd_sample(a, b, scratch, [&](int row, int j, float2 v) { *reinterpret_cast<float2*>(out + (long)row * 64 + 2 * j) = v; });
}

// transposed edge (short K <= 128) of one sample whose source rows sit in LDS (`xs`: DENSE_BWD_ROWS x 64, rows >= K zero);
// 8 waves walk the MT row tiles.
// Optional compaction (lists in LDS): `rlist` / n_rows -- only these output rows are computed (the live nodes of the layer
// below; the others are never read); `klist` / K_eff -- only these source rows are walked (the live nodes of this layer; the
// rows of dead ones are zero), padded with a.Kpad (a zero row of At and of xs) up to round_up(K_eff, 16) + 32 entries.
template <class Store>
__device__ __forceinline__ void dense_bwd_sample(const DenseLArgs& a, const float* xs_raw, Store store, const int* rlist = nullptr,
                                                 int n_rows = 0, const int* klist = nullptr, int K_eff = 0) {
  const float (*xs)[64] = reinterpret_cast<const float (*)[64]>(xs_raw);
  const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nch = (klist ? (K_eff + 15) / 16 * 16 : a.Kpad) / 16;      // chunks of 8 k-steps
  const int M = rlist ? n_rows : a.M, MT = rlist ? (n_rows + 31) / 32 : a.MT;
  for (int mt = wave; mt < MT; mt += 8) {
    const int arow = rlist ? rlist[mt * 32 + j < M ? mt * 32 + j : 0] : mt * 32 + j;
    const float* At = a.At + arow;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    // three A buffers in rotation: a chunk of 8 k-steps is only 16 MFMAs (~0.4 us), less than an L2 round trip, so the
    // weights are fetched TWO chunks ahead.  At carries 48 extra zero rows for the loads issued past the end.
    float a0[DL_CH], a1[DL_CH], a2[DL_CH];
    auto loadA = [&](float (&A)[DL_CH], int c) {
#pragma unroll
      for (int u = 0; u < DL_CH; ++u) A[u] = At[(long)(klist ? klist[c * 16 + 2 * u + h] : c * 16 + 2 * u + h) * a.ldA];
    };
    auto mma = [&](const float (&A)[DL_CH], int c) {
#pragma unroll
      for (int u = 0; u < DL_CH; ++u) {
        const float2 bv = *reinterpret_cast<const float2*>(&xs[klist ? klist[c * 16 + 2 * u + h] : c * 16 + 2 * u + h][2 * j]);
        acc0 = mfma32(A[u], bv.x, acc0);
        acc1 = mfma32(A[u], bv.y, acc1);
      }
    };
    loadA(a0, 0);
    loadA(a1, 1);
    for (int c = 0; c < nch; c += 3) {
      loadA(a2, c + 2);
      __builtin_amdgcn_sched_barrier(0);
      mma(a0, c);
      __builtin_amdgcn_sched_barrier(0);
      if (c + 1 >= nch) break;
      loadA(a0, c + 3);
      __builtin_amdgcn_sched_barrier(0);
      mma(a1, c + 1);
      __builtin_amdgcn_sched_barrier(0);
      if (c + 2 >= nch) break;
      loadA(a1, c + 4);
      __builtin_amdgcn_sched_barrier(0);
      mma(a2, c + 2);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row < M) store(rlist ? rlist[row] : row, j, make_float2(acc0[r], acc1[r]));
    }
  }
}

__global__ __launch_bounds__(512, 2) void k_dense_bwd_lds(DenseLArgs a) {
  __shared__ __attribute__((aligned(16))) float xs[DENSE_BWD_ROWS][64];       // 36 KB, rows >= K are zero
  const int b = blockIdx.x;
  const float* Xb = a.X + (long)b * a.K * 64;
  for (int e = threadIdx.x; e < DENSE_BWD_ROWS * 16; e += 512) {
    const int row = e >> 4, piece = e & 15;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < a.K) v = *reinterpret_cast<const f32x4*>(Xb + (long)row * 64 + piece * 4);
    *reinterpret_cast<f32x4*>(&xs[row][piece * 4]) = v;
  }
  __syncthreads();
  float* out = a.out + (long)b * a.M * 64;
  dense_bwd_sample(a, &xs[0][0], [&](int row, int j, float2 v) { *reinterpret_cast<float2*>(out + (long)row * 64 + 2 * j) = v; });
}

struct PropArgs {
  const float* pack; const float* mu_last; const float* prop_w; const float* prop_b;
  const float *lb, *ub, *z_out; float* mu_prop; float* nb_back; int B, N_last;
  const float *lbl, *ubl;   // bounds of the top ReLU layer (its rows have fc4_2 deferred: the bias term needs live_n)
};

// property node (graph_conv.py:194-210): nb = W_prop[b] . mu_L[b];
// mu_K = out3(relu(out2([relu(out1([l, u, z_out, c])), nb]))), then the backward edge from it (:324-326):
// nb_back[b, n, :] = W_prop[b][n] * mu_K[b, :].  One workgroup (4 waves) per sample, lane = channel: the waves split the
// rows of mu_L (partial sums combined in a fixed order through LDS), wave 0 runs the three small layers with the
// transposed weights in LDS (196 dependent FMA steps read LDS, not L2), all waves write the backward aggregate.
__global__ __launch_bounds__(256) void k_prop(PropArgs a) {
  __shared__ float wl[PackProp::FLOATS];
  __shared__ float xs[128];
  __shared__ float part[4][64];
  __shared__ float outv[64];
  __shared__ float spart[4];
  for (int i = threadIdx.x; i < PackProp::FLOATS; i += 256) wl[i] = a.pack[i];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.x;
  const float* mu = a.mu_last + (long)b * a.N_last * 64 + lane;
  const float* pw = a.prop_w + (long)b * a.N_last;
  float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int n = w; n < a.N_last; n += 16) {             // 4 independent loads in flight per wave
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int nn = n + 4 * u;
      if (nn < a.N_last) acc[u] = fmaf(pw[nn], mu[(long)nn * 64], acc[u]);
    }
  }
  part[w][lane] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  // sp = sum_n W_prop[n] live_n: the rows of mu_L hold E with mu = (fc4_2.E + b).live, so
  // out2[:, 64:].nb = (out2[:, 64:].fc4_2.W).(sum_n W_prop[n] E_n) + sp.(out2[:, 64:].fc4_2.b)   (folded in PackProp)
  float sp = 0.0f;
  for (int n = threadIdx.x; n < a.N_last; n += 256) {
    const long g = (long)b * a.N_last + n;
    sp += node_is_live(a.lbl[g], a.ubl[g]) ? pw[n] : 0.0f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sp += __shfl_xor(sp, o);
  if (lane == 0) spart[w] = sp;
  __syncthreads();
  if (w == 0) {
    const float nb = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    const float spt = (spart[0] + spart[1]) + (spart[2] + spart[3]);
    const float f[4] = {a.lb[b], a.ub[b], a.z_out[b], a.prop_b[b]};
    float h1 = wl[PackProp::B1 + lane];
#pragma unroll
    for (int k = 0; k < 4; ++k) h1 = fmaf(wl[PackProp::W1T + k * 64 + lane], f[k], h1);
    xs[lane] = relu_nan(h1);
    xs[64 + lane] = nb;
    __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): this wave's LDS writes are visible to its own reads
    float h2 = fmaf(spt, wl[PackProp::V2 + lane], wl[PackProp::B2 + lane]);
#pragma unroll 8
    for (int k = 0; k < 128; ++k) h2 = fmaf(wl[PackProp::W2T + k * 64 + lane], xs[k], h2);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    xs[lane] = relu_nan(h2);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    float o = wl[PackProp::B3 + lane];
#pragma unroll 8
    for (int k = 0; k < 64; ++k) o = fmaf(wl[PackProp::W3T + k * 64 + lane], xs[k], o);
    a.mu_prop[(long)b * 64 + lane] = o;
    outv[lane] = o;
  }
  __syncthreads();
  if (a.nb_back) {
    const float o = outv[lane];
    float* nbk = a.nb_back + (long)b * a.N_last * 64 + lane;
    for (int n = w; n < a.N_last; n += 4) nbk[(long)n * 64] = pw[n] * o;
  }
}

// ------------------------------------------------------------------------------------------
// k_top: the top of the network in one launch per round, one workgroup (8 waves) per sample.  When the last ReLU layer
// hangs on a Linear edge and has <= 128 nodes, everything between "layer L-1 forward-updated" and "layer L-1 can be
// backward-updated" is local to a sample and tiny:
//   F1  nb_L   = W_L . mu_{L-1}                        (dense forward edge, graph_conv.py:130-137)
//   F2  mu_L   <- forward node update                  (:139-186)
//   F3  mu_K   <- property node, nb_back = W_prop^T mu_K  (:194-210, :324-326)
//   B1  mu_L   <- backward node update                 (:253-350)
//   B2  nb_{L-1} = W_L^T . mu_L                        (dense transposed edge, :320-322)
// As five launches these cost ~160 us of mostly launch ramps, weight staging and latency; here the layer's rows never
// leave LDS.  LDS map (floats): A = weight pack of the running phase (first the dense-forward staging buffers),
// Bp = PackProp, C = the rows of layer L (DENSE_BWD_ROWS x 64, rows >= N zero), sm = small vectors.
// ------------------------------------------------------------------------------------------
struct TopArgs {
  DenseLArgs df;            // forward edge L (out unused)
  DenseLArgs db;            // transposed edge L (X unused: the rows come from LDS), out = aggregate rows of layer L-1
  const float *pack_f, *pack_b, *pack_p;
  const float *Pf, *Pb;     // cached P' rows of layer L (by node id), forward / backward
  const float* sf;          // bias-sum scalars of the forward edge (B, N)
  const float *lb, *ub;     // bounds of layer L, flat (B*N)
  const float *lbm, *ubm;   // bounds of layer L-1, flat (B*K): its dead rows (all zero) are skipped by the forward edge
  const float *prop_w, *prop_b, *lbK, *ubK, *z_out;
  float* mu_prop;           // (B, 64)
  float* mu;                // (B, N, 64) rows of layer L (backward-produced)
  int* status;
  int N;
};
#define TOP_A_FLOATS (PackUpd::FLOATS > DENSE_FWD_LDS_FLOATS ? PackUpd::FLOATS : DENSE_FWD_LDS_FLOATS)
#define TOP_FIXED_FLOATS (TOP_A_FLOATS + PackProp::FLOATS + DENSE_BWD_ROWS * 64 + 8 * 64 + 128 + 64 + 64)
#define TOP_LDS_FLOATS 40960                   // all 160 KB: what the fixed regions leave holds the live-row lists
#define TOP_K2_INTS (128 + 48)                  // live rows of layer L, padded for the chunks read ahead
#define TOP_LIST_INTS (TOP_LDS_FLOATS - TOP_FIXED_FLOATS - TOP_K2_INTS)

__device__ __forceinline__ void copy_to_lds_part(float* lds, const float* src, int nfloats, int tid, int nthr) {
  const f32x4* g = reinterpret_cast<const f32x4*>(src);
  f32x4* l = reinterpret_cast<f32x4*>(lds);
  const int n4 = nfloats / 4;
  for (int i0 = tid; i0 < n4; i0 += 8 * nthr) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * nthr;
      v[u] = g[i < n4 ? i : i0];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * nthr;
      if (i < n4) l[i] = v[u];
    }
  }
}

// needs 512 threads (threads beyond that idle) and TOP_LDS_FLOATS of LDS
__device__ __forceinline__ void top_sample(const TopArgs& a, const int b, float* lds) {
  float* A = lds;
  float* Bp = A + TOP_A_FLOATS;
  float* Cr = Bp + PackProp::FLOATS;
  float* part = Cr + DENSE_BWD_ROWS * 64;     // [8][64]
  float* xs = part + 8 * 64;                  // [128]
  float* outv = xs + 128;                     // [64]
  float* spart = outv + 64;                   // [8]
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = a.N;
  for (int i = tid; i < DENSE_BWD_ROWS * 16; i += 512) reinterpret_cast<f32x4*>(Cr)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // live source rows of layer L-1 (the dead ones are all zero: ~45 % of the forward edge's k-steps), compacted in node order
  // into the region PackProp takes afterwards
  // The list goes behind the fixed regions when it fits there (then the transposed edge B2 also uses it, to compute only the
  // live rows of layer L-1), else into the region PackProp takes after F1.
  int* k2list = reinterpret_cast<int*>(spart + 8);        // live rows of layer L (B2)
  int* tail = k2list + TOP_K2_INTS;
  const bool keep = a.df.K + 96 <= TOP_LIST_INTS;
  int* klist = keep ? tail : reinterpret_cast<int*>(Bp);
  int K_eff = 0;
  const bool compact = keep || a.df.K + 96 <= PackProp::FLOATS;
  if (compact) {
    int* wc = reinterpret_cast<int*>(part);              // per-wave counts
    const int K = a.df.K;
    for (int n0 = 0; n0 < K; n0 += 512) {
      const int n = n0 + tid;
      const long gm = (long)b * K + (n < K ? n : 0);
      const bool live = n < K && node_is_live(a.lbm[gm], a.ubm[gm]);
      const unsigned long long bal = __ballot(live);
      if (lane == 0) wc[wave] = __popcll(bal);
      __syncthreads();
      int before = 0, total = 0;
#pragma unroll
      for (int w8 = 0; w8 < 8; ++w8) { before += w8 < wave ? wc[w8] : 0; total += wc[w8]; }
      if (live) klist[K_eff + before + __popcll(bal & ((1ull << lane) - 1ull))] = n;
      K_eff += total;
      __syncthreads();
    }
    for (int i = K_eff + tid; i < (K_eff + 63) / 64 * 64 + 32; i += 512) klist[i] = a.df.Kpad;     // a zero row of At
  }
  __syncthreads();

  // ---- F1: rows of C <- W_L . mu_{L-1}
  dense_fwd_sample(a.df, b, A, [&](int row, int jj, float2 v) { *reinterpret_cast<float2*>(Cr + row * 64 + 2 * jj) = v; },
                   compact ? klist : nullptr, K_eff);
  __syncthreads();
  copy_to_lds(Bp, a.pack_p, PackProp::FLOATS);            // (read from F3 on, behind two more barriers)

  // per-lane node of the update phases (waves 0..3: one tile of 32 nodes each)
  const int n = wave * 32 + j;
  const bool upd_wave = wave * 32 < N;
  const bool valid = n < N;
  const long g = (long)b * N + (valid ? n : 0);
  const float* pw = a.prop_w + (long)b * N;
  Ratio r{};
  if (upd_wave) r = compute_ratio(a.lb[g], a.ub[g]);
  auto load_row = [&](Frag& x_, int row) {       // fragment <- LDS row (row-major 64 floats)
    const f32x4* p = reinterpret_cast<const f32x4*>(Cr + row * 64 + 4 * h);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 v = p[2 * q];
#pragma unroll
      for (int c = 0; c < 4; ++c) FRAG_AT(x_, 4 * q + c) = v[c];
    }
  };
  auto store_row = [&](const Frag& x_, float* base) {
    f32x4* p = reinterpret_cast<f32x4*>(base + 4 * h);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      f32x4 v;
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = FRAG_AT(x_, 4 * q + c);
      p[2 * q] = v;
    }
  };
  // general folded node chain on fragment X (k_node_update, kind 1); `sx`: small k-step input of a deferred projection
  auto chain = [&](const Frag& X, const float* Prow, bool deferred, float sx, Frag& H2) {
    Frag H;
    frag_bias(H, A + PackUpd::BA, h);
    if (deferred) {
      const float x1[1] = {sx};
      gemm_small<1>(A + PackUpd::VAW, lane, H, x1);
    }
    const float r0 = r.r0, r1 = r.r1;
    gemm_w64<64>(A + PackUpd::WA, lane, H, [&](int s) { return FRAG_AT(X, s & 31) * (s < 32 ? r0 : r1); });
    frag_relu(H);
    frag_load_rowptr(H2, Prow, h);
    gemm_w64<32>(A + PackUpd::WCB, lane, H2, [&](int s) { return FRAG_AT(H, s); });
    frag_relu(H2);
    frag_scale(H2, r.live);
  };

  // ---- F2: forward node update of layer L (rows stay in C)
  stage_pack(A, a.pack_f, PackUpd::FLOATS);
  if (upd_wave) {
    Frag X, E;
    load_row(X, valid ? n : 0);
    chain(X, r.amb != 0.0f ? a.Pf + g * 64 : a.pack_f + PackUpd::BCBROW, true, (h ? r.r1 : r.r0) * a.sf[g], E);
    if (valid) {
      if (frag_has_nan(E)) atomicOr(a.status, 1);
      store_row(E, Cr + n * 64);
    }
  }
  __syncthreads();

  // ---- F3: property node (k_prop) on the rows in C; meanwhile waves 1..7 stage the backward pack
  {
    float acc = 0.0f;
    for (int m = wave; m < N; m += 8) acc = fmaf(pw[m], Cr[m * 64 + lane], acc);
    part[wave * 64 + lane] = acc;
    float sp = 0.0f;
    for (int m = tid; m < N; m += 512) {
      const long gm = (long)b * N + m;
      sp += node_is_live(a.lb[gm], a.ub[gm]) ? pw[m] : 0.0f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sp += __shfl_xor(sp, o);
    if (lane == 0) spart[wave] = sp;
  }
  __syncthreads();
  if (wave == 0) {
    float nbv = 0.0f, spt = 0.0f;
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) { nbv += part[w8 * 64 + lane]; spt += spart[w8]; }
    const float f[4] = {a.lbK[b], a.ubK[b], a.z_out[b], a.prop_b[b]};
    float h1 = Bp[PackProp::B1 + lane];
#pragma unroll
    for (int k = 0; k < 4; ++k) h1 = fmaf(Bp[PackProp::W1T + k * 64 + lane], f[k], h1);
    xs[lane] = relu_nan(h1);
    xs[64 + lane] = nbv;
    __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): this wave's LDS writes are visible to its own reads
    float h2 = fmaf(spt, Bp[PackProp::V2 + lane], Bp[PackProp::B2 + lane]);
#pragma unroll 8
    for (int k = 0; k < 128; ++k) h2 = fmaf(Bp[PackProp::W2T + k * 64 + lane], xs[k], h2);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    xs[lane] = relu_nan(h2);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    float o = Bp[PackProp::B3 + lane];
#pragma unroll 8
    for (int k = 0; k < 64; ++k) o = fmaf(Bp[PackProp::W3T + k * 64 + lane], xs[k], o);
    a.mu_prop[(long)b * 64 + lane] = o;
    outv[lane] = o;
  } else {
    copy_to_lds_part(A, a.pack_b, PackUpd::FLOATS, tid - 64, 448);
  }
  __syncthreads();

  // ---- B1: backward node update of layer L; its aggregate is the rank-1 edge from the property node
  if (upd_wave) {
    Frag X, E;
    const float wn = valid ? pw[n] : 0.0f;
    {
      const f32x4* o4 = reinterpret_cast<const f32x4*>(outv + 4 * h);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const f32x4 v = o4[2 * q];
#pragma unroll
        for (int c = 0; c < 4; ++c) FRAG_AT(X, 4 * q + c) = wn * v[c];
      }
    }
    chain(X, r.amb != 0.0f ? a.Pb + g * 64 : a.pack_b + PackUpd::BCBROW, false, 0.0f, E);
    if (valid) {
      if (frag_has_nan(E)) atomicOr(a.status, 1);
      store_row(E, Cr + n * 64);
      store_row(E, a.mu + g * 64);
    }
  }
  __syncthreads();

  // ---- B2: aggregate rows of layer L-1 <- W_L^T . rows of C: only the live rows of layer L-1 (nothing reads the others), only
  // the live (non-zero) rows of layer L
  int K2 = 0;
  if (keep) {
    int* wc = reinterpret_cast<int*>(part);
    const unsigned long long bal = __ballot(upd_wave && valid && r.live != 0.0f && (lane < 32));
    if (lane == 0) wc[wave] = __popcll(bal);
    __syncthreads();
    int before = 0;
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) { before += w8 < wave ? wc[w8] : 0; K2 += wc[w8]; }
    if (upd_wave && valid && r.live != 0.0f && lane < 32) k2list[before + __popcll(bal & ((1ull << lane) - 1ull))] = n;
    for (int i = K2 + tid; i < TOP_K2_INTS; i += 512) k2list[i] = a.db.Kpad;   // zero row of At and of C
    __syncthreads();
  }
  float* out = a.db.out + (long)b * a.db.M * 64;
  auto put = [&](int row, int jj, float2 v) { *reinterpret_cast<float2*>(out + (long)row * 64 + 2 * jj) = v; };
  if (keep) dense_bwd_sample(a.db, Cr, put, klist, K_eff, k2list, K2);
  else dense_bwd_sample(a.db, Cr, put);
}

__global__ __launch_bounds__(512, 1) void k_top(TopArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  top_sample(a, blockIdx.x, lds);
}

// ------------------------------------------------------------------------------------------
// k_livesum: s[b, n'] = sum_n A[n', n] live[b, n] for every edge and direction (gnnb_pack.h "deferred projection"):
// the bias of a producer's deferred last layer reaches a consumer multiplied by this scalar.  Static over the rounds
// (live depends on the bounds only), so it runs once per forward.  Scalar stencil work (HBM/L2-bound, one thread per
// destination node), all edges in one launch.  Transposed conv edges are divided by the tap count exactly like their
// aggregate (graph_conv.py:306-312) unless the destination is the input layer (:361-372).
// ------------------------------------------------------------------------------------------
struct LiveSumJob {
  int kind;                 // 0 conv forward, 1 dense forward, 2 conv transposed, 3 dense transposed
  const float* w;           // conv fwd [ci][ky][kx][co]; conv bwd [co][ky][kx][ci]; dense (both directions) W[o][ld]
  const float* lf;          // live flags of the SOURCE layer (B, Nsrc), written by k_classify; null: all live (input layer)
  float* out;               // (B, Ndst)
  int Ndst, Nsrc, ld, normalise;
  int c_in, h_in, w_in, c_out, h_out, w_out, kh, kw, stride, pad;   // geometry of the conv edge (forward orientation)
  int wlds;                 // conv: number of weights to stage in LDS (0: read them from global memory)
};
struct LiveSumArgs { int njobs, B, lv_floats; LiveSumJob job[2 * MAXL]; };
#define LIVESUM_MAXW 16384    // conv weights staged in LDS (64 KB)

#define LIVESUM_MAXSRC 40000  // source nodes per sample that fit the 160 KB LDS (bind rejects larger layers)
#define LS_CC 4
#define LS_DO 9
// conv / transposed-conv stencil of one sample out of LDS.  KH, KW, S > 0: compile-time kernel size and stride, so the tap
// loops unroll completely and all LDS reads of a source channel are issued before their FMAs (masked, no branches);
// KH = 0: run-time geometry (any other conv).
// NT: threads taking part; lv == nullptr: every source node is live (the input layer)
template <int KH, int KW, int S, bool FWD, int NT>
__device__ __forceinline__ void livesum_conv(const LiveSumJob& jb, const float* lv, const float* W, float* out, int tid) {
  const int kh = KH ? KH : jb.kh, kw = KH ? KW : jb.kw, st = KH ? S : jb.stride;
  const int Hd = FWD ? jb.h_out : jb.h_in, Wd = FWD ? jb.w_out : jb.w_in, Cd = FWD ? jb.c_out : jb.c_in;   // destination side
  const int Hs = FWD ? jb.h_in : jb.h_out, Ws = FWD ? jb.w_in : jb.w_out, Cs = FWD ? jb.c_in : jb.c_out;   // source side
  const int npos = Hd * Wd;
  const int P = npos < NT ? npos : NT;                 // positions handled per pass
  const int ngrp = NT / P;                             // thread groups that split the channel chunks
  const int grp = tid / P;
  if (grp >= ngrp) return;
  const int nchunk = (Cd + LS_CC - 1) / LS_CC;
  // taps walked per axis.  forward: every ky, source row sy = y*s - p + ky; transposed: ky = ky0 + s*t, sy = sy0 - t
  const int TY = FWD ? kh : (kh + st - 1) / st, TX = FWD ? kw : (kw + st - 1) / st;
  constexpr int TYC = KH ? (FWD ? KH : (KH + S - 1) / S) : 1, TXC = KH ? (FWD ? KW : (KW + S - 1) / S) : 1;
  for (int pos = tid - grp * P; pos < npos; pos += NT) {      // one pass unless the layer has more than NT positions
    const int y = pos / Wd, x = pos - y * Wd;
    int ky0 = 0, kx0 = 0, sy0, sx0;
    if (FWD) {
      sy0 = y * st - jb.pad; sx0 = x * st - jb.pad;
    } else {
      ky0 = (y + jb.pad) % st; kx0 = (x + jb.pad) % st;
      sy0 = (y + jb.pad - ky0) / st; sx0 = (x + jb.pad - kx0) / st;
    }
    const int kstep = FWD ? 1 : st, sstep = FWD ? 1 : -1;
    int cnt_y = 0, cnt_x = 0;
    for (int t = 0; t < TY; ++t) cnt_y += ((unsigned)(sy0 + sstep * t) < (unsigned)Hs && ky0 + kstep * t < kh) ? 1 : 0;
    for (int t = 0; t < TX; ++t) cnt_x += ((unsigned)(sx0 + sstep * t) < (unsigned)Ws && kx0 + kstep * t < kw) ? 1 : 0;
    for (int ch = grp; ch < nchunk; ch += ngrp) {
      const int c0 = ch * LS_CC;
      float acc[LS_CC];
#pragma unroll
      for (int u = 0; u < LS_CC; ++u) acc[u] = 0.0f;
      for (int cs = 0; cs < Cs; ++cs) {
        if (KH) {
#pragma unroll
          for (int t = 0; t < TYC; ++t)
#pragma unroll
            for (int u2 = 0; u2 < TXC; ++u2) {
              const int sy = sy0 + sstep * t, ky = ky0 + kstep * t, sx = sx0 + sstep * u2, kx = kx0 + kstep * u2;
              const bool v = (unsigned)sy < (unsigned)Hs && (unsigned)sx < (unsigned)Ws && ky < kh && kx < kw;
              const float l = v ? (lv ? lv[(cs * Hs + (v ? sy : 0)) * Ws + (v ? sx : 0)] : 1.0f) : 0.0f;
              const float* wr = W + ((cs * kh + (v ? ky : 0)) * kw + (v ? kx : 0)) * Cd + c0;
#pragma unroll
              for (int u = 0; u < LS_CC; ++u) acc[u] = fmaf(c0 + u < Cd ? wr[u] : 0.0f, l, acc[u]);
            }
        } else {
          for (int t = 0; t < TY; ++t) {
            const int sy = sy0 + sstep * t, ky = ky0 + kstep * t;
            if ((unsigned)sy >= (unsigned)Hs || ky >= kh) continue;
            for (int u2 = 0; u2 < TX; ++u2) {
              const int sx = sx0 + sstep * u2, kx = kx0 + kstep * u2;
              if ((unsigned)sx >= (unsigned)Ws || kx >= kw) continue;
              const float l = lv ? lv[(cs * Hs + sy) * Ws + sx] : 1.0f;
              const float* wr = W + ((cs * kh + ky) * kw + kx) * Cd + c0;
#pragma unroll
              for (int u = 0; u < LS_CC; ++u) acc[u] = fmaf(c0 + u < Cd ? wr[u] : 0.0f, l, acc[u]);
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < LS_CC; ++u)
        if (c0 + u < Cd) {
          float v = acc[u];
          if (!FWD && jb.normalise) v = v / (float)(cnt_y * cnt_x);
          out[(c0 + u) * npos + pos] = v;
        }
    }
  }
}

// grid (sample, job): the live flags of the sample's source layer and the conv weights are staged in LDS once.  The stencil
// is ALU-bound (an FMA per tap), so the loops are built to spend few instructions per FMA: a thread owns one pixel position
// and 8 channels at a time (one LDS read of the flag + two 16-B reads of 8 consecutive weights feed 8 FMAs), positions and
// tap ranges are decoded once per thread, transposed edges walk only the taps of the lane's stride phase.
// one job (edge, direction) of one sample: lv = live flags of the source layer in LDS (nullptr: all live), W = the job's
// weights (conv: in LDS when staged), out = the sample's (Ndst) output row
template <int NT>
__device__ __forceinline__ void livesum_job(const LiveSumJob& jb, const float* lv, const float* W, float* out, int tid) {
  if (jb.kind == 0 || jb.kind == 2) {
    const bool fwd = jb.kind == 0;
    const int key = jb.kh * 100 + jb.kw * 10 + jb.stride;
    if (fwd) {
      if (key == 442) livesum_conv<4, 4, 2, true, NT>(jb, lv, W, out, tid);
      else if (key == 331) livesum_conv<3, 3, 1, true, NT>(jb, lv, W, out, tid);
      else livesum_conv<0, 0, 0, true, NT>(jb, lv, W, out, tid);
    } else {
      if (key == 442) livesum_conv<4, 4, 2, false, NT>(jb, lv, W, out, tid);
      else if (key == 331) livesum_conv<3, 3, 1, false, NT>(jb, lv, W, out, tid);
      else livesum_conv<0, 0, 0, false, NT>(jb, lv, W, out, tid);
    }
  } else if (jb.kind == 1) {
    // few outputs, long K: a wave per LS_DO outputs at a time (that many x 4 independent loads in flight), lanes stride over
    // the sources, shuffle reduction
    const int lane = tid & 63, wv = tid >> 6;
    for (int o0 = wv * LS_DO; o0 < jb.Ndst; o0 += (NT / 64) * LS_DO) {
      float acc[LS_DO];
      const float* wrow[LS_DO];
#pragma unroll
      for (int u = 0; u < LS_DO; ++u) { acc[u] = 0.0f; wrow[u] = jb.w + (long)(o0 + u < jb.Ndst ? o0 + u : o0) * jb.ld; }
#pragma unroll 4
      for (int i = lane; i < jb.Nsrc; i += 64) {
        const float l = lv ? lv[i] : 1.0f;
#pragma unroll
        for (int u = 0; u < LS_DO; ++u) acc[u] = fmaf(wrow[u][i], l, acc[u]);
      }
#pragma unroll
      for (int u = 0; u < LS_DO; ++u) {
        float v = acc[u];
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
        if (lane == 0 && o0 + u < jb.Ndst) out[o0 + u] = v;
      }
    }
  } else {
    // dense transposed: thread per input node, coalesced weight rows, broadcast live flags
    for (int n0 = tid; n0 < jb.Ndst; n0 += 4 * NT) {         // 4 nodes per thread at a time: 40 independent loads in flight
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      int nn[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) nn[u] = n0 + NT * u < jb.Ndst ? n0 + NT * u : n0;
#pragma unroll 10
      for (int o = 0; o < jb.Nsrc; ++o) {
        const float l = lv ? lv[o] : 1.0f;
        const float* wr = jb.w + (long)o * jb.ld;
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = fmaf(wr[nn[u]], l, acc[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (n0 + NT * u < jb.Ndst) out[n0 + NT * u] = acc[u];
    }
  }
}

__global__ __launch_bounds__(256) void k_livesum(LiveSumArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lv[];     // [lv_floats] flags, then the conv weights
  const LiveSumJob& jb = a.job[blockIdx.y];
  const long b = blockIdx.x;
  const int tid = threadIdx.x;
  if (jb.lf && (jb.Nsrc & 3) == 0) {
    copy_to_lds(lv, jb.lf + b * jb.Nsrc, jb.Nsrc);       // (B, Nsrc) rows stay 16-B aligned when Nsrc % 4 == 0
  } else if (jb.lf) {
    const float* lf = jb.lf + b * jb.Nsrc;
    for (int i = tid; i < jb.Nsrc; i += 256) lv[i] = lf[i];
  }
  float* wl = lv + a.lv_floats;
  if ((jb.wlds & 3) == 0) copy_to_lds(wl, jb.w, jb.wlds);
  else for (int i = tid; i < jb.wlds; i += 256) wl[i] = jb.w[i];
  const float* W = jb.wlds ? wl : jb.w;
  __syncthreads();
  float* out = jb.out + b * jb.Ndst;
  livesum_job<256>(jb, jb.lf ? lv : nullptr, W, out, tid);
}

// ------------------------------------------------------------------------------------------
// BaBSR ("KW") branching heuristic -- reference plnn/kw_score_conv.py choose_node_conv :41-113.
// A scalar `ratio` per node is swept backwards through the verified network (W^T / transposed conv, times the
// relaxation slope at every ReLU); each ReLU gets |max(b ratio (r0-1), b ratio r0) + min(ratio, 0) intercept| as score.
// One workgroup per subproblem, the ratio vector of the current layer lives in LDS (two buffers).
// ------------------------------------------------------------------------------------------
struct BabsrArgs {
  int L, R;
  const float* lb[MAXL]; const float* ub[MAXL]; const float* bias[MAXL];   // ReLU layer k at index k-1
  int N[MAXL], hw[MAXL], off[MAXL];
  // edge between layer k and k+1 at index k-1 (k = 1..L-1), walked transposed
  int ekind[MAXL];              // 0 conv, 1 linear
  const float* ew[MAXL];        // conv: [co][ky][kx][ci]; linear: W[o][i] with row stride ld
  int c_in[MAXL], h_in[MAXL], w_in[MAXL], c_out[MAXL], h_out[MAXL], w_out[MAXL], kh[MAXL], kw[MAXL], stride[MAXL], pad[MAXL], ld[MAXL];
  const float* prop_w;          // (B, N_L)
  const float* mask;            // (B, R): 1 where the BaB mask is -1
  float* scores;                // out (B, R): `score` of :103
  float* icp;                   // out (B, R): `intercept_tb` of :86
  int maxN;
};

__global__ __launch_bounds__(256) void k_babsr(BabsrArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* cur = lds;
  float* nxt = lds + a.maxN;
  const int b = blockIdx.x, tid = threadIdx.x;
  {
    const int NL = a.N[a.L - 1];
    for (int n = tid; n < NL; n += 256) cur[n] = a.prop_w[(long)b * NL + n];      // Linear(., 1)^T applied to ones(1), :73-77
  }
  for (int k = a.L - 1; k >= 0; --k) {
    __syncthreads();
    const int N = a.N[k];
    for (int n = tid; n < N; n += 256) {
      const long g = (long)b * N + n;
      const float lb = a.lb[k][g], ub = a.ub[k][g];
      const float lower_temp = lb - relu_nan(lb), upper_temp = relu_nan(ub);          // compute_ratio :23-27
      const float slope = upper_temp / (upper_temp - lower_temp);
      const float intercept = -1.0f * lower_temp * slope;
      const float rt = cur[n];
      const float icand = fminf(rt, 0.0f) * intercept;                               // :84-85
      const float m = a.mask[(long)b * a.R + a.off[k] + n];
      const float bb = a.bias[k][n / a.hw[k]];
      const float b1 = bb * (rt * (slope - 1.0f));                                   // :92-93
      const float rt2 = rt * slope;                                                  // :94
      const float b2 = bb * rt2;                                                     // :95
      a.scores[(long)b * a.R + a.off[k] + n] = fabsf(fmaxf(b1, b2) + icand) * m;     // :96-103
      a.icp[(long)b * a.R + a.off[k] + n] = icand * m;                               // :86
      cur[n] = rt2;
    }
    if (k == 0) break;                       // nothing reads the ratio below the first ReLU layer
    __syncthreads();
    const int e = k - 1;                     // edge between ReLU layers k-1+1 and k+1 in 1-based numbering
    const int Nin = a.N[k - 1];
    if (a.ekind[e] == 1) {                   // :74-77  ratio <- W^T ratio
      const int nout = N, ld = a.ld[e];
      const float* W = a.ew[e];
      for (int i = tid; i < Nin; i += 256) {
        float acc = 0.0f;
        for (int o = 0; o < nout; ++o) acc = fmaf(W[(long)o * ld + i], cur[o], acc);
        nxt[i] = acc;
      }
    } else {                                 // :109-111  ratio <- conv_transpose2d(ratio, W)
      const int CI = a.c_in[e], HI = a.h_in[e], WI = a.w_in[e], CO = a.c_out[e], HO = a.h_out[e], WO = a.w_out[e];
      const int KH = a.kh[e], KW = a.kw[e], S = a.stride[e], P = a.pad[e];
      const float* W = a.ew[e];
      for (int i = tid; i < Nin; i += 256) {
        const int x = i % WI, y = (i / WI) % HI, ci = i / (WI * HI);
        float acc = 0.0f;
        for (int ky = 0; ky < KH; ++ky) {
          const int ty = y + P - ky;
          if (ty < 0 || ty % S != 0 || ty / S >= HO) continue;
          const int oy = ty / S;
          for (int kx = 0; kx < KW; ++kx) {
            const int tx = x + P - kx;
            if (tx < 0 || tx % S != 0 || tx / S >= WO) continue;
            const int ox = tx / S;
            for (int co = 0; co < CO; ++co)
              acc = fmaf(W[((co * KH + ky) * KW + kx) * CI + ci], cur[(co * HO + oy) * WO + ox], acc);
          }
        }
        nxt[i] = acc;
      }
    }
    float* t = cur; cur = nxt; nxt = t;
  }
}

struct ArgmaxArgs { const float* scores; int* dec; int B, R, n_relu; int cum[16]; };

// torch.max(scores, 0) -> first maximal index; flat index -> [layer, idx]      graph_score.py:41-47
// first maximum of the sample's score row -> [layer, idx]; NT threads (a power of two), sv / si: NT floats / ints of LDS
template <int NT>
__device__ __forceinline__ void argmax_sample(const ArgmaxArgs& a, int b, float* sv, int* si) {
  float best = -INFINITY;
  int bi = 0x7fffffff;
  const float* s = a.scores + (long)b * a.R;
  for (int i = threadIdx.x; i < a.R; i += NT) {
    const float v = s[i];
    if (v > best) { best = v; bi = i; }     // strided ascending: keeps the first index per thread
  }
  sv[threadIdx.x] = best;
  si[threadIdx.x] = bi;
  __syncthreads();
  for (int st = NT / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
      const float v = sv[threadIdx.x + st];
      const int i = si[threadIdx.x + st];
      if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && i < si[threadIdx.x])) { sv[threadIdx.x] = v; si[threadIdx.x] = i; }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int lay = -1, idx = -1;
    if (si[0] != 0x7fffffff) {
      const int flat = si[0];
      lay = 0;
      while (lay < a.n_relu - 1 && a.cum[lay] <= flat) ++lay;
      idx = lay == 0 ? flat : flat - a.cum[lay - 1];
    }
    a.dec[b * 2] = lay;
    a.dec[b * 2 + 1] = idx;
  }
}

__global__ __launch_bounds__(256) void k_argmax(ArgmaxArgs a) {
  __shared__ float sv[256];
  __shared__ int si[256];
  argmax_sample<256>(a, blockIdx.x, sv, si);
}

#define N_PACKS 14   // == PK_COUNT
enum { PK_EMBED, PK_PRE_FWD, PK_PRE_BWD, PK_PRE_INP, PK_PROP, PK_UPD_FWD_E, PK_UPD_FWD_I, PK_UPD_FWD_F, PK_UPD_BWD, PK_UPD_BWD_B,
       PK_UPD_INP, PK_POST_INP, PK_SCORE_B, PK_SCORE_F, PK_COUNT };
static_assert(PK_COUNT == N_PACKS, "pack table");

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  fprintf(stderr, "[gnnb] error: %s\n", buf);   // the BaB harness swallows exceptions (bab_mip.py:73-76): log first
  return code;
}
#define HIPCHK(x)                                                                         \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) return fail(GNNB_E_HIP, "%s: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

enum ProfClass {
  PC_EMBED, PC_PRE, PC_PRE_INP, PC_CONV_FWD, PC_CONVT_BWD, PC_DENSE_AGG, PC_PROP_FWD,
  PC_NODE_UPDATE, PC_INPUT_UPDATE, PC_SCORE, PC_ARGMAX, PC_GATHER, PC_GATHER_INPUT, PC_CLASSIFY, PC_LIVESUM, PC_TOP, PC_COUNT
};
static const char* kProfNames[PC_COUNT] = {
    "k_embed", "k_pre", "k_pre_inp", "k_conv_fwd", "k_convT_bwd", "k_dense_agg", "k_prop",
    "k_node_update", "k_input_update", "k_score", "k_argmax", "k_gather", "k_gather_input_update", "k_classify", "k_livesum", "k_top"};

struct DevEdge {
  float *w_fwd = nullptr, *w_bwd = nullptr, *bias = nullptr;   // conv: tap-major copies; linear: W^T / W, zero-padded
  int ld_fwd = 0, mt_fwd = 0, ksq_fwd = 0, ld_bwd = 0, mt_bwd = 0, ksq_bwd = 0, kpad_fwd = 0, kpad_bwd = 0;
};

struct DevGather {          // one conv edge in one direction, as MFMA gather tables on the device
  bool ok = false;
  GatherGeom g;
  float* cmat = nullptr;
  int* koff = nullptr;
  int* ttab = nullptr;
};

struct gnnb_handle {
  int T = 2, p = 64, device = 0, n_cu = 256;
  bool use_gather = true;       // MFMA gather for conv edges (false: VALU gather kernels)
  int nu_waves = 12;            // waves per workgroup of k_node_update: 3 per SIMD at 168 VGPRs, measured 6 % faster
                                // than 8 (16 waves: 27 % slower); k_gather_input_update prefers 8, k_gather 8 x 2 workgroups
  int gather_occ = 2;           // workgroups per CU for k_gather (its LDS footprint is only the tap matrix)
  bool dense_lds = true;        // Linear edges: one workgroup per sample with the source rows in LDS (false: per-tile kernel)
  bool restrict_last = true;    // last backward step of layer 1 only for the scored nodes (nothing else reads it)
  bool bf3 = true;              // node update: 64x64 blocks on the bf16 matrix rate with three-piece operands (fp32 accuracy)
  int gather_sparse = 7;        // gathers behind a ReLU layer walk only the live rows of their window: bit 0 = 16-node forward
                                // gathers, bit 1 = 32-node gathers, bit 2 = the input-layer gather
  bool gather16 = true;         // forward conv edges: 16-node tiles on the 16x16x4 MFMA when their window is smaller
  bool embed_fuse = true;       // round 0: the first forward gather computes the input embedding itself (no k_embed, no mu[0] rows)
  bool use_top = true;          // fuse the top of the network (last Linear edge, last ReLU layer, property node) into k_top
  bool top_ok = false;          // ... which the bound network allows (set by gnnb_bind_network)
  int per_sample_min_b = 0;     // GNNB_PER_SAMPLE_MIN_B: batches below it take the per-tile dense kernel + separate launches
                                // instead of the one-workgroup-per-sample kernels (k_top, k_dense_*_lds), which need a batch
                                // that fills the CUs (B=2: 0.40 vs 0.49 ms, B=64: 0.64 vs 0.66, B=128: 0.96 vs 0.88 ms; 96 is
                                // the break-even).  Off by default: the two paths round differently, and with one path for
                                // every batch size a sample's scores do not depend on what it is batched or sharded with.
  Packs packs;
  std::vector<float> blob;      // the GNN parameters as handed to gnnb_create / gnnb_set_weights / left by gnnb_online_step
  gnnb_train::Trainer* trainer = nullptr;     // online learning (gnnb_online_create)
  float* d_pack[N_PACKS] = {nullptr};
  float* d_zero = nullptr;      // 64 zero floats: where masked gather loads point
  std::vector<int> proj;        // per graph layer: which Linear (LayerId) the rows of mu[k] still have to go through after
                                // the last enqueued kernel (-1: the rows are final) -- the "deferred projection" of gnnb_pack.h
  std::vector<DevGather> gf, gb;   // gf[k]: edge k forward (dst = layer k); gb[k]: edge k transposed (dst = layer k-1)
  bool bound = false;
  std::vector<Edge> edges;       // edges[k], k = 1..L (edges[0] unused)
  std::vector<DevEdge> dev;
  std::vector<int> N;            // graph layer sizes, N[0..L+1]
  std::vector<int> relu_q;       // fixed-layer index of the ReLU of graph layer k
  std::vector<int> hw;           // nodes per bias entry of layer k
  int n_fixed = 0, R = 0;
  int halfpass_limit = 0;
  bool prof = false;
  struct Ev { int cls; hipEvent_t a, b; };
  std::vector<Ev> pending;
  std::vector<hipEvent_t> pool;
  double prof_ms[PC_COUNT] = {0};
  int64_t prof_n[PC_COUNT] = {0};
  hipStream_t prof_stream = nullptr;
};


static int upload(float** d, const float* h, size_t n) {
  HIPCHK(hipMalloc((void**)d, n * sizeof(float)));
  HIPCHK(hipMemcpy(*d, h, n * sizeof(float), hipMemcpyHostToDevice));
  return 0;
}

// (re)build the operand packs of the scorer from a parameter blob and put them on the device
static int load_weights(gnnb_t* h, const float* w_blob, hipStream_t st) {
  h->blob.assign(w_blob, w_blob + blob_floats());
  build_packs(h->blob.data(), h->packs);
  const std::vector<float>* pv[N_PACKS] = {&h->packs.embed, &h->packs.pre_fwd, &h->packs.pre_bwd, &h->packs.pre_inp, &h->packs.prop,
                                           &h->packs.upd_fwd_e, &h->packs.upd_fwd_i, &h->packs.upd_fwd_f, &h->packs.upd_bwd,
                                           &h->packs.upd_bwd_b, &h->packs.upd_inp, &h->packs.post_inp, &h->packs.score_b,
                                           &h->packs.score_f};
  for (int i = 0; i < N_PACKS; ++i) {
    if (!h->d_pack[i]) {
      if (int rc = upload(&h->d_pack[i], pv[i]->data(), pv[i]->size())) return rc;
    } else {
      HIPCHK(hipMemcpyAsync(h->d_pack[i], pv[i]->data(), pv[i]->size() * sizeof(float), hipMemcpyHostToDevice, st));
    }
  }
  HIPCHK(hipStreamSynchronize(st));       // the host vectors are reused by the next call
  return 0;
}

extern "C" int gnnb_abi_version(void) { return GNNB_ABI_VERSION; }
extern "C" const char* gnnb_last_error(void) { return g_err.c_str(); }

extern "C" int gnnb_create(gnnb_t** out, const float* w_blob, size_t n_floats, int T, int p) {
  if (!out || !w_blob) return fail(GNNB_E_INVALID, "gnnb_create: null argument");
  if (p != P) return fail(GNNB_E_INVALID, "gnnb_create: embedding size %d unsupported (kernels are built for p=64)", p);
  if (T < 1 || T > 16) return fail(GNNB_E_INVALID, "gnnb_create: T=%d out of range", T);
  if (n_floats != blob_floats()) return fail(GNNB_E_INVALID, "gnnb_create: weight blob has %zu floats, expected %zu", n_floats, blob_floats());
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) return fail(GNNB_E_HIP, "gnnb_create: no HIP device (%s)", hipGetErrorString(e));
  gnnb_t* h = new gnnb_handle();
  h->T = T;
  h->p = p;
  HIPCHK(hipGetDevice(&h->device));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, h->device));
  h->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (int rc = load_weights(h, w_blob, nullptr)) return rc;
  HIPCHK(hipMalloc((void**)&h->d_zero, 256 * sizeof(float)));
  HIPCHK(hipMemset(h->d_zero, 0, 256 * sizeof(float)));
  // > 64 KiB of dynamic LDS needs the attribute
  HIPCHK(hipFuncSetAttribute((const void*)k_pre, hipFuncAttributeMaxDynamicSharedMemorySize, (PackPreFwd::FLOATS + PackPreBwd::FLOATS) * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_pre_inp, hipFuncAttributeMaxDynamicSharedMemorySize, PackPreInp::FLOATS * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_node_update<8, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (PackUpd::FLOATS + 4096) * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_node_update<12, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (PackUpd::FLOATS + 4096) * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_node_update<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (PackUpd::FLOATS + 4096) * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_node_update<12, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (PackUpd::FLOATS + 4096) * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_node_update<8, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (PackUpd::FLOATS + 4096) * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_node_update<8, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (PackUpd::FLOATS + 4096) * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_node_update<12, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (PackUpd::FLOATS + 4096) * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_node_update<12, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (PackUpd::FLOATS + 4096) * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_node_update<12, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (PackUpdL3::FLOATS + 6144) * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_node_update<12, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (PackUpdL3::FLOATS + 6144) * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_node_update<12, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (PackUpdL3::FLOATS + 6144) * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_node_update<12, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (PackUpdL3::FLOATS + 6144) * 4));
  if (const char* e = getenv("GNNB_BF3")) h->bf3 = e[0] == '1';
  if (const char* e = getenv("GNNB_NU_WAVES")) h->nu_waves = atoi(e) == 8 ? 8 : 12;
  if (const char* e = getenv("GNNB_GATHER_OCC")) h->gather_occ = atoi(e) < 1 ? 1 : atoi(e);
  HIPCHK(hipFuncSetAttribute((const void*)k_input_update, hipFuncAttributeMaxDynamicSharedMemorySize, PackUpdInp::FLOATS * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_score, hipFuncAttributeMaxDynamicSharedMemorySize, PackScore::FLOATS * 4));
  HIPCHK(hipFuncSetAttribute((const void*)k_gather<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_gather<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_gather16<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_gather16<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  if (const char* e = getenv("GNNB_NO_GATHER16")) h->gather16 = !(e[0] == '1');
  if (const char* e = getenv("GNNB_SPARSE")) h->gather_sparse = atoi(e);
  HIPCHK(hipFuncSetAttribute((const void*)k_gather16<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  if (const char* e = getenv("GNNB_NO_EMBED_FUSE")) h->embed_fuse = !(e[0] == '1');
  HIPCHK(hipFuncSetAttribute((const void*)k_livesum, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  if (const char* e = getenv("GNNB_NO_RESTRICT")) h->restrict_last = !(e[0] == '1');
  if (const char* e = getenv("GNNB_NO_DENSE_LDS")) h->dense_lds = !(e[0] == '1');
  HIPCHK(hipFuncSetAttribute((const void*)k_gather_input_update<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_gather_input_update<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_gather_input_update<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_gather_input_update<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  HIPCHK(hipFuncSetAttribute((const void*)k_gather<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  if (const char* e = getenv("GNNB_NO_GATHER")) h->use_gather = !(e[0] == '1');
  if (const char* e = getenv("GNNB_NO_TOP")) h->use_top = !(e[0] == '1');
  if (const char* e = getenv("GNNB_PER_SAMPLE_MIN_B")) h->per_sample_min_b = atoi(e);
  HIPCHK(hipFuncSetAttribute((const void*)k_top, hipFuncAttributeMaxDynamicSharedMemorySize, TOP_LDS_FLOATS * 4));
  *out = h;
  return GNNB_OK;
}

static void free_trainer(gnnb_t* h) {
  gnnb_train::Trainer* t = h->trainer;
  if (!t) return;
  for (float* p : {t->d_w, t->d_g, t->d_m, t->d_v, t->d_scores, t->d_ds, t->d_loss, t->d_imp})
    if (p) (void)hipFree(p);
  if (t->d_kw) (void)hipFree(t->d_kw);
  if (t->d_sel) (void)hipFree(t->d_sel);
  for (float* p : t->edge_w)
    if (p) (void)hipFree(p);
  t->arena.release();
  for (hipEvent_t e : t->events) (void)hipEventDestroy(e);
  if (t->side) (void)hipStreamDestroy(t->side);
  delete t;
  h->trainer = nullptr;
}

static void free_network(gnnb_t* h) {
  if (h->trainer) {                       // the edge weights of the trainer belong to the network that goes away
    for (float* p : h->trainer->edge_w)
      if (p) (void)hipFree(p);
    h->trainer->edge_w.clear();
  }
  for (auto& d : h->dev) {
    if (d.w_fwd) (void)hipFree(d.w_fwd);
    if (d.w_bwd) (void)hipFree(d.w_bwd);
    if (d.bias) (void)hipFree(d.bias);
  }
  for (auto* v : {&h->gf, &h->gb})
    for (auto& d : *v) {
      if (d.cmat) (void)hipFree(d.cmat);
      if (d.koff) (void)hipFree(d.koff);
      if (d.ttab) (void)hipFree(d.ttab);
    }
  h->gf.clear();
  h->gb.clear();
  h->dev.clear();
  h->edges.clear();
  h->N.clear();
  h->relu_q.clear();
  h->hw.clear();
  h->bound = false;
}

extern "C" int gnnb_destroy(gnnb_t* h) {
  if (!h) return GNNB_OK;
  free_network(h);
  for (int i = 0; i < N_PACKS; ++i)
    if (h->d_pack[i]) (void)hipFree(h->d_pack[i]);
  if (h->d_zero) (void)hipFree(h->d_zero);
  free_trainer(h);
  for (auto& ev : h->pending) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
  for (auto& ev : h->pool) (void)hipEventDestroy(ev);
  delete h;
  return GNNB_OK;
}

static bool conv_channels_ok(int c) { return c == 3 || c == 8 || c == 16 || c == 32; }

extern "C" int gnnb_bind_network(gnnb_t* h, const gnnb_layer_desc* L, int n, int c0, int h0, int w0) {
  if (!h || !L || n < 2) return fail(GNNB_E_INVALID, "gnnb_bind_network: bad arguments");
  free_network(h);
  int C = c0, H = h0, W = w0;
  bool flat = false;
  int nflat = c0 * h0 * w0;
  h->N.push_back(nflat);
  h->edges.emplace_back();
  h->relu_q.push_back(-1);
  h->hw.push_back(1);
  Edge pend;
  bool have = false;
  int pend_hw = 1;
  for (int q = 0; q < n; ++q) {
    const gnnb_layer_desc& d = L[q];
    if (d.kind == GNNB_CONV) {
      if (flat) return fail(GNNB_E_INVALID, "layer %d: conv after flatten", q);
      if (have) return fail(GNNB_E_INVALID, "layer %d: two linear maps without a ReLU between them", q);
      if (d.c_in != C) return fail(GNNB_E_INVALID, "layer %d: conv expects %d input channels, graph has %d", q, d.c_in, C);
      if (!d.weight || !d.bias) return fail(GNNB_E_INVALID, "layer %d: null weight/bias", q);
      Edge e;
      e.kind = 0;
      e.c_in = C; e.h_in = H; e.w_in = W; e.c_out = d.c_out; e.kh = d.kh; e.kw = d.kw; e.stride = d.stride; e.pad = d.pad;
      if (d.stride < 1 || d.kh < 1 || d.kw < 1) return fail(GNNB_E_INVALID, "layer %d: bad conv geometry", q);
      if ((H + 2 * d.pad - d.kh) % d.stride || (W + 2 * d.pad - d.kw) % d.stride)
        return fail(GNNB_E_INVALID, "layer %d: conv geometry leaves a remainder (conv_transpose2d of the reference would need output_padding)", q);
      e.h_out = (H + 2 * d.pad - d.kh) / d.stride + 1;
      e.w_out = (W + 2 * d.pad - d.kw) / d.stride + 1;
      e.n_in = C * H * W;
      e.n_out = e.c_out * e.h_out * e.w_out;
      e.w.assign(d.weight, d.weight + (size_t)e.c_out * e.c_in * e.kh * e.kw);
      e.b.assign(d.bias, d.bias + e.c_out);
      C = e.c_out; H = e.h_out; W = e.w_out;
      nflat = e.n_out;
      pend_hw = H * W;
      pend = e;
      have = true;
    } else if (d.kind == GNNB_LINEAR) {
      if (have) return fail(GNNB_E_INVALID, "layer %d: two linear maps without a ReLU between them", q);
      if (d.n_in != nflat) return fail(GNNB_E_INVALID, "layer %d: linear expects %d inputs, graph has %d", q, d.n_in, nflat);
      if (!d.weight || !d.bias) return fail(GNNB_E_INVALID, "layer %d: null weight/bias", q);
      Edge e;
      e.kind = 1;
      e.n_in = d.n_in; e.n_out = d.n_out;
      e.c_in = e.h_in = e.w_in = e.c_out = e.h_out = e.w_out = e.kh = e.kw = e.stride = e.pad = 0;
      e.w.assign(d.weight, d.weight + (size_t)d.n_out * d.n_in);
      e.b.assign(d.bias, d.bias + d.n_out);
      nflat = d.n_out;
      flat = true;
      pend_hw = 1;
      pend = e;
      have = true;
    } else if (d.kind == GNNB_RELU) {
      if (!have) return fail(GNNB_E_INVALID, "layer %d: ReLU without a preceding conv/linear", q);
      h->N.push_back(nflat);
      h->edges.push_back(pend);
      h->relu_q.push_back(q);
      h->hw.push_back(pend_hw);
      have = false;
    } else if (d.kind == GNNB_FLATTEN) {
      flat = true;
    } else {
      return fail(GNNB_E_INVALID, "layer %d: unknown kind %d", q, d.kind);
    }
  }
  if (have) return fail(GNNB_E_INVALID, "fixed layers must end after a ReLU (the property layer is passed per batch)");
  const int Lr = (int)h->N.size() - 1;
  if (Lr < 1 || Lr > MAXL) return fail(GNNB_E_INVALID, "unsupported number of ReLU layers %d (max %d)", Lr, MAXL);
  h->N.push_back(1);   // property node
  h->n_fixed = n;
  h->R = 0;
  for (int k = 1; k <= Lr; ++k) h->R += h->N[k];
  h->dev.resize(Lr + 1);
  h->proj.assign(Lr + 2, -1);
  for (int k = 0; k <= Lr; ++k)
    if (h->N[k] > LIVESUM_MAXSRC) return fail(GNNB_E_INVALID, "graph layer %d has %d nodes, more than the %d k_livesum holds in LDS", k, h->N[k], LIVESUM_MAXSRC);
  for (int k = 1; k <= Lr; ++k) {
    const Edge& e = h->edges[k];
    DevEdge& d = h->dev[k];
    if (int rc = upload(&d.bias, e.b.data(), e.b.size())) return rc;
    if (e.kind == 0) {
      std::vector<float> t(e.w.size());
      pack_conv_fwd(t.data(), e);
      if (int rc = upload(&d.w_fwd, t.data(), t.size())) return rc;
      pack_conv_bwd(t.data(), e);
      if (int rc = upload(&d.w_bwd, t.data(), t.size())) return rc;
    } else {
      // k_dense_agg operands At[k][i] = A[i][k], zero-padded to 32*MT columns and 8*ksq rows (ksq = k-steps per wave)
      auto ksq_of = [](int K) { return (((K + 1) / 2 + 3) / 4 + DENSE_CH - 1) / DENSE_CH * DENSE_CH; };
      d.mt_fwd = (e.n_out + 31) / 32;
      d.ld_fwd = d.mt_fwd * 32;
      d.ksq_fwd = ksq_of(e.n_in);
      d.kpad_fwd = (e.n_in + 63) / 64 * 64;
      d.kpad_bwd = (e.n_out + 15) / 16 * 16;
      const size_t rows_f = std::max<size_t>(8 * d.ksq_fwd + 2 * DENSE_CH, d.kpad_fwd + 32);
      std::vector<float> t(rows_f * d.ld_fwd, 0.f);           // forward: A = W, k = input node
      for (int o = 0; o < e.n_out; ++o)
        for (int i = 0; i < e.n_in; ++i) t[(size_t)i * d.ld_fwd + o] = e.w[(size_t)o * e.n_in + i];
      if (int rc = upload(&d.w_fwd, t.data(), t.size())) return rc;
      d.mt_bwd = (e.n_in + 31) / 32;
      d.ld_bwd = d.mt_bwd * 32;
      d.ksq_bwd = ksq_of(e.n_out);
      const size_t rows_b = std::max<size_t>(8 * d.ksq_bwd + 2 * DENSE_CH, d.kpad_bwd + 64);   // k_dense_bwd_lds reads up to 3 chunks past kpad
      t.assign(rows_b * d.ld_bwd, 0.f);                        // transposed: A = W^T, k = output node
      for (int o = 0; o < e.n_out; ++o)
        for (int i = 0; i < e.n_in; ++i) t[(size_t)o * d.ld_bwd + i] = e.w[(size_t)o * e.n_in + i];
      if (int rc = upload(&d.w_bwd, t.data(), t.size())) return rc;
    }
  }
  h->top_ok = Lr >= 2 && h->edges[Lr].kind == 1 && h->N[Lr] <= 128 && h->dense_lds && h->dev[Lr].mt_fwd <= 4 && h->dev[Lr].kpad_bwd <= 128;
  // MFMA gather tables for every conv edge, both directions (the input layer's transposed edge is not normalised)
  h->gf.assign(Lr + 1, DevGather());
  h->gb.assign(Lr + 1, DevGather());
  if (h->use_gather)
    for (int k = 1; k <= Lr; ++k) {
      if (h->edges[k].kind != 0) continue;
      for (int dir = 0; dir < 2; ++dir) {
        GatherHost gh;
        // the input layer's transposed gather is fused with its feature chain and update (132 MFMAs per tile)
        if (!build_gather(h->edges[k], dir, dir == 1 && k > 1, gh, (dir == 1 && k == 1) ? 132 : 0, h->gather16)) continue;
        DevGather& d = dir == 0 ? h->gf[k] : h->gb[k];
        d.g = gh.g;
        if (int rc = upload(&d.cmat, gh.cmat.data(), gh.cmat.size())) return rc;
        HIPCHK(hipMalloc((void**)&d.koff, gh.koff.size() * sizeof(int)));
        HIPCHK(hipMemcpy(d.koff, gh.koff.data(), gh.koff.size() * sizeof(int), hipMemcpyHostToDevice));
        {
          const TileMap& tm = gh.g.tm;
          if (tm.NCG > 255 || tm.NBY > 4095 || tm.NBX > 4095) return fail(GNNB_E_INVALID, "layer %d: tile table overflow", k);
          std::vector<int> tt(tm.TPS);
          for (int t = 0; t < tm.TPS; ++t) {
            const int cg = t / (tm.NBY * tm.NBX), rem = t % (tm.NBY * tm.NBX);
            tt[t] = cg | ((rem / tm.NBX) << 8) | ((rem % tm.NBX) << 20);
          }
          HIPCHK(hipMalloc((void**)&d.ttab, tt.size() * sizeof(int)));
          HIPCHK(hipMemcpy(d.ttab, tt.data(), tt.size() * sizeof(int), hipMemcpyHostToDevice));
        }
        d.ok = true;
      }
    }
  // an edge without MFMA gather tables falls back to the VALU gathers, which are compiled for a few channel counts only
  for (int k = 1; k <= Lr; ++k) {
    const Edge& e = h->edges[k];
    if (e.kind != 0) continue;
    if ((!h->gf[k].ok && !conv_channels_ok(e.c_out)) || (!h->gb[k].ok && !conv_channels_ok(e.c_in)))
      return fail(GNNB_E_INVALID, "conv edge %d (%d -> %d channels): no MFMA gather tables and the fallback kernels only cover channel counts "
                  "{3, 8, 16, 32}", k, e.c_in, e.c_out);
  }
  h->bound = true;
  return GNNB_OK;
}

// tile map of the input layer's update: the tiles of the transposed gather of edge 1 when it exists, else flat
static TileMap flat_map(int N) { TileMap t; t.mode = 0; t.N = N; return t; }
static TileMap bwd_map(const gnnb_t* h, int k) {
  const int L = (int)h->N.size() - 2;
  return (k + 1 <= L && h->gb[k + 1].ok) ? h->gb[k + 1].g.tm : flat_map(h->N[k]);
}
static long map_tiles(const TileMap& t, int B) { return t.mode ? (long)B * t.TPS : ((long)B * t.N + 31) / 32; }
static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static DTileMap to_dtm(const TileMap& t) {
  return DTileMap{t.mode, t.N, t.C, t.H, t.W, t.CT, t.PY, t.PX, t.ay, t.ax, t.NBY, t.NBX, t.NCG, t.TPS, ilog2(t.PY), ilog2(t.PX)};
}
static DGather to_dg(const DevGather& d, const float* zero) {
  const GatherGeom& g = d.g;
  return DGather{d.cmat, reinterpret_cast<const int2*>(d.koff), d.ttab, zero, g.K2, g.tm.NCG * g.K2, g.Hs, g.Ws, g.Ns, g.ystep, g.ybase,
                 g.xstep, g.xbase, g.WY, g.WX, g.normalise, g.kh, g.kw, g.stride, g.pad, g.lanes};
}
static size_t gather_lds_bytes(const DevGather& d, size_t pack_floats) {
  return (pack_floats + (size_t)d.g.tm.NCG * d.g.K2 * 64) * 4 + (size_t)gather_slots(d.g.K2, d.g.lanes) * 12 + (size_t)((d.g.tm.TPS + 3) & ~3) * 4;
}

// per-wave live-slot tables of the sparse gathers (behind the shared tables, 8-byte aligned)
static size_t sparse_tab_bytes(const DevGather& d) {
  return 8 + (size_t)WAVES_MLP * ((d.g.lanes == 16 ? 4 : 2) * d.g.K2 + 32) * 8;
}

extern "C" int gnnb_graph_info(const gnnb_t* h, int* n_graph, int* sizes, int* n_relu_total) {
  if (!h || !h->bound) return fail(GNNB_E_STATE, "gnnb_graph_info: no network bound");
  if (n_graph) *n_graph = (int)h->N.size();
  if (sizes)
    for (size_t k = 0; k < h->N.size(); ++k) sizes[k] = h->N[k];
  if (n_relu_total) *n_relu_total = h->R;
  return GNNB_OK;
}

// JSON description of the launch plan of one forward (per B=1): which kernel updates which layer, tile shapes and
// MFMA counts.  bench.py derives the algorithmic flops per kernel class from it; DESIGN.md quotes it.
extern "C" int gnnb_describe(const gnnb_t* h, char* buf, size_t cap) {
  if (!h || !h->bound || !buf || cap < 64) return fail(GNNB_E_INVALID, "gnnb_describe: bad arguments");
  const int L = (int)h->N.size() - 2;
  std::string o = "{\"T\": " + std::to_string(h->T) + ", \"bf3\": " + std::to_string(h->bf3 && h->nu_waves == 12 ? 1 : 0) + ", \"sizes\": [";
  for (size_t k = 0; k < h->N.size(); ++k) o += (k ? ", " : "") + std::to_string(h->N[k]);
  o += "], \"updates\": [";
  auto nnz = [&](int e) -> long {     // edges of the layer graph between layer e-1 and e (no-padding upper bound)
    if (e > L) return h->N[L];
    const Edge& ed = h->edges[e];
    return ed.kind == 0 ? (long)ed.c_out * ed.h_out * ed.w_out * ed.c_in * ed.kh * ed.kw : (long)ed.n_in * ed.n_out;
  };
  auto item = [&](const char* what, int k, const DevGather* d, const char* fallback, int n_src) {
    char t[640];
    const long ez = nnz(what[0] == 'f' ? k : k + 1);
    if (d && d->ok) {
      const GatherGeom& g = d->g;
      snprintf(t, sizeof t,
               "{\"update\": \"%s\", \"layer\": %d, \"kernel\": \"%s\", \"nodes\": %d, \"tiles_per_sample\": %d, \"tile_nodes\": %d, "
               "\"tile\": [%d, %d, %d], \"align\": [%d, %d], \"window\": [%d, %d], \"gather_ksteps\": %d, \"n_src\": %d, \"edge_nnz\": %ld}",
               what, k, k == 0 ? "k_gather_input_update" : "k_gather+k_node_update", h->N[k], g.tm.TPS, g.lanes, g.tm.CT, g.tm.PY, g.tm.PX,
               g.tm.ay, g.tm.ax, g.WY, g.WX, g.K2, n_src, ez);
    } else {
      snprintf(t, sizeof t, "{\"update\": \"%s\", \"layer\": %d, \"kernel\": \"%s\", \"nodes\": %d, \"n_src\": %d, \"edge_nnz\": %ld}",
               what, k, fallback, h->N[k], n_src, ez);
    }
    o += t;
  };
  const bool top = h->use_top && h->top_ok;     // k_top covers the edge into layer L, both updates of layer L and the edge back
  bool first = true;
  for (int k = 1; k <= L; ++k) {
    if (!first) o += ", ";
    first = false;
    if (top && k == L) item("fwd", k, nullptr, "k_top+k_top", h->N[k - 1]);
    else item("fwd", k, &h->gf[k], h->edges[k].kind == 0 ? "k_conv_fwd+k_node_update" : "k_dense_agg+k_node_update", h->N[k - 1]);
  }
  for (int k = L; k >= 1; --k) {
    o += ", ";
    if (k == L) item("bwd", k, nullptr, top ? "k_top+k_top" : "k_prop+k_node_update", 1);
    else if (top && k == L - 1) item("bwd", k, nullptr, "k_top+k_node_update", h->N[k + 1]);
    else item("bwd", k, &h->gb[k + 1], h->edges[k + 1].kind == 0 ? "k_convT_bwd+k_node_update" : "k_dense_agg+k_node_update", h->N[k + 1]);
  }
  o += ", ";
  item("input", 0, &h->gb[1], h->edges[1].kind == 0 ? "k_convT_bwd+k_input_update" : "k_dense_agg+k_input_update", h->N[1]);
  o += "]}";
  if (o.size() + 1 > cap) return fail(GNNB_E_NOMEM, "gnnb_describe: buffer too small (%zu needed)", o.size() + 1);
  memcpy(buf, o.c_str(), o.size() + 1);
  return GNNB_OK;
}

// ---- workspace layout (float offsets, every region 256-B aligned) ----
struct WsLayout {
  std::vector<size_t> mu, Pf, Pb, live, amb, score;
  std::vector<size_t> lf;       // live flags (B, N_k) as floats
  std::vector<size_t> sf, sb;   // k_livesum outputs: sf[k] (B, N_k) over edge k, sb[k] (B, N_k) over edge k+1 transposed
  size_t F1 = 0;                // rows of layer 1 after the producer-side map of the input update (PackPostInp)
  size_t cnt = 0, nb = 0, Q = 0, total = 0;
};
static size_t align64(size_t nfloats) { return (nfloats + 63) & ~(size_t)63; }
static WsLayout ws_layout(const gnnb_t* h, int B) {
  WsLayout w;
  const int K = (int)h->N.size() - 1;
  size_t off = 0;
  w.cnt = off; off += 64;                      // int counters: 4 per ReLU layer (live, amb, score, pad), zeroed every forward
  w.mu.resize(K + 1);
  for (int k = 0; k <= K; ++k) { w.mu[k] = off; off += align64((size_t)B * h->N[k] * 64); }
  size_t maxn = 0;
  for (int k = 0; k < K; ++k) maxn = std::max(maxn, (size_t)h->N[k]);
  w.nb = off; off += align64((size_t)B * maxn * 64);
  w.Pf.resize(K); w.Pb.resize(K); w.live.resize(K); w.amb.resize(K); w.score.resize(K);
  for (int k = 1; k < K; ++k) { w.Pf[k] = off; off += align64((size_t)B * h->N[k] * 64); }
  for (int k = 1; k < K; ++k) { w.Pb[k] = off; off += align64((size_t)B * h->N[k] * 64); }
  for (int k = 1; k < K; ++k) {
    w.live[k] = off; off += align64((size_t)B * h->N[k]);
    w.amb[k] = off; off += align64((size_t)B * h->N[k]);
    w.score[k] = off; off += align64((size_t)B * h->N[k]);
  }
  w.sf.assign(K, 0); w.sb.assign(K, 0); w.lf.assign(K, 0);
  for (int k = 1; k < K; ++k) { w.lf[k] = off; off += align64((size_t)B * h->N[k]); }
  for (int k = 1; k < K; ++k) { w.sf[k] = off; off += align64((size_t)B * h->N[k]); }
  for (int k = 0; k < K - 1; ++k) { w.sb[k] = off; off += align64((size_t)B * h->N[k]); }
  w.F1 = off; off += align64((size_t)B * h->N[1] * 64);
  w.Q = off; off += (size_t)map_tiles(bwd_map(h, 0), B) * 2048;
  w.total = off;
  return w;
}

extern "C" size_t gnnb_workspace_bytes(const gnnb_t* h, int B) {
  if (!h || !h->bound || B < 1) return 0;
  return ws_layout(h, B).total * sizeof(float);
}

extern "C" int gnnb_mu_location(const gnnb_t* h, int B, int k, size_t* offset_bytes, size_t* n_floats) {
  if (!h || !h->bound) return fail(GNNB_E_STATE, "gnnb_mu_location: no network bound");
  if (k < 0 || k >= (int)h->N.size() || B < 1) return fail(GNNB_E_INVALID, "gnnb_mu_location: bad layer/batch");
  WsLayout w = ws_layout(h, B);
  if (offset_bytes) *offset_bytes = w.mu[k] * sizeof(float);
  if (n_floats) *n_floats = (size_t)B * h->N[k] * 64;
  return GNNB_OK;
}

// Inspection: the rows of mu[k] written by the last forward are E with mu = W.E + b for the Linear `*linear_id`
// (index into the checkpoint's 26 Linear layers in state-dict order), or final embeddings when *linear_id = -1.
extern "C" int gnnb_mu_projection(const gnnb_t* h, int k, int* linear_id) {
  if (!h || !linear_id) return fail(GNNB_E_INVALID, "gnnb_mu_projection: null argument");
  if (k < 0 || k >= (int)h->N.size()) return fail(GNNB_E_INVALID, "gnnb_mu_projection: bad layer");
  *linear_id = k < (int)h->proj.size() ? h->proj[k] : -1;
  return GNNB_OK;
}

extern "C" int gnnb_set_halfpass_limit(gnnb_t* h, int n) {
  if (!h) return fail(GNNB_E_INVALID, "null handle");
  h->halfpass_limit = n;
  return GNNB_OK;
}

// ---- profiling ----
extern "C" int gnnb_profile_enable(gnnb_t* h, int on) {
  if (!h) return fail(GNNB_E_INVALID, "null handle");
  h->prof = on != 0;
  return GNNB_OK;
}
extern "C" int gnnb_profile_classes(void) { return PC_COUNT; }
extern "C" const char* gnnb_profile_class_name(int cls) { return (cls >= 0 && cls < PC_COUNT) ? kProfNames[cls] : ""; }
extern "C" int gnnb_profile_read(gnnb_t* h, double* total_ms, int64_t* launches, int n, int reset) {
  if (!h) return fail(GNNB_E_INVALID, "null handle");
  if (!h->pending.empty()) {
    for (auto& ev : h->pending) {
      HIPCHK(hipEventSynchronize(ev.b));        // launches may sit on several streams (batch pipelining)
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, ev.a, ev.b));
      h->prof_ms[ev.cls] += ms;
      h->prof_n[ev.cls] += 1;
      h->pool.push_back(ev.a);
      h->pool.push_back(ev.b);
    }
    h->pending.clear();
  }
  for (int i = 0; i < n && i < PC_COUNT; ++i) {
    if (total_ms) total_ms[i] = h->prof_ms[i];
    if (launches) launches[i] = h->prof_n[i];
  }
  if (reset)
    for (int i = 0; i < PC_COUNT; ++i) { h->prof_ms[i] = 0; h->prof_n[i] = 0; }
  return GNNB_OK;
}

struct Launcher {
  gnnb_t* h;
  hipStream_t st;
  int rc = 0;
  hipEvent_t get_event() {
    if (!h->pool.empty()) { hipEvent_t e = h->pool.back(); h->pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) rc = fail(GNNB_E_HIP, "hipEventCreate failed");
    return e;
  }
  template <class F>
  void run(int cls, F&& f) {
    if (rc) return;
    if (h->prof) {
      gnnb_handle::Ev ev{cls, get_event(), get_event()};
      if (rc) return;
      (void)hipEventRecord(ev.a, st);
      f();
      (void)hipEventRecord(ev.b, st);
      h->pending.push_back(ev);
      h->prof_stream = st;
    } else {
      f();
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) rc = fail(GNNB_E_HIP, "launch of %s failed: %s", kProfNames[cls], hipGetErrorString(e));
  }
};

static int mlp_grid(const gnnb_t* h, long ntiles) {
  long g = (ntiles + WAVES_MLP - 1) / WAVES_MLP;
  if (g > h->n_cu) g = h->n_cu;
  return (int)(g < 1 ? 1 : g);
}

template <int C>
static void launch_conv_fwd(const ConvArgs& a, hipStream_t st) {
  const long waves = (long)a.B * a.H_out * a.W_out;
  hipLaunchKernelGGL(k_conv_fwd<C>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a);
}
template <int C>
static void launch_convT(const ConvArgs& a, hipStream_t st) {
  const long waves = (long)a.B * a.H_in * a.W_in;
  hipLaunchKernelGGL(k_convT_bwd<C>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a);
}

extern "C" int gnnb_forward(gnnb_t* h, const gnnb_batch* in, int B, float* scores, int32_t* decisions, int32_t* status,
                            void* workspace, size_t workspace_bytes, void* stream) {
  if (!h || !in || !scores || !decisions || !status || !workspace) return fail(GNNB_E_INVALID, "gnnb_forward: null argument");
  if (!h->bound) return fail(GNNB_E_STATE, "gnnb_forward: call gnnb_bind_network first");
  if (B < 1) return fail(GNNB_E_INVALID, "gnnb_forward: B=%d", B);
  const int K = (int)h->N.size() - 1, L = K - 1;
  if (in->n_graph != K + 1 || in->n_relu != L || in->n_primal != h->n_fixed + 1)
    return fail(GNNB_E_INVALID, "gnnb_forward: batch has %d graph layers / %d dual / %d primal tensors, network needs %d / %d / %d",
                in->n_graph, in->n_relu, in->n_primal, K + 1, L, h->n_fixed + 1);
  for (int k = 0; k <= K; ++k)
    if (!in->lb[k] || !in->ub[k]) return fail(GNNB_E_INVALID, "gnnb_forward: null bounds pointer for graph layer %d", k);
  for (int k = 0; k < L; ++k)
    if (!in->dual[k]) return fail(GNNB_E_INVALID, "gnnb_forward: null dual pointer %d", k);
  for (int m = 0; m < in->n_primal; ++m)
    if (!in->primal[m]) return fail(GNNB_E_INVALID, "gnnb_forward: null primal pointer %d", m);
  if (!in->x_lp || !in->prop_w || !in->prop_b || !in->mask) return fail(GNNB_E_INVALID, "gnnb_forward: null input pointer");
  if ((long)B * h->N[0] * 64 >= (1L << 40)) return fail(GNNB_E_INVALID, "gnnb_forward: batch too large");
  const WsLayout w = ws_layout(h, B);
  if (workspace_bytes < w.total * sizeof(float))
    return fail(GNNB_E_NOMEM, "gnnb_forward: workspace %zu bytes < required %zu", workspace_bytes, w.total * sizeof(float));
  float* ws = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  Launcher lz{h, st};
  auto mu = [&](int k) { return ws + w.mu[k]; };
  float* nb = ws + w.nb;

  HIPCHK(hipMemsetAsync(status, 0, sizeof(int32_t), st));
  HIPCHK(hipMemsetAsync(ws + w.cnt, 0, 64 * sizeof(float), st));
  int* cnt = reinterpret_cast<int*>(ws + w.cnt);
  auto ilist = [&](size_t off) { return reinterpret_cast<int*>(ws + off); };
  std::vector<int> roff(L + 2, 0);          // offset of layer k inside the flat ReLU index
  for (int k = 2; k <= L + 1; ++k) roff[k] = roff[k - 1] + h->N[k - 1];

  const int total_halfpasses = 2 * h->T;
  const int limit = h->halfpass_limit > 0 ? std::min(h->halfpass_limit, total_halfpasses) : total_halfpasses;
  const bool debug_full = h->halfpass_limit > 0;   // with a limit set nothing is restricted or skipped as dead
  const bool per_sample = B >= h->per_sample_min_b;   // batch large enough for the one-workgroup-per-sample kernels
  // The rows of layer 1 the input-layer update aggregates went through its 64x64 map on the producer side (PackPostInp).
  // Outside inspection runs nothing else reads the plain rows of that half-pass, so the mapped rows simply take their place
  // in mu[1] (whose dead rows k_classify already zeroed); inspection runs keep both, the mapped ones in F1.
  float* const rows1_for_input = debug_full ? ws + w.F1 : mu(1);

  const bool embed_in_gather = h->embed_fuse && !debug_full && h->gf[1].ok;
  // ---- once per forward: classification lists, input embedding, embedding-independent feature chains ----
  {
    ClassifyArgs a{};
    a.L = L; a.mask = in->mask; a.scores = scores; a.cnt = cnt + 4; a.R = h->R;
    a.mu2 = debug_full ? ws + w.F1 : nullptr;      // inspection runs keep the plain rows in mu[1] and the mapped ones in F1
    int blk = 0;
    for (int k = 1; k <= L; ++k) {
      const int i = k - 1;
      a.lb[i] = in->lb[k]; a.ub[i] = in->ub[k]; a.mu[i] = mu(k);
      a.live[i] = ilist(w.live[k]); a.amb[i] = ilist(w.amb[k]); a.score[i] = ilist(w.score[k]);
      a.livef[i] = ws + w.lf[k];
      a.G[i] = (long)B * h->N[k]; a.N[i] = h->N[k]; a.off[i] = roff[k];
      a.blk0[i] = blk;
      blk += (int)((a.G[i] + CLS_THREADS - 1) / CLS_THREADS);
    }
    a.blk0[L] = blk;
    lz.run(PC_CLASSIFY, [&] { hipLaunchKernelGGL(k_classify, dim3((unsigned)blk), dim3(CLS_THREADS), 0, st, a); });
  }

  {   // bias-sum scalars of every edge and direction (the rows carry deferred projections)
    LiveSumArgs a{};
    a.B = B;
    int q = 0, maxw = 0;
    auto push = [&](int kind, const Edge& e, const float* wt, int ld, const float* lf, float* out, int Ndst, int Nsrc, int normalise) {
      LiveSumJob& j = a.job[q++];
      j.kind = kind; j.w = wt; j.lf = lf; j.out = out; j.Ndst = Ndst; j.Nsrc = Nsrc; j.ld = ld; j.normalise = normalise;
      j.c_in = e.c_in; j.h_in = e.h_in; j.w_in = e.w_in; j.c_out = e.c_out; j.h_out = e.h_out; j.w_out = e.w_out;
      j.kh = e.kh; j.kw = e.kw; j.stride = e.stride; j.pad = e.pad;
      const long nw = (long)e.c_in * e.c_out * e.kh * e.kw;
      j.wlds = (e.kind == 0 && nw <= LIVESUM_MAXW) ? (int)nw : 0;
      maxw = std::max(maxw, j.wlds);
    };
    for (int k = 1; k <= L; ++k) {            // forward edge k: source layer k-1 (the input layer is all live)
      const Edge& e = h->edges[k];
      push(e.kind == 0 ? 0 : 1, e, e.kind == 0 ? h->dev[k].w_fwd : h->dev[k].w_bwd, h->dev[k].ld_bwd, k > 1 ? ws + w.lf[k - 1] : nullptr,
           ws + w.sf[k], h->N[k], h->N[k - 1], 0);
    }
    if (limit >= 2)
      for (int k = 0; k < L; ++k) {           // edge k+1 transposed: source layer k+1
        const Edge& e = h->edges[k + 1];
        push(e.kind == 0 ? 2 : 3, e, h->dev[k + 1].w_bwd, h->dev[k + 1].ld_bwd, ws + w.lf[k + 1], ws + w.sb[k], h->N[k], h->N[k + 1],
             k >= 1 ? 1 : 0);
      }
    a.njobs = q;
    if (const char* e = getenv("GNNB_LS_ONLY")) {       // dev: time one job (results are wrong)
      const int only = atoi(e);
      if (only >= 0 && only < q) { a.job[0] = a.job[only]; q = 1; a.njobs = 1; }
    }
    int maxn = 0;
    for (int k = 0; k <= L; ++k) maxn = std::max(maxn, h->N[k]);
    a.lv_floats = (maxn + 3) & ~3;
    if ((size_t)(a.lv_floats + maxw) * 4 > 160 * 1024) {      // very wide layers: leave the weights in global memory
      for (int i = 0; i < q; ++i) a.job[i].wlds = 0;
      maxw = 0;
    }
    // (running this and k_pre on a side stream under k_embed / the first aggregation was measured: 1.72 ms vs 1.59 ms in-line)
    lz.run(PC_LIVESUM, [&] { hipLaunchKernelGGL(k_livesum, dim3((unsigned)B, (unsigned)q), dim3(256), (size_t)(a.lv_floats + maxw) * sizeof(float), st, a); });
  }
  {
    const long G = (long)B * h->N[0];
    EmbedArgs a{h->d_pack[PK_EMBED] + PackEmbed::W, h->d_pack[PK_EMBED] + PackEmbed::B, in->lb[0], in->x_lp, in->ub[0], mu(0), G};
    long grid = (G + 16 * EMBED_UNROLL - 1) / (16 * EMBED_UNROLL);
    if (grid > (long)h->n_cu * 16) grid = (long)h->n_cu * 16;
    // with the MFMA gather on the first edge, round 0 computes the embedding inside that gather (k_gather<true>): nothing
    // else reads mu[0] before the input-layer update overwrites it.  Inspection runs keep the rows.
    if (!embed_in_gather) lz.run(PC_EMBED, [&] { hipLaunchKernelGGL(k_embed, dim3((unsigned)grid), dim3(256), 0, st, a); });
    for (auto& pj : h->proj) pj = -1;
    h->proj[0] = L_INP_F_1;
  }
  {
    PreAllArgs a{};
    a.pack_f = h->d_pack[PK_PRE_FWD]; a.pack_b = h->d_pack[PK_PRE_BWD];
    a.L = L; a.do_bwd = limit >= 2 ? 1 : 0; a.cnt = cnt + 4;
    long nt = 0;                                      // upper bound: the kernel reads the real counts on the device
    for (int k = 1; k <= L; ++k) {
      const int i = k - 1, q = h->relu_q[k];
      a.lb[i] = in->lb[k]; a.ub[i] = in->ub[k]; a.dual[i] = in->dual[k - 1];
      a.z_pre[i] = in->primal[q - 1]; a.z_post[i] = in->primal[q]; a.bias[i] = h->dev[k].bias;
      a.Pf[i] = ws + w.Pf[k]; a.Pb[i] = ws + w.Pb[k]; a.list[i] = ilist(w.amb[k]);
      a.N[i] = h->N[k]; a.hw[i] = h->hw[k];
      nt += (((long)B * h->N[k] + 31) / 32) * 2;
    }
    const size_t lds = (PackPreFwd::FLOATS + PackPreBwd::FLOATS) * 4;
    lz.run(PC_PRE, [&] { hipLaunchKernelGGL(k_pre, dim3(mlp_grid(h, nt / 8)), dim3(WG_MLP), lds, st, a); });
  }
  const bool need_inp = (limit >= 2) && (h->T > 1 || debug_full) && !h->gb[1].ok;    // the fused input kernel computes Q itself
  if (need_inp) {
    const long G = (long)B * h->N[0];
    const TileMap tm = bwd_map(h, 0);
    const long nt = map_tiles(tm, B);
    PreArgs a{h->d_pack[PK_PRE_INP], in->lb[0], in->ub[0], nullptr, nullptr, nullptr, nullptr, ws + w.Q, G, nt, h->N[0], 1,
              to_dtm(tm), nullptr, nullptr};
    lz.run(PC_PRE_INP, [&] { hipLaunchKernelGGL(k_pre_inp, dim3(mlp_grid(h, nt)), dim3(WG_MLP), PackPreInp::FLOATS * 4, st, a); });
  }

  auto conv_args = [&](const Edge& e, const float* src, float* dst, const float* wt, int normalise) {
    return ConvArgs{src, dst, wt, B, e.c_in, e.h_in, e.w_in, e.c_out, e.h_out, e.w_out, e.kh, e.kw, e.stride, e.pad, normalise};
  };
  auto gather = [&](const DevGather& d, int k, const float* src, bool scored, bool embed_src, int src_layer) {      // phase A over a conv edge, MFMA
    const long nt = map_tiles(d.g.tm, B);
    // sparse: a 16-node forward gather behind a ReLU layer skips the (zero) rows of that layer's dead nodes
    const bool sparse = (h->gather_sparse & (d.g.lanes == 16 ? 1 : 2)) && !embed_src && src_layer >= 1;
    GArgs a{in->lb[k], in->ub[k], in->mask, src, nb, nt, scored ? 1 : 0, h->R, roff[k], to_dtm(d.g.tm), to_dg(d, h->d_zero),
            EmbedSrc{in->lb[0], in->x_lp, in->ub[0], h->d_pack[PK_EMBED]}, sparse ? in->lb[src_layer] : nullptr, sparse ? in->ub[src_layer] : nullptr};
    const size_t lds = gather_lds_bytes(d, 0) + (sparse ? sparse_tab_bytes(d) : 0);
    long grid = (nt + WAVES_MLP - 1) / WAVES_MLP;
    if (grid > (long)h->n_cu * h->gather_occ) grid = (long)h->n_cu * h->gather_occ;
    lz.run(PC_GATHER, [&] {
      if (d.g.lanes == 16) {
        if (embed_src) hipLaunchKernelGGL((k_gather16<true>), dim3((unsigned)grid), dim3(WG_MLP), lds, st, a);
        else if (sparse) hipLaunchKernelGGL((k_gather16<false, true>), dim3((unsigned)grid), dim3(WG_MLP), lds, st, a);
        else hipLaunchKernelGGL((k_gather16<false>), dim3((unsigned)grid), dim3(WG_MLP), lds, st, a);
      } else if (embed_src) hipLaunchKernelGGL(k_gather<true>, dim3((unsigned)grid), dim3(WG_MLP), lds, st, a);
      else if (sparse) hipLaunchKernelGGL((k_gather<false, true>), dim3((unsigned)grid), dim3(WG_MLP), lds, st, a);
      else hipLaunchKernelGGL(k_gather<false>, dim3((unsigned)grid), dim3(WG_MLP), lds, st, a);
    });
  };
  // phase A: nb <- A_k mu[k-1]
  auto agg_fwd = [&](int k) {
    const Edge& e = h->edges[k];
    if (h->gf[k].ok) { gather(h->gf[k], k, mu(k - 1), false, k == 1 && embed_in_gather && h->proj[0] == L_INP_F_1, k - 1); return; }
    if (e.kind == 0) {
      ConvArgs a = conv_args(e, mu(k - 1), nb, h->dev[k].w_fwd, 0);
      lz.run(PC_CONV_FWD, [&] {
        switch (e.c_out) {
          case 3: launch_conv_fwd<3>(a, st); break;
          case 8: launch_conv_fwd<8>(a, st); break;
          case 16: launch_conv_fwd<16>(a, st); break;
          default: launch_conv_fwd<32>(a, st); break;
        }
      });
    } else {
      const DevEdge& de = h->dev[k];
      if (h->dense_lds && per_sample && de.mt_fwd <= 4) {        // one workgroup per sample, source rows staged in LDS
        DenseLArgs a{de.w_fwd, mu(k - 1), nb, B, e.n_in, e.n_out, de.ld_fwd, de.mt_fwd, de.kpad_fwd};
        lz.run(PC_DENSE_AGG, [&] { hipLaunchKernelGGL(k_dense_fwd_lds, dim3(B), dim3(512), 0, st, a); });
        return;
      }
      DenseArgs a{de.w_fwd, mu(k - 1), nb, h->d_zero, B, e.n_in, e.n_out, de.ld_fwd, de.mt_fwd, de.ksq_fwd};
      const long tiles = (long)B * a.MT;
      lz.run(PC_DENSE_AGG, [&] {
        if (a.K >= 512) hipLaunchKernelGGL(k_dense_agg<true>, dim3((unsigned)tiles), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_dense_agg<false>, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, a);
      });
    }
  };
  // phase A: nb <- A_{k+1}^T mu[k+1]  (k+1 <= L), conv case divided by the tap count when `normalise`
  auto agg_bwd = [&](int k, int normalise, bool scored) {
    const Edge& e = h->edges[k + 1];
    // the input layer (k = 0) aggregates the rows of layer 1 that already went through its 64x64 map (PackPostInp)
    const float* srcb = k == 0 ? rows1_for_input : mu(k + 1);
    if (k >= 1 && h->gb[k + 1].ok) { gather(h->gb[k + 1], k, mu(k + 1), scored, false, k + 1); return; }
    if (e.kind == 0) {
      ConvArgs a = conv_args(e, srcb, nb, h->dev[k + 1].w_bwd, normalise);
      lz.run(PC_CONVT_BWD, [&] {
        switch (e.c_in) {
          case 3: launch_convT<3>(a, st); break;
          case 8: launch_convT<8>(a, st); break;
          case 16: launch_convT<16>(a, st); break;
          default: launch_convT<32>(a, st); break;
        }
      });
    } else {
      const DevEdge& de = h->dev[k + 1];
      if (h->dense_lds && per_sample && de.kpad_bwd <= 128) {    // one workgroup per sample, the whole source layer in LDS
        DenseLArgs a{de.w_bwd, srcb, nb, B, e.n_out, e.n_in, de.ld_bwd, de.mt_bwd, de.kpad_bwd};
        lz.run(PC_DENSE_AGG, [&] { hipLaunchKernelGGL(k_dense_bwd_lds, dim3(B), dim3(512), 0, st, a); });
        return;
      }
      DenseArgs a{de.w_bwd, srcb, nb, h->d_zero, B, e.n_out, e.n_in, de.ld_bwd, de.mt_bwd, de.ksq_bwd};
      const long tiles = (long)B * a.MT;
      lz.run(PC_DENSE_AGG, [&] {
        if (a.K >= 512) hipLaunchKernelGGL(k_dense_agg<true>, dim3((unsigned)tiles), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_dense_agg<false>, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, a);
      });
    }
  };
  // phase B: node MLP over a compacted list of nodes
  // post_input: this is the backward update of layer 1 and an input-layer update follows -- the kernel also applies the input
  // update's 64x64 map to its rows (PackPostInp) and writes them to F1; only inspection runs still need the plain rows
  auto node_update = [&](int k, bool fwd, bool scored, bool post_input = false) {
    const long nt = ((long)B * h->N[k] + 31) / 32;
    // the aggregate in `nb` was built from rows whose last Linear is deferred (gnnb_pack.h), except the one k_prop writes
    const int src_proj = fwd ? h->proj[k - 1] : (k < L ? h->proj[k + 1] : -1);
    int pack = PK_UPD_BWD;
    const float* sarr = nullptr;
    if (fwd) {
      pack = src_proj == L_INP_F_1 ? PK_UPD_FWD_E : (src_proj == L_INP_B2_2 ? PK_UPD_FWD_I : PK_UPD_FWD_F);
      sarr = ws + w.sf[k];
    } else if (k < L) {
      pack = PK_UPD_BWD_B;
      sarr = ws + w.sb[k];
    }
    const bool deferred = sarr != nullptr;
    // normal: list0 = live non-ambiguous nodes (short chain), list1 = ambiguous nodes; restricted: the scored nodes, general chain
    UpdArgs a{h->d_pack[pack], in->lb[k], in->ub[k], nb, ws + (fwd ? w.Pf[k] : w.Pb[k]), (post_input && !debug_full) ? nullptr : mu(k), status,
              ilist(w.live[k]), cnt + 4 * k + (scored ? 3 : 0), ilist(scored ? w.score[k] : w.amb[k]), cnt + 4 * k + (scored ? 2 : 1), sarr,
              post_input ? rows1_for_input : nullptr, nullptr};
    const int wv = h->nu_waves;  // waves per workgroup (one workgroup per CU shares the LDS weights)
    const bool bf3 = h->bf3 && wv == 12;
    a.wp = h->d_pack[PK_POST_INP] + (bf3 ? (h->gb[1].ok ? PackPostInp::WPG3 : PackPostInp::WPN3) : (h->gb[1].ok ? PackPostInp::WPG : PackPostInp::WPN));
    const size_t ldsb = bf3 ? (size_t)(PackUpdL3::FLOATS + (post_input ? 6144 : 0)) * 4 : (size_t)(PackUpd::FLOATS + (post_input ? 4096 : 0)) * 4;
    long grid = (nt + wv - 1) / wv;
    if (grid > h->n_cu) grid = h->n_cu;
    lz.run(PC_NODE_UPDATE, [&] {
      const dim3 g((unsigned)grid), b12(768), b8(512);
      if (bf3) {
        if (post_input && deferred) hipLaunchKernelGGL((k_node_update<12, true, true, true>), g, b12, ldsb, st, a);
        else if (post_input) hipLaunchKernelGGL((k_node_update<12, false, true, true>), g, b12, ldsb, st, a);
        else if (deferred) hipLaunchKernelGGL((k_node_update<12, true, false, true>), g, b12, ldsb, st, a);
        else hipLaunchKernelGGL((k_node_update<12, false, false, true>), g, b12, ldsb, st, a);
      } else if (post_input) {
        if (wv == 12 && deferred) hipLaunchKernelGGL((k_node_update<12, true, true>), g, b12, ldsb, st, a);
        else if (wv == 12) hipLaunchKernelGGL((k_node_update<12, false, true>), g, b12, ldsb, st, a);
        else if (deferred) hipLaunchKernelGGL((k_node_update<8, true, true>), g, b8, ldsb, st, a);
        else hipLaunchKernelGGL((k_node_update<8, false, true>), g, b8, ldsb, st, a);
      } else if (wv == 12 && deferred) hipLaunchKernelGGL((k_node_update<12, true>), g, b12, ldsb, st, a);
      else if (wv == 12) hipLaunchKernelGGL((k_node_update<12, false>), g, b12, ldsb, st, a);
      else if (deferred) hipLaunchKernelGGL((k_node_update<8, true>), g, b8, ldsb, st, a);
      else hipLaunchKernelGGL((k_node_update<8, false>), g, b8, ldsb, st, a);
    });
    h->proj[k] = fwd ? L_FC4_2 : L_BC4_1;
  };
  auto update_input = [&]() {
    h->proj[0] = L_INP_B2_2;
    if (h->gb[1].ok) {
      const DevGather& d = h->gb[1];
      const long nt = map_tiles(d.g.tm, B);
      const bool sparse = (h->gather_sparse & 4) != 0;
      GIArgs a{h->d_pack[PK_PRE_INP], h->d_pack[PK_UPD_INP], in->lb[0], in->ub[0], rows1_for_input, ws + w.sb[0], mu(0), nt, to_dtm(d.g.tm), to_dg(d, h->d_zero),
               in->lb[1], in->ub[1]};
      const size_t lds = gather_lds_bytes(d, PackUpdInp::FLOATS + PackPreInp::FLOATS) + (sparse ? sparse_tab_bytes(d) : 0);
      lz.run(PC_GATHER_INPUT, [&] {
        if (sparse && h->bf3) hipLaunchKernelGGL((k_gather_input_update<true, true>), dim3(mlp_grid(h, nt)), dim3(WG_MLP), lds, st, a);
        else if (sparse) hipLaunchKernelGGL((k_gather_input_update<true, false>), dim3(mlp_grid(h, nt)), dim3(WG_MLP), lds, st, a);
        else if (h->bf3) hipLaunchKernelGGL((k_gather_input_update<false, true>), dim3(mlp_grid(h, nt)), dim3(WG_MLP), lds, st, a);
        else hipLaunchKernelGGL((k_gather_input_update<false, false>), dim3(mlp_grid(h, nt)), dim3(WG_MLP), lds, st, a);
      });
      return;
    }
    agg_bwd(0, 0, false);
    const long G = (long)B * h->N[0], nt = (G + 31) / 32;
    UpdInpArgs a{h->d_pack[PK_UPD_INP], nb, ws + w.Q, ws + w.sb[0], mu(0), G, nt};
    lz.run(PC_INPUT_UPDATE, [&] { hipLaunchKernelGGL(k_input_update, dim3(mlp_grid(h, nt)), dim3(WG_MLP), PackUpdInp::FLOATS * 4, st, a); });
  };

  // the top of the network as one launch per round (k_top); with a half-pass limit (inspection) the separate kernels run
  const bool top_fused = h->use_top && h->top_ok && !debug_full && per_sample;
  auto top = [&]() {
    const Edge& e = h->edges[L];
    const DevEdge& de = h->dev[L];
    TopArgs a{};
    a.df = DenseLArgs{de.w_fwd, mu(L - 1), nullptr, B, e.n_in, e.n_out, de.ld_fwd, de.mt_fwd, de.kpad_fwd};
    a.db = DenseLArgs{de.w_bwd, nullptr, nb, B, e.n_out, e.n_in, de.ld_bwd, de.mt_bwd, de.kpad_bwd};
    a.pack_f = h->d_pack[PK_UPD_FWD_F]; a.pack_b = h->d_pack[PK_UPD_BWD]; a.pack_p = h->d_pack[PK_PROP];
    a.Pf = ws + w.Pf[L]; a.Pb = ws + w.Pb[L]; a.sf = ws + w.sf[L];
    a.lb = in->lb[L]; a.ub = in->ub[L];
    a.lbm = in->lb[L - 1]; a.ubm = in->ub[L - 1];
    a.prop_w = in->prop_w; a.prop_b = in->prop_b; a.lbK = in->lb[K]; a.ubK = in->ub[K]; a.z_out = in->primal[in->n_primal - 1];
    a.mu_prop = mu(K); a.mu = mu(L); a.status = status; a.N = h->N[L];
    lz.run(PC_TOP, [&] { hipLaunchKernelGGL(k_top, dim3(B), dim3(512), TOP_LDS_FLOATS * 4, st, a); });
    h->proj[L] = L_BC4_1;
  };

  int done = 0;
  for (int t = 0; t < h->T && done < limit; ++t) {
    if (top_fused) {
      for (int k = 1; k < L; ++k) {
        agg_fwd(k);
        node_update(k, true, false);
      }
      top();                                     // F1 .. B2: both half-passes of layer L, aggregate of layer L-1 in `nb`
      for (int k = L - 1; k >= 1; --k) {
        const bool scored = h->restrict_last && t == h->T - 1 && k == 1;
        if (k < L - 1) agg_bwd(k, 1, scored);
        node_update(k, false, scored, k == 1 && t < h->T - 1);
      }
      if (t < h->T - 1) update_input();
      done += 2;
      continue;
    }
    // forward sweep (graph_conv.py:107-192) + property node (:194-210)
    for (int k = 1; k <= L; ++k) {
      agg_fwd(k);
      node_update(k, true, false);
    }
    {
      // the backward sweep starts with the edge from the property node: its aggregate is written by the same kernel
      const bool bwd_follows = done + 1 < limit;
      PropArgs a{h->d_pack[PK_PROP], mu(L), in->prop_w, in->prop_b, in->lb[K], in->ub[K], in->primal[in->n_primal - 1], mu(K),
                 bwd_follows ? nb : nullptr, B, h->N[L], in->lb[L], in->ub[L]};
      lz.run(PC_PROP_FWD, [&] { hipLaunchKernelGGL(k_prop, dim3(B), dim3(256), 0, st, a); });
    }
    if (++done >= limit) break;
    // backward sweep (:222-350), Gauss-Seidel order: layer k reads the already-updated mu[k+1]
    for (int k = L; k >= 1; --k) {
      // after the last backward step mu[1] is only read by the score head, i.e. at the scored nodes
      const bool scored = h->restrict_last && !debug_full && t == h->T - 1 && k == 1;
      if (k < L) agg_bwd(k, 1, scored);          // (k == L: k_prop already wrote the aggregate from the property node)
      node_update(k, false, scored, k == 1 && (t < h->T - 1 || debug_full));
    }
    // input layer (:360-385): its last-round result is never read, so it only runs when another round follows
    if (t < h->T - 1 || debug_full) update_input();
    ++done;
  }

  // scores (graph_conv.py:442-450) and decision (graph_score.py:41-47)
  ArgmaxArgs am{scores, decisions, B, h->R, L, {0}};
  {
    ScoreArgs a{};
    a.pack = h->d_pack[h->proj[1] == L_FC4_2 ? PK_SCORE_F : PK_SCORE_B]; a.scores = scores; a.L = L; a.R = h->R; a.cnt = cnt + 4;
    long nt = 0;
    for (int k = 1; k <= L; ++k) {
      const int i = k - 1;
      a.mu[i] = mu(k); a.list[i] = ilist(w.score[k]); a.N[i] = h->N[k]; a.off[i] = roff[k];
      a.lb[i] = in->lb[k]; a.ub[i] = in->ub[k];
      nt += ((long)B * h->N[k] + 31) / 32;
      am.cum[k - 1] = roff[k] + h->N[k];
    }
    lz.run(PC_SCORE, [&] { hipLaunchKernelGGL(k_score, dim3(mlp_grid(h, nt / 4)), dim3(WG_MLP), PackScore::FLOATS * 4, st, a); });
  }
  lz.run(PC_ARGMAX, [&] { hipLaunchKernelGGL(k_argmax, dim3(B), dim3(256), 0, st, am); });
  return lz.rc;
}

// BaBSR scores of a batch (reference plnn/kw_score_conv.py choose_node_conv :41-113; the decision rule :115-156 stays
// on the host).  lb/ub: HOST tables of n_graph DEVICE pointers exactly as in gnnb_batch; prop_w (B, N_L), mask (B, R),
// scores/intercepts (B, R) device.  Stream-ordered, no allocation.
extern "C" int gnnb_babsr(gnnb_t* h, const float* const* lb, const float* const* ub, int n_graph, const float* prop_w,
                          const float* mask, int B, float* scores, float* intercepts, void* stream) {
  if (!h || !lb || !ub || !prop_w || !mask || !scores || !intercepts) return fail(GNNB_E_INVALID, "gnnb_babsr: null argument");
  if (!h->bound) return fail(GNNB_E_STATE, "gnnb_babsr: call gnnb_bind_network first");
  const int K = (int)h->N.size() - 1, L = K - 1;
  if (n_graph != K + 1 || B < 1) return fail(GNNB_E_INVALID, "gnnb_babsr: %d graph layers given, network has %d", n_graph, K + 1);
  BabsrArgs a{};
  a.L = L; a.R = h->R; a.prop_w = prop_w; a.mask = mask; a.scores = scores; a.icp = intercepts;
  int off = 0, maxN = 0;
  for (int k = 1; k <= L; ++k) {
    const int i = k - 1;
    if (!lb[k] || !ub[k]) return fail(GNNB_E_INVALID, "gnnb_babsr: null bounds pointer for graph layer %d", k);
    a.lb[i] = lb[k]; a.ub[i] = ub[k]; a.bias[i] = h->dev[k].bias; a.N[i] = h->N[k]; a.hw[i] = h->hw[k]; a.off[i] = off;
    off += h->N[k];
    maxN = std::max(maxN, h->N[k]);
    if (k < L) {                              // edge k+1 (between graph layers k and k+1)
      const Edge& e = h->edges[k + 1];
      a.ekind[i] = e.kind; a.ew[i] = h->dev[k + 1].w_bwd;
      a.c_in[i] = e.c_in; a.h_in[i] = e.h_in; a.w_in[i] = e.w_in; a.c_out[i] = e.c_out; a.h_out[i] = e.h_out; a.w_out[i] = e.w_out;
      a.kh[i] = e.kh; a.kw[i] = e.kw; a.stride[i] = e.stride; a.pad[i] = e.pad; a.ld[i] = h->dev[k + 1].ld_bwd;
    }
  }
  a.maxN = maxN;
  hipLaunchKernelGGL(k_babsr, dim3(B), dim3(256), (size_t)2 * maxN * sizeof(float), (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(GNNB_E_HIP, "launch of k_babsr failed: %s", hipGetErrorString(e));
  return GNNB_OK;
}


// ================================================================================================================
// Online learning (reference graphnet/graph_score_online.py:9-23, :62-77)
// ================================================================================================================
extern "C" int gnnb_get_weights(const gnnb_t* h, float* w_blob, size_t n_floats) {
  if (!h || !w_blob) return fail(GNNB_E_INVALID, "gnnb_get_weights: null argument");
  if (n_floats != blob_floats()) return fail(GNNB_E_INVALID, "gnnb_get_weights: %zu floats, expected %zu", n_floats, blob_floats());
  memcpy(w_blob, h->blob.data(), n_floats * sizeof(float));
  return GNNB_OK;
}

extern "C" int gnnb_set_weights(gnnb_t* h, const float* w_blob, size_t n_floats) {
  if (!h || !w_blob) return fail(GNNB_E_INVALID, "gnnb_set_weights: null argument");
  if (n_floats != blob_floats()) return fail(GNNB_E_INVALID, "gnnb_set_weights: %zu floats, expected %zu", n_floats, blob_floats());
  HIPCHK(hipDeviceSynchronize());          // no forward may still be reading the packs
  if (int rc = load_weights(h, w_blob, nullptr)) return rc;
  if (h->trainer) HIPCHK(hipMemcpy(h->trainer->d_w, w_blob, n_floats * sizeof(float), hipMemcpyHostToDevice));
  return GNNB_OK;
}

// torch.optim.Adam(model.parameters(), lr, weight_decay) of graph_score_online.py:15; the moments start at zero
extern "C" int gnnb_online_create(gnnb_t* h, float lr, float weight_decay) {
  if (!h) return fail(GNNB_E_INVALID, "gnnb_online_create: null handle");
  free_trainer(h);
  gnnb_train::Trainer* t = new gnnb_train::Trainer();
  h->trainer = t;
  t->lr = lr; t->wd = weight_decay;
  const size_t n = blob_floats();
  for (float** p : {&t->d_w, &t->d_g, &t->d_m, &t->d_v}) {
    HIPCHK(hipMalloc((void**)p, n * sizeof(float)));
    HIPCHK(hipMemset(*p, 0, n * sizeof(float)));
  }
  HIPCHK(hipMemcpy(t->d_w, h->blob.data(), n * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(hipFuncSetAttribute((const void*)gnnb_train::k_tlin_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  if (!(getenv("GNNB_ONLINE_ONE_STREAM") && getenv("GNNB_ONLINE_ONE_STREAM")[0] == '1'))
    HIPCHK(hipStreamCreateWithFlags(&t->side, hipStreamNonBlocking));
  return GNNB_OK;
}

extern "C" int gnnb_online_grad(const gnnb_t* h, float* grad, size_t n_floats) {
  if (!h || !grad || !h->trainer) return fail(GNNB_E_STATE, "gnnb_online_grad: no trainer (gnnb_online_create)");
  if (n_floats != blob_floats()) return fail(GNNB_E_INVALID, "gnnb_online_grad: %zu floats, expected %zu", n_floats, blob_floats());
  HIPCHK(hipMemcpy(grad, h->trainer->d_g, n_floats * sizeof(float), hipMemcpyDeviceToHost));
  return GNNB_OK;
}

// One GraphChoice.online_learning step (graph_score_online.py:62-77) for B subproblems (the reference: B = 1):
//   loss = sum_b ( max_j scores_b[j] - scores_b[kw_b] + improvement_b );  backward;  Adam step;  scorer packs rebuilt.
// in: the batch exactly as for gnnb_forward.  kw_index (HOST, B): the KW decision as a flat index into the R ReLU nodes
// (trans_len[lay-1] + idx, :63-67), which must be an undecided node of the mask.  improvement (HOST, B).  loss (HOST, B,
// may be NULL).  scores_padded (DEVICE (B, R), may be NULL): the scores of the training-form forward BEFORE the update.
// apply = 0: gradient only (gnnb_online_grad), the parameters and the Adam state stay as they are.
extern "C" int gnnb_online_step(gnnb_t* h, const gnnb_batch* in, int B, const int32_t* kw_index, const float* improvement,
                                float* loss, float* scores_padded, int apply, void* stream) {
  using namespace gnnb_train;
  if (!h || !in || !kw_index || !improvement) return fail(GNNB_E_INVALID, "gnnb_online_step: null argument");
  if (!h->bound) return fail(GNNB_E_STATE, "gnnb_online_step: call gnnb_bind_network first");
  if (!h->trainer) return fail(GNNB_E_STATE, "gnnb_online_step: call gnnb_online_create first");
  const int K = (int)h->N.size() - 1, L = K - 1, R = h->R, T = h->T;
  if (B < 1 || in->n_graph != K + 1 || in->n_relu != L || in->n_primal < h->n_fixed)
    return fail(GNNB_E_INVALID, "gnnb_online_step: batch does not match the bound network");
  for (int b = 0; b < B; ++b)
    if (kw_index[b] < 0 || kw_index[b] >= R) return fail(GNNB_E_INVALID, "gnnb_online_step: kw_index[%d] = %d outside [0, %d)", b, kw_index[b], R);
  Trainer& t = *h->trainer;
  hipStream_t st = (hipStream_t)stream;
  t.st = st;
  t.ev_next = 0;
  t.tape.clear();
  if (t.arena.reset(st)) return fail(GNNB_E_HIP, "gnnb_online_step: arena reset failed");
  if (t.edge_w.empty()) {                                  // torch-layout copies of the verified network's weights
    t.edge_w.assign(h->edges.size(), nullptr);
    for (int k = 1; k <= L; ++k)
      if (int rc = upload(&t.edge_w[k], h->edges[k].w.data(), h->edges[k].w.size())) return rc;
  }
  if (t.cap_B < B) {
    for (float** p : {&t.d_scores, &t.d_ds, &t.d_loss, &t.d_imp}) { if (*p) (void)hipFree(*p); *p = nullptr; }
    if (t.d_kw) (void)hipFree(t.d_kw);
    if (t.d_sel) (void)hipFree(t.d_sel);
    HIPCHK(hipMalloc((void**)&t.d_sel, (size_t)B * 8));
    HIPCHK(hipMalloc((void**)&t.d_scores, (size_t)B * R * 4));
    HIPCHK(hipMalloc((void**)&t.d_ds, (size_t)B * R * 4));
    HIPCHK(hipMalloc((void**)&t.d_loss, (size_t)B * 4));
    HIPCHK(hipMalloc((void**)&t.d_imp, (size_t)B * 4));
    HIPCHK(hipMalloc((void**)&t.d_kw, (size_t)B * 4));
    t.cap_B = B;
  }
  HIPCHK(hipMemcpyAsync(t.d_kw, kw_index, (size_t)B * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(t.d_imp, improvement, (size_t)B * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemsetAsync(t.d_ds, 0, (size_t)B * R * 4, st));
  HIPCHK(hipMemsetAsync(t.d_g, 0, blob_floats() * 4, st));

  // ---- per-node constants ----
  struct LC { float *r0, *r1, *amb, *live, *nd2, *d1, *ff, *fb; Trainer::List ambl, livel; };
  std::vector<LC> lc(L + 1);
  for (int k = 1; k <= L; ++k) {
    const long n = (long)B * h->N[k];
    LC& c = lc[k];
    for (float** p : {&c.r0, &c.r1, &c.amb, &c.live, &c.nd2, &c.d1}) *p = t.arena.alloc(n);
    c.ff = t.arena.alloc(7 * n);
    c.fb = t.arena.alloc(7 * n);
    if (t.arena.err) return fail(GNNB_E_NOMEM, "gnnb_online_step: out of device memory");
    const int q = h->relu_q[k];
    TPrepArgs a{in->lb[k], in->ub[k], in->dual[k - 1], in->primal[q - 1], in->primal[q], h->dev[k].bias, h->N[k], h->hw[k], n,
                c.r0, c.r1, c.amb, c.live, c.nd2, c.d1, c.ff, c.fb};
    hipLaunchKernelGGL(k_tprep, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    // node lists: the relaxation chains run over the ambiguous nodes, the update chains over the live ones
    int* buf = reinterpret_cast<int*>(t.arena.alloc(2 * n + 2));
    if (!buf) return fail(GNNB_E_NOMEM, "gnnb_online_step: out of device memory");
    c.ambl = Trainer::List{buf, buf + 2 * n, n};
    c.livel = Trainer::List{buf + n, buf + 2 * n + 1, n};
    hipLaunchKernelGGL(k_tcompact, dim3(1), dim3(256), 0, st, TCompact{c.amb, buf, buf + 2 * n, n});
    hipLaunchKernelGGL(k_tcompact, dim3(1), dim3(256), 0, st, TCompact{c.live, buf + n, buf + 2 * n + 1, n});
  }
  auto cols = [&](std::initializer_list<const float*> cs, long n) {
    TColsArgs a{};
    int w = 0;
    for (const float* c : cs) a.c[w++] = c;
    a.w = w; a.n = n; a.dst = t.arena.alloc((size_t)n * w);
    hipLaunchKernelGGL(k_tcols, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    return (const float*)a.dst;
  };
  const long n0 = (long)B * h->N[0];
  const float* inp3 = cols({in->lb[0], in->x_lp, in->ub[0]}, n0);                       // graph_conv.py:90-93
  const float* inp2 = cols({in->lb[0], in->ub[0]}, n0);                                 // :380-381
  const float* featp = cols({in->lb[K], in->ub[K], in->primal[in->n_primal - 1], in->prop_b}, B);     // :202-205

  auto edge = [&](int k, int dir, int norm, const TT& src) {       // nb = A_k src (dir 0) or A_k^T src (dir 1, / tap count if norm)
    const Edge& e = h->edges[k];
    TT y = t.rows((long)B * (dir == 0 ? h->N[k] : h->N[k - 1]));
    if (e.kind == 0) {
      TConv a{src.v, y.v, t.edge_w[k], B, e.c_in, e.h_in, e.w_in, e.c_out, e.h_out, e.w_out, e.kh, e.kw, e.stride, e.pad, dir, norm, 0};
      hipLaunchKernelGGL(k_tconv, dim3((unsigned)((y.n + 3) / 4)), dim3(256), 0, st, a);
      TConv b = a;
      b.src = y.g; b.dst = src.g; b.dir = 1 - dir; b.acc = 1;
      const long nsrc = src.n;
      t.tape.push_back([b, nsrc, st]() { hipLaunchKernelGGL(k_tconv, dim3((unsigned)((nsrc + 3) / 4)), dim3(256), 0, st, b); });
    } else {
      TDense a{t.edge_w[k], 0, src.v, y.v, B, e.n_out, e.n_in, dir, 0};
      hipLaunchKernelGGL(k_tdense, dim3((unsigned)y.n), dim3(256), 0, st, a);
      TDense b = a;
      b.src = y.g; b.dst = src.g; b.dir = 1 - dir; b.acc = 1;
      const long nsrc = src.n;
      t.tape.push_back([b, nsrc, st]() { hipLaunchKernelGGL(k_tdense, dim3((unsigned)nsrc), dim3(256), 0, st, b); });
    }
    return y;
  };
  auto prop_edge = [&](int dir, const TT& src) {                   // the property layer: one (1, N_L) matrix per sample
    TT y = t.rows(dir == 0 ? (long)B : (long)B * h->N[L]);
    TDense a{in->prop_w, (long)h->N[L], src.v, y.v, B, 1, h->N[L], dir, 0};
    hipLaunchKernelGGL(k_tdense, dim3((unsigned)y.n), dim3(256), 0, st, a);
    TDense b = a;
    b.src = y.g; b.dst = src.g; b.dir = 1 - dir; b.acc = 1;
    const long nsrc = src.n;
    t.tape.push_back([b, nsrc, st]() { hipLaunchKernelGGL(k_tdense, dim3((unsigned)nsrc), dim3(256), 0, st, b); });
    return y;
  };
  auto S = [](const TT& x, const float* s = nullptr) { return Trainer::seg(x, s); };
  auto SF = [](const TT& x, const float* s = nullptr) { return Trainer::seg(x, s, true); };     // a segment addressed by node

  // ---- relaxation terms (graph_conv.py:153-161, :273-293): functions of the node features only, so the same in every round
  // -- computed once (the reference recomputes them per round; their gradient contributions from all rounds add up in
  // relax.g before the chain is walked back once) and only for the ambiguous nodes (`* amb` zeroes every other row)
  std::vector<TT> relax_f(L + 1), relax_b(L + 1);
  for (int k = 1; k <= L; ++k) {
    const LC& c = lc[k];
    const long n = (long)B * h->N[k];
    TT a = t.lin(L_FC1, {}, c.ff, n, true, nullptr, &c.ambl);
    relax_f[k] = t.lin(L_FC1_1, {S(a)}, nullptr, n, false, c.amb, &c.ambl, true);      // :160-161
    TT a1 = t.lin(L_BC1, {}, c.fb, n, true, nullptr, &c.ambl);
    TT a2 = t.lin(L_BC1_1, {S(a1)}, nullptr, n, true, nullptr, &c.ambl);
    TT sb = t.lin(L_BC1_2, {S(a2)}, nullptr, n, false, nullptr, &c.ambl);              // :285
    TT b1 = t.lin(L_BC2, {S(sb), S(sb, c.nd2), S(sb, c.d1)}, nullptr, n, true, nullptr, &c.ambl);    // :287-291
    relax_b[k] = t.lin(L_BC2_1, {S(b1)}, nullptr, n, false, c.amb, &c.ambl, true);     // :293
  }

  // ---- the forward of graph_conv.py:77-388, every Linear on the tape ----
  std::vector<TT> mu(K + 1);
  for (int r = 0; r < T; ++r) {
    if (r == 0) {
      TT a = t.lin(L_INP_F, {}, inp3, n0, true, nullptr);
      mu[0] = t.lin(L_INP_F_1, {S(a)}, nullptr, n0, false, nullptr);                     // :94
    }
    for (int k = 1; k <= L; ++k) {                                                       // :107-192
      const LC& c = lc[k];
      const long n = (long)B * h->N[k];
      TT nb = edge(k, 0, 0, mu[k - 1]);
      // the update chain over the live nodes only (`* live` zeroes the rows of the others, :178)
      TT e1 = t.lin(L_FC3, {SF(nb, c.r0), SF(nb, c.r1)}, nullptr, n, true, nullptr, &c.livel);      // :169-170
      TT e = t.lin(L_FC3_2, {S(e1)}, nullptr, n, false, nullptr, &c.livel);
      TT d = t.lin(L_FC4, {SF(relax_f[k]), S(e)}, nullptr, n, true, nullptr, &c.livel);             // :176-177
      mu[k] = t.lin(L_FC4_2, {S(d)}, nullptr, n, false, c.live, &c.livel, true);                     // :178
    }
    {                                                                                    // :194-210
      TT nb = prop_edge(0, mu[L]);
      TT hh = t.lin(L_OUT1, {}, featp, B, true, nullptr);
      TT o = t.lin(L_OUT2, {S(hh), S(nb)}, nullptr, B, true, nullptr);
      mu[K] = t.lin(L_OUT3, {S(o)}, nullptr, B, false, nullptr);
    }
    for (int k = L; k >= 1; --k) {                                                       // :222-350
      const LC& c = lc[k];
      const long n = (long)B * h->N[k];
      TT nb = k == L ? prop_edge(1, mu[K]) : edge(k + 1, 1, h->edges[k + 1].kind == 0 ? 1 : 0, mu[k + 1]);   // :299-326
      TT e1 = t.lin(L_BC3, {SF(nb, c.r0), SF(nb, c.r1)}, nullptr, n, true, nullptr, &c.livel);      // :331-336
      TT e = t.lin(L_BC3_1, {S(e1)}, nullptr, n, false, nullptr, &c.livel);
      TT d = t.lin(L_BC4, {SF(relax_b[k]), S(e)}, nullptr, n, true, nullptr, &c.livel);             // :344-345
      mu[k] = t.lin(L_BC4_1, {S(d)}, nullptr, n, false, c.live, &c.livel, true);                     // :347
    }
    if (r + 1 < T) {                                                                     // :360-385 (the last round's input rows feed nothing)
      TT nb = edge(1, 1, 0, mu[1]);
      TT a = t.lin(L_INP_B, {}, inp2, n0, true, nullptr);
      TT relax = t.lin(L_INP_B_1, {S(a)}, nullptr, n0, false, nullptr);
      TT c2 = t.lin(L_INP_B2, {S(relax), S(nb)}, nullptr, n0, true, nullptr);
      mu[0] = t.lin(L_INP_B2_2, {S(c2)}, nullptr, n0, false, nullptr);
    }
  }
  // ---- scores (:442-450) and the loss ----
  int off = 0;
  for (int k = 1; k <= L; ++k) {
    const long n = (long)B * h->N[k];
    TT hk = t.lin(L_FNODE, {S(mu[k])}, nullptr, n, true, nullptr);
    TScore a{hk.v, hk.g, t.d_w + weight_offset(L_FSCORE), t.d_w + bias_offset(L_FSCORE), in->mask, t.d_scores, t.d_ds, h->N[k], R, off, n,
             t.d_g + weight_offset(L_FSCORE), t.d_g + bias_offset(L_FSCORE), t.d_sel, B};
    hipLaunchKernelGGL(k_tscore_fwd, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, a);
    t.tape.push_back([a, n, st]() {
      hipLaunchKernelGGL(k_tscore_bwd, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, a);
      hipLaunchKernelGGL(k_tscore_bwd_w, dim3(1), dim3(64), 0, st, a);
    });
    off += h->N[k];
  }
  if (t.arena.err) return fail(GNNB_E_NOMEM, "gnnb_online_step: out of device memory");
  if (scores_padded) HIPCHK(hipMemcpyAsync(scores_padded, t.d_scores, (size_t)B * R * 4, hipMemcpyDeviceToDevice, st));
  TLoss la{t.d_scores, t.d_ds, t.d_kw, t.d_imp, t.d_loss, R, t.d_sel};
  hipLaunchKernelGGL(k_tloss, dim3(B), dim3(256), 0, st, la);
  // ---- backward: the tape in reverse ----
  for (auto it = t.tape.rbegin(); it != t.tape.rend(); ++it) (*it)();
  t.tape.clear();
  if (t.join()) return fail(GNNB_E_HIP, "gnnb_online_step: joining the weight-gradient stream failed");
  if (t.arena.err) return fail(GNNB_E_NOMEM, "gnnb_online_step: out of device memory");
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(GNNB_E_HIP, "gnnb_online_step: a launch failed: %s", hipGetErrorString(e));
  if (loss) {
    t.h_loss.resize(B);
    HIPCHK(hipMemcpyAsync(t.h_loss.data(), t.d_loss, (size_t)B * 4, hipMemcpyDeviceToHost, st));
  }
  if (apply) {
    t.step += 1;
    const double b1 = 0.9, b2 = 0.999;
    const double bc1 = 1.0 - std::pow(b1, t.step), bc2 = 1.0 - std::pow(b2, t.step);
    TAdam a{t.d_w, t.d_g, t.d_m, t.d_v, (int)blob_floats(), (float)(t.lr / bc1), t.wd, (float)b1, (float)b2, 1e-8f, (float)std::sqrt(bc2)};
    hipLaunchKernelGGL(k_tadam, dim3((unsigned)((blob_floats() + 255) / 256)), dim3(256), 0, st, a);
    std::vector<float> nw(blob_floats());
    HIPCHK(hipMemcpyAsync(nw.data(), t.d_w, nw.size() * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (int rc = load_weights(h, nw.data(), st)) return rc;      // the scorer's folded packs follow the new parameters
  } else {
    HIPCHK(hipStreamSynchronize(st));
  }
  if (loss) memcpy(loss, t.h_loss.data(), (size_t)B * 4);
  return GNNB_OK;
}
