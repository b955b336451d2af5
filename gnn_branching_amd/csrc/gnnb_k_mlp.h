// gnnb_k_mlp.h -- part of libgnnb.so, included by gnnb.hip (one translation unit; see its header comment).
// setup and node-MLP kernels: k_embed, k_classify, k_pre, k_pre_inp, k_node_update, k_input_update.
#pragma once

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
        o[c] = relu_max(fmaf(u[r], w[c][2], fmaf(x[r], w[c][1], fmaf(l[r], w[c][0], bias[c]))));
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
  int zero[MAXL];                  // 1: some consumer of this layer's rows reads the rows of dead nodes too -> they are zeroed here
  int* live[MAXL]; int* amb[MAXL]; int* score[MAXL];
  float* livef[MAXL];              // (B*N_k) 1.0 / 0.0: [r0 != 0], read by k_livesum
  long G[MAXL];
  int N[MAXL], off[MAXL], blk0[MAXL + 1];   // first workgroup of each layer
  const float* mask;
  float* scores;                   // (B, R)
  int* cnt;                        // 4 ints per layer: plain (live, not ambiguous), ambiguous, scored, 0 -- ZERO on entry (gnnb_handle::d_ctl)
  int R;
  // the words a forward starts from zero and nothing touches before the kernels that use them: the status word, k_score's
  // decision keys and its finished-workgroup counter, k_top's arrival counters (this is the first kernel of a forward)
  int32_t* status; unsigned long long* best; int* done; int B; int* topflag; int nflag;
};

#define CLS_THREADS 1024
#ifndef CLS_NPT
#define CLS_NPT 2             // nodes per thread: a block's load -> ballot -> atomic -> write chain is ~5 us of latency whatever its
                              // size, and with one node per thread the 800 blocks of base B=256 ran as two rounds of it
                              // (base / wide, us: 1 node 15.1 / 24.5, 2 nodes 11.3 / 17.5, 4: 11.8 / 19.6, 8: 17.5 / 18.5)
#endif
#define CLS_BLOCK (CLS_THREADS * CLS_NPT)
// one global atomic per list and workgroup (a single counter word only sustains ~90 atomics/us)
// amb_local != nullptr (k_classify_pre): the block's ambiguous nodes are also listed there (LDS), in block order; returns their
// number and, in *layer, the block's layer.
__device__ __forceinline__ int classify_block(const ClassifyArgs& a, int* amb_local, int* layer) {
  __shared__ int wcnt[3][CLS_NPT][CLS_THREADS / 64];
  __shared__ int wbase[3][CLS_NPT][CLS_THREADS / 64];
  __shared__ int amb_total;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {
    const long gid = (long)blockIdx.x * CLS_THREADS + threadIdx.x, nthr = (long)gridDim.x * CLS_THREADS;
    for (long i = gid; i < a.B; i += nthr) a.best[i] = 0ull;
    for (long i = gid; i < a.nflag; i += nthr) a.topflag[i] = 0;
    if (gid == 0) { *a.status = 0; *a.done = 0; }
  }
  int k = 0;
  while (k + 1 < a.L && (int)blockIdx.x >= a.blk0[k + 1]) ++k;
  const long G = a.G[k];
  const int N = a.N[k];
  const long g0 = (long)(blockIdx.x - a.blk0[k]) * CLS_BLOCK + threadIdx.x;
  bool flag[CLS_NPT][3], live[CLS_NPT], valid[CLS_NPT];
  unsigned long long bal[CLS_NPT][3];
#pragma unroll
  for (int i = 0; i < CLS_NPT; ++i) {
    const long g = g0 + (long)i * CLS_THREADS;
    valid[i] = g < G;
    const long gc = valid[i] ? g : G - 1;
    const Ratio r = compute_ratio(a.lb[k][gc], a.ub[k][gc]);
    const long b = gc / N;
    const long sidx = b * a.R + a.off[k] + (gc - b * N);
    live[i] = valid[i] && r.live != 0.0f;
    flag[i][1] = valid[i] && r.amb != 0.0f;                 // ambiguous (a subset of live)
    flag[i][0] = live[i] && !flag[i][1];                    // live with r0 == r1: the cheap update path
    flag[i][2] = valid[i] && a.mask[sidx] != 0.0f;
    if (valid[i]) {
      a.scores[sidx] = -INFINITY;
      a.livef[k][g] = live[i] ? 1.0f : 0.0f;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      bal[i][c] = __ballot(flag[i][c]);
      if (lane == 0) wcnt[c][i][wave] = __popcll(bal[i][c]);
    }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int c = threadIdx.x;
    int total = 0;
    for (int i = 0; i < CLS_NPT; ++i)
      for (int w = 0; w < CLS_THREADS / 64; ++w) { wbase[c][i][w] = total; total += wcnt[c][i][w]; }
    const int base = total ? atomicAdd(a.cnt + 4 * k + c, total) : 0;
    if (c == 1) amb_total = total;
    if (amb_local && c == 1) {                             // block-local positions first, then the global ones
      for (int i = 0; i < CLS_NPT; ++i)
        for (int w = 0; w < CLS_THREADS / 64; ++w) wcnt[c][i][w] = wbase[c][i][w];
    }
    for (int i = 0; i < CLS_NPT; ++i)
      for (int w = 0; w < CLS_THREADS / 64; ++w) wbase[c][i][w] += base;
  }
  __syncthreads();
  int* lists[3] = {a.live[k], a.amb[k], a.score[k]};
  float* mu = a.mu[k];
  float* mu2 = k == 0 ? a.mu2 : nullptr;
#pragma unroll
  for (int i = 0; i < CLS_NPT; ++i) {
    const long g = g0 + (long)i * CLS_THREADS;
#pragma unroll
    for (int c = 0; c < 3; ++c)
      if (flag[i][c]) lists[c][wbase[c][i][wave] + __popcll(bal[i][c] & ((1ull << lane) - 1ull))] = (int)g;
    if (amb_local && flag[i][1]) amb_local[wcnt[1][i][wave] + __popcll(bal[i][1] & ((1ull << lane) - 1ull))] = (int)g;
    unsigned long long dead = a.zero[k] ? __ballot(valid[i] && !live[i]) : 0ull;
    while (dead) {                       // the whole wave zeroes one dead row per iteration (coalesced 256 B)
      const int l = __ffsll((long long)dead) - 1;
      dead &= dead - 1;
      const long row = g - lane + l;
      mu[row * 64 + lane] = 0.0f;
      if (mu2) mu2[row * 64 + lane] = 0.0f;
    }
  }
  if (layer) *layer = k;
  return amb_total;
}

__global__ __launch_bounds__(CLS_THREADS) void k_classify(ClassifyArgs a) { (void)classify_block(a, nullptr, nullptr); }

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
// BF3: the 64x64 blocks (W2 forward; W2, W3, W5 backward) on the bf16 matrix rate with three-piece operands (LDS images
// PackPreFwdL3 / PackPreBwdL3); the 192-wide W4 and the feature layers stay on the fp32 MFMA
// (k_pre's 16 waves sit at the 128-register step with the sequential blocks: the pipelined form spills 16 registers there)
#ifndef PRE_PIPE
#define PRE_PIPE false
#endif
template <bool BF3>
__device__ __forceinline__ void pre_tile(const PreAllArgs& a, const float* lds, const float* lds_b, int k, bool bwd, const int* list, int count,
                                         long t, int lane) {
  constexpr int F_W1 = BF3 ? (int)PackPreFwdL3::W1 : (int)PackPreFwd::W1, F_B1 = BF3 ? (int)PackPreFwdL3::B1 : (int)PackPreFwd::B1;
  constexpr int F_B2 = BF3 ? (int)PackPreFwdL3::B2 : (int)PackPreFwd::B2;
  constexpr int B_W1 = BF3 ? (int)PackPreBwdL3::W1 : (int)PackPreBwd::W1, B_B1 = BF3 ? (int)PackPreBwdL3::B1 : (int)PackPreBwd::B1;
  constexpr int B_B2 = BF3 ? (int)PackPreBwdL3::B2 : (int)PackPreBwd::B2, B_B3 = BF3 ? (int)PackPreBwdL3::B3 : (int)PackPreBwd::B3;
  constexpr int B_W4 = (int)PackPreBwd::W4, B_B4 = BF3 ? (int)PackPreBwdL3::B4 : (int)PackPreBwd::B4;
  constexpr int B_B5 = BF3 ? (int)PackPreBwdL3::B5 : (int)PackPreBwd::B5;
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
      frag_bias(H, lds + F_B1, h);
      gemm_small<4>(lds + F_W1, lane, H, x);
      frag_relu(H);
      Frag Pf;                                   // fc1_1 and the first half of fc4 are one folded 64x64 map
      frag_bias(Pf, lds + F_B2, h);
      if (BF3) gemm_w64_bf3<1, PRE_PIPE>(lds + PackPreFwdL3::W23, lane, Pf, [&](int s) { return FRAG_AT(H, s); });
      else gemm_w64<32>(lds + PackPreFwd::W2, lane, Pf, [&](int s) { return FRAG_AT(H, s); });
      if (valid) frag_store_rows(Pf, a.Pf[k], gc, h);
    } else {
      // feat7' = [l, u, beta, -d2+d1, z_post, z_pre, c]
      x[0] = h ? ub : lb;
      x[1] = h ? (-d2 + d1) : r.beta;
      x[2] = h ? zpre : zpost;
      x[3] = h ? 0.0f : c;
      Frag H1;
      frag_bias(H1, lds_b + B_B1, h);
      gemm_small<4>(lds_b + B_W1, lane, H1, x);
      frag_relu(H1);
      Frag H2;
      frag_bias(H2, lds_b + B_B2, h);
      if (BF3) gemm_w64_bf3<1, PRE_PIPE>(lds_b + PackPreBwdL3::W23, lane, H2, [&](int s) { return FRAG_AT(H1, s); });
      else gemm_w64<32>(lds_b + PackPreBwd::W2, lane, H2, [&](int s) { return FRAG_AT(H1, s); });
      frag_relu(H2);
      Frag S;
      frag_bias(S, lds_b + B_B3, h);
      if (BF3) gemm_w64_bf3<1, PRE_PIPE>(lds_b + PackPreBwdL3::W33, lane, S, [&](int s) { return FRAG_AT(H2, s); });
      else gemm_w64<32>(lds_b + PackPreBwd::W3, lane, S, [&](int s) { return FRAG_AT(H2, s); });
      // bc2 on [s, s*(-d2), s*d1]  (:287-291)
      const float nd2 = -d2;
      Frag H4;
      frag_bias(H4, lds_b + B_B4, h);
      auto in4 = [&](int s) {
        const float v = FRAG_AT(S, s & 31);
        return s < 32 ? v : (s < 64 ? v * nd2 : v * d1);
      };
      if (BF3) gemm_w64_bf3<3, PRE_PIPE>(lds_b + PackPreBwdL3::W43, lane, H4, in4);
      else gemm_w64<96>(lds_b + B_W4, lane, H4, in4);
      frag_relu(H4);
      Frag Pb;                                   // bc2_1 and the first half of bc4 are one folded 64x64 map
      frag_bias(Pb, lds_b + B_B5, h);
      if (BF3) gemm_w64_bf3<1, PRE_PIPE>(lds_b + PackPreBwdL3::W53, lane, Pb, [&](int s) { return FRAG_AT(H4, s); });
      else gemm_w64<32>(lds_b + PackPreBwd::W5, lane, Pb, [&](int s) { return FRAG_AT(H4, s); });
      if (valid) frag_store_rows(Pb, a.Pb[k], gc, h);
    }
  }
}

#ifndef PRE_WAVES
#define PRE_WAVES 16      // the weights (150 KB) allow one workgroup per CU: 16 waves of <= 128 registers fill it
#endif
template <bool BF3>
__global__ __launch_bounds__(PRE_WAVES * 64) void k_pre(PreAllArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long nhalf = 0;
  for (int k = 0; k < a.L; ++k) nhalf += (long)((a.cnt[4 * k + 1] + 31) / 32);
  // tile t of one direction: which layer (wave-uniform)
  auto run = [&](long t, bool bwd, const float* lf, const float* lb) {
    int k = 0, count = 0;
    for (; k < a.L; ++k) {
      count = a.cnt[4 * k + 1];
      const long tk = (count + 31) / 32;
      if (t < tk) break;
      t -= tk;
    }
    pre_tile<BF3>(a, lf, lb, k, bwd, a.list[k], count, t, lane);
  };
  if (BF3) {
    // forward image + everything of the backward image behind its head; forward tiles; then the head; backward tiles
    copy_to_lds(lds + PackPreFwdL3::W1, a.pack_f + PackPreFwd::W1, 512 + 64);
    copy_to_lds(lds + PackPreFwdL3::B2, a.pack_f + PackPreFwd::B2, 64);
    copy_to_lds(lds + PackPreFwdL3::W23, a.pack_f + PackPreFwd::W23, 6144);
    if (a.do_bwd) {
      copy_to_lds(lds + PackPreBwdL3::B3, a.pack_b + PackPreBwd::B3, 64);
      copy_to_lds(lds + PackPreBwdL3::B4, a.pack_b + PackPreBwd::B4, 64);
      copy_to_lds(lds + PackPreBwdL3::B5, a.pack_b + PackPreBwd::B5, 64);
      copy_to_lds(lds + PackPreBwdL3::W33, a.pack_b + PackPreBwd::W33, 6144);
      copy_to_lds(lds + PackPreBwdL3::W43, a.pack_b + PackPreBwd::W43, 18432);
      copy_to_lds(lds + PackPreBwdL3::W53, a.pack_b + PackPreBwd::W53, 6144);
    }
    __syncthreads();
    for (long t = (long)wave * gridDim.x + blockIdx.x; t < nhalf; t += (long)gridDim.x * PRE_WAVES) run(t, false, lds, lds);
    if (!a.do_bwd) return;
    __syncthreads();
    copy_to_lds(lds + PackPreBwdL3::W1, a.pack_b + PackPreBwd::W1, 512 + 64);                      // W1, B1
    copy_to_lds(lds + PackPreBwdL3::B2, a.pack_b + PackPreBwd::B2, 64);
    copy_to_lds(lds + PackPreBwdL3::W23, a.pack_b + PackPreBwd::W23, 6144);
    __syncthreads();
    for (long t = (long)wave * gridDim.x + blockIdx.x; t < nhalf; t += (long)gridDim.x * PRE_WAVES) run(t, true, lds, lds);
    return;
  }
  float* lds_b = lds + (int)PackPreFwd::FLOATS;
  copy_to_lds(lds_b, a.pack_b, PackPreBwd::FLOATS);
  stage_pack(lds, a.pack_f, PackPreFwd::FLOATS);
  const long ntiles = nhalf * (a.do_bwd ? 2 : 1);
  for (long tile = (long)wave * gridDim.x + blockIdx.x; tile < ntiles; tile += (long)gridDim.x * PRE_WAVES) {
    const bool bwd = a.do_bwd && tile < nhalf;
    run((a.do_bwd && !bwd) ? tile - nhalf : tile, bwd, lds, lds_b);
  }
}

// k_classify_pre: small batches -- k_classify and k_pre in ONE launch.  A block classifies its nodes (classify_block, the code
// of k_classify) and then runs the hoisted feature chains of ITS OWN ambiguous nodes (kept in an LDS list) with its 16 waves; P'
// rows are addressed by node id, so who computes them does not matter.  One launch less for the BaB loop's own call (B = 1: 27.5 us
// against 7.6 + 22.1); from B = 2 on k_pre's even tile dealing wins (a block's share of the ambiguous nodes varies: B = 8 42.5 us
// against 31.8), so the host only uses it for a single subproblem (GNNB_CLSPRE_MAX_B).
// Dynamic LDS: the bf16 x 3 image of the feature chains (PackPreBwdL3) + CLS_BLOCK ints.
static_assert(CLS_THREADS == PRE_WAVES * 64, "k_classify_pre: one block = the 16 waves of k_pre");
#define CLSPRE_LDS_BYTES ((size_t)(PackPreBwdL3::FLOATS + CLS_BLOCK) * 4)
static_assert(CLSPRE_LDS_BYTES + 2048 <= 160 * 1024, "k_classify_pre: image + list + the classification's static LDS must fit one CU");
__global__ __launch_bounds__(CLS_THREADS) void k_classify_pre(ClassifyArgs ca, PreAllArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int* amb_local = reinterpret_cast<int*>(lds + PackPreBwdL3::FLOATS);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the forward image and the tail of the backward image go out first: their loads fly under the classification
  copy_to_lds(lds + PackPreFwdL3::W1, a.pack_f + PackPreFwd::W1, 512 + 64);
  copy_to_lds(lds + PackPreFwdL3::B2, a.pack_f + PackPreFwd::B2, 64);
  copy_to_lds(lds + PackPreFwdL3::W23, a.pack_f + PackPreFwd::W23, 6144);
  if (a.do_bwd) {
    copy_to_lds(lds + PackPreBwdL3::B3, a.pack_b + PackPreBwd::B3, 64);
    copy_to_lds(lds + PackPreBwdL3::B4, a.pack_b + PackPreBwd::B4, 64);
    copy_to_lds(lds + PackPreBwdL3::B5, a.pack_b + PackPreBwd::B5, 64);
    copy_to_lds(lds + PackPreBwdL3::W33, a.pack_b + PackPreBwd::W33, 6144);
    copy_to_lds(lds + PackPreBwdL3::W43, a.pack_b + PackPreBwd::W43, 18432);
    copy_to_lds(lds + PackPreBwdL3::W53, a.pack_b + PackPreBwd::W53, 6144);
  }
  int k_ = 0;
  const int count = __builtin_amdgcn_readfirstlane(classify_block(ca, amb_local, &k_));      // (block-uniform: scalar registers, not a spilled vector pair)
  const int k = __builtin_amdgcn_readfirstlane(k_);
  __syncthreads();                                   // the block's list and the images are in LDS
  const long nt = (count + 31) / 32;
  for (long t = wave; t < nt; t += PRE_WAVES) pre_tile<true>(a, lds, lds, k, false, amb_local, count, t, lane);
  if (!a.do_bwd) return;
  __syncthreads();
  copy_to_lds(lds + PackPreBwdL3::W1, a.pack_b + PackPreBwd::W1, 512 + 64);                      // W1, B1
  copy_to_lds(lds + PackPreBwdL3::B2, a.pack_b + PackPreBwd::B2, 64);
  copy_to_lds(lds + PackPreBwdL3::W23, a.pack_b + PackPreBwd::W23, 6144);
  __syncthreads();
  for (long t = wave; t < nt; t += PRE_WAVES) pre_tile<true>(a, lds, lds, k, true, amb_local, count, t, lane);
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
  int smod;                 // > 0: sarr is one (N) table shared by all samples (source layer all live: the input layer), index g % smod
  // POST (layer 1, backward, an input-layer update follows): the consumer's 64x64 map inp_b2[:, 64:].bc4_1.W is applied here,
  // on the ~3x fewer producer nodes: F = WP.E goes to `post` (rows by node id), and `mu` may be null (nothing else reads E)
  float* post;
  const float* wp;          // PackPostInp block (WPN or WPG), staged behind the update pack
  void* post3;              // POST: F is written as three bf16 pieces (rows3: the input update's aggregate runs on the bf16 matrix rate) instead of `post`
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
// (gemm_w64_bf3; LDS image PackUpdL3), and so does the general chain: its 128-wide first layer as WAS.(r0 x) + Wa[:, 64:].((r1 - r0) x).
// STAGE = false: the caller (k_gather_update) has the LDS image in place already
template <bool DEFERRED, bool POST = false, bool BF3 = false, bool STAGE = true>
__device__ __forceinline__ void node_update_loop(const UpdArgs& a, float* lds, int c0, int c1, long tile, long stride, int lane) {
  constexpr int O_WA = (int)PackUpd::WA, O_BA = BF3 ? (int)PackUpdL3::BA : (int)PackUpd::BA;
  constexpr int O_BCB = BF3 ? (int)PackUpdL3::BCB : (int)PackUpd::BCB, O_VAW = BF3 ? (int)PackUpdL3::VAW : (int)PackUpd::VAW;
  constexpr int O_END = BF3 ? (int)PackUpdL3::FLOATS : (int)PackUpd::FLOATS;
  constexpr bool NU_PIPE = (GEMM_BF3_PIPE != 0) && !(DEFERRED && POST);      // (the pipelined blocks spill two registers in the DEFERRED + POST instantiation)
  const int h = lane >> 5, j = lane & 31;
  // the general tiles (1.5-3x the work of a short-chain tile) come FIRST in the tile order, so they are never a SIMD's tail
  const long n1 = (c1 + 31) / 32, n0 = (c0 + 31) / 32, ntiles = n0 + n1;
  const float* bias_row = a.pack + PackUpd::BCBROW;
  long gc = 0, gc_n = 0;
  bool valid = false, valid_n = false;
  float lb = 0.0f, ub = 0.0f, lb_n = 0.0f, ub_n = 0.0f, sw = 0.0f, sw_n = 0.0f;
  Frag X;
  constexpr bool deferred = DEFERRED;            // the aggregate is built from rows with a deferred projection (gnnb_pack.h)
  // The inputs of the next tile are fetched under this tile's chain in two stages: the list entry and the per-node scalars at
  // the top (a handful of registers), the 32-register aggregate row only once the first GEMM has consumed X -- it lands in X
  // itself, so no second fragment is alive across the chain (a separate prefetch fragment put every bf16x3 instantiation
  // 2-14 VGPRs over the 168 of three waves per SIMD: scratch spills) and the two dependent round trips (list -> row) are
  // split over the two stages.
  auto fetch_scalars = [&](long tl, long& g_, bool& v_, float& l_, float& u_, float& s_) {
    const bool k0 = tl >= n1;
    const long idx = (k0 ? tl - n1 : tl) * 32 + j;
    v_ = idx < (k0 ? c0 : c1);
    g_ = (k0 ? a.list0 : a.list1)[v_ ? idx : 0];
    l_ = a.lb[g_];
    u_ = a.ub[g_];
    if (deferred) s_ = a.sarr[a.smod > 0 ? g_ % a.smod : g_];
  };
  if (tile < ntiles) {
    fetch_scalars(tile, gc, valid, lb, ub, sw);
    frag_load_rows(X, a.nb, gc, h);
  }
  constexpr bool post = POST;
  if (STAGE) {
    if (post) copy_to_lds(lds + O_END, a.wp, BF3 ? 6144 : 4096);
    if (BF3) {
      copy_to_lds(lds + PackUpdL3::BA, a.pack + PackUpd::BA, 64);
      copy_to_lds(lds + PackUpdL3::BCB, a.pack + PackUpd::BCB, 64 + 64 + 128);          // BCB, BCBROW, VAW
      stage_pack(lds + PackUpdL3::WAS3, a.pack + PackUpd::WAS3, 3 * 6144);               // WAS3, WCB3, WA1S3
    } else stage_pack(lds, a.pack, PackUpd::FLOATS);
  }
  if (tile >= ntiles) return;
  FT_DECL;
  for (;;) {
    FT_MARK(0);                                // loop overhead
    const Ratio r = compute_ratio(lb, ub);
    const bool kind0 = tile >= n1;             // wave-uniform
    const long next = tile + stride;
    const bool has_next = next < ntiles;
    if (has_next) fetch_scalars(next, gc_n, valid_n, lb_n, ub_n, sw_n);
    Frag H, H2;
    frag_bias(H, lds + O_BA, h);
    if (kind0) {
      const float r0 = r.r0;
      if (deferred) {                        // + s.(r0 Wa0.bp + r1 Wa1.bp), r0 == r1: one small k-step
        const float x[1] = {r0 * sw};
        gemm_small<1>(lds + O_VAW, lane, H, x);
      }
      if (BF3) gemm_w64_bf3<1, NU_PIPE>(lds + PackUpdL3::WAS3, lane, H, [&](int s) { return FRAG_AT(X, s) * r0; });
      else gemm_w64<32>(lds + PackUpd::WAS, lane, H, [&](int s) { return FRAG_AT(X, s) * r0; });
      frag_bias(H2, lds + O_BCB, h);
    } else {
      // nodes without a relaxation term (amb = 0) read the bias row instead of their (never written) P' row
      frag_load_rowptr(H2, r.amb != 0.0f ? a.P + gc * 64 : bias_row, h);
      const float r0 = r.r0, r1 = r.r1;
      if (deferred) {
        const float x[1] = {(h ? r1 : r0) * sw};
        gemm_small<1>(lds + O_VAW, lane, H, x);
      }
      if (BF3) {                             // Wa.[r0 x, r1 x] = WAS.(r0 x) + Wa[:, 64:].((r1 - r0) x), both blocks bf16 x 3 (PackUpdL3)
        const float dr = r1 - r0;
        gemm_w64_bf3<1, NU_PIPE>(lds + PackUpdL3::WAS3, lane, H, [&](int s) { return FRAG_AT(X, s) * r0; });
        gemm_w64_bf3<1, NU_PIPE>(lds + PackUpdL3::WA1S3, lane, H, [&](int s) { return FRAG_AT(X, s) * dr; });
      } else gemm_w64<64>(lds + O_WA, lane, H, [&](int s) { return FRAG_AT(X, s & 31) * (s < 32 ? r0 : r1); });
    }
    FT_MARK(1);                                // wait for this tile's rows + first GEMM
    __builtin_amdgcn_sched_barrier(0);         // X is dead from here: the next tile's row loads go into it, under the second GEMM
    if (has_next) frag_load_rows(X, a.nb, gc_n, h);
    __builtin_amdgcn_sched_barrier(0);
    FT_MARK(2);                                // wait for the next tile's list entry / scalars + issue of its row loads
    frag_relu(H);
    if (BF3) gemm_w64_bf3<1, NU_PIPE>(lds + PackUpdL3::WCB3, lane, H2, [&](int s) { return FRAG_AT(H, s); });
    else gemm_w64<32>(lds + PackUpd::WCB, lane, H2, [&](int s) { return FRAG_AT(H, s); });
    frag_relu(H2);
    FT_MARK(3);                                // second GEMM
    if (r.live == 0.0f) {                      // a dead node's row is zero whatever its (possibly never written) aggregate held
#pragma unroll
      for (int R = 0; R < 32; ++R) FRAG_AT(H2, R) = 0.0f;
    }
    // (row stores through an LDS image, whole 256-B rows per instruction instead of 32-B pieces: measured, no gain here)
    if (valid && frag_has_nan(H2)) atomicOr(a.status, 1);      // a NaN here is a NaN in mu = Wd.E + bd (:184-186, :339-341)
    if (valid && a.mu) frag_store_rows(H2, a.mu, gc, h);
    if (post) {
#pragma unroll
      for (int R = 0; R < 32; ++R) FRAG_AT(H, R) = 0.0f;
      if (BF3) gemm_w64_bf3<1, NU_PIPE>(lds + O_END, lane, H, [&](int s) { return FRAG_AT(H2, s); });
      else gemm_w64<32>(lds + O_END, lane, H, [&](int s) { return FRAG_AT(H2, s); });
      if (valid) {
        if (a.post3) frag_store_rows3(H, a.post3, gc, h);
        else frag_store_rows(H, a.post, gc, h);
      }
    }
    FT_MARK(4);                                // stores (+ POST block)
    if (!has_next) break;
    tile = next; gc = gc_n; valid = valid_n; lb = lb_n; ub = ub_n; sw = sw_n;
  }
#ifdef FUSED_TIMING
  if (FUSED_TIMING == 3 && !POST && STAGE) FT_FLUSH();
#endif
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
