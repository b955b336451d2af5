// gnnb_k_edges.h -- part of libgnnb.so, included by gnnb.hip (one translation unit; see its header comment).
// the other edges and the top of the network: VALU conv kernels, dense edges (per tile / per sample out of LDS), k_prop, k_top.
#pragma once

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
// s_ret (only with klist, i.e. when exactly the live source rows are walked): the wave that stores row tile mt also returns, in
// lane j, s[row mt*32 + j] = sum over the walked rows of W[row][k] = sum_k W[row][k] live_k -- the bias sum of this edge
// (gnnb_pack.h "deferred projection"); needs 256 more floats of scratch behind the 16384 of the tiles
__device__ __forceinline__ void dense_fwd_sample(const DenseLArgs& a, int b, float* scratch, Store store, const int* klist = nullptr,
                                                 int K_eff = 0, float* s_ret = nullptr) {
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
  float sacc = 0.0f;
  auto mma = [&](const float (&A)[16], int buf) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const float2 bv = *reinterpret_cast<const float2*>(&xs[buf][kh][2 * u + h][2 * j]);
      if (s_ret) sacc += A[u];               // (list padding points at a zero row of At)
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
  float (*sred)[64] = reinterpret_cast<float (*)[64]>(scratch + 2 * 2 * 32 * 64 + 4 * 32 * 64);
  if (s_ret) sacc += __shfl_xor(sacc, 32);
  if (kh == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { red[mt][r][lane] = acc0[r]; red[mt][16 + r][lane] = acc1[r]; }
    if (s_ret) sred[mt][lane] = sacc;
  }
  __syncthreads();
  if (s_ret && kh == 0) *s_ret = sacc + sred[mt][lane];
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
// sout (only with klist): also writes sout[row] = sum over the walked k of W[k][row] = the bias sum of the transposed edge
__device__ __forceinline__ void dense_bwd_sample(const DenseLArgs& a, const float* xs_raw, Store store, const int* rlist = nullptr,
                                                 int n_rows = 0, const int* klist = nullptr, int K_eff = 0, float* sout = nullptr) {
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
    float sacc = 0.0f;
    auto mma = [&](const float (&A)[DL_CH], int c) {
#pragma unroll
      for (int u = 0; u < DL_CH; ++u) {
        const float2 bv = *reinterpret_cast<const float2*>(&xs[klist ? klist[c * 16 + 2 * u + h] : c * 16 + 2 * u + h][2 * j]);
        if (sout) sacc += A[u];
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
    if (sout) {
      sacc += __shfl_xor(sacc, 32);
      if (h == 0 && mt * 32 + j < M) sout[arow] = sacc;
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

// ---- the two Linear edges of k_top on the bf16 matrix rate ----------------------------------------------------------------------
// Both operands in three bf16 pieces, six products per k-step (gemm_w64_bf3's arithmetic: fp32-grade sums, 2.7x fewer matrix-pipe
// cycles than v_mfma_f32_32x32x2_f32).  The weights stay fp32 in memory and are split in registers next to the MFMAs (a
// pre-split image would double the bytes every sample pulls out of L2).  v_mfma_f32_32x32x16_bf16 operands: lane (i = l & 31,
// kg = l >> 5) holds k = 8 kg .. 8 kg + 7 of row / column i; WHICH source rows a k-step's 16 slots stand for is free, so a lane
// simply takes 8 consecutive entries of the live-row list.
struct Split3 { unsigned u1, u2, u3; };
__device__ __forceinline__ Split3 split3(float a, float b) {
  Split3 r;
  split_pair_bf3(a, b, r.u1, r.u2, r.u3);
  return r;
}
__device__ __forceinline__ void split3_to(u32x4 (&d)[3], int q, float a, float b) {
  const Split3 r = split3(a, b);
  d[0][q] = r.u1; d[1][q] = r.u2; d[2][q] = r.u3;
}
__device__ __forceinline__ f32x16 mfma6(const u32x4 (&w)[3], const u32x4 (&x)[3], f32x16 acc) {
  const bf16x8 w1 = __builtin_bit_cast(bf16x8, w[0]), w2 = __builtin_bit_cast(bf16x8, w[1]), w3 = __builtin_bit_cast(bf16x8, w[2]);
  const bf16x8 x1 = __builtin_bit_cast(bf16x8, x[0]), x2 = __builtin_bit_cast(bf16x8, x[1]), x3 = __builtin_bit_cast(bf16x8, x[2]);
  acc = mfma_bf16(w3, x1, acc);
  acc = mfma_bf16(w2, x2, acc);
  acc = mfma_bf16(w1, x3, acc);
  acc = mfma_bf16(w2, x1, acc);
  acc = mfma_bf16(w1, x2, acc);
  return mfma_bf16(w1, x1, acc);
}
// the same six products in the same order with the operand roles exchanged: acc = (x pieces as rows) . (w pieces as columns), i.e. the
// TRANSPOSE of mfma6's tile, sum for sum (dense_bwd_sample_bf3<FUSE>: the tile then is a node-update fragment)
__device__ __forceinline__ f32x16 mfma6t(const u32x4 (&w)[3], const u32x4 (&x)[3], f32x16 acc) {
  const bf16x8 w1 = __builtin_bit_cast(bf16x8, w[0]), w2 = __builtin_bit_cast(bf16x8, w[1]), w3 = __builtin_bit_cast(bf16x8, w[2]);
  const bf16x8 x1 = __builtin_bit_cast(bf16x8, x[0]), x2 = __builtin_bit_cast(bf16x8, x[1]), x3 = __builtin_bit_cast(bf16x8, x[2]);
  acc = mfma_bf16(x1, w3, acc);
  acc = mfma_bf16(x2, w2, acc);
  acc = mfma_bf16(x3, w1, acc);
  acc = mfma_bf16(x1, w2, acc);
  acc = mfma_bf16(x2, w1, acc);
  return mfma_bf16(x1, w1, acc);
}
typedef unsigned short top_idx_t;      // live-row lists of k_top: 16-bit (layers up to 65535 nodes), so that they fit beside the weights

// forward edge (long K) of sample b: out[m][c] = sum_k W[m][k] X[k][c], m < 128 (a.ldA <= 128 columns of At), c < 64.
// 8 waves split K: wave w takes k-steps (16 list entries each) w, w + 8, ... for ALL rows and channels (8 accumulator tiles), so
// every X row and every weight is loaded and split by exactly one wave -- no staging, no barrier in the loop.
// Row tile t holds rows 4 i + t (i < 32) and channel tile nt channels 2 j + nt: one 16-B load per lane and list entry brings the
// weights of all four row tiles (a whole 512-B row of At per half-wave; as 4-B loads -- 32 instead of 8 per k-step -- the
// loop was bound by the address rate of the texture path, 23 us of loads for 460 KB), one 8-B load both channel tiles of X.
// The 8 partial sums meet in LDS in wave order (4 phases of 2 tiles, `scratch`: 16384 floats; every wave adds up a quarter tile).
// klist (LDS, or null: all a.K rows): the K_eff live source rows, padded with a.Kpad (a zero row of At) up to K_eff + 1.
// s_fin (LDS, 128 floats) != null: s_fin[m] = sum over the walked rows of W[m][k] (the bias sum of this edge); spart: 256 floats.
// store(row, channel, value).  Requires 512 threads.
// TS (4 / 2 / 1): row tiles this workgroup computes, t0 .. t0 + TS - 1 (k_top_split: a sample's rows spread over 4 / TS workgroups by
// OUTPUT tile, every sum in the order of the unsplit kernel -- the results do not depend on the split); the weight load narrows
// with it (16 / 8 / 4 bytes per lane and list entry).
template <int TS, class Store1>
__device__ __forceinline__ void dense_fwd_sample_bf3(const DenseLArgs& a, int b, float* scratch, float* spart, Store1 store,
                                                     const top_idx_t* klist, int K_eff, float* s_fin, int t0 = 0) {
  const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int Kw = klist ? K_eff : a.K;
  const int nsteps = (Kw + 15) >> 4;
  // buffer loads: a row's address is one 32-bit multiply, and a slot past the end of the list (row Kpad >= K) is out of range
  // of this sample's X -> reads zero, while At has a zero row there
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)(a.X + (long)b * a.K * 64), 0, a.K * 256, 0x00020000);
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.At, 0, 0x7fffffff, 0x00020000);
  const unsigned ldb = (unsigned)a.ldA * 4u;
  const unsigned acol = (4 * j < a.ldA ? 16u * (unsigned)j : 0u) + 4u * (unsigned)t0;       // (rows 4 j + t >= ldA do not exist: any finite weights will do)
  f32x16 acc[TS][2];
#pragma unroll
  for (int t = 0; t < TS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[t][0][r] = 0.0f; acc[t][1][r] = 0.0f; }
  float sacc[TS];
#pragma unroll
  for (int t = 0; t < TS; ++t) sacc[t] = 0.0f;
  // One set of raw registers: they are free as soon as they are split, and the loads of the wave's next k-step go out right
  // there, under this step's MFMAs (8 accumulator tiles leave no room for a second buffer).
  float2 x[8];
  float A[8][TS];
  auto rowof = [&](int s, int q) {
    const int i = s * 16 + 8 * h + q;
    const bool v = i < Kw;
    return (unsigned)(klist ? (int)klist[v ? i : Kw] : (v ? i : a.Kpad));      // (list entry Kw is padding = Kpad)
  };
  auto issueX = [&](int s) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const auto v = __builtin_amdgcn_raw_buffer_load_b64(rX, rowof(s, q) * 256u + 8u * (unsigned)j, 0, 0);
      x[q] = make_float2(__uint_as_float(v[0]), __uint_as_float(v[1]));
    }
  };
  auto issueA = [&](int s) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const unsigned off = rowof(s, q) * ldb + acol;
      if constexpr (TS == 4) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rA, off, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) A[q][t] = __uint_as_float(v[t]);
      } else if constexpr (TS == 2) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(rA, off, 0, 0);
        A[q][0] = __uint_as_float(v[0]); A[q][1] = __uint_as_float(v[1]);
      } else {
        A[q][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rA, off, 0, 0));
      }
    }
  };
  auto splitA = [&](u32x4 (&wa)[3], int t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      split3_to(wa, q, A[2 * q][t], A[2 * q + 1][t]);
      sacc[t] += A[2 * q][t] + A[2 * q + 1][t];
    }
  };
  int s = wave;
  if (s < nsteps) {
    issueX(s);
    issueA(s);
  }
  for (; s < nsteps; s += 8) {
    u32x4 xb0[3], xb1[3];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      split3_to(xb0, q, x[2 * q].x, x[2 * q + 1].x);
      split3_to(xb1, q, x[2 * q].y, x[2 * q + 1].y);
    }
    __builtin_amdgcn_sched_barrier(0);
    issueX(s + 8);                             // (past the end: padding entries -> zeros)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (TS == 4) {
      {
        u32x4 wa[3];
        splitA(wa, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc[0][0] = mfma6(wa, xb0, acc[0][0]);
        acc[0][1] = mfma6(wa, xb1, acc[0][1]);
        __builtin_amdgcn_sched_barrier(0);
        splitA(wa, 1);
        __builtin_amdgcn_sched_barrier(0);
        acc[1][0] = mfma6(wa, xb0, acc[1][0]);
        acc[1][1] = mfma6(wa, xb1, acc[1][1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      {
        u32x4 wa[3], wb[3];
        splitA(wa, 2);
        splitA(wb, 3);
        __builtin_amdgcn_sched_barrier(0);
        issueA(s + 8);                           // all four tiles' weights are split: their registers take the next step's
        __builtin_amdgcn_sched_barrier(0);
        acc[2][0] = mfma6(wa, xb0, acc[2][0]);
        acc[2][1] = mfma6(wa, xb1, acc[2][1]);
        acc[3][0] = mfma6(wb, xb0, acc[3][0]);
        acc[3][1] = mfma6(wb, xb1, acc[3][1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      u32x4 wa[TS][3];
#pragma unroll
      for (int t = 0; t < TS; ++t) splitA(wa[t], t);
      __builtin_amdgcn_sched_barrier(0);
      issueA(s + 8);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < TS; ++t) {
        acc[t][0] = mfma6(wa[t], xb0, acc[t][0]);
        acc[t][1] = mfma6(wa[t], xb1, acc[t][1]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- the 8 partial sums of every tile meet in LDS, two tiles per phase; every wave adds up a quarter of a tile in wave order
  // (phase p: row tile t0 + p % TS ... the pair (row tile, channel tile 0 / 1): the same two tiles, the same order, for every TS)
  float (*sc)[2][16][64] = reinterpret_cast<float (*)[2][16][64]>(scratch);
#pragma unroll
  for (int p = 0; p < TS; ++p) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[wave][u][r][lane] = acc[p][u][r];
    }
    if (s_fin) {
      const float sp = sacc[p] + __shfl_xor(sacc[p], 32);
      if (h == 0) spart[wave * 32 + j] = sp;
    }
    __syncthreads();
    {
      const int u = wave & 1, rq = wave >> 1, t = t0 + p, nt = u;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        float tot = sc[0][u][4 * rq + rr][lane];
#pragma unroll
        for (int w8 = 1; w8 < 8; ++w8) tot += sc[w8][u][4 * rq + rr][lane];
        const int row = 4 * (rr + 8 * rq + 4 * h) + t;
        if (row < a.M) store(row, 2 * j + nt, tot);
      }
      if (s_fin && wave == 0 && h == 0) {
        float tot = spart[j];
#pragma unroll
        for (int w8 = 1; w8 < 8; ++w8) tot += spart[w8 * 32 + j];
        s_fin[4 * j + t] = tot;
      }
    }
    __syncthreads();
  }
}

// transposed edge (short K <= 128) of one sample: out[row][c] = sum_k W[row][k] C[k][c], the rows of C (layer L) in LDS (`Cr`,
// fp32, rows >= K zero).  First the rows of C are split into three bf16 pieces and laid down as the B operands of every k-step
// (`img`: 12288 floats; one ds_read_b128 per operand and lane afterwards); then the waves walk the row tiles.
// rlist / n_rows (LDS): only these output rows (the live nodes of the layer below); null: every row.
// The weights come k-contiguous out of the FORWARD edge's image Wk[row][k] (ldK floats per row): a lane's 8 k of a k-step are 32
// contiguous bytes of its row, two 16-B loads.  Every k < a.Kpad is walked (the rows of dead layer-L nodes are zero in C);
// livek[k] (LDS, 0 / 1) says which k count for the bias sums `sout` (see dense_bwd_sample; null: not wanted).
// ALL k-steps of a row tile are requested at once and the NEXT tile's weights are requested before this tile's MFMAs start (two
// register sets in ping-pong, out-of-range offsets instead of control flow around the loads): with the loads only three k-steps
// ahead -- a k-step is 0.15 us of work, an L2 round trip under load ~1 us -- the edge spent 4 of its 5 us per tile waiting
// (base B = 256: 19.7 -> see DESIGN 5.6).  part / nparts: this workgroup takes row tiles part, part + nparts, ... (k_top_split).
// FUSE (k_top with the node update of layer L-1 inside): a row tile is computed TRANSPOSED (operand roles exchanged, channel tile nt =
// channels 32 nt .. 32 nt + 31) so that its accumulators ARE the node-update fragment of its 32 nodes (lane (j, h): node j, register
// 4 q + c = channel 8 q + 4 h + c) -- every sum has the terms and the order of the plain form, so the aggregate is bit-identical to
// the rows the plain form stores.  node(arow, in_range) is called when a tile starts (it requests what the node's chain needs from
// memory), epi(arow, in_range, X, s, what node() returned) when its MFMAs are done; mid() runs once behind the barrier that ends the
// image build (Cr is dead from there on).  `store` is unused.
struct NoHook { __device__ void operator()() const {} };
// k_top keeps the rows of layer L in LDS as C[row][64]: a row is 64 banks, and the update phases put node 4 j + t on lane j, so the 16
// lanes of a ds_read_b128 group all hit the same four banks (F2 + B1: 1.1 M conflict cycles per launch, profiles/r04_k_top_lds_conflicts_by_phase.txt).
// The column index of row n is therefore XOR-ed with 4 ((n >> 2) & 15): 16-byte pieces stay whole, the 16 lanes of a group spread over
// all 64 banks.  Every access to C goes through csw().
__device__ __forceinline__ int csw(int row) { return ((row >> 2) & 15) << 2; }
template <bool FUSE = false, class Store, class Mid = NoHook, class Node = NoHook, class Epi = NoHook>
__device__ __forceinline__ void dense_bwd_sample_bf3(const DenseLArgs& a, const float* Cr, float* img, Store store, const top_idx_t* rlist,
                                                     int n_rows, float* sout, const float* Wk, int ldK, const float* livek,
                                                     int part = 0, int nparts = 1, Mid mid = Mid(), Node node = Node(), Epi epi = Epi()) {
  constexpr int NST = 8;                           // k-steps of 16: Kpad <= 128 (bind: kpad_bwd <= 128)
  const int lane = threadIdx.x & 63, h = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nst = a.Kpad / 16;
  const int M = rlist ? n_rows : a.M, MT = (M + 31) / 32;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)Wk, 0, 0x7fffffff, 0x00020000);
  struct Raw { f32x4 v[NST][2]; };
  // the weights of row tile mt (all k-steps); a tile past the end reads nothing (out-of-range offsets return zeros)
  auto arow_of = [&](int mt) { return rlist ? (int)rlist[mt * 32 + j < M ? mt * 32 + j : 0] : mt * 32 + j; };
  auto load_tile = [&](Raw& R, int mt) {
    const bool in = mt < MT;
    const unsigned acol = (unsigned)arow_of(in ? mt : 0) * (unsigned)ldK * 4u + 32u * (unsigned)h;
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      const unsigned off = (in && st < nst) ? acol + 64u * (unsigned)st : 0x80000000u;
      const auto v0 = __builtin_amdgcn_raw_buffer_load_b128(rA, off, 0, 0), v1 = __builtin_amdgcn_raw_buffer_load_b128(rA, off, 16, 0);
      R.v[st][0] = f32x4{__uint_as_float(v0[0]), __uint_as_float(v0[1]), __uint_as_float(v0[2]), __uint_as_float(v0[3])};
      R.v[st][1] = f32x4{__uint_as_float(v1[0]), __uint_as_float(v1[1]), __uint_as_float(v1[2]), __uint_as_float(v1[3])};
    }
  };
  const int first = part + nparts * wave, stride = nparts * 8;
  Raw R0, R1;
  load_tile(R0, first);                            // (in flight under the image build)
  unsigned* im = reinterpret_cast<unsigned*>(img);
  for (int e = threadIdx.x; e < nst * 256; e += 512) {
    const int q = e & 3, n = (e >> 2) & 31, kg = (e >> 7) & 1, st = e >> 8;
    const int k0 = st * 16 + kg * 8 + 2 * q;
    float2 v0, v1;                                   // .x / .y: channel tile 0 / 1 of column n (plain: channels 2 n, 2 n + 1; FUSE: n, 32 + n)
    if (FUSE) {
      const int s0 = csw(k0), s1 = csw(k0 + 1);
      v0 = make_float2(Cr[k0 * 64 + (n ^ s0)], Cr[k0 * 64 + ((32 + n) ^ s0)]);
      v1 = make_float2(Cr[(k0 + 1) * 64 + (n ^ s1)], Cr[(k0 + 1) * 64 + ((32 + n) ^ s1)]);
    } else {
      v0 = *reinterpret_cast<const float2*>(Cr + k0 * 64 + ((2 * n) ^ csw(k0)));
      v1 = *reinterpret_cast<const float2*>(Cr + (k0 + 1) * 64 + ((2 * n) ^ csw(k0 + 1)));
    }
    const Split3 sx = split3(v0.x, v1.x), sy = split3(v0.y, v1.y);
    const unsigned u[2][3] = {{sx.u1, sx.u2, sx.u3}, {sy.u1, sy.u2, sy.u3}};
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int p = 0; p < 3; ++p) im[((((st * 2 + nt) * 3 + p) * 64) + kg * 32 + n) * 4 + q] = u[nt][p];
  }
  __syncthreads();
  if constexpr (FUSE) mid();
  const u32x4* im4 = reinterpret_cast<const u32x4*>(img) + lane;
  auto compute = [&](const Raw& R, int mt) {
    [[maybe_unused]] const int arow_t = arow_of(mt);
    [[maybe_unused]] const bool in_t = mt * 32 + j < M;
    [[maybe_unused]] auto nd = [&]() { if constexpr (FUSE) return node(arow_t, in_t); else return 0; }();
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    float sacc = 0.0f;
    auto split = [&](u32x4 (&w)[3], int st) {          // the weights of k-step st -> three bf16 pieces (+ the bias sum)
      float lk[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) lk[q] = 1.0f;
      if (sout && livek) {
        const f32x4* l4 = reinterpret_cast<const f32x4*>(livek + 16 * st + 8 * h);
        const f32x4 u0 = l4[0], u1 = l4[1];
#pragma unroll
        for (int q = 0; q < 4; ++q) { lk[q] = u0[q]; lk[4 + q] = u1[q]; }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float w0 = R.v[st][q >> 1][2 * (q & 1)], w1 = R.v[st][q >> 1][2 * (q & 1) + 1];
        split3_to(w, q, w0, w1);
        if (sout) sacc += w0 * lk[2 * q] + w1 * lk[2 * q + 1];
      }
    };
    u32x4 w[2][3];
    split(w[0], 0);
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      if (st < nst) {                                  // (wave-uniform)
        __builtin_amdgcn_sched_barrier(0);
        const u32x4 xa[3] = {im4[((st * 2 + 0) * 3 + 0) * 64], im4[((st * 2 + 0) * 3 + 1) * 64], im4[((st * 2 + 0) * 3 + 2) * 64]};
        const u32x4 xb[3] = {im4[((st * 2 + 1) * 3 + 0) * 64], im4[((st * 2 + 1) * 3 + 1) * 64], im4[((st * 2 + 1) * 3 + 2) * 64]};
        if (st + 1 < NST) split(w[(st + 1) & 1], st + 1);      // next k-step's pieces under this step's MFMAs (5 vector instructions per MFMA)
        if (FUSE) {
          acc0 = mfma6t(w[st & 1], xa, acc0);
          acc1 = mfma6t(w[st & 1], xb, acc1);
        } else {
          acc0 = mfma6(w[st & 1], xa, acc0);
          acc1 = mfma6(w[st & 1], xb, acc1);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);       // the six operand reads first
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);     // five vector instructions under it
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const int arow = arow_of(mt);
    if constexpr (!FUSE) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < M) store(rlist ? (int)rlist[row] : row, j, make_float2(acc0[r], acc1[r]));
      }
    }
    if (sout) {
      sacc += __shfl_xor(sacc, 32);
      if (h == 0 && mt * 32 + j < M) sout[arow] = sacc;
    }
    if constexpr (FUSE) {
      Frag X;
      X.t[0] = acc0; X.t[1] = acc1;
      epi(arow, in_t, X, sacc, nd);
    }
  };
  for (int mt = first; mt < MT; mt += 2 * stride) {
    load_tile(R1, mt + stride);
    __builtin_amdgcn_sched_barrier(0);
    compute(R0, mt);
    __builtin_amdgcn_sched_barrier(0);
    if (mt + stride >= MT) break;
    load_tile(R0, mt + 2 * stride);
    __builtin_amdgcn_sched_barrier(0);
    compute(R1, mt + stride);
    __builtin_amdgcn_sched_barrier(0);
  }
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
// leave LDS.  LDS map (floats): W = A + Bp (29568) = what the running phase needs: F1's reduction scratch (and, in Bp, a live-row
// list too long for the tail), then the bf16 x 3 image of the forward / backward node-update pack (PackUpdL3), then B2's operand
// image; C = the rows of layer L (DENSE_BWD_ROWS x 64, rows >= N zero), sm = small vectors, the lists.
// ------------------------------------------------------------------------------------------
struct TopArgs {
  DenseLArgs df;            // forward edge L (out unused)
  DenseLArgs db;            // transposed edge L (X unused: the rows come from LDS), out = aggregate rows of layer L-1
  const float *pack_f, *pack_b, *pack_p;
  const float *Pf, *Pb;     // cached P' rows of layer L (by node id), forward / backward
  const float* sf;          // bias-sum scalars of the forward edge (B, N); null: computed here (live-row walk of F1)
  float* sb_out;            // (B, K) bias-sum scalars of the transposed edge written by B2 (its live-row walk), or null (k_livesum has them)
  const float *lb, *ub;     // bounds of layer L, flat (B*N)
  const float *lbm, *ubm;   // bounds of layer L-1, flat (B*K): its dead rows (all zero) are skipped by the forward edge
  const float *prop_w, *prop_b, *lbK, *ubK, *z_out;
  float* mu_prop;           // (B, 64)
  float* mu;                // (B, N, 64) rows of layer L (backward-produced)
  int* status;
  int N;
  // k_top_split (S > 1 workgroups per sample): exchange buffer (B, 8, 64) for the property node's partial sums, one arrival
  // counter per sample (zeroed by k_classify) and the count it stood at when this launch started (2 S per earlier launch)
  float* xbuf; int* xflag; int xbase;
  // fuse_um != 0: the backward node update of layer L-1 runs on B2's row tiles in here (its aggregate never reaches memory, the
  // k_node_update launch behind k_top is gone); um = that update's arguments (list / nb fields unused)
  UpdArgs um; int fuse_um;
};
#define TOP_A_FLOATS (PackUpd::FLOATS > DENSE_FWD_LDS_FLOATS ? PackUpd::FLOATS : DENSE_FWD_LDS_FLOATS)
#define TOP_FIXED_FLOATS (TOP_A_FLOATS + PackProp::FLOATS + DENSE_BWD_ROWS * 64 + 8 * 64 + 128 + 64 + 64)
#define TOP_LDS_FLOATS 40960                   // all 160 KB: what the fixed regions leave holds the live-row lists
#define TOP_K2_INTS (128 + 48)                  // spare
#define TOP_LIST_INTS (TOP_LDS_FLOATS - TOP_FIXED_FLOATS - TOP_K2_INTS)
// the live-row list of layer L-1 (16-bit entries): behind the fixed regions when it fits (then the transposed edge walks only
// live rows too), else in the region PackProp takes after F1, else the edge walks every row
#define TOP_LIST_KEEP_OK(K) ((K) + 96 <= 2 * (int)TOP_LIST_INTS && (K) < 65535)
#define TOP_LIST_OK(K) (TOP_LIST_KEEP_OK(K) || ((K) + 96 <= 2 * (int)PackProp::FLOATS && (K) < 65535))
#define TOP_POLL_CAP (1 << 20)
static_assert(TOP_A_FLOATS >= 16384 && TOP_A_FLOATS >= 12288, "k_top: F1's reduction scratch and B2's operand image live in region A");
static_assert(TOP_A_FLOATS + PackProp::FLOATS >= PackUpdL3::FLOATS, "k_top: the bf16 x 3 node-update image spans regions A and Bp");
static_assert(DENSE_BWD_ROWS >= 128 + 16, "k_top: rows 128.. of C are the property node's scratch");
// fuse_um: B2's operand image takes the first TOP_IMG_FLOATS of region A; the node-update image of layer L-1 goes behind it, over
// the rest of A, region Bp and the first rows of C (dead once the operand image is built); the live-row list must be the kept one
#define TOP_IMG_FLOATS 12288
static_assert(TOP_IMG_FLOATS + PackUpdL3::FLOATS <= TOP_A_FLOATS + PackProp::FLOATS + DENSE_BWD_ROWS * 64, "k_top: the L-1 update image fits behind B2's operand image");

// One sample's top of the network on workgroup `part` of S = 4 / TS (TS row tiles of the last ReLU layer per workgroup; S = 1: the
// whole sample, no exchange).  Layer L's node n sits in row tile n % 4 (lane n / 4 of wave n % 4 in the update phases, rows
// 4 i + t of F1's tile t), so a workgroup owns whole tiles through F1, F2, B1; what crosses workgroups goes through global memory.
// Ordering contract of a hand-off (no fences: the XCDs' L2s are not coherent with each other and a fence per arrival cost 40-50 us at
// B = 128): EVERY store of the handed-off bytes is a write-through `sc1` store, every storing wave drains them (`s_waitcnt vmcnt(0)`)
// before the workgroup barrier behind which ONE lane adds to the sample's arrival counter (agent-scope atomic); the consumer polls that
// counter with `sc1` loads, joins a workgroup barrier, and EVERY load of the handed-off bytes is an `sc1` load to registers
// (MI355X_MICROARCH.md, "Valid forms": the `sc1` row of the hand-off table).  tests/test_gpu_parity.py::test_top_workgroup_split_is_bit_identical
// runs S = 1 / 2 / 4 on a NaN-poisoned exchange buffer and is the guard of this contract.  What crosses:
//   after F2: the 8 per-wave partial sums of the property node's aggregate (wave w sums rows w, w + 8, ...: all of one tile) --
//             every workgroup then evaluates the property node itself, from the same 8 vectors in the same order;
//   after B1: the rows of layer L (they go to a.mu anyway) -- every workgroup loads the rows it does not own and computes its
//             share of B2's row tiles.
// Every sum keeps the order of the unsplit kernel, so the results do not depend on S (tests: bit-identical for S = 1, 2, 4).
// The S workgroups of a sample must be resident together: the host only splits when B x S workgroups fit the chip at one per CU,
// and every wait has an iteration cap that raises status bit 1 instead of hanging.  A caller that shares the GPU with long kernels of
// other streams or processes (so that the partner workgroups may be dispatched late) should set GNNB_TOP_SPLIT=1.
// needs 512 threads and TOP_LDS_FLOATS of LDS
template <int TS>
__device__ __forceinline__ void top_sample(const TopArgs& a, const int b, const int part, float* lds) {
  constexpr int S = 4 / TS;
  const int t0 = part * TS;
  float* A = lds;
  float* Bp = A + TOP_A_FLOATS;
  float* Cr = Bp + PackProp::FLOATS;
  float* part_ = Cr + DENSE_BWD_ROWS * 64;    // [8][64]
  float* xs = part_ + 8 * 64;                 // [128]
  float* outv = xs + 128;                     // [64]
  float* spart = outv + 64;                   // [8]
  float* part2 = Cr + 128 * 64;               // [4][64] + [8][64]: the property node's partial sums (rows >= 128 of C are never read)
  float* part3 = part2 + 4 * 64;
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = a.N;
  FT_DECL;
  for (int i = tid; i < DENSE_BWD_ROWS * 16; i += 512) reinterpret_cast<f32x4*>(Cr)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  // live source rows of layer L-1 (the dead ones are all zero: ~45 % of the forward edge's k-steps), compacted in node order.
  // The list goes behind the fixed regions when it fits there (then the transposed edge B2 also uses it, to compute only the
  // live rows of layer L-1), else into the region PackProp takes after F1.
  int* tail = reinterpret_cast<int*>(spart + 8) + TOP_K2_INTS;
  const bool keep = TOP_LIST_KEEP_OK(a.df.K);
  top_idx_t* klist = reinterpret_cast<top_idx_t*>(keep ? tail : reinterpret_cast<int*>(Bp));
  int K_eff = 0;
  const bool compact = TOP_LIST_OK(a.df.K);
  if (compact) {
    int* wc = reinterpret_cast<int*>(part_);             // per-wave counts
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
      if (live) klist[K_eff + before + __popcll(bal & ((1ull << lane) - 1ull))] = (top_idx_t)n;
      K_eff += total;
      __syncthreads();
    }
    for (int i = K_eff + tid; i < (K_eff + 63) / 64 * 64 + 32; i += 512) klist[i] = (top_idx_t)a.df.Kpad;     // a zero row of At
  }
  __syncthreads();
  FT_MARK(0);        // zero C, live-row list of layer L-1

#if defined(TOP_STOP) && TOP_STOP == 1     // dev, timing only (TOP_STOP = 1 / 2 / 3: leave before F1 / after F1 / before B2)
  if (a.N > 0) return;
#endif
  // ---- F1: this workgroup's rows of C <- W_L . mu_{L-1}
  const bool own_s = compact && !a.sf;                    // the bias sums of the forward edge come out of F1's own walk (-> xs)
#if defined(TOP_ABL) && (TOP_ABL & 2)     // dev, timing only: no F1
  if (K_eff < 0)
#endif
  dense_fwd_sample_bf3<TS>(a.df, b, A, part_, [&](int row, int ch, float v) { Cr[row * 64 + (ch ^ csw(row))] = v; }, compact ? klist : nullptr, K_eff,
                           own_s ? xs : nullptr, t0);
#if defined(TOP_STOP) && TOP_STOP == 2
  if (a.N > 0) { if (Cr[tid] == 12345.0f) a.mu_prop[0] = 1.0f; return; }      // (dev stop: any read of C keeps F1 alive)
#endif
  FT_MARK(1);        // F1 dense forward edge

  // per-lane node of the update phases: wave t < 4 = row tile t, lane j = node 4 j + t
  const int n = 4 * j + wave;
  const bool upd_wave = wave < 4 && wave >= t0 && wave < t0 + TS && wave < N;
  const bool valid = upd_wave && n < N;
  const long g = (long)b * N + (valid ? n : 0);
  const float* pw = a.prop_w + (long)b * N;
  const float s_own = (own_s && valid) ? xs[n] : 0.0f;      // (xs is rewritten in F3, behind a barrier)
  Ratio r{};
  if (upd_wave) r = compute_ratio(a.lb[g], a.ub[g]);
  auto load_row = [&](Frag& x_, int row) {       // fragment <- row of C in LDS (columns swizzled: csw)
    const float* p = Cr + row * 64;
    const int sw = csw(row);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p + ((8 * q + 4 * h) ^ sw));
#pragma unroll
      for (int c = 0; c < 4; ++c) FRAG_AT(x_, 4 * q + c) = v[c];
    }
  };
  auto store_row = [&](const Frag& x_, float* base, int sw = 0) {      // sw: csw(row) for a row of C, 0 for a row in memory
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      f32x4 v;
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = FRAG_AT(x_, 4 * q + c);
      *reinterpret_cast<f32x4*>(base + ((8 * q + 4 * h) ^ sw)) = v;
    }
  };
  // general folded node chain on fragment X, the arithmetic of k_node_update / k_gather_update_q (bf16 x 3 blocks, LDS image
  // PackUpdL3 in W: Wa.[r0 x, r1 x] = WAS.(r0 x) + Wa[:, 64:].((r1 - r0) x)); `sx`: small k-step input of a deferred projection.
  auto chain = [&](const Frag& X, const float* Prow, bool deferred, float sx, Frag& H2) {
    Frag H;
    frag_load_rowptr(H2, Prow, h);             // P' row (global): requested before the first block, wanted after it
    frag_bias(H, A + PackUpdL3::BA, h);
    if (deferred) {
      const float x1[1] = {sx};
      gemm_small<1>(A + PackUpdL3::VAW, lane, H, x1);
    }
    const float r0 = r.r0, dr = r.r1 - r.r0;
    gemm_w64_bf3<1>(A + PackUpdL3::WAS3, lane, H, [&](int s) { return FRAG_AT(X, s) * r0; });
    if (__any(r.amb != 0.0f)) gemm_w64_bf3<1>(A + PackUpdL3::WA1S3, lane, H, [&](int s) { return FRAG_AT(X, s) * dr; });
    frag_relu(H);
    gemm_w64_bf3<1>(A + PackUpdL3::WCB3, lane, H2, [&](int s) { return FRAG_AT(H, s); });
    frag_relu(H2);
    frag_scale(H2, r.live);
  };
  // the bf16 x 3 image of a node-update pack -> W, by threads t0 .. t0 + nthr - 1 (no barrier)
  auto stage_l3 = [&](const float* pack, int th0, int nthr) { stage_updl3(A, pack, nullptr, tid - th0, nthr); };
  // Hand-off between the S workgroups of this sample WITHOUT cache maintenance (MI355X_MICROARCH.md "Valid forms", first row of
  // its table): every handed-off byte is stored write-through (`sc1`: relaxed agent-scope atomic stores) and loaded `sc1`; every
  // storing wave drains its stores (vmcnt(0)), the workgroup meets at a barrier, ONE lane adds to the sample's counter; the
  // consumer's lane 0 polls the counter with `sc1` loads, the workgroup meets at a barrier, then everybody loads.  (With agent-scope
  // release / acquire fences instead -- a write-back of the XCD's whole L2 per arrival -- each hand-off cost 40-50 us at B = 128.)
  auto st_sc1 = [](float* p, float2 v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto ld_sc1 = [](const float* p) {
    return __builtin_bit_cast(float2, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  };
  auto arrive = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(a.xflag + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto await = [&](int target) -> bool {
    int* okp = reinterpret_cast<int*>(spart + 8);      // (first of the TOP_K2_INTS spare words)
    if (tid == 0) {
      int ok = 0;
      for (int it = 0; it < TOP_POLL_CAP; ++it) {
        if (__hip_atomic_load(a.xflag + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = 1; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      if (!ok) atomicOr(a.status, 2);
      *okp = ok;
    }
    __syncthreads();
    return *okp != 0;
  };
  // The backward pack's image (74 KB) is fetched into the REGISTERS of waves 4..7 while waves 0..3 run the forward chain (they
  // would idle), and written to LDS behind the chain's barrier: the property node then has no staging beside it.
  constexpr int PRE_N = PackUpdL3::FLOATS / 4 / 256 + 1;           // f32x4 per thread of waves 4..7 (18 of the blocks + 1 of the small vectors)
  static_assert((PackUpdL3::FLOATS - PackUpdL3::WAS3) % (4 * 256) == 0 && (PackUpdL3::WAS3 / 4) <= 256, "k_top: backward image prefetch");
  f32x4 pre[PRE_N];
  auto prefetch_b = [&]() {                      // (waves 4..7)
    const int t = tid - 256;
    const f32x4* big = reinterpret_cast<const f32x4*>(a.pack_b + PackUpd::WAS3);
#pragma unroll
    for (int u = 0; u < PRE_N - 1; ++u) pre[u] = big[t + 256 * u];
    // the small vectors: BA (16 f32x4) then BCB, BCBROW, VAW (64 f32x4)
    const int sidx = t < 16 ? PackUpd::BA / 4 + t : PackUpd::BCB / 4 + (t < 80 ? t - 16 : 0);
    pre[PRE_N - 1] = reinterpret_cast<const f32x4*>(a.pack_b)[sidx];
  };
  auto commit_b = [&]() {                        // (waves 4..7, behind the barrier that ends the forward chain)
    const int t = tid - 256;
    f32x4* big = reinterpret_cast<f32x4*>(A + PackUpdL3::WAS3);
#pragma unroll
    for (int u = 0; u < PRE_N - 1; ++u) big[t + 256 * u] = pre[u];
    if (t < 16) reinterpret_cast<f32x4*>(A + PackUpdL3::BA)[t] = pre[PRE_N - 1];
    else if (t < 80) reinterpret_cast<f32x4*>(A + PackUpdL3::BCB)[t - 16] = pre[PRE_N - 1];
  };

  // ---- F2: forward node update of layer L (rows stay in C)
  stage_l3(a.pack_f, 0, 512);
  __syncthreads();
#if defined(TOP_STOP) && TOP_STOP == 5
  if (a.N > 0) return;
#endif
  FT_MARK(2);        // staging the forward pack
  if (wave >= 4) prefetch_b();
  if (upd_wave) {
    Frag X, E;
    load_row(X, valid ? n : 0);
    chain(X, r.amb != 0.0f ? a.Pf + g * 64 : a.pack_f + PackUpd::BCBROW, true, (h ? r.r1 : r.r0) * (a.sf ? a.sf[g] : s_own), E);
    if (valid) {
      if (frag_has_nan(E)) atomicOr(a.status, 1);
      store_row(E, Cr + n * 64, csw(n));
    }
  }
  __syncthreads();
  if (wave >= 4) commit_b();                   // (nothing reads the image again before the barriers of F3)
#if defined(TOP_STOP) && TOP_STOP == 6
  if (a.N > 0) return;
#endif
  FT_MARK(3);        // F2 chain

  // ---- F3: property node (k_prop).  Wave w sums W_prop[m] . row m over m = w, w + 8, ... (one row tile: w % 4), the 8 partial
  // vectors are added in wave order; its three small layers are split over the waves by k (every weight row is requested at
  // once: one L2 round trip per layer instead of a serial 196-step chain on one wave).
  {
    const bool mine = (wave & 3) >= t0 && (wave & 3) < t0 + TS;
    if (mine) {
      float acc = 0.0f;
      for (int m = wave; m < N; m += 8) acc = fmaf(pw[m], Cr[m * 64 + (lane ^ csw(m))], acc);
      if (S == 1) part_[wave * 64 + lane] = acc;
      else __hip_atomic_store(a.xbuf + ((long)b * 8 + wave) * 64 + lane, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    float sp = 0.0f;
    for (int m = tid; m < N; m += 512) {
      const long gm = (long)b * N + m;
      sp += node_is_live(a.lb[gm], a.ub[gm]) ? pw[m] : 0.0f;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sp += __shfl_xor(sp, o);
    if (lane == 0) spart[wave] = sp;
  }
  if (S > 1) {
    arrive();
    if (!await(a.xbase + S)) return;
    part_[wave * 64 + lane] = __hip_atomic_load(a.xbuf + ((long)b * 8 + wave) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const float* Pp = a.pack_p;                 // the property node's weights (PackProp, 51 KB) come straight from L2
  float w3r[8];                                // this wave's 8 rows of the last layer, requested now
#pragma unroll
  for (int q = 0; q < 8; ++q) w3r[q] = Pp[PackProp::W3T + (8 * wave + q) * 64 + lane];
  if (wave < 4) {
    float w2r[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) w2r[q] = Pp[PackProp::W2T + (32 * wave + q) * 64 + lane];
    float nbv = 0.0f, spt = 0.0f;
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) { nbv += part_[w8 * 64 + lane]; spt += spart[w8]; }
    const float f[4] = {a.lbK[b], a.ubK[b], a.z_out[b], a.prop_b[b]};
    float h1 = Pp[PackProp::B1 + lane];
#pragma unroll
    for (int k = 0; k < 4; ++k) h1 = fmaf(Pp[PackProp::W1T + k * 64 + lane], f[k], h1);
    // element k of the hidden layer's input [relu(h1), nb] is held by lane k % 64: through LDS (every wave writes the same values)
    xs[lane] = relu_nan(h1);
    xs[64 + lane] = nbv;
    __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): this wave's LDS writes are visible to its own reads
    float h2 = wave == 0 ? fmaf(spt, Pp[PackProp::V2 + lane], Pp[PackProp::B2 + lane]) : 0.0f;
#pragma unroll
    for (int q = 0; q < 32; ++q) h2 = fmaf(w2r[q], xs[32 * wave + q], h2);
    part2[wave * 64 + lane] = h2;
  }
  __syncthreads();
  {
    const float h2 = ((part2[lane] + part2[64 + lane]) + part2[128 + lane]) + part2[192 + lane];
    xs[lane] = relu_nan(h2);                           // (every wave writes the same values)
    __builtin_amdgcn_s_waitcnt(0xc07f);
    float o = wave == 0 ? Pp[PackProp::B3 + lane] : 0.0f;
#pragma unroll
    for (int q = 0; q < 8; ++q) o = fmaf(w3r[q], xs[8 * wave + q], o);
    part3[wave * 64 + lane] = o;
  }
  __syncthreads();
  if (wave == 0) {
    float o = part3[lane];
#pragma unroll
    for (int w8 = 1; w8 < 8; ++w8) o += part3[w8 * 64 + lane];
    if (part == 0) a.mu_prop[(long)b * 64 + lane] = o;
    outv[lane] = o;
  }
  __syncthreads();
#if defined(TOP_STOP) && TOP_STOP == 7
  if (a.N > 0) return;
#endif
  FT_MARK(4);        // F3 property node + staging backward pack

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
      store_row(E, Cr + n * 64, csw(n));
      if (S == 1) store_row(E, a.mu + g * 64);
      else {                                           // the other workgroups of the sample read these rows: write-through stores
        float* base = a.mu + g * 64 + 4 * h;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          st_sc1(base + 8 * q, make_float2(FRAG_AT(E, 4 * q), FRAG_AT(E, 4 * q + 1)));
          st_sc1(base + 8 * q + 2, make_float2(FRAG_AT(E, 4 * q + 2), FRAG_AT(E, 4 * q + 3)));
        }
      }
    }
  }
  // which rows of C count for B2's bias sums (xs: free since F3): the live nodes of layer L
  if (tid < 128) {
    const long gm = (long)b * N + (tid < N ? tid : 0);
    xs[tid] = (tid < N && node_is_live(a.lb[gm], a.ub[gm])) ? 1.0f : 0.0f;
  }
  if (S > 1) {
    arrive();                                          // this workgroup's rows of layer L are in a.mu
    if (!await(a.xbase + 2 * S)) return;
    for (int e = tid; e < N * 32; e += 512) {          // the rows the other workgroups own -> C (8 bytes per load)
      const int t = (e >> 5) & 3;
      if (t < t0 || t >= t0 + TS) *reinterpret_cast<float2*>(Cr + (e >> 5) * 64 + ((2 * (e & 31)) ^ csw(e >> 5))) = ld_sc1(a.mu + (long)b * N * 64 + 2 * e);
    }
  }
  __syncthreads();
#if defined(TOP_STOP) && TOP_STOP == 8
  if (a.N > 0) return;
#endif
  FT_MARK(5);        // B1 chain

  // ---- B2: aggregate rows of layer L-1 <- W_L^T . rows of C: only the live rows of layer L-1 (nothing reads the others); every
  // row of C is walked (the dead ones are zero), the weights come k-contiguous out of the forward edge's image
#if defined(TOP_STOP) && TOP_STOP == 3
  if (a.N > 0) return;
#endif
  float* out = a.db.out + (long)b * a.db.M * 64;
#ifdef TOP_ABL_NOSTORE   // dev, timing only
  auto put = [&](int row, int jj, float2 v) { if (v.x > 1e30f) *reinterpret_cast<float2*>(out + (long)row * 64 + 2 * jj) = v; };
#else
  auto put = [&](int row, int jj, float2 v) { *reinterpret_cast<float2*>(out + (long)row * 64 + 2 * jj) = v; };
#endif
#if defined(TOP_ABL) && (TOP_ABL & 1)     // dev, timing only: no B2
  if (N < 0) dense_bwd_sample_bf3(a.db, Cr, A, put, nullptr, 0, nullptr, a.df.At, a.df.ldA, nullptr);
#else
  if (a.fuse_um && keep) {
    // the image of layer L-1's update: requested now (registers), written behind the operand image's barrier (mid)
    float* W2 = A + TOP_IMG_FLOATS;
    constexpr int N4 = PackUpdL3::FLOATS / 4;                       // 16 f32x4 BA | 64 BCB, BCBROW, VAW | 3 x 1536 blocks
    constexpr int PER = (N4 + 511) / 512;
    f32x4 st2[PER];
    const f32x4* p4 = reinterpret_cast<const f32x4*>(a.um.pack);
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int i = tid + 512 * u;                                  // index into the LDS image
      int src = PackUpd::WAS3 / 4 + (i - PackUpdL3::WAS3 / 4);      // the three blocks are contiguous in both
      if (i < 16) src = PackUpd::BA / 4 + i;
      else if (i < PackUpdL3::WAS3 / 4) src = PackUpd::BCB / 4 + (i - 16);
      st2[u] = p4[i < N4 ? src : 0];
    }
    const long mbase = (long)b * a.db.M;
    struct NodeIn { float lb, ub, s; };
    dense_bwd_sample_bf3<true>(
        a.db, Cr, A, put, klist, K_eff, a.sb_out ? a.sb_out + mbase : nullptr, a.df.At, a.df.ldA, xs, part, S,
        [&]() {
#pragma unroll
          for (int u = 0; u < PER; ++u) {
            const int i = tid + 512 * u;
            if (i < N4) reinterpret_cast<f32x4*>(W2)[i] = st2[u];
          }
          __syncthreads();
        },
        [&](int arow, bool in) {
          const long gm = mbase + (in ? arow : 0);
          return NodeIn{a.lbm[gm], a.ubm[gm], a.sb_out ? 0.0f : a.um.sarr[gm]};
        },
        [&](int arow, bool in, const Frag& X, float ssum, const NodeIn& nd) {
          const Ratio rt = compute_ratio(nd.lb, nd.ub);
          upd_chain_frag<false>(a.um, W2, X, in ? (int)(mbase + arow) : 0, in ? rt.r0 : 0.0f, in ? rt.r1 : 0.0f, in && rt.amb != 0.0f,
                                in ? (a.sb_out ? ssum : nd.s) : 0.0f, in, lane);
        });
  } else if (keep) dense_bwd_sample_bf3(a.db, Cr, A, put, klist, K_eff, a.sb_out ? a.sb_out + (long)b * a.db.M : nullptr, a.df.At, a.df.ldA, xs, part, S);
  else dense_bwd_sample_bf3(a.db, Cr, A, put, nullptr, 0, nullptr, a.df.At, a.df.ldA, xs, part, S);
#endif
#ifdef FUSED_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  FT_MARK(6);        // B2 dense transposed edge
  if (FUSED_TIMING == 4) FT_FLUSH();
#endif
}

// TS = 4: one workgroup per sample; TS = 2 / 1: 2 / 4 workgroups per sample (grid = B x 4 / TS, the parts of a sample adjacent)
template <int TS>
__global__ __launch_bounds__(512, 1) void k_top(TopArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int S = 4 / TS;
  top_sample<TS>(a, blockIdx.x / S, blockIdx.x % S, lds);
}
