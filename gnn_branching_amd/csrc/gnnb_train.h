// Online-learning step of the branching GNN on the device.
//
// Reference: graphnet/graph_score_online.py:62-77 (GraphChoice.online_learning) --
//     loss = gnn_score - kw_score + improvement;  loss.backward();  Adam(lr, weight_decay).step()
// where gnn_score = max of the scores of one subproblem and kw_score = the score of the node the BaBSR heuristic chose
// (relu_conv_online.py:58-276 calls it once per branch whose KW decision beat the GNN's).  torch.autograd does the backward
// pass there; here the forward of graph_conv.py:77-388 / :442-470 is re-run in "training form" -- every Linear of the GNN as
// its own kernel with its output kept -- on a tape, then the tape is walked backwards with hand-written adjoint kernels, and
// one Adam kernel updates the 117 825 parameters in place.  The scorer's fused inference kernels are not used here: their
// folded weight packs are rebuilt from the new parameters after the step (gnnb_online_step ends with build_packs + upload).
//
// This is a latency path (the reference takes one subproblem per step; B > 1 sums the B losses), so the kernels are plain
// VALU/LDS kernels, one launch per op, a few thousand rows each -- not the MFMA pipeline of gnnb_forward.
//
// Included at the end of gnnb.hip (one translation unit: it uses the bound network of gnnb_handle).
#pragma once
#include <functional>

namespace gnnb_train {
using namespace gnnb;

// ---- device memory: a bump arena, zeroed at the start of every step (gradient buffers start at 0) ----
struct Arena {
  struct Chunk { char* p; size_t cap, used; };
  std::vector<Chunk> chunks;
  size_t chunk_bytes = (size_t)256 << 20;
  int err = 0;
  float* alloc(size_t nfloats) {
    const size_t bytes = (nfloats * 4 + 255) & ~(size_t)255;
    for (auto& c : chunks)
      if (c.cap - c.used >= bytes) { float* r = (float*)(c.p + c.used); c.used += bytes; return r; }
    Chunk c{nullptr, bytes > chunk_bytes ? bytes : chunk_bytes, 0};
    if (hipMalloc((void**)&c.p, c.cap) != hipSuccess) { err = 1; return nullptr; }
    if (hipMemset(c.p, 0, c.cap) != hipSuccess) { err = 1; return nullptr; }
    c.used = bytes;
    chunks.push_back(c);
    return (float*)c.p;
  }
  int reset(hipStream_t st) {          // everything handed out so far back to zero, arena empty
    for (auto& c : chunks) {
      if (c.used && hipMemsetAsync(c.p, 0, c.used, st) != hipSuccess) return 1;
      c.used = 0;
    }
    return 0;
  }
  void release() { for (auto& c : chunks) (void)hipFree(c.p); chunks.clear(); }
};

struct TT { float* v = nullptr; float* g = nullptr; long n = 0; };   // rows (n, 64): value and gradient

// ---- per-node constants of a ReLU layer (graph_conv.py:149-159, :261-279, :499-514) ----
struct TPrepArgs {
  const float *lb, *ub, *dual, *z_pre, *z_post, *bias;
  int N, hw; long n;
  float *r0, *r1, *amb, *live, *nd2, *d1, *featf, *featb;   // (n) each, features (n, 7)
};
__global__ void k_tprep(TPrepArgs a) {
  const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= a.n) return;
  const float lb = a.lb[g], ub = a.ub[g];
  const float lower_temp = lb - fmaxf(lb, 0.0f), upper_temp = fmaxf(ub, 0.0f);
  const float r0 = upper_temp / (upper_temp - lower_temp);
  const float beta = -1.0f * lower_temp * r0;
  const float amb = beta > 0.0f ? 1.0f : 0.0f;
  const float r1 = (1.0f - 2.0f * (r0 * amb)) * amb + r0;
  const float d1 = a.dual[3 * g + 1], d2 = a.dual[3 * g + 2];
  const float c = a.bias[(int)(g % a.N) / a.hw];
  const float zp = a.z_pre[g], zq = a.z_post[g];
  a.r0[g] = r0; a.r1[g] = r1; a.amb[g] = amb; a.live[g] = r0 != 0.0f ? 1.0f : 0.0f; a.nd2[g] = -d2; a.d1[g] = d1;
  float* f = a.featf + 7 * g;                      // :153-159
  f[0] = beta; f[1] = lb; f[2] = ub; f[3] = d1 - d2; f[4] = zp; f[5] = zq; f[6] = c;
  float* b = a.featb + 7 * g;                      // :273-279
  b[0] = lb; b[1] = ub; b[2] = beta; b[3] = -d2 + d1; b[4] = zq; b[5] = zp; b[6] = c;
}

// columns interleaved into (n, w) feature rows: dst[g][j] = col_j[g]
struct TColsArgs { const float* c[4]; int w; long n; float* dst; };
__global__ void k_tcols(TColsArgs a) {
  const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= a.n) return;
  for (int j = 0; j < a.w; ++j) a.dst[g * a.w + j] = (j == 0 ? a.c[0] : j == 1 ? a.c[1] : j == 2 ? a.c[2] : a.c[3])[g];
}

// ---- one Linear of the GNN:  y = omask * act(b + sum_seg W[:, 64 seg : 64 seg + 64] (s_seg * x_seg))  ----
// inputs: up to three 64-wide row segments (a torch.cat along dim 1; each optionally scaled per row by a constant), or
// one (n, kf) block of scalar node features.  act = relu or identity; omask = a per-row constant (amb / live) or null.
// Compact form (ridx != null): the op runs over the first *n_dev entries of a node list only -- the ambiguous nodes for the
// relaxation chains, the live nodes for the update chains; every other row of its output is (and stays) zero, exactly what
// the `* amb` / `* live` masks of the reference produce.  Row i of a compact tensor belongs to node ridx[i]; features,
// per-row constants, and the segments / outputs flagged `full` are addressed by node.
struct TSeg { const float* x; const float* s; float* gx; int full; };
struct TLin {
  const float* W; const float* b; float* gW; float* gb;    // W (64, K) row-major as in the checkpoint
  int K, nseg, kf;
  TSeg seg[3];
  const float* feat;
  const float* omask; int relu;
  float* y; const float* gy; long n;      // n: rows (compact form: capacity of the list)
  const int* ridx; const int* n_dev; int out_full;
  float* part; int nchunks;
  int jbase;      // k_tlin_bwd_x: first segment of this launch
};
#define TL_ROWS 32
#define TL_CHUNK 64

__device__ __forceinline__ TSeg tl_seg(const TLin& a, int j) { return j == 0 ? a.seg[0] : j == 1 ? a.seg[1] : a.seg[2]; }
__device__ __forceinline__ long tl_rows(const TLin& a) {
  if (!a.n_dev) return a.n;
  const long m = *a.n_dev;
  return m < a.n ? m : a.n;
}
// node of each local row (or -1 past the end) into LDS
__device__ __forceinline__ void tl_stage_rows(const TLin& a, long row0, long n, int* Rs, int rows) {
  for (int r = threadIdx.x; r < rows; r += blockDim.x) {
    const long i = row0 + r;
    Rs[r] = i < n ? (a.ridx ? a.ridx[i] : (int)i) : -1;
  }
}

__global__ __launch_bounds__(256) void k_tlin_fwd(TLin a) {
  extern __shared__ float tl_lds[];
  __shared__ int Rs[TL_ROWS];
  const long n = tl_rows(a), row0 = (long)blockIdx.x * TL_ROWS;
  if (row0 >= n) return;
  const int K = a.K, KP = a.K | 1, tid = threadIdx.x;
  float* Ws = tl_lds;                    // [64][KP]: row-major like the checkpoint, odd row stride (lane c reads bank c + k)
  float* Xs = tl_lds + 64 * KP;          // [TL_ROWS][K]
  tl_stage_rows(a, row0, n, Rs, TL_ROWS);
  for (int i = tid; i < 64 * K; i += 256) { const int c = i / K, k = i - c * K; Ws[c * KP + k] = a.W[i]; }
  __syncthreads();
  for (int i = tid; i < TL_ROWS * K; i += 256) {
    const int r = i / K, k = i - r * K;
    const int node = Rs[r];
    float v = 0.0f;
    if (node >= 0) {
      if (a.nseg) {
        const TSeg sg = tl_seg(a, k >> 6);
        v = sg.x[(sg.full ? (long)node : row0 + r) * 64 + (k & 63)];
        if (sg.s) v *= sg.s[node];
      } else v = a.feat[(long)node * a.kf + k];
    }
    Xs[i] = v;
  }
  __syncthreads();
  const int c = tid & 63, rq = tid >> 6;
  float acc[8];
  const float bias = a.b[c];
#pragma unroll
  for (int r = 0; r < 8; ++r) acc[r] = bias;
  const float* xr = Xs + rq * 8 * K;
  const float* wr = Ws + c * KP;
  for (int k = 0; k < K; ++k) {
    const float w = wr[k];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = fmaf(xr[r * K + k], w, acc[r]);
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int node = Rs[rq * 8 + r];
    if (node < 0) continue;
    float v = a.relu ? fmaxf(acc[r], 0.0f) : acc[r];
    if (a.omask) v *= a.omask[node];
    a.y[(a.out_full ? (long)node : row0 + rq * 8 + r) * 64 + c] = v;
  }
}

// gradient that reaches the pre-activation: dym = gy * relu'(y) * omask   (orow: row of y / gy, node: its node)
__device__ __forceinline__ float tl_dym(const TLin& a, long orow, int node, int c) {
  float d = a.gy[orow * 64 + c];
  if (a.relu && !(a.y[orow * 64 + c] > 0.0f)) d = 0.0f;
  if (a.omask) d *= a.omask[node];
  return d;
}

// gx_seg += s_seg * (dym . W[:, seg]);  blockIdx.y = segment
__global__ __launch_bounds__(256) void k_tlin_bwd_x(TLin a) {
  __shared__ float Ws[64 * 64];          // [c][k]
  __shared__ float Ds[TL_ROWS * 64];     // [r][c]
  __shared__ int Rs[TL_ROWS];
  const long n = tl_rows(a), row0 = (long)blockIdx.x * TL_ROWS;
  if (row0 >= n) return;
  const int j = blockIdx.y + a.jbase, tid = threadIdx.x;
  const TSeg sg = tl_seg(a, j);
  if (!sg.gx) return;
  tl_stage_rows(a, row0, n, Rs, TL_ROWS);
  for (int i = tid; i < 4096; i += 256) Ws[i] = a.W[(i >> 6) * a.K + 64 * j + (i & 63)];
  __syncthreads();
  for (int i = tid; i < TL_ROWS * 64; i += 256) {
    const int r = i >> 6, node = Rs[r];
    Ds[i] = node >= 0 ? tl_dym(a, a.out_full ? (long)node : row0 + r, node, i & 63) : 0.0f;
  }
  __syncthreads();
  const int k = tid & 63, rq = tid >> 6;
  float acc[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) acc[r] = 0.0f;
  for (int c = 0; c < 64; ++c) {
    const float w = Ws[c * 64 + k];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = fmaf(Ds[(rq * 8 + r) * 64 + c], w, acc[r]);
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int node = Rs[rq * 8 + r];
    if (node < 0) continue;
    const float s = sg.s ? sg.s[node] : 1.0f;
    sg.gx[(sg.full ? (long)node : row0 + rq * 8 + r) * 64 + k] += s * acc[r];
  }
}

// partial weight gradients of one chunk of rows: part[chunk][c][col] = sum_r dym[r][c] * (s x)[r][col], col K = bias
// blockIdx.y = segment (or 0 for a feature block)
__global__ __launch_bounds__(256) void k_tlin_bwd_w(TLin a) {
  __shared__ __attribute__((aligned(16))) float Ds[TL_CHUNK * 64];
  __shared__ __attribute__((aligned(16))) float Xs[TL_CHUNK * 64];
  __shared__ int Rs[TL_CHUNK];
  const long n = tl_rows(a), row0 = (long)blockIdx.x * TL_CHUNK;
  if (row0 >= n) return;
  const int j = blockIdx.y, tid = threadIdx.x;
  const int KW = a.nseg ? 64 : a.kf;
  const TSeg sg = tl_seg(a, j);
  tl_stage_rows(a, row0, n, Rs, TL_CHUNK);
  __syncthreads();
  for (int i = tid; i < TL_CHUNK * 64; i += 256) {
    const int r = i >> 6, node = Rs[r];
    Ds[i] = node >= 0 ? tl_dym(a, a.out_full ? (long)node : row0 + r, node, i & 63) : 0.0f;
  }
  for (int i = tid; i < TL_CHUNK * KW; i += 256) {
    const int r = i / KW, k = i - r * KW;
    const int node = Rs[r];
    float v = 0.0f;
    if (node >= 0) {
      if (a.nseg) { v = sg.x[(sg.full ? (long)node : row0 + r) * 64 + k]; if (sg.s) v *= sg.s[node]; }
      else v = a.feat[(long)node * a.kf + k];
    }
    Xs[i] = v;
  }
  __syncthreads();
  float* part = a.part + (long)blockIdx.x * 64 * (a.K + 1);
  if (a.nseg) {                    // 64 x 64 outputs: a 4 x 4 block per thread, two b128 LDS reads per row
    const int c0 = (tid >> 4) * 4, k0 = (tid & 15) * 4;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[i][q] = 0.0f;
    for (int r = 0; r < TL_CHUNK; ++r) {
      const float4 d = *reinterpret_cast<const float4*>(&Ds[r * 64 + c0]);
      const float4 x = *reinterpret_cast<const float4*>(&Xs[r * 64 + k0]);
      const float dv[4] = {d.x, d.y, d.z, d.w}, xv[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[i][q] = fmaf(dv[i], xv[q], acc[i][q]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) part[(c0 + i) * (a.K + 1) + 64 * j + k0 + q] = acc[i][q];
  } else
  for (int o = tid; o < 64 * KW; o += 256) {
    const int c = o / KW, k = o - c * KW;
    float s = 0.0f;
    for (int r = 0; r < TL_CHUNK; ++r) s = fmaf(Ds[r * 64 + c], Xs[r * KW + k], s);
    part[c * (a.K + 1) + 64 * j + k] = s;
  }
  if (j == 0 && tid < 64) {
    float s = 0.0f;
    for (int r = 0; r < TL_CHUNK; ++r) s += Ds[r * 64 + tid];
    part[tid * (a.K + 1) + a.K] = s;
  }
}
// gW / gb += the chunk partials, in chunk order (deterministic)
__global__ void k_tlin_reduce(TLin a) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= 64 * (a.K + 1)) return;
  const int nch = (int)((tl_rows(a) + TL_CHUNK - 1) / TL_CHUNK);
  float s = 0.0f;
  for (int ch = 0; ch < nch; ++ch) s += a.part[(long)ch * 64 * (a.K + 1) + o];
  const int c = o / (a.K + 1), col = o - c * (a.K + 1);
  if (col < a.K) a.gW[c * a.K + col] += s;
  else a.gb[c] += s;
}

// ordered list of the rows with a non-zero flag, and their number: one workgroup
struct TCompact { const float* flag; int* idx; int* cnt; long n; };
__global__ __launch_bounds__(256) void k_tcompact(TCompact a) {
  __shared__ int wsum[4];
  __shared__ int base;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  if (tid == 0) base = 0;
  __syncthreads();
  for (long start = 0; start < a.n; start += 256) {
    const long i = start + tid;
    const bool f = i < a.n && a.flag[i] != 0.0f;
    const unsigned long long bal = __ballot(f);
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int off = base + __popcll(bal & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) off += wsum[w];
    if (f) a.idx[off] = (int)i;
    __syncthreads();
    if (tid == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
  }
  if (tid == 0) *a.cnt = base;
}

// ---- edges of the layer graph (graph_conv.py:110-137, :299-326, :361-376) and their adjoints ----
__device__ __forceinline__ int t_taps(int y, int n_out, int k, int s, int p) {      // taps of a transposed conv reaching y
  int c = 0;
  for (int ky = 0; ky < k; ++ky) {
    const int t = y + p - ky;
    if (t >= 0 && t % s == 0 && t / s < n_out) ++c;
  }
  return c;
}
struct TConv {
  const float* src; float* dst; const float* w;      // w: torch layout (C_out, C_in, kh, kw)
  int B, C_in, H_in, W_in, C_out, H_out, W_out, kh, kw, stride, pad;
  int dir;    // 0: dst (B, C_out H_out W_out, 64) = A src;  1: dst (B, C_in H_in W_in, 64) = A^T src
  int norm;   // tap-count division (:306-312): dir 1: of the result; dir 0 (its adjoint): of the source rows
  int acc;    // dst += instead of dst =
};
// one wave per destination node, lane = embedding channel
__global__ __launch_bounds__(256) void k_tconv(TConv a) {
  const int lane = threadIdx.x & 63;
  const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  float acc = 0.0f;
  long out;
  if (a.dir == 0) {
    if (wv >= (long)a.B * a.C_out * a.H_out * a.W_out) return;
    const int ox = (int)(wv % a.W_out), oy = (int)((wv / a.W_out) % a.H_out);
    const int co = (int)((wv / ((long)a.W_out * a.H_out)) % a.C_out), b = (int)(wv / ((long)a.W_out * a.H_out * a.C_out));
    for (int ky = 0; ky < a.kh; ++ky) {
      const int iy = oy * a.stride - a.pad + ky;
      if ((unsigned)iy >= (unsigned)a.H_in) continue;
      for (int kx = 0; kx < a.kw; ++kx) {
        const int ix = ox * a.stride - a.pad + kx;
        if ((unsigned)ix >= (unsigned)a.W_in) continue;
        const float f = a.norm ? (float)(t_taps(iy, a.H_out, a.kh, a.stride, a.pad) * t_taps(ix, a.W_out, a.kw, a.stride, a.pad)) : 1.0f;
        for (int ci = 0; ci < a.C_in; ++ci) {
          float v = a.src[((((long)b * a.C_in + ci) * a.H_in + iy) * a.W_in + ix) * 64 + lane];
          if (a.norm) v = v / f;
          acc = fmaf(a.w[(((long)co * a.C_in + ci) * a.kh + ky) * a.kw + kx], v, acc);
        }
      }
    }
    out = wv;
  } else {
    if (wv >= (long)a.B * a.C_in * a.H_in * a.W_in) return;
    const int x = (int)(wv % a.W_in), y = (int)((wv / a.W_in) % a.H_in);
    const int ci = (int)((wv / ((long)a.W_in * a.H_in)) % a.C_in), b = (int)(wv / ((long)a.W_in * a.H_in * a.C_in));
    for (int ky = 0; ky < a.kh; ++ky) {
      const int ty = y + a.pad - ky;
      if (ty < 0 || ty % a.stride != 0 || ty / a.stride >= a.H_out) continue;
      const int oy = ty / a.stride;
      for (int kx = 0; kx < a.kw; ++kx) {
        const int tx = x + a.pad - kx;
        if (tx < 0 || tx % a.stride != 0 || tx / a.stride >= a.W_out) continue;
        const int ox = tx / a.stride;
        for (int co = 0; co < a.C_out; ++co)
          acc = fmaf(a.w[(((long)co * a.C_in + ci) * a.kh + ky) * a.kw + kx],
                     a.src[((((long)b * a.C_out + co) * a.H_out + oy) * a.W_out + ox) * 64 + lane], acc);
      }
    }
    if (a.norm) acc = acc / (float)(t_taps(y, a.H_out, a.kh, a.stride, a.pad) * t_taps(x, a.W_out, a.kw, a.stride, a.pad));
    out = wv;
  }
  if (a.acc) a.dst[out * 64 + lane] += acc;
  else a.dst[out * 64 + lane] = acc;
}

struct TDense {
  const float* A; long a_bstride;     // A (n_out, n_in) row-major; a_bstride: floats between the matrices of two samples (0: shared)
  const float* src; float* dst;
  int B, n_out, n_in;
  int dir;    // 0: dst (B, n_out, 64) = A src (B, n_in, 64);  1: dst (B, n_in, 64) = A^T src (B, n_out, 64)
  int acc;
};
// one workgroup per destination row: its 4 waves take every 4th source row, partial sums added in wave order
__global__ __launch_bounds__(256) void k_tdense(TDense a) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long wv = blockIdx.x;
  const int nd = a.dir == 0 ? a.n_out : a.n_in, ns = a.dir == 0 ? a.n_in : a.n_out;
  const int i = (int)(wv % nd), b = (int)(wv / nd);
  const float* A = a.A + (long)b * a.a_bstride;
  const float* src = a.src + (long)b * ns * 64 + lane;
  float acc = 0.0f;
  if (a.dir == 0) for (int k = wave; k < ns; k += 4) acc = fmaf(A[(long)i * a.n_in + k], src[(long)k * 64], acc);
  else for (int k = wave; k < ns; k += 4) acc = fmaf(A[(long)k * a.n_in + i], src[(long)k * 64], acc);
  red[wave][lane] = acc;
  __syncthreads();
  if (wave) return;
  acc = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
  if (a.acc) a.dst[wv * 64 + lane] += acc;
  else a.dst[wv * 64 + lane] = acc;
}

// ---- score head (graph_conv.py:442-450): s = fscore(relu(fnode(mu))) for the masked nodes, -inf elsewhere ----
struct TScore {
  const float* h; float* gh;       // relu(fnode(mu_k)) rows (B N_k, 64) and their gradient
  const float* w; const float* b;  // fscore (1, 64), (1)
  const float* mask;               // (B, R)
  float* scores; float* ds;        // (B, R)
  int N, R, off; long n;
  float* gw; float* gb;
  const int* sel; int B;           // (B, 2): the two nodes of every sample the loss reads (argmax, KW), flat ReLU indices
};
__global__ __launch_bounds__(256) void k_tscore_fwd(TScore a) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.n) return;
  float v = a.h[row * 64 + lane] * a.w[lane];
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  const long f = (row / a.N) * a.R + a.off + row % a.N;
  if (lane == 0) a.scores[f] = a.mask[f] != 0.0f ? v + a.b[0] : -INFINITY;
}
__global__ __launch_bounds__(256) void k_tscore_bwd(TScore a) {        // gh = ds w^T  (ds is zero except at <= 2 nodes per sample)
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.n) return;
  const float d = a.ds[(row / a.N) * a.R + a.off + row % a.N];
  a.gh[row * 64 + lane] += d * a.w[lane];
}
// d loss / d fscore: ds is +1 at the argmax and -1 at the KW node of every sample and zero elsewhere -- one wave walks those
// 2 B nodes in order (deterministic) and takes the ones that belong to this layer
__global__ __launch_bounds__(64) void k_tscore_bwd_w(TScore a) {
  const int lane = threadIdx.x;
  float gw = 0.0f, gb = 0.0f;
  for (int b = 0; b < a.B; ++b)
    for (int q = 0; q < 2; ++q) {
      const int f = a.sel[2 * b + q] - a.off;
      if (f < 0 || f >= a.N) continue;
      const float d = q == 0 ? 1.0f : -1.0f;
      gw = fmaf(d, a.h[((long)b * a.N + f) * 64 + lane], gw);
      gb += d;
    }
  a.gw[lane] += gw;
  if (lane == 0) a.gb[0] += gb;
}

// loss_b = max_j s_b[j] - s_b[kw_b] + improvement_b (graph_score_online.py:73); ds = d loss / d scores
struct TLoss { const float* scores; float* ds; const int* kw; const float* imp; float* loss; int R; int* sel; };
__global__ __launch_bounds__(256) void k_tloss(TLoss a) {
  __shared__ float sv[256];
  __shared__ int si[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* s = a.scores + (long)b * a.R;
  float best = -INFINITY; int bi = 0x7fffffff;
  for (int i = tid; i < a.R; i += 256) { const float v = s[i]; if (v > best) { best = v; bi = i; } }   // first maximum of a strided slice
  sv[tid] = best; si[tid] = bi;
  __syncthreads();
  for (int o = 128; o; o >>= 1) {
    if (tid < o) {
      const float v = sv[tid + o]; const int i = si[tid + o];
      if (v > sv[tid] || (v == sv[tid] && i < si[tid])) { sv[tid] = v; si[tid] = i; }
    }
    __syncthreads();
  }
  if (tid == 0) {
    const int am = si[0], kw = a.kw[b];
    a.loss[b] = sv[0] - s[kw] + a.imp[b];
    a.ds[(long)b * a.R + am] += 1.0f;
    a.ds[(long)b * a.R + kw] -= 1.0f;
    a.sel[2 * b] = am; a.sel[2 * b + 1] = kw;
  }
}

// torch.optim.Adam (not AdamW): g += wd p; m, v moments; p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
struct TAdam { float *p, *g, *m, *v; int n; float step_size, wd, b1, b2, eps, bc2s; };   // step_size = lr / (1 - b1^t), bc2s = sqrt(1 - b2^t)
__global__ void k_tadam(TAdam a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const float g = a.g[i] + a.wd * a.p[i];
  const float m = a.m[i] + (1.0f - a.b1) * (g - a.m[i]);        // exp_avg.lerp_(grad, 1 - beta1)
  const float v = a.b2 * a.v[i] + (1.0f - a.b2) * g * g;
  a.m[i] = m; a.v[i] = v;
  const float denom = sqrtf(v) / a.bc2s + a.eps;
  a.p[i] = a.p[i] - a.step_size * (m / denom);
}

// ---- the tape ----
struct Trainer {
  float *d_w = nullptr, *d_g = nullptr, *d_m = nullptr, *d_v = nullptr;
  int step = 0;
  float lr = 1e-4f, wd = 1e-4f;
  Arena arena;
  std::vector<float*> edge_w;            // torch-layout weights of the bound network's edges, device
  std::vector<std::function<void()>> tape;
  hipStream_t st = nullptr;
  // weight-gradient kernels never feed the rest of the backward pass: they run on a side stream, ordered behind the point of
  // the main stream where their gy is final (same order among themselves as on one stream, so the sums stay reproducible)
  hipStream_t side = nullptr;
  std::vector<hipEvent_t> events;
  size_t ev_next = 0;
  hipStream_t fork() {
    if (!side) return st;
    if (ev_next == events.size()) {
      hipEvent_t e;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return st;
      events.push_back(e);
    }
    hipEvent_t e = events[ev_next++];
    if (hipEventRecord(e, st) != hipSuccess || hipStreamWaitEvent(side, e, 0) != hipSuccess) return st;
    return side;
  }
  int join() {                     // the main stream continues behind everything on the side stream
    if (!side) return 0;
    if (ev_next == events.size()) {
      hipEvent_t e;
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return 1;
      events.push_back(e);
    }
    hipEvent_t e = events[ev_next++];
    return hipEventRecord(e, side) != hipSuccess || hipStreamWaitEvent(st, e, 0) != hipSuccess;
  }
  float *d_scores = nullptr, *d_ds = nullptr, *d_loss = nullptr, *d_imp = nullptr; int *d_kw = nullptr, *d_sel = nullptr;
  int cap_B = 0;
  std::vector<float> h_loss;

  TT rows(long n) { TT t; t.n = n; t.v = arena.alloc((size_t)n * 64); t.g = arena.alloc((size_t)n * 64); return t; }

  // a node list: idx (cap) ordered node ids, *cnt their number
  struct List { const int* idx = nullptr; const int* cnt = nullptr; long cap = 0; };

  // y = omask * act(W x + b); records the adjoint.  list: compact form over that node list (rows of `y` = list entries,
  // or nodes when out_full); n: rows of the plain form / nodes of the layer.
  TT lin(int layer, std::vector<TSeg> segs, const float* feat, long n, bool relu, const float* omask, const List* list = nullptr,
         bool out_full = false) {
    TT y = rows(list && !out_full ? list->cap : n);
    TLin a{};
    a.W = d_w + weight_offset(layer); a.b = d_w + bias_offset(layer);
    a.gW = d_g + weight_offset(layer); a.gb = d_g + bias_offset(layer);
    a.K = kLin[layer].in; a.nseg = (int)segs.size(); a.kf = feat ? a.K : 0;
    for (int j = 0; j < a.nseg; ++j) a.seg[j] = segs[j];
    a.feat = feat; a.omask = omask; a.relu = relu ? 1 : 0; a.y = y.v; a.gy = y.g;
    a.n = list ? list->cap : n;
    if (list) { a.ridx = list->idx; a.n_dev = list->cnt; a.out_full = out_full ? 1 : 0; }
    const unsigned nblk = (unsigned)((a.n + TL_ROWS - 1) / TL_ROWS);
    const size_t lds = ((size_t)(a.K | 1) * 64 + (size_t)TL_ROWS * a.K) * 4;
    hipLaunchKernelGGL(k_tlin_fwd, dim3(nblk), dim3(256), lds, st, a);
    tape.push_back([this, a, nblk]() mutable {
      a.nchunks = (int)((a.n + TL_CHUNK - 1) / TL_CHUNK);
      a.part = arena.alloc((size_t)a.nchunks * 64 * (a.K + 1));
      if (!a.part) return;
      hipStream_t ws = fork();
      hipLaunchKernelGGL(k_tlin_bwd_w, dim3((unsigned)a.nchunks, a.nseg ? a.nseg : 1), dim3(256), 0, ws, a);
      hipLaunchKernelGGL(k_tlin_reduce, dim3((64 * (a.K + 1) + 255) / 256), dim3(256), 0, ws, a);
      bool any = false, alias = false;
      for (int j = 0; j < a.nseg; ++j) {
        any = any || a.seg[j].gx;
        for (int i = 0; i < j; ++i) alias = alias || (a.seg[j].gx && a.seg[j].gx == a.seg[i].gx);
      }
      if (!any) return;
      if (!alias) hipLaunchKernelGGL(k_tlin_bwd_x, dim3(nblk, a.nseg), dim3(256), 0, st, a);
      else                                 // segments of the same tensor (r0 nb | r1 nb) add into the same rows: one after the other
        for (int j = 0; j < a.nseg; ++j) { a.jbase = j; hipLaunchKernelGGL(k_tlin_bwd_x, dim3(nblk, 1), dim3(256), 0, st, a); }
    });
    return y;
  }
  static TSeg seg(const TT& t, const float* s = nullptr, bool full = false) { return TSeg{t.v, s, t.g, full ? 1 : 0}; }
};

}  // namespace gnnb_train
