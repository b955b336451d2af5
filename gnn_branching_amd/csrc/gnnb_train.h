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
// VALU/LDS kernels over a few thousand rows each -- not the MFMA pipeline of gnnb_forward.  Round 3: the Linears of one MLP
// chain over the same rows (fc3 -> fc3_2 -> fc4 -> fc4_2, bc1 -> ... -> bc2_1, ...) are ONE launch forward (k_tchain_fwd: a row
// tile goes through all of them, each output kept in memory for the backward pass and handed to the next layer through LDS) and
// ONE launch backward (k_tchain_bwd_x); all weight gradients of a step are two launches at its end (k_tlin_bwd_w_all over every
// (op, row chunk), k_tlin_reduce_all adding the partials in tape order).  Every sum keeps the order of the one-launch-per-op form,
// so values and gradients are bit-identical to it.
//
// Included at the end of gnnb.hip (one translation unit: it uses the bound network of gnnb_handle).
#pragma once
#include <functional>

namespace gnnb_train {
using namespace gnnb;
typedef float tf4 __attribute__((ext_vector_type(4)));

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
#define T_MAXL 8        // ReLU layers the merged per-layer launches of a step carry (gnnb_online_step rejects deeper networks)
struct TPrepMulti { TPrepArgs a[T_MAXL]; };          // every ReLU layer in one launch: blockIdx.y = layer
__global__ void k_tprep(TPrepMulti m) {
  const TPrepArgs& a = m.a[blockIdx.y];
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
struct TColsMulti { TColsArgs a[4]; };               // blockIdx.y = job
__global__ void k_tcols(TColsMulti m) {
  const TColsArgs& a = m.a[blockIdx.y];
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
  int layer;      // index of this Linear in the checkpoint (k_tlin_reduce_all adds up the ops of one layer in tape order)
  unsigned prev;  // bit j: segment j is the output tile of the PREVIOUS op of the chain (read from LDS, not from memory)
};
#define TC_MAXOPS 5
struct TChain { TLin op[TC_MAXOPS]; int nops; };
#define TL_ROWS 32      // rows per block of a chain kernel when there are many (TL_ROWS_SMALL otherwise: more, lighter blocks)
#define TL_ROWS_SMALL 8
#define TL_ROWS_TINY 4
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

// A chain of Linears over one set of rows, forward: a tile of TL_ROWS rows goes through op 0 .. nops-1; every output is written
// to memory (the backward pass needs it) and stays in LDS (`Ys`) for the segments of the next op that read it (TLin.prev).
// Per op the arithmetic of the former one-launch-per-Linear kernel: acc = bias, then one fma per input feature in order.
template <int R>
__device__ __forceinline__ void tchain_fwd_body(const TChain& c, float* tl_lds) {
  __shared__ int Rs[R];
  const long n = tl_rows(c.op[0]), row0 = (long)blockIdx.x * R;
  if (row0 >= n) return;
  const int tid = threadIdx.x;
  int Kmax = 0;
  for (int i = 0; i < c.nops; ++i) Kmax = c.op[i].K > Kmax ? c.op[i].K : Kmax;
  float* Ws = tl_lds;                         // [64][KP]: row-major like the checkpoint, odd row stride (lane c reads bank c + k)
  float* Xs = Ws + 64 * (Kmax | 1);           // [R][K]
  float* Ys = Xs + R * Kmax;            // [R][64]: the previous op's output tile
  tl_stage_rows(c.op[0], row0, n, Rs, R);
  for (int i = 0; i < c.nops; ++i) {
    const TLin& a = c.op[i];
    const int K = a.K, KP = a.K | 1;
    __syncthreads();                          // Rs is there (i = 0); the previous op is done with Ws / Xs and its Ys is complete
    // Staging with every load of a thread in flight at once (a plain copy loop keeps one load in flight per thread: 16-32
    // dependent L2 round trips per op were most of the former kernels' ~10 us)
    if ((K & 63) == 0) {                      // weights of a 64 / 128 / 192-wide layer: 16-B loads, 4 KB per wave instruction
      const tf4* W4 = reinterpret_cast<const tf4*>(a.W);
      const int n4 = 16 * K, k4n = K >> 2;    // tf4 per matrix, per row
      tf4 v[12];                            // (n4 <= 3072 = 12 x 256)
#pragma unroll
      for (int u = 0; u < 12; ++u) { const int q = tid + 256 * u; if (q < n4) v[u] = W4[q]; }
#pragma unroll
      for (int u = 0; u < 12; ++u) {
        const int q = tid + 256 * u;
        if (q < n4) {
          const int cc = q / k4n, k = (q - cc * k4n) * 4;
          float* d = Ws + cc * KP + k;
          d[0] = v[u][0]; d[1] = v[u][1]; d[2] = v[u][2]; d[3] = v[u][3];
        }
      }
    } else {
      for (int q = tid; q < 64 * K; q += 256) { const int cc = q / K, k = q - cc * K; Ws[cc * KP + k] = a.W[q]; }      // (K <= 7)
    }
    if (a.nseg) {                             // 64-wide row segments: one 16-B load per (row, segment, 4 features)
      const int items = R * a.nseg * 16;                  // <= 1536 = 6 x 256
      tf4 v[6];
      float sc[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int q = tid + 256 * u;
        v[u] = tf4{0.f, 0.f, 0.f, 0.f};
        sc[u] = 1.0f;
        if (q < items) {
          const int f4 = q & 15, j = (q >> 4) % a.nseg, r = q / (16 * a.nseg);
          const int node = Rs[r];
          if (node >= 0) {
            const TSeg sg = tl_seg(a, j);
            if ((a.prev >> j) & 1u) v[u] = *reinterpret_cast<const tf4*>(Ys + r * 64 + 4 * f4);
            else v[u] = *reinterpret_cast<const tf4*>(sg.x + (sg.full ? (long)node : row0 + r) * 64 + 4 * f4);
            if (sg.s) sc[u] = sg.s[node];
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int q = tid + 256 * u;
        if (q < items) {
          const int f4 = q & 15, j = (q >> 4) % a.nseg, r = q / (16 * a.nseg);
          float* d = Xs + r * K + 64 * j + 4 * f4;
          // (x * s: the product the one-load-per-element form made, 0 for a row past the end)
          d[0] = v[u][0] * sc[u]; d[1] = v[u][1] * sc[u]; d[2] = v[u][2] * sc[u]; d[3] = v[u][3] * sc[u];
        }
      }
    } else {
      for (int q = tid; q < R * K; q += 256) {
        const int r = q / K, k = q - r * K;
        const int node = Rs[r];
        Xs[q] = node >= 0 ? a.feat[(long)node * a.kf + k] : 0.0f;
      }
    }
    __syncthreads();
    const int cc = tid & 63, rq = tid >> 6;
    float acc[R / 4];
    const float bias = a.b[cc];
#pragma unroll
    for (int r = 0; r < R / 4; ++r) acc[r] = bias;
    const float* xr = Xs + rq * (R / 4) * K;
    const float* wr = Ws + cc * KP;
    for (int k = 0; k < K; ++k) {
      const float w = wr[k];
#pragma unroll
      for (int r = 0; r < R / 4; ++r) acc[r] = fmaf(xr[r * K + k], w, acc[r]);
    }
#pragma unroll
    for (int r = 0; r < R / 4; ++r) {
      const int node = Rs[rq * (R / 4) + r];
      float v = 0.0f;
      if (node >= 0) {
        v = a.relu ? fmaxf(acc[r], 0.0f) : acc[r];
        if (a.omask) v *= a.omask[node];
        a.y[(a.out_full ? (long)node : row0 + rq * (R / 4) + r) * 64 + cc] = v;
      }
      Ys[(rq * (R / 4) + r) * 64 + cc] = v;
    }
  }
}

// gradient that reaches the pre-activation: dym = gy * relu'(y) * omask   (orow: row of y / gy, node: its node)
__device__ __forceinline__ float tl_dym(const TLin& a, long orow, int node, int c) {
  float d = a.gy[orow * 64 + c];
  if (a.relu && !(a.y[orow * 64 + c] > 0.0f)) d = 0.0f;
  if (a.omask) d *= a.omask[node];
  return d;
}

// The same chain backwards (input gradients): op nops-1 .. 0 on one row tile.  gx_seg += s_seg * (dym . W[:, seg]), the segments
// of an op in order (the order of the former one-launch-per-segment form).  The gradient of a segment that is the previous op's
// output -- whose total is that op's gy -- is added to what memory already holds for it (contributions of consumers outside the
// chain, which ran earlier on the tape), written back, and handed to the next iteration through LDS (`Gs`).
template <int R>
__global__ __launch_bounds__(256) void k_tchain_fwd(TChain c) {
  extern __shared__ __attribute__((aligned(16))) float tl_lds[];
  tchain_fwd_body<R>(c, tl_lds);
}
// several independent chains (the same chain of different layers) in one launch: blockIdx.y = chain, descriptors in memory
template <int R>
__global__ __launch_bounds__(256) void k_tchain_fwd_multi(const TChain* cs) {
  extern __shared__ __attribute__((aligned(16))) float tl_lds[];
  tchain_fwd_body<R>(cs[blockIdx.y], tl_lds);
}

template <int R>
__device__ __forceinline__ void tchain_bwd_x_body(const TChain& c) {
  __shared__ __attribute__((aligned(16))) float Ws[64 * 64];          // [c][k]
  __shared__ __attribute__((aligned(16))) float Ds[R * 64];     // [r][c]
  __shared__ __attribute__((aligned(16))) float Gs[R * 64];     // [r][k]: gy of the op about to be walked, when it was produced by the op above it
  __shared__ int Rs[R];
  const long n = tl_rows(c.op[0]), row0 = (long)blockIdx.x * R;
  if (row0 >= n) return;
  const int tid = threadIdx.x;
  tl_stage_rows(c.op[0], row0, n, Rs, R);
  bool carried = false;
  for (int i = c.nops - 1; i >= 0; --i) {
    const TLin& a = c.op[i];
    bool any = false;
    for (int j = 0; j < a.nseg; ++j) any = any || tl_seg(a, j).gx != nullptr;
    if (!any) { carried = false; continue; }
    __syncthreads();                          // Rs (first pass); Gs of the iteration before is complete; Ds free
    {                                         // dym of the tile: 2 x 16-B items per thread, their loads issued together
      tf4 gv[2], yv[2];
      float om[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = tid + 256 * u, r = q >> 4, f4 = q & 15;
        const int node = q < R * 16 ? Rs[r] : -1;
        gv[u] = tf4{0.f, 0.f, 0.f, 0.f}; yv[u] = tf4{1.f, 1.f, 1.f, 1.f}; om[u] = 1.0f;
        if (node >= 0) {
          const long orow = a.out_full ? (long)node : row0 + r;
          gv[u] = carried ? *reinterpret_cast<const tf4*>(Gs + r * 64 + 4 * f4) : *reinterpret_cast<const tf4*>(a.gy + orow * 64 + 4 * f4);
          if (a.relu) yv[u] = *reinterpret_cast<const tf4*>(a.y + orow * 64 + 4 * f4);
          if (a.omask) om[u] = a.omask[node];
        }
      }
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int q = tid + 256 * u, r = q >> 4, f4 = q & 15;
        if (q >= R * 16) continue;
        float* d = Ds + r * 64 + 4 * f4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float dv = gv[u][e];
          if (a.relu && !(yv[u][e] > 0.0f)) dv = 0.0f;
          if (a.omask) dv *= om[u];
          d[e] = dv;
        }
      }
    }
    const int k = tid & 63, rq = tid >> 6;
    float g[R / 4];
    bool have_prev = false;
    long paddr[R / 4];
    float* pgx = nullptr;
    for (int j = 0; j < a.nseg; ++j) {
      const TSeg sg = tl_seg(a, j);
      if (!sg.gx) continue;
      const bool pv = ((a.prev >> j) & 1u) != 0;
      __syncthreads();                        // Ds is complete (first segment); the segment before is done with Ws
      {                                       // W[:, 64 j .. 64 j + 63] -> Ws: four 16-B loads per thread in flight
        tf4 wv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int q = tid + 256 * u; wv[u] = *reinterpret_cast<const tf4*>(a.W + (q >> 4) * a.K + 64 * j + 4 * (q & 15)); }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int q = tid + 256 * u; *reinterpret_cast<tf4*>(Ws + (q >> 4) * 64 + 4 * (q & 15)) = wv[u]; }
      }
      __syncthreads();
      float acc[R / 4];
#pragma unroll
      for (int r = 0; r < R / 4; ++r) acc[r] = 0.0f;
      for (int cc = 0; cc < 64; ++cc) {
        const float w = Ws[cc * 64 + k];
#pragma unroll
        for (int r = 0; r < R / 4; ++r) acc[r] = fmaf(Ds[(rq * (R / 4) + r) * 64 + cc], w, acc[r]);
      }
#pragma unroll
      for (int r = 0; r < R / 4; ++r) {
        const int node = Rs[rq * (R / 4) + r];
        if (node < 0) continue;
        const float sc = sg.s ? sg.s[node] : 1.0f;
        const long addr = (sg.full ? (long)node : row0 + rq * (R / 4) + r) * 64 + k;
        if (pv) {
          if (!have_prev) { g[r] = sg.gx[addr]; paddr[r] = addr; }
          g[r] += sc * acc[r];
        } else sg.gx[addr] += sc * acc[r];
      }
      if (pv) { have_prev = true; pgx = sg.gx; }
    }
    if (have_prev) {
      __syncthreads();                        // everybody has read Gs (through Ds) before it is rewritten
#pragma unroll
      for (int r = 0; r < R / 4; ++r) {
        const int node = Rs[rq * (R / 4) + r];
        if (node >= 0) pgx[paddr[r]] = g[r];
        Gs[(rq * (R / 4) + r) * 64 + k] = node >= 0 ? g[r] : 0.0f;
      }
    }
    carried = have_prev;
  }
}

template <int R>
__global__ __launch_bounds__(256) void k_tchain_bwd_x(TChain c) { tchain_bwd_x_body<R>(c); }
template <int R>
__global__ __launch_bounds__(256) void k_tchain_bwd_x_multi(const TChain* cs) { tchain_bwd_x_body<R>(cs[blockIdx.y]); }

// Weight gradients of EVERY op of a step in one launch: block -> (op, chunk of TL_CHUNK rows), blockIdx.y = segment (or 0 for a
// feature block).  part[chunk][c][col] = sum_r dym[r][c] * (s x)[r][col], col K = bias.  Runs after the whole backward walk, when
// every gy is final; x, y, gy of all ops are still in the arena.  first[o] = first block of op o, first[nops] = the grid size.
__global__ __launch_bounds__(256) void k_tlin_bwd_w_all(const TLin* ops, const int* first, int nops) {
  __shared__ __attribute__((aligned(16))) float Ds[TL_CHUNK * 64];
  __shared__ __attribute__((aligned(16))) float Xs[TL_CHUNK * 64];
  __shared__ int Rs[TL_CHUNK];
  __shared__ int s_op;
  __shared__ TLin s_a;            // the op's descriptor in LDS: as a private copy its run-time indexed segment table lived in scratch (248 B per lane)
  const int tid = threadIdx.x;
  if (tid == 0) {
    int o = 0;
    while (o + 1 < nops && first[o + 1] <= (int)blockIdx.x) ++o;
    s_op = o;
    s_a = ops[o];
  }
  __syncthreads();
  const TLin& a = s_a;
  const int chunk = (int)blockIdx.x - first[s_op];
  const long n = tl_rows(a), row0 = (long)chunk * TL_CHUNK;
  const int j = blockIdx.y;
  if (row0 >= n || j >= (a.nseg ? a.nseg : 1)) return;
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
  float* part = a.part + (long)chunk * 64 * (a.K + 1);
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
    const int cq = o / KW, k = o - cq * KW;
    float sm = 0.0f;
    for (int r = 0; r < TL_CHUNK; ++r) sm = fmaf(Ds[r * 64 + cq], Xs[r * KW + k], sm);
    part[cq * (a.K + 1) + 64 * j + k] = sm;
  }
  if (j == 0 && tid < 64) {
    float sm = 0.0f;
    for (int r = 0; r < TL_CHUNK; ++r) sm += Ds[r * 64 + tid];
    part[tid * (a.K + 1) + a.K] = sm;
  }
}
// gW / gb += the chunk partials of every op of a layer: the ops in tape order, the chunks of an op in chunk order (deterministic,
// and the order of the former one-reduction-per-op form).  blockIdx.y = Linear of the checkpoint.
__global__ void k_tlin_reduce_all(const TLin* ops, int nops) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x, layer = blockIdx.y;
  for (int q = 0; q < nops; ++q) {
    if (ops[q].layer != layer) continue;
    const TLin a = ops[q];
    if (o >= 64 * (a.K + 1)) return;
    const int nch = (int)((tl_rows(a) + TL_CHUNK - 1) / TL_CHUNK);
    float sm = 0.0f;
    for (int ch = 0; ch < nch; ++ch) sm += a.part[(long)ch * 64 * (a.K + 1) + o];
    const int cq = o / (a.K + 1), col = o - cq * (a.K + 1);
    if (col < a.K) a.gW[cq * a.K + col] += sm;
    else a.gb[cq] += sm;
  }
}

// ordered list of the rows with a non-zero flag, and their number: one workgroup
struct TCompact { const float* flag; int* idx; int* cnt; long n; };
struct TCompactMulti { TCompact a[2 * T_MAXL]; };      // one workgroup per list: blockIdx.x = list
__global__ __launch_bounds__(256) void k_tcompact(TCompactMulti m) {
  const TCompact& a = m.a[blockIdx.x];
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
// one wave per destination node, lane = embedding channel.  The node's valid taps are listed first (lane-parallel, kept in the
// order of the plain loops: ky, kx, channel), then walked in batches of TCONV_BATCH independent row loads -- the sums and their
// order are those of the plain loops (which kept one dependent load in flight per tap: 23 us per launch).
#define TCONV_MAXTAPS 512
#define TCONV_BATCH 16
__global__ __launch_bounds__(256) void k_tconv(TConv a) {
  __shared__ int s_row[4][TCONV_MAXTAPS];
  __shared__ float s_w[4][TCONV_MAXTAPS];
  __shared__ float s_f[4][TCONV_MAXTAPS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long wv = (long)blockIdx.x * 4 + wave;
  const long ndst = a.dir == 0 ? (long)a.B * a.C_out * a.H_out * a.W_out : (long)a.B * a.C_in * a.H_in * a.W_in;
  if (wv >= ndst) return;
  const int C = a.dir == 0 ? a.C_in : a.C_out;          // channels of the source side
  const int ntaps = a.kh * a.kw * C;
  int nl = 0;
  long sbase;                                            // first source row of this sample
  float post = 1.0f;
  if (a.dir == 0) {
    const int ox = (int)(wv % a.W_out), oy = (int)((wv / a.W_out) % a.H_out);
    const int co = (int)((wv / ((long)a.W_out * a.H_out)) % a.C_out), b = (int)(wv / ((long)a.W_out * a.H_out * a.C_out));
    sbase = (long)b * a.C_in * a.H_in * a.W_in;
    for (int t0 = 0; t0 < ntaps; t0 += 64) {
      const int t = t0 + lane;
      const int ci = t % C, kx = (t / C) % a.kw, ky = t / (C * a.kw);
      const int iy = oy * a.stride - a.pad + ky, ix = ox * a.stride - a.pad + kx;
      const bool ok = t < ntaps && (unsigned)iy < (unsigned)a.H_in && (unsigned)ix < (unsigned)a.W_in;
      const unsigned long long bal = __ballot(ok);
      if (ok) {
        const int p = nl + __popcll(bal & ((1ull << lane) - 1ull));
        s_row[wave][p] = (ci * a.H_in + iy) * a.W_in + ix;
        s_w[wave][p] = a.w[(((long)co * a.C_in + ci) * a.kh + ky) * a.kw + kx];
        s_f[wave][p] = a.norm ? (float)(t_taps(iy, a.H_out, a.kh, a.stride, a.pad) * t_taps(ix, a.W_out, a.kw, a.stride, a.pad)) : 1.0f;
      }
      nl += __popcll(bal);
    }
  } else {
    const int x = (int)(wv % a.W_in), y = (int)((wv / a.W_in) % a.H_in);
    const int ci = (int)((wv / ((long)a.W_in * a.H_in)) % a.C_in), b = (int)(wv / ((long)a.W_in * a.H_in * a.C_in));
    sbase = (long)b * a.C_out * a.H_out * a.W_out;
    for (int t0 = 0; t0 < ntaps; t0 += 64) {
      const int t = t0 + lane;
      const int co = t % C, kx = (t / C) % a.kw, ky = t / (C * a.kw);
      const int ty = y + a.pad - ky, tx = x + a.pad - kx;
      const bool ok = t < ntaps && ty >= 0 && ty % a.stride == 0 && ty / a.stride < a.H_out && tx >= 0 && tx % a.stride == 0 && tx / a.stride < a.W_out;
      const unsigned long long bal = __ballot(ok);
      if (ok) {
        const int p = nl + __popcll(bal & ((1ull << lane) - 1ull));
        s_row[wave][p] = (co * a.H_out + ty / a.stride) * a.W_out + tx / a.stride;
        s_w[wave][p] = a.w[(((long)co * a.C_in + ci) * a.kh + ky) * a.kw + kx];
        s_f[wave][p] = 1.0f;
      }
      nl += __popcll(bal);
    }
    if (a.norm) post = (float)(t_taps(y, a.H_out, a.kh, a.stride, a.pad) * t_taps(x, a.W_out, a.kw, a.stride, a.pad));
  }
  __builtin_amdgcn_wave_barrier();
  const float* src = a.src + sbase * 64 + lane;
  const bool nrm0 = a.dir == 0 && a.norm;
  float acc = 0.0f;
  for (int q = 0; q < nl; q += TCONV_BATCH) {
    float v[TCONV_BATCH], w[TCONV_BATCH], f[TCONV_BATCH];
#pragma unroll
    for (int u = 0; u < TCONV_BATCH; ++u) {
      const bool in = q + u < nl;                        // (wave-uniform)
      w[u] = in ? s_w[wave][q + u] : 0.0f;
      f[u] = in ? s_f[wave][q + u] : 1.0f;
      v[u] = in ? src[(long)s_row[wave][q + u] * 64] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < TCONV_BATCH; ++u) {
      if (q + u < nl) {
        float vv = v[u];
        if (nrm0) vv = vv / f[u];
        acc = fmaf(w[u], vv, acc);
      }
    }
  }
  if (a.dir == 1 && a.norm) acc = acc / post;
  if (a.acc) a.dst[wv * 64 + lane] += acc;
  else a.dst[wv * 64 + lane] = acc;
}

struct TDense {
  const float* A; long a_bstride;     // A (n_out, n_in) row-major; a_bstride: floats between the matrices of two samples (0: shared)
  const float* src; float* dst;
  int B, n_out, n_in;
  int dir;    // 0: dst (B, n_out, 64) = A src (B, n_in, 64);  1: dst (B, n_in, 64) = A^T src (B, n_out, 64)
  int acc;
};
// one workgroup of TD_WAVES waves per destination row: wave w takes the source rows w, w + TD_WAVES, ... (16 independent loads at a
// time, their fmas in row order), the partial sums are added in wave order.  (4 waves walked the 1024 source rows of the Linear
// edge in 16 dependent batches: 20 us per launch.)
#define TD_WAVES 16
__global__ __launch_bounds__(TD_WAVES * 64) void k_tdense(TDense a) {
  __shared__ float red[TD_WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long wv = blockIdx.x;
  const int nd = a.dir == 0 ? a.n_out : a.n_in, ns = a.dir == 0 ? a.n_in : a.n_out;
  const int i = (int)(wv % nd), b = (int)(wv / nd);
  const float* A = a.A + (long)b * a.a_bstride;
  const float* src = a.src + (long)b * ns * 64 + lane;
  float acc = 0.0f;
  for (int k0 = wave; k0 < ns; k0 += 16 * TD_WAVES) {
    float w[16], v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int k = k0 + TD_WAVES * u;
      const bool in = k < ns;                            // (wave-uniform)
      w[u] = in ? (a.dir == 0 ? A[(long)i * a.n_in + k] : A[(long)k * a.n_in + i]) : 0.0f;
      v[u] = in ? src[(long)k * 64] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (k0 + TD_WAVES * u < ns) acc = fmaf(w[u], v[u], acc);
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave) return;
  acc = red[0][lane];
#pragma unroll
  for (int w = 1; w < TD_WAVES; ++w) acc += red[w][lane];
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
struct TScoreMulti { TScore a[T_MAXL]; };              // blockIdx.y = ReLU layer
__global__ __launch_bounds__(256) void k_tscore_fwd(TScoreMulti m) {
  const TScore& a = m.a[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.n) return;
  float v = a.h[row * 64 + lane] * a.w[lane];
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  const long f = (row / a.N) * a.R + a.off + row % a.N;
  if (lane == 0) a.scores[f] = a.mask[f] != 0.0f ? v + a.b[0] : -INFINITY;
}
__global__ __launch_bounds__(256) void k_tscore_bwd(TScoreMulti m) {        // gh = ds w^T  (ds is zero except at <= 2 nodes per sample)
  const TScore& a = m.a[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.n) return;
  const float d = a.ds[(row / a.N) * a.R + a.off + row % a.N];
  a.gh[row * 64 + lane] += d * a.w[lane];
}
// d loss / d fscore: ds is +1 at the argmax and -1 at the KW node of every sample and zero elsewhere -- one wave walks those
// 2 B nodes in order (deterministic) and takes the ones that belong to this layer
// (all layers in one launch: the layers' contributions are added in the order of the former one-launch-per-layer form, last layer first)
__global__ __launch_bounds__(64) void k_tscore_bwd_w(TScoreMulti m, int nlayers) {
  const int lane = threadIdx.x;
  for (int l = nlayers - 1; l >= 0; --l) {
    const TScore& a = m.a[l];
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
  int n_cu = 256;
  float *d_scores = nullptr, *d_ds = nullptr, *d_loss = nullptr, *d_imp = nullptr; int *d_kw = nullptr, *d_sel = nullptr;
  int cap_B = 0;
  std::vector<float> h_loss;

  TT rows(long n) { TT t; t.n = n; t.v = arena.alloc((size_t)n * 64); t.g = arena.alloc((size_t)n * 64); return t; }

  // a node list: idx (cap) ordered node ids, *cnt their number
  struct List { const int* idx = nullptr; const int* cnt = nullptr; long cap = 0; };

  // One Linear of a chain: y = omask * act(W [segments] + b).  A segment with x == nullptr is the previous op's output.
  struct Spec { int layer; std::vector<TSeg> segs; const float* feat; bool relu; const float* omask; bool out_full; };
  std::vector<TLin> wops;                // every op of the step, in tape (backward) order: the weight-gradient launches walk it

  // A chain of Linears over the same rows (all of them the plain form over n rows, or the compact form over `list`): one launch
  // forward, one backward.  Returns the outputs of the ops.  list: compact form (rows of an output = list entries, or nodes when
  // its op is out_full); n: rows of the plain form / nodes of the layer.
  std::vector<TT> chain(const std::vector<Spec>& specs, long n, const List* list = nullptr) {
    std::vector<TT> out;
    TChain c{};
    c.nops = (int)specs.size();
    int Kmax = 0;
    for (int i = 0; i < c.nops; ++i) {
      const Spec& sp = specs[i];
      TT y = rows(list && !sp.out_full ? list->cap : n);
      TLin& a = c.op[i];
      a.layer = sp.layer;
      a.W = d_w + weight_offset(sp.layer); a.b = d_w + bias_offset(sp.layer);
      a.gW = d_g + weight_offset(sp.layer); a.gb = d_g + bias_offset(sp.layer);
      a.K = kLin[sp.layer].in; a.nseg = (int)sp.segs.size(); a.kf = sp.feat ? a.K : 0;
      for (int j = 0; j < a.nseg; ++j) {
        a.seg[j] = sp.segs[j];
        if (!sp.segs[j].x) {                       // the previous op's output tile
          a.prev |= 1u << j;
          a.seg[j].x = out[i - 1].v; a.seg[j].gx = out[i - 1].g; a.seg[j].full = specs[i - 1].out_full ? 1 : 0;
        }
      }
      a.feat = sp.feat; a.omask = sp.omask; a.relu = sp.relu ? 1 : 0; a.y = y.v; a.gy = y.g;
      a.n = list ? list->cap : n;
      if (list) { a.ridx = list->idx; a.n_dev = list->cnt; a.out_full = sp.out_full ? 1 : 0; }
      a.nchunks = (int)((a.n + TL_CHUNK - 1) / TL_CHUNK);
      a.part = arena.alloc((size_t)a.nchunks * 64 * (a.K + 1));
      Kmax = a.K > Kmax ? a.K : Kmax;
      out.push_back(y);
    }
    // few rows (the latency case: one subproblem has ~1000 live nodes per layer): 4-row tiles -- eight times the blocks, each an
    // eighth of the inner loop, the weights staged per block either way (base B = 1, ms per step: 32-row tiles 3.31, 16: 2.64, 8:
    // 2.56, 4: 2.21); a few thousand rows: 8-row tiles; many rows: 32-row tiles amortise the staging
    const long nrows = list ? list->cap : n;
    const int R = nrows <= 16L * n_cu ? TL_ROWS_TINY : (nrows <= 128L * n_cu ? TL_ROWS_SMALL : TL_ROWS);
    const unsigned nblk = (unsigned)((nrows + R - 1) / R);
    const size_t lds = ((size_t)(Kmax | 1) * 64 + (size_t)R * Kmax + (size_t)R * 64) * 4;
    if (R == TL_ROWS_TINY) hipLaunchKernelGGL(k_tchain_fwd<TL_ROWS_TINY>, dim3(nblk), dim3(256), lds, st, c);
    else if (R == TL_ROWS_SMALL) hipLaunchKernelGGL(k_tchain_fwd<TL_ROWS_SMALL>, dim3(nblk), dim3(256), lds, st, c);
    else hipLaunchKernelGGL(k_tchain_fwd<TL_ROWS>, dim3(nblk), dim3(256), lds, st, c);
    tape.push_back([this, c, nblk, R]() {
      bool any = false;
      for (int i = 0; i < c.nops; ++i)
        for (int j = 0; j < c.op[i].nseg; ++j) any = any || c.op[i].seg[j].gx;
      if (any) {
        if (R == TL_ROWS_TINY) hipLaunchKernelGGL(k_tchain_bwd_x<TL_ROWS_TINY>, dim3(nblk), dim3(256), 0, st, c);
        else if (R == TL_ROWS_SMALL) hipLaunchKernelGGL(k_tchain_bwd_x<TL_ROWS_SMALL>, dim3(nblk), dim3(256), 0, st, c);
        else hipLaunchKernelGGL(k_tchain_bwd_x<TL_ROWS>, dim3(nblk), dim3(256), 0, st, c);
      }
      for (int i = c.nops - 1; i >= 0; --i) wops.push_back(c.op[i]);
    });
    return out;
  }
  // Several independent chains (the same chain of different ReLU layers) as ONE launch forward and one backward: blockIdx.y =
  // chain, the descriptors uploaded once and read from memory by both launches.  Returns the outputs per chain.
  struct Job { std::vector<Spec> specs; long n; const List* list; };
  TChain* desc = nullptr;                // pinned, device-visible descriptor slots of the step in flight (every step ends in a sync)
  int ndesc = 0;
  static constexpr int kDescCap = 8 * T_MAXL;
  std::vector<std::vector<TT>> chain_multi(const std::vector<Job>& jobs) {
    std::vector<std::vector<TT>> outs;
    const int nj = (int)jobs.size();
    if (!desc || ndesc + nj > kDescCap) { arena.err = true; return std::vector<std::vector<TT>>(nj, std::vector<TT>(TC_MAXOPS)); }
    const int base = ndesc;
    long maxrows = 0;
    int Kmax = 0;
    for (const Job& jb : jobs) {
      std::vector<TT> out;
      TChain c{};
      c.nops = (int)jb.specs.size();
      for (int i = 0; i < c.nops; ++i) {
        const Spec& sp = jb.specs[i];
        TT y = rows(jb.list && !sp.out_full ? jb.list->cap : jb.n);
        TLin& a = c.op[i];
        a.layer = sp.layer;
        a.W = d_w + weight_offset(sp.layer); a.b = d_w + bias_offset(sp.layer);
        a.gW = d_g + weight_offset(sp.layer); a.gb = d_g + bias_offset(sp.layer);
        a.K = kLin[sp.layer].in; a.nseg = (int)sp.segs.size(); a.kf = sp.feat ? a.K : 0;
        for (int j = 0; j < a.nseg; ++j) {
          a.seg[j] = sp.segs[j];
          if (!sp.segs[j].x) {
            a.prev |= 1u << j;
            a.seg[j].x = out[i - 1].v; a.seg[j].gx = out[i - 1].g; a.seg[j].full = jb.specs[i - 1].out_full ? 1 : 0;
          }
        }
        a.feat = sp.feat; a.omask = sp.omask; a.relu = sp.relu ? 1 : 0; a.y = y.v; a.gy = y.g;
        a.n = jb.list ? jb.list->cap : jb.n;
        if (jb.list) { a.ridx = jb.list->idx; a.n_dev = jb.list->cnt; a.out_full = sp.out_full ? 1 : 0; }
        a.nchunks = (int)((a.n + TL_CHUNK - 1) / TL_CHUNK);
        a.part = arena.alloc((size_t)a.nchunks * 64 * (a.K + 1));
        Kmax = a.K > Kmax ? a.K : Kmax;
        out.push_back(y);
      }
      const long nrows = jb.list ? jb.list->cap : jb.n;
      maxrows = nrows > maxrows ? nrows : maxrows;
      desc[ndesc++] = c;
      outs.push_back(out);
    }
    const TChain* d_cs = desc + base;
    const int R = maxrows <= 16L * n_cu ? TL_ROWS_TINY : (maxrows <= 128L * n_cu ? TL_ROWS_SMALL : TL_ROWS);
    const dim3 grid((unsigned)((maxrows + R - 1) / R), (unsigned)nj);
    const size_t lds = ((size_t)(Kmax | 1) * 64 + (size_t)R * Kmax + (size_t)R * 64) * 4;
    if (R == TL_ROWS_TINY) hipLaunchKernelGGL(k_tchain_fwd_multi<TL_ROWS_TINY>, grid, dim3(256), lds, st, d_cs);
    else if (R == TL_ROWS_SMALL) hipLaunchKernelGGL(k_tchain_fwd_multi<TL_ROWS_SMALL>, grid, dim3(256), lds, st, d_cs);
    else hipLaunchKernelGGL(k_tchain_fwd_multi<TL_ROWS>, grid, dim3(256), lds, st, d_cs);
    std::vector<TChain> cs(desc + base, desc + ndesc);
    tape.push_back([this, cs, d_cs, grid, R]() {
      bool any = false;
      for (const TChain& c : cs)
        for (int i = 0; i < c.nops; ++i)
          for (int j = 0; j < c.op[i].nseg; ++j) any = any || c.op[i].seg[j].gx;
      if (any) {
        if (R == TL_ROWS_TINY) hipLaunchKernelGGL(k_tchain_bwd_x_multi<TL_ROWS_TINY>, grid, dim3(256), 0, st, d_cs);
        else if (R == TL_ROWS_SMALL) hipLaunchKernelGGL(k_tchain_bwd_x_multi<TL_ROWS_SMALL>, grid, dim3(256), 0, st, d_cs);
        else hipLaunchKernelGGL(k_tchain_bwd_x_multi<TL_ROWS>, grid, dim3(256), 0, st, d_cs);
      }
      for (int q = (int)cs.size() - 1; q >= 0; --q)
        for (int i = cs[q].nops - 1; i >= 0; --i) wops.push_back(cs[q].op[i]);
    });
    return outs;
  }
  TT lin(int layer, std::vector<TSeg> segs, const float* feat, long n, bool relu, const float* omask, const List* list = nullptr,
         bool out_full = false) {
    return chain({Spec{layer, std::move(segs), feat, relu, omask, out_full}}, n, list)[0];
  }
  // the weight gradients of every recorded op: two launches behind the backward walk
  std::vector<int> h_first;
  int weight_grads() {
    const int nops = (int)wops.size();
    if (!nops) return 0;
    h_first.assign(nops + 1, 0);
    for (int o = 0; o < nops; ++o) h_first[o + 1] = h_first[o] + wops[o].nchunks;
    const size_t op_floats = (sizeof(TLin) * nops + 3) / 4, first_floats = nops + 1;
    TLin* d_ops = reinterpret_cast<TLin*>(arena.alloc(op_floats));
    int* d_first = reinterpret_cast<int*>(arena.alloc(first_floats));
    if (!d_ops || !d_first) return 1;
    if (hipMemcpyAsync(d_ops, wops.data(), sizeof(TLin) * nops, hipMemcpyHostToDevice, st) != hipSuccess) return 1;
    if (hipMemcpyAsync(d_first, h_first.data(), sizeof(int) * (nops + 1), hipMemcpyHostToDevice, st) != hipSuccess) return 1;
    hipLaunchKernelGGL(k_tlin_bwd_w_all, dim3((unsigned)h_first[nops], 3), dim3(256), 0, st, d_ops, d_first, nops);
    hipLaunchKernelGGL(k_tlin_reduce_all, dim3((64 * 193 + 255) / 256, (unsigned)L_COUNT), dim3(256), 0, st, d_ops, nops);
    return 0;
  }
  static TSeg seg(const TT& t, const float* s = nullptr, bool full = false) { return TSeg{t.v, s, t.g, full ? 1 : 0}; }
  static TSeg prev(const float* s = nullptr) { return TSeg{nullptr, s, nullptr, 0}; }      // the previous op's output (chain)
};

}  // namespace gnnb_train
