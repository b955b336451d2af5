// gnnb_k_misc.h -- part of libgnnb.so, included by gnnb.hip (one translation unit; see its header comment).
// k_livesum (bias sums of the deferred projections), k_babsr (BaBSR heuristic), k_gather_scored.
#pragma once

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

// ------------------------------------------------------------------------------------------
// k_gather_scored: the restricted last step's transposed conv aggregate (graph_conv.py:299-318), list-driven.
// After the last backward sweep mu[1] is read by the score head only, i.e. at the scored nodes (3-6 % of layer 1), but 62 % of
// the 32-node gather tiles hold one: the tile gather spent 38 us (base, deep) on them.  Here ONE WAVE PER SCORED NODE:
// lane = embedding channel.  The node's window (C_out x (k/s)^2 source nodes of the layer above) is evaluated lane-parallel
// first -- slot validity, liveness of the source node (a dead row is zero and need not be in memory), tap weight -- and
// compacted into a per-wave LDS list; then one coalesced 256-B row load + one FMA per live slot, in slot order (deterministic).
// Divided by the tap count like the tile gather (:306-312); also writes the bias-sum scalar of the deferred projection.
// ------------------------------------------------------------------------------------------
struct GSArgs {
  const int* list; const int* cnt;          // scored nodes of the dst layer (flat ids b * N + n) and their number
  const float* mu_src;                      // (B, Ns, 64) rows of the layer above
  const float* w;                           // conv weight as [co][ky][kx][ci] (pack_conv_bwd)
  const float *src_lb, *src_ub;             // bounds of the layer above (B, Ns)
  float* nb; float* sout;                   // out: aggregate rows by node id; bias sums (or null)
  int N, C, H, W, Co, Ho, Wo, kh, kw, stride, pad, normalise;
};
#define GS_WAVES 8
#ifndef GS_INFLIGHT
#define GS_INFLIGHT 16       // row loads in flight per wave (base, us: 8 -> 29.5, 16 -> 26.5, 32 -> 32.6)
#endif
#define GS_SLOT_LIMIT 128    // C_out x (k/s)^2 window slots at most (the host's limit: cifar_wide_kw's second conv has 32 x 2 x 2)
#define GS_MAXSLOTS (GS_SLOT_LIMIT + 16)      // + padding to a multiple of GS_INFLIGHT
// the aggregate row of ONE scored node by one wave (lane = channel): acc = its channel of the (normalised) aggregate, ssum = the
// bias sum; s_row / s_w: this wave's GS_MAXSLOTS-entry LDS lists.  Shared by k_gather_scored and k_scored_tail.
__device__ __forceinline__ void gather_scored_node(const GSArgs& a, int gc, int lane, int* s_row, float* s_w, float& acc_out, float& ssum_out) {
  const int Ns = a.Co * a.Ho * a.Wo;
  // window slots (co, dy, dx): the taps ky = (y + pad) % s + s dy that can hit a source pixel at all (k/s per axis, not k)
  const int ty_n = (a.kh + a.stride - 1) / a.stride, tx_n = (a.kw + a.stride - 1) / a.stride;
  const int nslots = a.Co * ty_n * tx_n;
  const int b = gc / a.N, n = gc - b * a.N;
  const int ci = n / (a.H * a.W), y = (n / a.W) % a.H, x = n % a.W;
  const float* slb = a.src_lb + (long)b * Ns;
  const float* sub = a.src_ub + (long)b * Ns;
  const int ky0 = (y + a.pad) % a.stride, kx0 = (x + a.pad) % a.stride;
  int nlive = 0, freq = 0;
  for (int s0 = 0; s0 < nslots; s0 += 64) {
    const int sl = s0 + lane;
    const int co = sl / (ty_n * tx_n), dy = (sl / tx_n) % ty_n, dx = sl % tx_n;
    const int ky = ky0 + a.stride * dy, kx = kx0 + a.stride * dx;
    const int ty = y + a.pad - ky, tx = x + a.pad - kx;              // multiples of the stride by construction
    const int oy = ty / a.stride, ox = tx / a.stride;
    const bool hit = sl < nslots && ky < a.kh && kx < a.kw && ty >= 0 && tx >= 0 && oy < a.Ho && ox < a.Wo;
    const int row = hit ? (co * a.Ho + oy) * a.Wo + ox : 0;
    if (s0 == 0) freq = __popcll(__ballot(hit && co == 0));           // taps that touch this pixel (the reference's `freq`)
    // (the tap weight is requested together with the bounds, not behind the liveness test: one memory round trip instead of two)
    const float wv = hit ? a.w[((co * a.kh + ky) * a.kw + kx) * a.C + ci] : 0.0f;
    const bool live = hit && node_is_live(slb[row], sub[row]);
    const unsigned long long bal = __ballot(live);
    if (live) {
      const int p = nlive + __popcll(bal & ((1ull << lane) - 1ull));
      s_row[p] = row;
      s_w[p] = wv;
    }
    nlive += __popcll(bal);
  }
  // pad to a multiple of GS_INFLIGHT with (row 0, weight 0): whole rounds of independent row loads, no remainder loop
  const int npad = (nlive + GS_INFLIGHT - 1) / GS_INFLIGHT * GS_INFLIGHT;
  if (lane < npad - nlive) { s_row[nlive + lane] = 0; s_w[nlive + lane] = 0.0f; }
  __builtin_amdgcn_wave_barrier();
  const float* src = a.mu_src + (long)b * Ns * 64 + lane;
  float acc = 0.0f, ssum = 0.0f;
  for (int q = 0; q < npad; q += GS_INFLIGHT) {
    float v[GS_INFLIGHT], wq[GS_INFLIGHT];
#pragma unroll
    for (int u = 0; u < GS_INFLIGHT; ++u) {
      wq[u] = s_w[q + u];
      v[u] = q + u < nlive ? src[(long)s_row[q + u] * 64] : 0.0f;     // (wave-uniform; a padding slot must not touch memory:
                                                                      //  row 0 may be a dead node's never-written row)
    }
#pragma unroll
    for (int u = 0; u < GS_INFLIGHT; ++u) {
      acc = fmaf(wq[u], v[u], acc);
      ssum += wq[u];
    }
  }
  if (a.normalise) {
    const float f = (float)(freq > 0 ? freq : 1);
    acc = acc / f;
    ssum = ssum / f;
  }
  acc_out = acc;
  ssum_out = ssum;
  __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(GS_WAVES * 64) void k_gather_scored(GSArgs a) {
  __shared__ int s_row[GS_WAVES][GS_MAXSLOTS];
  __shared__ float s_w[GS_WAVES][GS_MAXSLOTS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int total = *a.cnt;
  // The list holds a sample's scored nodes next to each other (k_classify: one block per 2048 nodes), and neighbouring scored nodes
  // share window rows.  Each workgroup takes ONE contiguous segment of the list, and the segments of the workgroups of one XCD
  // (blockIdx % 8) are neighbours: a sample's rows are then fetched into ONE L2 and re-read there.  (Round 3 dealt list entries
  // round-robin over the workgroups, i.e. one sample over all eight XCDs: 131 MB fetched for ~42 MB of distinct rows, L2 hit rate 26 %.)
  int wg = blockIdx.x;
  const int nwg = gridDim.x;
  if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);
  const int seg = (total + nwg - 1) / nwg;
  const int i0 = wg * seg, i1 = i0 + seg < total ? i0 + seg : total;
  for (int idx = i0 + wave; idx < i1; idx += GS_WAVES) {
    const int gc = a.list[idx];
    float acc, ssum;
    gather_scored_node(a, gc, lane, s_row[wave], s_w[wave], acc, ssum);
    a.nb[(long)gc * 64 + lane] = acc;
    if (lane == 0 && a.sout) a.sout[gc] = ssum;
  }
}


// ------------------------------------------------------------------------------------------
// k_scored_tail: the end of a forward in ONE launch instead of three (k_gather_scored, the restricted k_node_update, k_score).  After
// the last backward sweep mu[1] is read by the score head only, at the scored nodes, so their transposed aggregate, the folded node
// update (q_chain: the arithmetic of k_node_update) and the score head (score_rows) run back to back per tile of 32 scored nodes (the
// rows are still written to mu[1], as the three kernels did: inspection reads them, the score head does not); the scored nodes of the
// other layers go through the score head as in k_score, and the workgroup that finishes last turns the per-sample keys into
// decisions.  Same arithmetic per node as the three kernels: bit-identical scores (tests).
// Round 3's form (one wave per node, wave 0 alone running chain + score head) only paid up to B = 8: each launch of the three was
// its own ramp there, but at B = 256 it took 76 us against 61.  This form scales with the batch:
//   * a workgroup takes contiguous SEGMENTS of the scored list (16 .. 64 nodes; a sample's scored nodes are neighbours in the list, and
//     the segments of the workgroups of one XCD are neighbours: window rows shared by scored nodes are re-read in one L2), as many
//     rounds of equal segments as the batch needs;
//   * waves 0 / 1 are CHAIN waves: one 32-node tile of the segment each (chain + score head); the other 14 are GATHER waves that take
//     FOUR scored nodes at a time (gather_scored_multi: lane = (node, channel quad): one 16-B piece of a 256-B row per lane, 16 rows in
//     flight per node), so that a CU keeps 56 nodes' loads in flight at 4 waves per SIMD -- what k_gather_scored needed 8 waves per SIMD for;
//   * the rows go through an LDS buffer (two: round r + 1 is gathered while the chain waves work on round r); after its last round
//     a gather wave goes straight to its share of the other layers' score tiles.
// ------------------------------------------------------------------------------------------
struct TailArgs { GSArgs g; FArgs f; ScoreArgs s; int sp; };      // sp: entries per node in the gather waves' slot lists (tail_slots_pad)
#ifndef TAIL_WAVES
#define TAIL_WAVES 12          // 3 per SIMD: 168 registers (the chain's pipelined blocks and 64 registers of rows in flight per gather lane without spills)
#endif
#define TAIL_SEG 40           // scored nodes of layer 1 per workgroup and round at most: two chain tiles (32 + 8) and ONE group of four per gather wave
                              // (10 gather waves; with 48 two waves gathered two groups in a row and every round took twice as long: wide 163 us)
#ifndef TAIL_PIPE
#define TAIL_PIPE true
#endif
#ifndef TAIL_INF_REGS
#define TAIL_INF_REGS 48      // (64: three registers spilled)
#endif
#define TAIL_FIXED_FLOATS (PackUpdL3::FLOATS + PackScore::FLOATS + 2 * TAIL_SEG * QROW + 16)
// slot-list entries per node (window slots + padding to whole rounds of GS_INFLIGHT) and nodes per gather wave (4, or 2 when four lists
// per wave do not fit beside the weights: windows of more than 96 slots)
__host__ __device__ inline int tail_slots_pad(int nslots) { return ((nslots + GS_INFLIGHT - 1) / GS_INFLIGHT) * GS_INFLIGHT + GS_INFLIGHT; }
__host__ __device__ inline int tail_npw(int nslots) { return (size_t)TAIL_FIXED_FLOATS * 4 + (size_t)(TAIL_WAVES - 1) * 4 * tail_slots_pad(nslots) * 6 + 64 <= 160 * 1024 - 256 ? 4 : 2; }
__host__ __device__ inline size_t tail_lds_bytes(int nslots) {
  return (size_t)TAIL_FIXED_FLOATS * 4 + (size_t)(TAIL_WAVES - 1) * tail_npw(nslots) * tail_slots_pad(nslots) * 6 + 64;
}
static_assert(TAIL_FIXED_FLOATS * 4 + (TAIL_WAVES - 1) * 2 * (GS_SLOT_LIMIT + GS_INFLIGHT) * 6 + 64 <= 160 * 1024 - 256, "k_scored_tail: LDS (two nodes per wave)");

// NPW (4 or 2) scored nodes at once by one wave.  Lane (g = lane / LPN, c = lane % LPN): node gcs[g] (wave-uniform ids, -1: none),
// channels NPW c .. NPW c + NPW - 1 (LPN = 64 / NPW lanes per node hold its 64 channels: a 16-B or 8-B piece of a 256-B row per lane).
// Per node exactly gather_scored_node's evaluation: window slots in slot order, liveness of the source node, tap weight, compaction
// into an LDS list, then acc = fma(w, row, acc) over the live slots in slot order and division by the tap count -- the same value
// per channel bit for bit.  s_row / s_w: this wave's [NPW][sp] lists (16-bit row indices: the host checks Ns < 65536).
template <int NPW>
__device__ __forceinline__ void gather_scored_multi(const GSArgs& a, const int (&gcs)[NPW], int lane, unsigned short* s_row, float* s_w, int sp,
                                                    float (&acc_out)[NPW], float& ssum_out) {
  constexpr int LPN = 64 / NPW, CPL = NPW;
  typedef float vec_t __attribute__((ext_vector_type(CPL)));
  const int Ns = a.Co * a.Ho * a.Wo;
  const int ty_n = (a.kh + a.stride - 1) / a.stride, tx_n = (a.kw + a.stride - 1) / a.stride;
  const int nslots = a.Co * ty_n * tx_n;
  const int npass = (nslots + 63) >> 6;                       // <= 2 (GS_SLOT_LIMIT)
  // ---- every load of the table builds first (independent), then the ballots
  float lbv[NPW][2], ubv[NPW][2], wvv[NPW][2];
  int rowv[NPW][2];
  bool hitv[NPW][2];
  int freq[NPW], nlive[NPW];
#pragma unroll
  for (int g = 0; g < NPW; ++g) {
    const int gc = gcs[g] < 0 ? 0 : gcs[g];
    const int b = gc / a.N, n = gc - b * a.N;
    const int ci = n / (a.H * a.W), y = (n / a.W) % a.H, x = n % a.W;
    const float* slb = a.src_lb + (long)b * Ns;
    const float* sub = a.src_ub + (long)b * Ns;
    const int ky0 = (y + a.pad) % a.stride, kx0 = (x + a.pad) % a.stride;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int sl = 64 * p + lane;
      const int co = sl / (ty_n * tx_n), dy = (sl / tx_n) % ty_n, dx = sl % tx_n;
      const int ky = ky0 + a.stride * dy, kx = kx0 + a.stride * dx;
      const int ty = y + a.pad - ky, tx = x + a.pad - kx;              // multiples of the stride by construction
      const int oy = ty / a.stride, ox = tx / a.stride;
      const bool hit = p < npass && gcs[g] >= 0 && sl < nslots && ky < a.kh && kx < a.kw && ty >= 0 && tx >= 0 && oy < a.Ho && ox < a.Wo;
      const int row = hit ? (co * a.Ho + oy) * a.Wo + ox : 0;
      hitv[g][p] = hit; rowv[g][p] = row;
      lbv[g][p] = hit ? slb[row] : 0.0f;
      ubv[g][p] = hit ? sub[row] : 0.0f;
      wvv[g][p] = hit ? a.w[((co * a.kh + ky) * a.kw + kx) * a.C + ci] : 0.0f;
      if (p == 0) freq[g] = __popcll(__ballot(hit && co == 0));        // taps that touch this pixel (the reference's `freq`)
    }
  }
#pragma unroll
  for (int g = 0; g < NPW; ++g) {
    unsigned short* sr = s_row + g * sp;
    float* sw = s_w + g * sp;
    int nl = 0;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const bool live = hitv[g][p] && node_is_live(lbv[g][p], ubv[g][p]);
      const unsigned long long bal = __ballot(live);
      if (live) {
        const int q = nl + __popcll(bal & ((1ull << lane) - 1ull));
        sr[q] = (unsigned short)rowv[g][p];
        sw[q] = wvv[g][p];
      }
      nl += __popcll(bal);
    }
    nlive[g] = nl;
  }
  __builtin_amdgcn_wave_barrier();
  // ---- the walks of the nodes side by side
  const int g = lane / LPN, c = lane % LPN;
  int mygc = gcs[0], myn = nlive[0], myf = freq[0], nmax = nlive[0];
#pragma unroll
  for (int g2 = 1; g2 < NPW; ++g2) {
    mygc = g == g2 ? gcs[g2] : mygc;
    myn = g == g2 ? nlive[g2] : myn;
    myf = g == g2 ? freq[g2] : myf;
    nmax = nmax > nlive[g2] ? nmax : nlive[g2];
  }
  const int b = (mygc < 0 ? 0 : mygc) / a.N;
  const float* src = a.mu_src + (long)b * Ns * 64 + CPL * c;
  const unsigned short* sr = s_row + g * sp;
  const float* sw = s_w + g * sp;
  float acc[CPL];
#pragma unroll
  for (int k = 0; k < CPL; ++k) acc[k] = 0.0f;
  float ssum = 0.0f;
  constexpr int INF = TAIL_INF_REGS / NPW;      // rows in flight per node: TAIL_INF_REGS registers of row data per lane either way
  for (int q = 0; q < nmax; q += INF) {
    vec_t v[INF];
    float wq[INF];
#pragma unroll
    for (int u = 0; u < INF; ++u) {
      const bool in = q + u < myn;
      wq[u] = in ? sw[q + u] : 0.0f;
      // (a slot beyond this node's list must not touch memory: it may be a dead node's never-written row)
      vec_t z;
#pragma unroll
      for (int k = 0; k < CPL; ++k) z[k] = 0.0f;
      v[u] = in ? *reinterpret_cast<const vec_t*>(src + (long)sr[q + u] * 64) : z;
    }
#pragma unroll
    for (int u = 0; u < INF; ++u) {
#pragma unroll
      for (int k = 0; k < CPL; ++k) acc[k] = fmaf(wq[u], v[u][k], acc[k]);
      ssum += wq[u];
    }
  }
  if (a.normalise) {
    const float f = (float)(myf > 0 ? myf : 1);
#pragma unroll
    for (int k = 0; k < CPL; ++k) acc[k] = acc[k] / f;
    ssum = ssum / f;
  }
#pragma unroll
  for (int k = 0; k < CPL; ++k) acc_out[k] = acc[k];
  ssum_out = ssum;
  __builtin_amdgcn_wave_barrier();
}

// the gather waves' part of one segment: groups of NPW nodes qd, qd + ngw, ... of the segment -> rows of the LDS buffer rb.
// The aggregate of a dead node marked undecided is gathered like any other (its source rows are live rows) and zeroed at the end: the
// table build then does not wait for the destination's bounds (one memory round trip less per group).
template <int NPW>
__device__ __forceinline__ void tail_gather_segment(const TailArgs& a, float* rb, int sg, int seg, int total, int first, int ngw, int lane,
                                                    unsigned short* s_row, float* s_w) {
  constexpr int LPN = 64 / NPW, CPL = NPW;
  typedef float vec_t __attribute__((ext_vector_type(CPL)));
  for (int qd = first; NPW * qd < seg; qd += ngw) {
    const int i0 = sg * seg + NPW * qd;
    if (i0 >= total) break;
    int gcs[NPW];
#pragma unroll
    for (int g = 0; g < NPW; ++g) gcs[g] = i0 + g < total ? __builtin_amdgcn_readfirstlane(a.g.list[i0 + g]) : -1;
    const int g = lane / LPN;
    int mygc = gcs[0];
#pragma unroll
    for (int g2 = 1; g2 < NPW; ++g2) mygc = g == g2 ? gcs[g2] : mygc;
    const int gcl = mygc < 0 ? 0 : mygc;
    const float dlb = a.f.u.lb[gcl], dub = a.f.u.ub[gcl];      // (requested here, used after the gather)
    float acc[CPL];
    float ssum;
    gather_scored_multi<NPW>(a.g, gcs, lane, s_row, s_w, a.sp, acc, ssum);
    const Ratio rt = compute_ratio(dlb, dub);
    if (mygc >= 0) {
      float* row = rb + (NPW * qd + g) * QROW;
      const bool dead = rt.live == 0.0f;
      vec_t o;
#pragma unroll
      for (int k = 0; k < CPL; ++k) o[k] = dead ? 0.0f : acc[k];
      *reinterpret_cast<vec_t*>(row + CPL * (lane % LPN)) = o;
      if (lane % LPN == 0)
        *reinterpret_cast<f32x4*>(row + 64) = f32x4{__int_as_float(mygc | (rt.amb != 0.0f ? (int)0x80000000 : 0)), rt.r0, rt.r1, dead ? 0.0f : ssum};
    }
  }
}

#if defined(FUSED_TIMING) && FUSED_TIMING == 5      // dev: per-phase cycle sums of k_scored_tail (slots 0-4: first gather wave, 5-9: chain wave 0)
#define TT_DECL unsigned long long tt_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tt_last = __builtin_readcyclecounter()
#define TT_MARK(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); tt_[i] += n_ - tt_last; tt_last = n_; } while (0)
#define TT_FLUSH(c) do { if ((c) && (threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 10; ++i_) atomicAdd(&g_fused_t[i_], tt_[i_]); atomicAdd(&g_fused_t[15], 1ull); } } while (0)
#else
#define TT_DECL
#define TT_MARK(i)
#define TT_FLUSH(c)
#endif
__global__ __launch_bounds__(TAIL_WAVES * 64, 1) void k_scored_tail(TailArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lds_sc = lds + PackUpdL3::FLOATS;
  float* rows = lds_sc + PackScore::FLOATS;                       // [2][TAIL_SEG][QROW]
  const int npw = tail_npw(a.g.Co * ((a.g.kh + a.g.stride - 1) / a.g.stride) * ((a.g.kw + a.g.stride - 1) / a.g.stride));
  float* s_w_all = rows + 2 * TAIL_SEG * QROW;                     // [TAIL_WAVES - 1][npw][sp] tap weights, then as many 16-bit row indices
  unsigned short* s_row_all = reinterpret_cast<unsigned short*>(s_w_all + (TAIL_WAVES - 1) * npw * a.sp);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  TT_DECL;
  const int total = *a.g.cnt;                                      // scored nodes of layer 1
  int wg = blockIdx.x;
  const int nwg = gridDim.x;
  if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);      // the workgroups of one XCD (blockIdx % 8) take neighbouring segments
  // rounds of equal segments: as few rounds as TAIL_SEG allows, every workgroup the same share (a multiple of 16 nodes)
  const int rounds = total > 0 ? (total + TAIL_SEG * nwg - 1) / (TAIL_SEG * nwg) : 0;
  int seg = rounds ? (total + rounds * nwg - 1) / (rounds * nwg) : npw;
  seg = (seg + npw - 1) / npw * npw;                               // whole groups of npw nodes
  if (seg < 16 && total >= 16) seg = 16;                           // (small batches: at least half a chain tile per workgroup)
  const int nseg = (total + seg - 1) / seg;
  const int nchain = (seg + 31) >> 5;                              // chain waves: one 32-node tile of the segment each (1 or 2)
  const int ngw = TAIL_WAVES - nchain;
  // The weights (92 KB) are staged by the waves that have no group of nodes to gather in the first round -- the chain waves and the
  // gather waves beyond the segment's groups (a segment has at most 12 groups) -- while the others already gather;
  // the barrier behind the first round (or the one below, for a workgroup without a segment) publishes them.
  {
    const int ngroups = wg < nseg ? (seg + npw - 1) / npw : 0;
    const int nbusy = ngroups < ngw ? ngroups : ngw;               // gather waves nchain .. nchain + nbusy - 1 have a group
    const int nst = TAIL_WAVES - nbusy;
    if (wave < nchain || wave >= nchain + nbusy) {
      const int si = wave < nchain ? wave : wave - nbusy;
      const int ct = si * 64 + lane, cn = nst * 64;
      stage_updl3(lds, a.f.u.pack, nullptr, ct, cn);
      copy_to_lds_part(lds_sc, a.s.pack, PackScore::FLOATS, ct, cn);
    }
  }
  TT_MARK(wave < nchain ? 5 : 0);             // staging (chain / idle waves) or nothing
  // ---- layer 1: gather -> chain -> score, segment by segment
  int buf = 0;
  for (int sg = wg; sg < nseg; sg += nwg, buf ^= 1) {
    float* rb = rows + buf * TAIL_SEG * QROW;
    if (wave >= nchain) {
      const int gw = wave - nchain;
      if (npw == 4) tail_gather_segment<4>(a, rb, sg, seg, total, gw, ngw, lane, s_row_all + gw * 4 * a.sp, s_w_all + gw * 4 * a.sp);
      else tail_gather_segment<2>(a, rb, sg, seg, total, gw, ngw, lane, s_row_all + gw * 2 * a.sp, s_w_all + gw * 2 * a.sp);
    }
    TT_MARK(wave < nchain ? 5 : 1);           // gather of this segment (gather waves); chain waves: idle
    __syncthreads();                          // (first pass: also the weights; every pass: this segment's rows are in LDS, the previous segment's chains are done)
    TT_MARK(wave < nchain ? 6 : 2);           // barrier wait
    if (wave < nchain) {
      const int left = total - sg * seg - 32 * wave;
      const int lim = seg - 32 * wave < 32 ? seg - 32 * wave : 32;
      const int nvalid = left < lim ? left : lim;
      if (nvalid > 0) {
        const float* rt_ = rb + 32 * wave * QROW;
        Frag E;
        q_chain<false, TAIL_PIPE>(a.f, lds, rt_, nvalid, lane, [] {}, &E);
        const int j = lane & 31;
        const bool valid = j < nvalid;
        const int gc = valid ? (__float_as_int(rt_[j * QROW + 64]) & 0x7fffffff) : 0;
        const float live = valid && node_is_live(a.f.u.lb[gc], a.f.u.ub[gc]) ? 1.0f : 0.0f;
        if (live == 0.0f) {                     // a dead node marked undecided: its row is zero by definition
#pragma unroll
          for (int R = 0; R < 32; ++R) FRAG_AT(E, R) = 0.0f;
        }
        score_rows(a.s, lds_sc, 0, gc, valid, live, E, lane);
      }
    }
  }
  TT_MARK(wave < nchain ? 7 : 0);             // chain + score head of the last segment (chain waves)
  if (wg >= nseg) __syncthreads();            // a workgroup without a segment: the staged weights become visible here
  // ---- the scored nodes of the other layers: the score head on their rows in memory (k_score's tiles), dealt from the LAST
  // workgroup backwards -- those have the fewest layer-1 segments -- and to the gather waves first (they are free while the chain
  // waves finish the last segment: no barrier in front of this phase)
  long ntiles = 0;
  for (int k = 1; k < a.s.L; ++k) ntiles += (a.s.cnt[4 * k + 2] + 31) / 32;
  const int slot = wave >= nchain ? wave - nchain : ngw + wave;
  for (long tile = (long)(gridDim.x - 1 - blockIdx.x) * TAIL_WAVES + slot; tile < ntiles; tile += (long)gridDim.x * TAIL_WAVES) {
    int k = 1, count = 0;
    long t = tile;
    for (; k < a.s.L; ++k) {
      count = a.s.cnt[4 * k + 2];
      const long tk = (count + 31) / 32;
      if (t < tk) break;
      t -= tk;
    }
    score_tile(a.s, lds_sc, k, a.s.list[k], count, t, lane);
  }
  TT_MARK(wave < nchain ? 8 : 3);             // the other layers' score tiles
  score_finish(a.s);
  TT_MARK(wave < nchain ? 9 : 4);             // finish
  TT_FLUSH(wave == 0 || wave == nchain);
}

// ---- k_scatter_amb: the compact records of a host-fed batch (gnnb_pack_amb_records) -> full-size dual / primal arrays --------------------
// image (32-bit words): hdr[16] = {magic, L, n_records, B, 0 ..}; n_records x {layer k - 1, flat node index b N_k + n, dual[:, 1], dual[:, 2],
// primal_pre, primal_post} in any order; then zout[B] = primals[-1].  One thread per record / per z_out entry, grid-stride (the record count
// is only known on the device).
#define AMBREC_MAGIC 0x414d4252
#define AMBREC_WORDS 6
struct ScatterArgs {
  const int* image;
  float* dual[MAXL]; float* z_pre[MAXL]; float* z_post[MAXL]; float* z_out;
  long G[MAXL];          // nodes of ReLU layer k in the batch (B N_k): a record's flat index must be below it
  long max_rec;          // records the image can hold for this binding and B (sum of G)
  int* status;           // may be null; bit 2 (value 4): the image does not belong to this binding / batch size, or holds a record outside its arrays
  int L, B;
};
__global__ void k_scatter_amb(ScatterArgs a) {
  const int* hdr = a.image;
  // an image packed under another binding or batch size (or a stale device copy) must not be scattered: nothing is written, bit 2 is raised
  if (hdr[0] != AMBREC_MAGIC || hdr[1] != a.L || hdr[3] != a.B || hdr[2] < 0 || (long)hdr[2] > a.max_rec) {
    if (a.status && blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.status, 4);
    return;
  }
  const long nrec = hdr[2];
  const long total = nrec + a.B;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    if (i < nrec) {
      const int* r = a.image + 16 + i * AMBREC_WORDS;
      const int k = r[0], g = r[1];
      if ((unsigned)k >= (unsigned)a.L || g < 0 || (long)g >= a.G[k]) {      // (a corrupt record must not write out of bounds)
        if (a.status) atomicOr(a.status, 4);
        continue;
      }
      a.dual[k][(long)g * 3 + 1] = __int_as_float(r[2]);
      a.dual[k][(long)g * 3 + 2] = __int_as_float(r[3]);
      a.z_pre[k][g] = __int_as_float(r[4]);
      a.z_post[k][g] = __int_as_float(r[5]);
    } else {
      a.z_out[i - nrec] = __int_as_float(a.image[16 + nrec * AMBREC_WORDS + (i - nrec)]);
    }
  }
}

// ---- k_occupy: inspection hook (gnnb_debug_occupy) -- workgroups that hold their CU (its LDS, through the dynamic allocation) until `ticks`
// of the chip-wide 100 MHz clock have passed since the first wave looked at it.  tests/test_gpu_dist_safety.py runs forwards beside it to
// show that no kernel of a forward needs workgroups co-resident once the k_top split is off.  Every wave leaves when the time is up: the
// grid always drains.
__global__ void k_occupy(unsigned long long ticks) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (threadIdx.x == 0) lds[0] = 0.0f;                       // (touch the allocation)
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
