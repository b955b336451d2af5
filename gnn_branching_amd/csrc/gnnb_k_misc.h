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
#define GS_MAXSLOTS 112      // C_out x (k/s)^2 <= 96 window slots (the host's limit) + padding to a multiple of GS_INFLIGHT
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
    const bool live = hit && node_is_live(slb[row], sub[row]);
    const float wv = live ? a.w[((co * a.kh + ky) * a.kw + kx) * a.C + ci] : 0.0f;
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
  for (int idx = blockIdx.x * GS_WAVES + wave; idx < total; idx += gridDim.x * GS_WAVES) {
    const int gc = a.list[idx];
    float acc, ssum;
    gather_scored_node(a, gc, lane, s_row[wave], s_w[wave], acc, ssum);
    a.nb[(long)gc * 64 + lane] = acc;
    if (lane == 0 && a.sout) a.sout[gc] = ssum;
  }
}


// ------------------------------------------------------------------------------------------
// k_scored_tail: the end of a forward at SMALL batch sizes in one launch instead of three (k_gather_scored, the restricted
// k_node_update, k_score): after the last backward sweep mu[1] is read by the score head only, at the scored nodes, so their
// transposed aggregate (one wave per node, lane = channel: gather_scored_node, the code of k_gather_scored), the folded node
// update (q_chain, the arithmetic of k_node_update) and the score head (score_rows) run back to back on a tile of 16 scored
// nodes (the rows are still written to mu[1], as the three kernels did: inspection reads them, the score head does not); the scored nodes of the other layers go through the score head as in k_score, and
// the workgroup that finishes last turns the per-sample keys into decisions.  Same arithmetic per node as the three kernels:
// bit-identical scores (tests).  At B <= 8 each of the three launches was its own ramp (weights staged, one tile's dependent
// loads): 10 + 13 + 15 us at B = 1.
// Per workgroup (16 waves): waves 1..15 gather the 32 nodes of a tile into an LDS row buffer (two buffers: the next tile is
// gathered while wave 0 runs chain + score head of this one).
// ------------------------------------------------------------------------------------------
struct TailArgs { GSArgs g; FArgs f; ScoreArgs s; };
#define TAIL_WAVES 16
#define TAIL_TILE 16          // scored nodes of layer 1 per tile: half-filled MFMA tiles, but twice the workgroups gather in parallel and
                              // a wave gathers one node, not two or three in a row (B = 1: 38 -> see DESIGN 5.3)
#define TAIL_LDS_FLOATS (PackUpdL3::FLOATS + PackScore::FLOATS + 2 * TAIL_TILE * QROW + TAIL_WAVES * GS_MAXSLOTS * 2 + 16)

__global__ __launch_bounds__(TAIL_WAVES * 64, 1) void k_scored_tail(TailArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lds_sc = lds + PackUpdL3::FLOATS;
  float* rows = lds_sc + PackScore::FLOATS;                       // [2][TAIL_TILE][QROW]
  int* s_row = reinterpret_cast<int*>(rows + 2 * TAIL_TILE * QROW);      // [TAIL_WAVES][GS_MAXSLOTS]
  float* s_w = reinterpret_cast<float*>(s_row + TAIL_WAVES * GS_MAXSLOTS);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  copy_to_lds(lds + PackUpdL3::BA, a.f.u.pack + PackUpd::BA, 64);
  copy_to_lds(lds + PackUpdL3::BCB, a.f.u.pack + PackUpd::BCB, 64 + 64 + 128);
  copy_to_lds(lds + PackUpdL3::WAS3, a.f.u.pack + PackUpd::WAS3, 3 * 6144);
  copy_to_lds(lds_sc, a.s.pack, PackScore::FLOATS);
  const int total = *a.g.cnt;                                      // scored nodes of layer 1
  const int T1 = (total + TAIL_TILE - 1) / TAIL_TILE;
  // ---- layer 1: gather -> chain -> score, tile by tile
  int buf = 0;
  for (int tile = blockIdx.x; tile < T1; tile += gridDim.x, buf ^= 1) {
    float* rb = rows + buf * TAIL_TILE * QROW;
    if (wave >= 1) {
      for (int l = wave - 1; l < TAIL_TILE; l += TAIL_WAVES - 1) {
        const int idx = tile * TAIL_TILE + l;
        float* row = rb + l * QROW;
        if (idx < total) {
          const int gc = a.g.list[idx];
          const float lb = a.f.u.lb[gc], ub = a.f.u.ub[gc];
          const Ratio rt = compute_ratio(lb, ub);
          float acc = 0.0f, ssum = 0.0f;
          if (rt.live != 0.0f) gather_scored_node(a.g, gc, lane, s_row + wave * GS_MAXSLOTS, s_w + wave * GS_MAXSLOTS, acc, ssum);
          row[lane] = acc;
          if (lane == 0) *reinterpret_cast<f32x4*>(row + 64) = f32x4{__int_as_float(gc | (rt.amb != 0.0f ? (int)0x80000000 : 0)), rt.r0, rt.r1, ssum};
        }
      }
    }
    __syncthreads();                          // (first pass: also the weights; every pass: this tile's rows are in LDS, the previous tile's chain is done)
    if (wave == 0) {
      const int nvalid = total - tile * TAIL_TILE < TAIL_TILE ? total - tile * TAIL_TILE : TAIL_TILE;
      Frag E;
      q_chain<false>(a.f, lds, rb, nvalid, lane, [] {}, &E);
      const int j = lane & 31;
      const bool valid = j < nvalid;
      const int gc = valid ? (__float_as_int(rb[j * QROW + 64]) & 0x7fffffff) : 0;
      const float live = valid && node_is_live(a.f.u.lb[gc], a.f.u.ub[gc]) ? 1.0f : 0.0f;
      if (live == 0.0f) {                     // a dead node marked undecided: its row is zero by definition
#pragma unroll
        for (int R = 0; R < 32; ++R) FRAG_AT(E, R) = 0.0f;
      }
      score_rows(a.s, lds_sc, 0, gc, valid, live, E, lane);
    }
  }
  __syncthreads();
  // ---- the scored nodes of the other layers: the score head on their rows in memory (k_score's tiles), dealt from the LAST
  // workgroup backwards -- those have no layer-1 tile and start here at once
  long ntiles = 0;
  for (int k = 1; k < a.s.L; ++k) ntiles += (a.s.cnt[4 * k + 2] + 31) / 32;
  for (long tile = (long)(gridDim.x - 1 - blockIdx.x) * TAIL_WAVES + wave; tile < ntiles; tile += (long)gridDim.x * TAIL_WAVES) {
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
  score_finish(a.s);
}
