// gnnb_k_gather.h -- part of libgnnb.so, included by gnnb.hip (one translation unit; see its header comment).
// conv-edge message passing on the MFMA: k_gather, k_gather16, k_gather_input_update; the score head k_score.
#pragma once

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
  const uint2* taps3;    // 32-node tiles: the tap matrix in three bf16 pieces, [NCG][2 K2 slots][32 dst] x {p1 | p2 << 16, p3} (fill_gather_tables)
};

#define STAGE16_FLOATS (16 * 68 + 16)      // per-wave LDS image of the gathers' row stores (store_tile16 / frag_store_rows_*_staged<16>)
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
                                            __amdgpu_buffer_rsrc_t rsrc, int j, int wy0, int wx0, int Hs, int Ws, int lane, bool keep = false) {
  const int h = lane >> 5;
  if (!keep) {                                      // keep: the aggregate is added onto what X already holds
#pragma unroll
    for (int R = 0; R < 32; ++R) FRAG_AT(X, R) = 0.0f;
  }
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
                                                   const float* slb, const float* sub, int j, int wy0, int wx0, int Hs, int Ws, int lane,
                                                   bool keep = false, float* ssum = nullptr) {
  // ssum: also return s[dst node j] = sum of the tap weights of the live window slots = sum_n A[n', n] live_n, the scalar the
  // bias of the source rows' deferred projection is multiplied with (gnnb_pack.h; otherwise computed by k_livesum)
  const int h = lane >> 5;
  float sacc = 0.0f;
  if (!keep) {
#pragma unroll
    for (int R = 0; R < 32; ++R) FRAG_AT(X, R) = 0.0f;
  }
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
      c.v[u] = buf_load2(rsrc, e.x + lane_off);      // (a padding entry's BUF_OOB + lane_off is still past the buffer's end: the load returns 0)
      c.cr[u] = e.y;
    }
  };
  auto mma = [&](const Chunk& c, int s0) {
    float b[GATHER_CHS];
#pragma unroll
    for (int u = 0; u < GATHER_CHS; ++u) {
      b[u] = cm[c.cr[u] + j];
      X.t[0] = mfma32(c.v[u].x, b[u], X.t[0]);
      X.t[1] = mfma32(c.v[u].y, b[u], X.t[1]);
    }
    if (ssum) {
      // padding entries point at tap row 0 and must not count for the bias sum: only the chunk that holds entry n can have them -- a wave-uniform
      // test, so every other chunk adds its taps without the per-entry compare and select (2 of its ~6 vector instructions per k-step)
      if (2 * (s0 + GATHER_CHS) <= n) {
#pragma unroll
        for (int u = 0; u < GATHER_CHS; ++u) sacc += b[u];
      } else {
#pragma unroll
        for (int u = 0; u < GATHER_CHS; ++u) sacc += 2 * (s0 + u) + h < n ? b[u] : 0.0f;
      }
    }
  };
  load(cur, 0);
  const int npairs = K2e / (2 * GATHER_CHS);
  int s0 = 0;
  for (int pr = 0; pr < npairs; ++pr, s0 += 2 * GATHER_CHS) {
    load(nxt, s0 + GATHER_CHS);
    __builtin_amdgcn_sched_barrier(0);
    mma(cur, s0);
    __builtin_amdgcn_sched_barrier(0);
    load(cur, s0 + 2 * GATHER_CHS);
    __builtin_amdgcn_sched_barrier(0);
    mma(nxt, s0 + GATHER_CHS);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (K2e & GATHER_CHS) mma(cur, s0);
  if (ssum) *ssum = sacc + __shfl_xor(sacc, 32);
}


// The sparse walk on the bf16 matrix rate: the source rows come as three bf16 pieces (rows3, gnnb_dev.h), the taps too (DGather::taps3),
// and a group of 16 live window slots is ONE k-step of v_mfma_f32_32x32x16_bf16 per channel tile and product -- six products
// x1 w3 + x2 w2 + x3 w1 + x1 w2 + x2 w1 + x1 w1 as in gemm_w64_bf3 (fp32-grade sums): 12 MFMAs of 32 matrix-pipe cycles per 16 slots
// instead of 16 fp32 MFMAs of 64 VECTOR-pipe cycles.  Lane (m = lane & 31, kg = lane >> 5) takes table entries 16 G + 8 kg .. + 7: one
// 12-byte load per slot brings the three pieces of channels 2m, 2m + 1 (the fp32 form: 8 bytes per slot), four v_perm per operand
// put 8 slots of one channel and piece side by side.  Result layout = gather_tile_sparse's (X.t[t][r] = channel 2 k + t, k = (r & 3) + 8 (r >> 2) + 4 h).
// taps3: this channel group's [slot][32 dst] uint2 in LDS; cm: the fp32 tap matrix (bias sums only).
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
#define PERM_LO 0x05040100u      // v_perm_b32(hi_src, lo_src): {lo_src.lo16, hi_src.lo16}
#define PERM_HI 0x07060302u      //                              {lo_src.hi16, hi_src.hi16}
__device__ __forceinline__ void gather_tile_sparse_bf3(Frag& X, const float* cm, const uint2* taps3, const int2* ko, uint2* tab, int K2,
                                                       __amdgpu_buffer_rsrc_t rsrc3, const float* slb, const float* sub, int j, int wy0, int wx0,
                                                       int Hs, int Ws, int lane, bool keep = false, float* ssum = nullptr) {
  const int h = lane >> 5;
  float sacc = 0.0f;
  if (!keep) {
#pragma unroll
    for (int R = 0; R < 32; ++R) FRAG_AT(X, R) = 0.0f;
  }
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
    if (live) tab[n + __popcll(bal & ((1ull << lane) - 1ull))] = make_uint2((unsigned)row * ROW3_BYTES, (unsigned)sl * 32u);
    n += __popcll(bal);
  }
  const int npad = (n + 15) & ~15;
  for (int q = n + lane; q < npad + 16; q += 64) tab[q] = make_uint2(BUF_OOB, 0u);       // out-of-range offset: the load returns 0
  const int ng = npad >> 4;
  const unsigned lane_off = 12u * (unsigned)j;
  u32x3 v[8];
  auto load = [&](int G) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned o = tab[16 * G + 8 * h + u].x;
      v[u] = __builtin_amdgcn_raw_buffer_load_b96(rsrc3, o == BUF_OOB ? BUF_OOB : o + lane_off, 0, 0);
    }
  };
  load(0);
  for (int G = 0; G < ng; ++G) {
    // the taps of this lane's 8 slots for dst node j: B operands (three pieces); the fp32 taps for the bias sum
    u32x4 b1, b2, b3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned c0 = tab[16 * G + 8 * h + 2 * q].y, c1 = tab[16 * G + 8 * h + 2 * q + 1].y;
      const uint2 t0 = taps3[c0 + j], t1 = taps3[c1 + j];
      if (ssum) {        // (padding entries point at tap row 0)
        sacc += 16 * G + 8 * h + 2 * q < n ? cm[c0 + j] : 0.0f;
        sacc += 16 * G + 8 * h + 2 * q + 1 < n ? cm[c1 + j] : 0.0f;
      }
      b1[q] = __builtin_amdgcn_perm(t1.x, t0.x, PERM_LO);
      b2[q] = __builtin_amdgcn_perm(t1.x, t0.x, PERM_HI);
      b3[q] = __builtin_amdgcn_perm(t1.y, t0.y, PERM_LO);
    }
    const bf16x8 w1 = __builtin_bit_cast(bf16x8, b1), w2 = __builtin_bit_cast(bf16x8, b2), w3 = __builtin_bit_cast(bf16x8, b3);
    // A operands: channel 2m + t of the 8 slots, piece p (the loads of group G are waited for here); one channel tile at a time, the
    // next group's loads are issued as soon as the second tile's operands are built: they fly under its MFMAs
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      u32x4 a[3];
#pragma unroll
      for (int pp = 0; pp < 3; ++pp)
#pragma unroll
        for (int q = 0; q < 4; ++q) a[pp][q] = __builtin_amdgcn_perm(v[2 * q + 1][pp], v[2 * q][pp], t2 ? PERM_HI : PERM_LO);
      if (t2 == 1) {
        __builtin_amdgcn_sched_barrier(0);
        load(G + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      const bf16x8 x1 = __builtin_bit_cast(bf16x8, a[0]), x2 = __builtin_bit_cast(bf16x8, a[1]), x3 = __builtin_bit_cast(bf16x8, a[2]);
      X.t[t2] = mfma_bf16(x1, w3, X.t[t2]);
      X.t[t2] = mfma_bf16(x2, w2, X.t[t2]);
      X.t[t2] = mfma_bf16(x3, w1, X.t[t2]);
      X.t[t2] = mfma_bf16(x1, w2, X.t[t2]);
      X.t[t2] = mfma_bf16(x2, w1, X.t[t2]);
      X.t[t2] = mfma_bf16(x1, w1, X.t[t2]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (ssum) *ssum = sacc + __shfl_xor(sacc, 32);
}

// `sbase` = first row of this sample's source layer; must be built from wave-uniform values
// tab != nullptr: sparse walk (slb / sub = bounds of the source layer of this sample)
__device__ __forceinline__ void gather_dispatch(Frag& X, const float* cm, const int2* ko, const unsigned* kvo, const DGather& g,
                                                const float* sbase, int j, int wy0, int wx0, int lane,
                                                uint2* tab = nullptr, const float* slb = nullptr, const float* sub = nullptr, bool keep = false,
                                                float* ssum = nullptr) {
  const int uy = __builtin_amdgcn_readfirstlane(wy0), ux = __builtin_amdgcn_readfirstlane(wx0);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)sbase, 0, g.Ns * 256, 0x00020000);
  if (tab) { gather_tile_sparse(X, cm, ko, tab, g.K2, rsrc, slb, sub, j, uy, ux, g.Hs, g.Ws, lane, keep, ssum); return; }
  if (uy >= 0 && ux >= 0 && uy + g.WY <= g.Hs && ux + g.WX <= g.Ws)
    gather_tile<true>(X, cm, ko, kvo, g.K2, rsrc, j, uy, ux, g.Hs, g.Ws, lane, keep);
  else
    gather_tile<false>(X, cm, ko, kvo, g.K2, rsrc, j, uy, ux, g.Hs, g.Ws, lane, keep);
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
      float e0 = relu_max(fmaf(c.u[q], w[0][2], fmaf(c.x[q], w[0][1], fmaf(c.l[q], w[0][0], bias[0]))));
      float e1 = relu_max(fmaf(c.u[q], w[1][2], fmaf(c.x[q], w[1][1], fmaf(c.l[q], w[1][0], bias[1]))));
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
                                                     int Hs, int Ws, int lane, float* ssum = nullptr) {
  const int g = lane >> 4, i = lane & 15;
  float sacc = 0.0f;                                  // ssum: see gather_tile_sparse
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
      const unsigned o = e.x + lane_off;               // (a padding entry's BUF_OOB + lane_off is still past the buffer's end: the load returns 0)
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0);
      c.v[u] = f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
      c.cr[u] = e.y;
    }
  };
  auto mma = [&](const Chunk& c, int s0) {
    float b[GATHER_CHS16];
#pragma unroll
    for (int u = 0; u < GATHER_CHS16; ++u) {
      b[u] = cm[c.cr[u] + i];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mfma16(c.v[u][t], b[u], acc[t]);
    }
    if (ssum) {
      if (4 * (s0 + GATHER_CHS16) <= n) {              // (wave-uniform: no padding entry in this chunk, see gather_tile_sparse)
#pragma unroll
        for (int u = 0; u < GATHER_CHS16; ++u) sacc += b[u];
      } else {
#pragma unroll
        for (int u = 0; u < GATHER_CHS16; ++u) sacc += 4 * (s0 + u) + g < n ? b[u] : 0.0f;
      }
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
  if (ssum) {
    sacc += __shfl_xor(sacc, 16);
    *ssum = sacc + __shfl_xor(sacc, 32);
  }
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
        float e = relu_max(fmaf(c.u[q], w[t][2], fmaf(c.x[q], w[t][1], fmaf(c.l[q], w[t][0], bias[t]))));
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
  float* sout;                    // SPARSE: (B, N) bias-sum scalars s = sum_n A[n', n] live_n written beside nb (null: k_livesum has them)
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

// the aggregate of one 32-node tile into X (gather channel map), tap-count division included; ssum: the bias-sum scalar of the
// sparse walk (see gather_tile_sparse), normalised like the aggregate
template <bool EMBED, bool SPARSE>
__device__ __forceinline__ void gather_compute_tile(const GArgs& a, const TileCtx& tc, int sample, const float* lds_cm, const int2* lds_ko,
                                                    const unsigned* lds_kvo, uint2* tab, const EmbedLane& el, int lane, Frag& X, float& ssum) {
  const int j = lane & 31;
  ssum = 0.0f;
  const int wy0 = tc.by * a.g.ystep + a.g.ybase, wx0 = tc.bx * a.g.xstep + a.g.xbase;
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
                      a.src_lb + (long)sample * a.g.Ns, a.src_ub + (long)sample * a.g.Ns, false, a.sout ? &ssum : nullptr);
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
    ssum = ssum / freq;                      // the bias sum of a transposed conv edge is normalised like its aggregate
  }
}

// one tile of phase A: the aggregate rows of the tile's dst nodes that will be updated
template <bool EMBED, bool SPARSE>
__device__ __forceinline__ void gather_process_tile(const GArgs& a, const TileCtx& tc, int sample, const float* lds_cm, const int2* lds_ko,
                                                    const unsigned* lds_kvo, uint2* tab, const EmbedLane& el, int lane) {
  const int h = lane >> 5;
  const long gc = tc.sample * a.tm.N + tc.n;
  float ssum = 0.0f;
  bool need;
  if (a.need_scored) need = tc.valid && a.mask[tc.sample * a.R + a.off + tc.n] != 0.0f;
  else need = tc.valid && node_is_live(a.lb[gc], a.ub[gc]);    // (one load of k_classify's live flag instead: measured 1.7 % slower)
  if (!__any(need)) return;
  Frag X;
  gather_compute_tile<EMBED, SPARSE>(a, tc, sample, lds_cm, lds_ko, lds_kvo, tab, el, lane, X, ssum);
  if (need) frag_store_rows_gathered(X, a.nb, gc, h);
  if (SPARSE && need && h == 0 && a.sout) a.sout[gc] = ssum;
}

// The embedding on the matrix pipe: E0 = relu(inp_f [l, x, u] + b) is itself a K = 4 product ([l, x, u, 1] against [W | b]),
// and the result layout of v_mfma_f32_16x16x4_f32 (lane (i, g), register r = row 4g + r, column i) is the A-operand layout
// of the tap MFMA (lane (i, g) = channel of row i at the slot of k-index g) if the embedding MFMA's row 4g + r is the window
// slot that k-step 4 grp + r wants at k-index g, i.e. slot 16 grp + 4 r + g.  So per group of 4 k-steps: one scalar load per
// lane (lane group 0 / 1 / 2 reads l / x / u of its row's slot, group 3 supplies the 1 of the bias), 4 embedding MFMAs (one
// per 16-channel tile), 16 v_max, 16 tap MFMAs -- instead of 64 FMAs + 16 v_max + 12 loads per lane.  An out-of-range slot
// feeds zeros (including its "1"), so its embedding is relu(0) = 0.
// The scalars: every lane needs ONE of l / x / u (its k-index) per group, so it loads just that one through its own pointer,
// and a window of up to EMB_PRE groups (the 4x4 and 5x5 first convolutions: 3 and 5) is loaded whole before the first MFMA: one
// memory round trip per tile.  (Three buffer loads per lane and group, each prefetched one group = 20 MFMAs ahead, left the
// gather waves waiting on memory half of their time: SQ_WAIT_INST_ANY 50 %, profiles/r02c.)
#define EMB_PRE 6
__device__ __forceinline__ void gather_tile16_embed_mfma(f32x4 (&acc)[4], const float* cm, const int2* ko, int K2, const float* pl,
                                                         const float* px, const float* pu, const float (&bw)[4],
                                                         int wy0, int wx0, int Hs, int Ws, int lane) {
  const int m = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int origin = wy0 * Ws + wx0;
  const int sl = 4 * (m & 3) + (m >> 2);            // slot of embedding row m inside a group of 16
  const float* src = kq == 0 ? pl : kq == 1 ? px : pu;
  auto load = [&](int grp) -> float {
    const unsigned long long ev = reinterpret_cast<const unsigned long long*>(ko)[16 * grp + sl];
    const int ex = (int)(unsigned)ev, ey = (int)(unsigned)(ev >> 32);
    const int wy = wy0 + (ey & 0xffff), wx = wx0 + (ey >> 16);
    const bool inb = (unsigned)wy < (unsigned)Hs && (unsigned)wx < (unsigned)Ws;      // (table padding: 0x7fff, never in range)
    float v = inb ? 1.0f : 0.0f;                    // k-index 3: the 1 of the bias; an out-of-range slot feeds zeros
    if (inb && kq < 3) v = src[origin + ex];
    return v;
  };
  auto group = [&](float cur, int grp) {
    f32x4 e[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      e[t] = mfma16(cur, bw[t], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int r = 0; r < 4; ++r) e[t][r] = relu_max(e[t][r]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float b = cm[(4 * grp + r) * 64 + lane];
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = mfma16(e[t][r], b, acc[t]);
    }
  };
  const int ngrp = K2 / 4;
  if (ngrp <= EMB_PRE) {
    float ain[EMB_PRE];
#pragma unroll
    for (int g = 0; g < EMB_PRE; ++g) ain[g] = g < ngrp ? load(g) : 0.0f;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < EMB_PRE; ++g) {
      if (g < ngrp) group(ain[g], g);
      __builtin_amdgcn_sched_barrier(0);
    }
    return;
  }
  float ain = load(0);
  for (int grp = 0; grp < ngrp; ++grp) {
    const float cur = ain;
    if (grp + 1 < ngrp) ain = load(grp + 1);
    __builtin_amdgcn_sched_barrier(0);
    group(cur, grp);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// the aggregate of one 16-node tile (forward edges only: no tap-count division): acc[t][r] of lane (j, g') = channel 16 g' + 4 r + t
template <bool EMBED, bool SPARSE>
__device__ __forceinline__ void gather_compute_tile16(const GArgs& a, const TileCtx& tc, int sample, const float* lds_cm, const int2* lds_ko,
                                                      const unsigned* lds_kvo, uint2* tab, const float (&ew)[4][3], const float (&eb)[4], int lane,
                                                      f32x4 (&acc)[4], float& ssum) {
  ssum = 0.0f;
  const int wy0 = tc.by * a.g.ystep + a.g.ybase, wx0 = tc.bx * a.g.xstep + a.g.xbase;
  const int uy = __builtin_amdgcn_readfirstlane(wy0), ux = __builtin_amdgcn_readfirstlane(wx0);
  const bool interior = uy >= 0 && ux >= 0 && uy + a.g.WY <= a.g.Hs && ux + a.g.WX <= a.g.Ws;
  const float* cmt = lds_cm + tc.cg * a.g.K2 * 64;
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
      gather_tile16_embed_mfma(acc, cmt, lds_ko, a.g.K2, a.es.lb + sb, a.es.x + sb, a.es.ub + sb, bw, uy, ux, a.g.Hs, a.g.Ws, lane);
    } else if (interior) gather_tile16_embed<true>(acc, cmt, lds_ko, lds_kvo, a.g.K2, rl, rx, ru, ew, eb, uy, ux, a.g.Hs, a.g.Ws, lane);
    else gather_tile16_embed<false>(acc, cmt, lds_ko, lds_kvo, a.g.K2, rl, rx, ru, ew, eb, uy, ux, a.g.Hs, a.g.Ws, lane);
  } else {
    const float* sbase = a.mu_src + (long)sample * a.g.Ns * 64;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)sbase, 0, a.g.Ns * 256, 0x00020000);
    if (SPARSE) {
      const long sb = (long)sample * a.g.Ns;
      gather_tile16_sparse(acc, cmt, lds_ko, tab, a.g.K2, rsrc, a.src_lb + sb, a.src_ub + sb, uy, ux, a.g.Hs, a.g.Ws, lane,
                           a.sout ? &ssum : nullptr);
    } else if (interior) gather_tile16<true>(acc, cmt, lds_ko, lds_kvo, a.g.K2, rsrc, uy, ux, a.g.Hs, a.g.Ws, lane);
    else gather_tile16<false>(acc, cmt, lds_ko, lds_kvo, a.g.K2, rsrc, uy, ux, a.g.Hs, a.g.Ws, lane);
  }
}

// one 16-node tile of phase A
// The aggregate rows of a 16-node tile -> HBM.  Lane (j, g') holds channels 16 g' + 4 r + t of node j; stored straight from
// there one wave instruction writes 16-B pieces at a 64-B stride (every 64-B sector a quarter full, four times).  Through a
// 4.3 KB per-wave LDS image (rows padded to 68 floats) each instruction writes four whole 256-B rows instead.
__device__ __forceinline__ void store_tile16(const f32x4 (&acc)[4], float* stage, float* nb, long gc, bool need, int lane) {
  const int j = lane & 15, gq = lane >> 4;
  if (!stage) {
    if (need) {
      f32x4* p = reinterpret_cast<f32x4*>(nb + gc * 64 + 16 * gq);
#pragma unroll
      for (int r = 0; r < 4; ++r) p[r] = f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
    }
    return;
  }
  f32x4* srow = reinterpret_cast<f32x4*>(stage + j * 68 + 16 * gq);
#pragma unroll
  for (int r = 0; r < 4; ++r) srow[r] = f32x4{acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
  int* sgc = reinterpret_cast<int*>(stage + 16 * 68);
  if (gq == 0) sgc[j] = need ? (int)gc : -1;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 4 * i + gq;
    const int g = sgc[row];
    const f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * 68 + 4 * j);
    if (g >= 0) *reinterpret_cast<f32x4*>(nb + (long)g * 64 + 4 * j) = v;
  }
  __builtin_amdgcn_wave_barrier();
}

template <bool EMBED, bool SPARSE>
__device__ __forceinline__ void gather_process_tile16(const GArgs& a, const TileCtx& tc, int sample, const float* lds_cm, const int2* lds_ko,
                                                      const unsigned* lds_kvo, uint2* tab, const float (&ew)[4][3], const float (&eb)[4], int lane,
                                                      float* stage = nullptr) {
  const int gq = lane >> 4;
  const long gc = tc.sample * a.tm.N + tc.n;
  float ssum = 0.0f;
  bool need;
  if (a.need_scored) need = tc.valid && a.mask[tc.sample * a.R + a.off + tc.n] != 0.0f;
  else need = tc.valid && node_is_live(a.lb[gc], a.ub[gc]);
  if (!__any(need)) return;
  f32x4 acc[4];
  gather_compute_tile16<EMBED, SPARSE>(a, tc, sample, lds_cm, lds_ko, lds_kvo, tab, ew, eb, lane, acc, ssum);
  if (SPARSE && need && gq == 0 && a.sout) a.sout[gc] = ssum;
  store_tile16(acc, stage, a.nb, gc, need, lane);
}

// LDS image of a gather's tables: tap matrix, window offsets (two forms), tile table
struct GatherLds { float* cm; int2* ko; int* tt; unsigned* kvo; uint2* t3; };
// with3: the bf16 x 3 tap matrix (DGather::taps3, twice the bytes of cm) sits in front of the fp32 one
__device__ __forceinline__ GatherLds gather_lds(float* base, const DGather& g, int TPS, bool with3 = false) {
  GatherLds l;
  l.t3 = reinterpret_cast<uint2*>(base);
  l.cm = base + (with3 ? g.ncg_k2 * 128 : 0);
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
  // rounds of 8 tiles dealt round-robin over the workgroups in the XCD-grouped order (see k_gather16): the 64 workgroups of an XCD
  // work on neighbouring rounds, i.e. on a handful of samples whose source rows stay in that L2.  With one contiguous chunk per
  // workgroup (round 3) an XCD had 32 samples open at once and the counters saw every source row of the transposed layer-1
  // aggregate fetched twice (profiles/r04a_base_aggonly_pmc_summary.json: 150 MB for 110 MB of rows moved)
  int wg = blockIdx.x;
  const int nwg = gridDim.x;
  if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);
  const long nrounds = (a.ntiles + WAVES_MLP - 1) / WAVES_MLP;
  for (long r = wg; r < nrounds; r += nwg) {
    const long tile = r * WAVES_MLP + wave;
    if (tile >= a.ntiles) break;
    // tile, sample and t are wave-uniform by construction; make them provably so (scalar registers, scalar base address)
    const int sample = tile_sample(a.tm, tile);
    const int t = __builtin_amdgcn_readfirstlane((int)(tile - (long)sample * a.tm.TPS));
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
#ifdef GATHER_STAGGER      // dev: start the waves of a workgroup GATHER_STAGGER x 64 cycles apart (they otherwise run their phases in lockstep)
  for (int w_ = 0; w_ < __builtin_amdgcn_readfirstlane(wave); ++w_) __builtin_amdgcn_s_sleep(GATHER_STAGGER);
#endif
  // per wave, behind the shared tables: the staging image of store_tile16, then (SPARSE) the table of the live window slots
  float* wsc = reinterpret_cast<float*>(gl.kvo + ((gather_slots(a.g.K2, 16) + 3) & ~3));
  float* stage = wsc + wave * STAGE16_FLOATS;
  uint2* tab = reinterpret_cast<uint2*>(wsc + WAVES_MLP * STAGE16_FLOATS) + wave * (4 * a.g.K2 + 32);
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
  FT_DECL;
  for (long r = wg; r < nrounds; r += nwg) {
    const long tile = r * WAVES_MLP + wave;
    if (tile >= a.ntiles) break;
    const int sample = tile_sample(a.tm, tile);
    const int t = __builtin_amdgcn_readfirstlane((int)(tile - (long)sample * a.tm.TPS));
    const TileCtx tc = block_decode(a.tm, gl.tt, sample, t, j);
#ifdef FUSED_TIMING
    {
      FT_MARK(0);                               // loop overhead + decode
      const int gq = lane >> 4;
      const long gc = tc.sample * a.tm.N + tc.n;
      float ssum = 0.0f;
      const bool need = tc.valid && node_is_live(a.lb[gc], a.ub[gc]);
      if (!__any(need)) { FT_MARK(1); continue; }
      FT_MARK(1);                               // bounds of the dst nodes (one memory round trip)
      f32x4 acc[4];
      gather_compute_tile16<EMBED, SPARSE>(a, tc, sample, gl.cm, gl.ko, gl.kvo, tab, ew, eb, lane, acc, ssum);
      FT_MARK(2);                               // walk
      if (need) {
        f32x4* p = reinterpret_cast<f32x4*>(a.nb + gc * 64 + 16 * gq);
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) p[r4] = f32x4{acc[0][r4], acc[1][r4], acc[2][r4], acc[3][r4]};
        if (SPARSE && gq == 0 && a.sout) a.sout[gc] = ssum;
      }
      FT_MARK(3);                               // store issue
#if FUSED_TIMING > 1
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      FT_MARK(4);                               // store completion (perturbs: the next tile's loads wait for it anyway)
#endif
    }
#else
    gather_process_tile16<EMBED, SPARSE>(a, tc, sample, gl.cm, gl.ko, gl.kvo, tab, ew, eb, lane, stage);
#endif
  }
#ifdef FUSED_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  FT_MARK(5);
  if (FUSED_TIMING < 3 && !EMBED && !SPARSE) FT_FLUSH();       // (the dense layer-1 gather only)
#endif
}

struct GIArgs {
  const float* pack_pre;    // PackPreInp
  const float* pack;        // PackUpdInp (gather variant)
  const float *lb, *ub;     // input bounds, flat (B*N0)
  const float* mu_src; const float* sarr; float* mu; long ntiles; DTileMap tm; DGather g;
  const float *src_lb, *src_ub;     // SPARSE: bounds of ReLU layer 1 (the rows of its dead nodes are zero and skipped)
  int s_from_gather;                // SPARSE: the bias-sum scalar comes out of this kernel's own gather instead of sarr (k_livesum)
  const void* mu_src3;              // SPARSE: the source rows as three bf16 pieces (rows3) -> the aggregate runs on the bf16 matrix rate; null: fp32 rows, fp32 MFMAs
};

// input layer: E_0 = relu(Q + inp_b2[:, 64:] . (A_1^T mu_1)),  Q = inp_b2[:, :64] . inp_b_1(relu(inp_b([l0,u0]))) + b;
// mu_0 = inp_b2_2(E_0) is deferred into the next round's forward update of ReLU layer 1 (gnnb_pack.h).
// graph_conv.py:361-385; the aggregate, the feature chain and the update stay in registers.
// one tile of the fused input-layer update; lds_upd / lds_pre: PackUpdInp / PackPreInp in LDS
template <bool SPARSE, bool BF3, bool R3 = false>
__device__ __forceinline__ void input_update_tile(const GIArgs& a, const TileCtx& tc, int sample, const float* lds_upd, const float* lds_pre,
                                                  const GatherLds& gl, uint2* tab, int lane) {
  const int h = lane >> 5, j = lane & 31;
  if (!__any(tc.valid)) return;
  const long gc = tc.sample * a.tm.N + tc.n;
  const int wy0 = tc.by * a.g.ystep + a.g.ybase, wx0 = tc.bx * a.g.xstep + a.g.xbase;
  float x[1];
  x[0] = h ? a.ub[gc] : a.lb[gc];
  Frag H0;
  frag_bias(H0, lds_pre + PackPreInp::B1, h);
  gemm_small<1>(lds_pre + PackPreInp::W1, lane, H0, x);
  frag_relu(H0);
  Frag H;                                  // inp_b_1 and the first half of inp_b2 are folded into one 64x64 map
  frag_bias(H, lds_pre + PackPreInp::B2, h);
#if defined(GIU_ABL) && (GIU_ABL & 2)      // dev, timing only: no feature chain
#else
  if (BF3) gemm_w64_bf3<1>(lds_pre + PackPreInp::W23, lane, H, [&](int s) { return FRAG_AT(H0, s); });
  else gemm_w64<32>(lds_pre + PackPreInp::W2, lane, H, [&](int s) { return FRAG_AT(H0, s); });
#endif
  float ssum = 0.0f;
  const bool own_s = SPARSE && a.s_from_gather;
  if (!own_s) {                              // bias term of the projection deferred in the rows of mu_1
    const float xs[1] = {h ? 0.0f : a.sarr[gc]};
    gemm_small<1>(lds_upd + PackUpdInp::VC, lane, H, xs);
  }
  // The aggregate already went through inp_b2[:, 64:].bc4_1.W on the producer side, with its rows permuted to this fragment
  // layout (PackPostInp::WPG), register for register: the gather accumulates straight onto H (one fragment less alive).
#if defined(GIU_ABL) && (GIU_ABL & 4)      // dev, timing only: no gather
  if (wy0 > 100000)
#endif
  if (SPARSE && R3) {
    const int uy = __builtin_amdgcn_readfirstlane(wy0), ux = __builtin_amdgcn_readfirstlane(wx0);
    const __amdgpu_buffer_rsrc_t rsrc3 = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(a.mu_src3) + (long)sample * a.g.Ns * ROW3_BYTES),
                                                                           0, a.g.Ns * ROW3_BYTES, 0x00020000);
    gather_tile_sparse_bf3(H, gl.cm + tc.cg * a.g.K2 * 64, gl.t3 + (long)tc.cg * a.g.K2 * 64, gl.ko, tab, a.g.K2, rsrc3,
                           a.src_lb + (long)sample * a.g.Ns, a.src_ub + (long)sample * a.g.Ns, j, uy, ux, a.g.Hs, a.g.Ws, lane, true,
                           &ssum);
  } else if (SPARSE)
    gather_dispatch(H, gl.cm + tc.cg * a.g.K2 * 64, gl.ko, gl.kvo, a.g, a.mu_src + (long)sample * a.g.Ns * 64, j, wy0, wx0, lane, tab,
                    a.src_lb + (long)sample * a.g.Ns, a.src_ub + (long)sample * a.g.Ns, true, &ssum /* (always: a pointer chosen at run time would pin `ssum` to the stack; it is only used when own_s) */);
  else
    gather_dispatch(H, gl.cm + tc.cg * a.g.K2 * 64, gl.ko, gl.kvo, a.g, a.mu_src + (long)sample * a.g.Ns * 64, j, wy0, wx0, lane, nullptr,
                    nullptr, nullptr, true);
  if (own_s) {
    const float xs[1] = {h ? 0.0f : ssum};
    gemm_small<1>(lds_upd + PackUpdInp::VC, lane, H, xs);
  }
  frag_relu(H);
#if defined(GIU_ABL) && (GIU_ABL & 1)      // dev, timing only: no row stores
  if (tc.valid && FRAG_AT(H, 0) > 1e30f) frag_store_rows(H, a.mu, gc, h);
#else
  if (tc.valid) frag_store_rows(H, a.mu, gc, h);
#endif
}

// R3: the source rows are three bf16 pieces (GIArgs::mu_src3) and the aggregate runs on the bf16 matrix rate (gather_tile_sparse_bf3); its
// 24-byte loads and operand pieces need ~165 registers: 12 waves per workgroup (3 per SIMD), one workgroup per CU.  !R3: fp32 rows,
// fp32 MFMAs, two 8-wave workgroups per CU (128 registers).
#define GIU_R3_WAVES 12
template <bool SPARSE, bool BF3, bool R3 = false>
__global__ __launch_bounds__(R3 ? GIU_R3_WAVES * 64 : WG_MLP, R3 ? 3 : 2) void k_gather_input_update(GIArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int NW = R3 ? GIU_R3_WAVES : WAVES_MLP;
  float* lds_pre = lds + PackUpdInp::FLOATS;
  const GatherLds gl = gather_lds(lds_pre + PackPreInp::FLOATS, a.g, a.tm.TPS, R3);
  stage_gather(gl.cm, gl.ko, gl.tt, gl.kvo, a.g, a.tm.TPS);
  if (R3) copy_to_lds(reinterpret_cast<float*>(gl.t3), reinterpret_cast<const float*>(a.g.taps3), a.g.ncg_k2 * 128);
  copy_to_lds(lds_pre, a.pack_pre, PackPreInp::FLOATS);
  stage_pack(lds, a.pack, PackUpdInp::FLOATS);
  const int lane = threadIdx.x & 63, j = lane & 31, wave = threadIdx.x >> 6;
  uint2* tab = reinterpret_cast<uint2*>(gl.kvo + ((gather_slots(a.g.K2, 32) + 1) & ~1)) + wave * (2 * a.g.K2 + 32);      // SPARSE
  // rounds of NW tiles round-robin over the workgroups, XCD-grouped (see k_gather / k_gather16): an XCD keeps a handful of samples
  // open instead of 32 (contiguous chunks: the counters saw the rows of layer 1 fetched 2.2 times, profiles/r04a_base_aggonly_*)
  int wg = blockIdx.x;
  const int nwg = gridDim.x;
  if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);
  const long nrounds = (a.ntiles + NW - 1) / NW;
  for (long r = wg; r < nrounds; r += nwg) {
    const long tile = r * NW + wave;
    if (tile >= a.ntiles) break;
    const int sample = tile_sample(a.tm, tile);
    const int t = __builtin_amdgcn_readfirstlane((int)(tile - (long)sample * a.tm.TPS));
    const TileCtx tc = block_decode(a.tm, gl.tt, sample, t, j);
    input_update_tile<SPARSE, BF3, R3>(a, tc, sample, lds, lds_pre, gl, tab, lane);
  }
}

struct ScoreArgs {        // every ReLU layer in one launch
  const float* pack; float* scores;
  int L, R;
  const float* mu[MAXL]; const int* list[MAXL];
  const float* lb[MAXL]; const float* ub[MAXL];
  const int* cnt;         // cnt[4k + 2] = number of scored nodes of layer k
  int* cnt_all;           // the forward's whole counter block (64 ints): this is the last kernel that reads it and leaves it zero
  int N[MAXL], off[MAXL]; // nodes per sample in layer k, offset of layer k in the flat ReLU index
  // the decision (graph_score.py:41-47: first maximal score -> [layer, idx]) in the same launch: every tile folds its scores into
  // best[b] = max over (order-preserving score bits << 32 | ~flat index) -- ties go to the lower index -- and the workgroup that
  // finishes last turns the B keys into decisions.  best / done are zeroed by k_classify.
  unsigned long long* best; int* done; int* dec; int B, n_relu; int cum[16];
};

// score = fscore(relu(fnode(mu_g))) for the nodes g whose BaB mask is -1 (the rest stays -inf)    graph_conv.py:445-450
// the rows hold E_g with mu_g = (Wp.E_g + bp).live: fnode is pre-multiplied by Wp, fnode.bp.live enters as a small k-step
// one tile (32 scored nodes `list[32 t ..]` of layer k); lds: PackScore
// the score head on the rows X of a tile (lane (j, h): node gc of layer k, `live` its [r0 != 0]); writes the scores and folds them
// into the per-sample keys
__device__ __forceinline__ void score_rows(const ScoreArgs& a, const float* lds, int k, long gc, bool valid, float live, const Frag& X, int lane) {
  const int h = lane >> 5;
  const float bs = lds[PackScore::BS];
  const int N = a.N[k];
  const long b = gc / N;
  Frag H;
  frag_bias(H, lds + PackScore::B1, h);
  {
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
  const int flat = a.off[k] + (int)(gc - b * N);
  if (valid && h == 0) a.scores[b * a.R + flat] = part + bs;
  // fold into the per-sample best key: lanes of one sample are reduced in the wave first (a tile spans at most a few samples)
  const unsigned u = __float_as_uint(part + bs);
  const unsigned ord = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  unsigned long long key = (valid && h == 0) ? (((unsigned long long)ord << 32) | (0xffffffffu - (unsigned)flat)) : 0ull;
  unsigned long long todo = __ballot(key != 0ull);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const long bl = __shfl((int)b, leader);                 // (B < 2^31)
    const bool mine = key != 0ull && b == bl;
    unsigned long long m = mine ? key : 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long other = ((unsigned long long)__shfl_xor((unsigned)(m >> 32), o) << 32) | (unsigned)__shfl_xor((unsigned)m, o);
      m = other > m ? other : m;
    }
    if (lane == leader) atomicMax(a.best + bl, m);
    todo &= ~__ballot(mine);
  }
}

__device__ __forceinline__ void score_tile(const ScoreArgs& a, const float* lds, int k, const int* list, int count, long t, int lane) {
  const int h = lane >> 5, j = lane & 31;
  const long idx = t * 32 + j;
  const bool valid = idx < count;
  const long gc = list[valid ? idx : 0];
  Frag X;
  frag_load_rows(X, a.mu[k], gc, h);
  const float live = node_is_live(a.lb[k][gc], a.ub[k][gc]) ? 1.0f : 0.0f;
  if (live == 0.0f) {                      // a dead node marked undecided: its row is zero by definition (and need not be in memory)
#pragma unroll
    for (int R = 0; R < 32; ++R) FRAG_AT(X, R) = 0.0f;
  }
  score_rows(a, lds, k, gc, valid, live, X, lane);
}

// the workgroup that finishes last converts the per-sample keys into decisions and leaves the forward's counters zero
__device__ __forceinline__ void score_finish(const ScoreArgs& a) {
  __shared__ int last;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    last = atomicAdd(a.done, 1) == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (threadIdx.x < 64) a.cnt_all[threadIdx.x] = 0;      // every workgroup has read its counts: the block is zero again for the next forward
  for (int b = threadIdx.x; b < a.B; b += blockDim.x) {
    const unsigned long long key = __hip_atomic_load(a.best + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int lay = -1, idx = -1;
    if (key != 0ull) {
      const int flat = (int)(0xffffffffu - (unsigned)key);
      lay = 0;
      while (lay < a.n_relu - 1 && a.cum[lay] <= flat) ++lay;
      idx = lay == 0 ? flat : flat - a.cum[lay - 1];
    }
    a.dec[b * 2] = lay;
    a.dec[b * 2 + 1] = idx;
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
  score_finish(a);
}
