// gnnb_k_fused.h -- part of libgnnb.so, included by gnnb.hip (one translation unit; see its header comment).
// k_gather_update: one half-pass over a conv edge in ONE kernel -- the MFMA gather of the neighbour aggregate and the folded node
// update of the nodes it feeds, with the aggregate rows never written to HBM (reference graph_conv.py:110-181 forward,
// :299-349 backward: aggregate -> update of one layer).
//
// A gather tile is a block of 16 / 32 dst nodes of one sample, of which only the live ones (~55 %) are updated, and the node
// MLP wants 32 nodes on its lanes.  So every wave keeps ONE pending chain tile P in registers and compacts the live nodes of
// each gathered tile into it, register for register, with ds_bpermute (the LDS crossbar, no LDS memory): the destination
// lane of pending slot d pulls the accumulator registers of the source lane that holds the (d - cnt)-th live node of the tile
// (a 32-entry per-wave table in LDS, written by the live lanes at their rank, turns ranks into lanes).  When P is full the
// chain runs on it and writes the rows of mu; nodes of the tile that did not fit start the next P; what is left in P after a
// wave's last tile goes through the chain as a partly filled tile.  The channel order of P is whatever the gather's
// accumulators hold -- the first layer's weights are packed for that order (PackUpd::WAS3_G.. / WA1S3_G..).
// Ambiguous nodes (r0 != r1, a cached P' row: 3-19 % of the live ones) ride in the same tiles as the others:
//   Wa.[r0 x, r1 x] = WAS.(r0 x) + Wa[:, 64:].((r1 - r0) x),     WAS = Wa[:, :64] + Wa[:, 64:],
// a third bf16x3 block that a tile runs only if it holds such a node; for every other node of that tile it adds exact zeros,
// so a node's result does not depend on what it shares a tile with (batched == per-sample, bit for bit).
// (First version: ambiguous nodes and the partly filled tiles went to HBM and through a per-workgroup tail pass behind a
// barrier -- 45 % of the kernel's time for a few tiles per workgroup.)
#pragma once

struct FArgs {
  GArgs g;             // the gather (k_gather / k_gather16 arguments; g.nb is unused; g.sout != null: the sparse walk computes the bias sums)
  UpdArgs u;           // the node update (k_node_update arguments; list0 / list1 / cnt0 / cnt1, nb are unused)
  int sw_from_gather;  // 1: the bias-sum scalar of a node comes out of its (sparse) gather; 0: u.sarr holds it (table / k_livesum)
};

#ifndef FUSED_WAVES
#define FUSED_WAVES 8       // two waves per SIMD at up to 256 VGPRs (three at 168 spill 30-90 registers: measured 15-40 % slower)
#endif
#ifndef FUSED_ABL
#define FUSED_ABL 0      // dev (tools/mk_abl_fused.sh): 1 no row loads, 2 no gather MFMAs, 4 no chain, 8 no compaction -- results are wrong
#endif
#define FUSED_MAX_LDS (160 * 1024)

#ifndef FW_NB
#define FW_NB 3
#endif
// FW_NB: chunks (4 k-steps each) of source-row loads a wave keeps in flight
#define FW_MAXIT 4       // 64-slot sweeps over the window of a tile (the host does not fuse edges with larger windows)
// per-wave LDS scratch of the fused kernel: rank -> lane table (32 ints), then the slot table of the tile being walked
// ((lanes == 16 ? 4 : 2) * K2 window slots + the always-masked entries the in-flight chunks past the end read)
__host__ __device__ inline size_t fused_wave_bytes(int K2, int lanes, bool sparse) {
  (void)sparse;
  return 32 * 4 + (size_t)((lanes == 16 ? 4 : 2) * K2 + 16 * (2 * FW_NB + 1)) * 8;
}

// ---- the gather of the fused kernel: a software-pipelined walk over a per-tile slot table --------------------------------
// The stand-alone gathers keep one chunk of 4 k-steps in flight behind the one being multiplied and start a tile by waiting for
// its nodes' bounds: per tile two to three dependent memory round trips against ~1 us of MFMAs, which only many waves per
// SIMD hide -- and the fused kernel has three.  Here a wave (a) fetches everything the NEXT tile needs to get started (its
// nodes' bounds and, for a sparse walk, the bounds of every window slot) while it multiplies the current one, (b) turns the
// window into a table of {row byte offset, tap-matrix row} of the slots that are inside the layer and, behind a ReLU layer, live
// (a dead source row is exactly zero), and (c) walks the table with FW_NB chunks of row loads in flight.
template <int NIT>           // NIT: 64-slot sweeps that cover the window (1, 2 or 4)
struct FwNext {
  int sample, t;           // wave-uniform tile coordinates
  bool has;
  float lb, ub;            // bounds of this lane's dst node
  int row[NIT];            // source row of window slot 64 it + lane; -1: outside the layer / padding
  float sl[NIT], su[NIT];  // SPARSE: bounds of that source node
};

template <int LANES, bool SPARSE, int NIT>
__device__ __forceinline__ void fw_prefetch(const GArgs& a, const GatherLds& gl, long tile, int lane, int jn, FwNext<NIT>& nx) {
  nx.has = tile < a.ntiles;
  if (!nx.has) return;
  nx.sample = __builtin_amdgcn_readfirstlane((int)(tile / a.tm.TPS));
  nx.t = __builtin_amdgcn_readfirstlane((int)(tile - (long)nx.sample * a.tm.TPS));
  const TileCtx tc = block_decode(a.tm, gl.tt, nx.sample, nx.t, jn);
  const long gc = tc.sample * a.tm.N + tc.n;
  nx.lb = a.lb[gc];
  nx.ub = a.ub[gc];
  const int wy0 = __builtin_amdgcn_readfirstlane(tc.by * a.g.ystep + a.g.ybase), wx0 = __builtin_amdgcn_readfirstlane(tc.bx * a.g.xstep + a.g.xbase);
  const int origin = wy0 * a.g.Ws + wx0;
  const int nslots = (LANES == 16 ? 4 : 2) * a.g.K2;
  const float* slb = SPARSE ? a.src_lb + (long)nx.sample * a.g.Ns : nullptr;
  const float* sub = SPARSE ? a.src_ub + (long)nx.sample * a.g.Ns : nullptr;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    nx.row[it] = -1;
    nx.sl[it] = nx.su[it] = 0.0f;
    if (it * 64 < nslots) {
      const int sl = it * 64 + lane;
      const unsigned long long ev = reinterpret_cast<const unsigned long long*>(gl.ko)[sl < nslots ? sl : nslots];     // (entry nslots: padding)
      const int ex = (int)(unsigned)ev, ey = (int)(unsigned)(ev >> 32);
      const int wy = wy0 + (ey & 0xffff), wx = wx0 + (ey >> 16);
      const bool inb = sl < nslots && (unsigned)wy < (unsigned)a.g.Hs && (unsigned)wx < (unsigned)a.g.Ws;     // (table padding: 0x7fff, never in range)
      nx.row[it] = inb ? origin + ex : -1;
      if (SPARSE) {
        const int rr = inb ? origin + ex : 0;
        nx.sl[it] = slb[rr];
        nx.su[it] = sub[rr];
      }
    }
  }
}

// slot table of the current tile: entries {row byte offset, tap-matrix row offset} of its usable window slots; returns their number
template <int LANES, bool SPARSE, int NIT>
__device__ __forceinline__ int fw_build_tab(const FwNext<NIT>& cu, uint2* tab, int nslots, int lane) {
  int n = 0;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    if (it * 64 < nslots) {
      const bool use = cu.row[it] >= 0 && (!SPARSE || node_is_live(cu.sl[it], cu.su[it]));
      const unsigned long long bal = __ballot(use);
      if (use) tab[n + __popcll(bal & ((1ull << lane) - 1ull))] = make_uint2((unsigned)cu.row[it] * 256u, (unsigned)(it * 64 + lane) * (unsigned)LANES);
      n += __popcll(bal);
    }
  }
  constexpr int CS = LANES == 16 ? 16 : 8;                // slots per chunk of 4 k-steps
  const int npad = (n + CS - 1) / CS * CS;
  for (int q = n + lane; q < npad + 2 * FW_NB * CS; q += 64) tab[q] = make_uint2(BUF_OOB, 0u);   // out-of-range offset: the load returns 0
  __builtin_amdgcn_wave_barrier();
  return n;
}

struct FwChunk16 { f32x4 v[4]; unsigned cr[4]; };
struct FwChunk32 { float2 v[4]; unsigned cr[4]; };

// 16-node tile: acc[t][r] of lane (i, g) = channel 16 g + 4 r + t of dst node i (see gather_tile16).  `after_issue` runs once the
// first FW_NB chunks are on their way (the next tile's prefetch goes there); ssum: see gather_tile_sparse.
template <bool want_s, class F>
__device__ __forceinline__ void fw_walk16(f32x4 (&acc)[4], const float* cm, const uint2* tab, int n, __amdgpu_buffer_rsrc_t rsrc, int lane,
                                          float& ssum, F after_issue) {
  const int g = lane >> 4, i = lane & 15;
  float sacc = 0.0f;
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const unsigned lane_off = 16u * (unsigned)i;
  const int nch = (n + 15) / 16;
  auto load = [&](FwChunk16& c, int ch) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint2 e = tab[4 * (4 * ch + u) + g];
      const unsigned o = (e.x == BUF_OOB || (FUSED_ABL & 1)) ? BUF_OOB : e.x + lane_off;
      const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0);
      c.v[u] = f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
      c.cr[u] = e.y;
    }
  };
  auto mma = [&](const FwChunk16& c, int ch) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float b = cm[c.cr[u] + i];
      if (want_s) sacc += 4 * (4 * ch + u) + g < n ? b : 0.0f;        // (padding entries point at tap row 0)
      if (FUSED_ABL & 2) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t][u] += c.v[u][t] * b;
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma16(c.v[u][t], b, acc[t]);
      }
    }
  };
  FwChunk16 b0, b1, b2;
  load(b0, 0);
  load(b1, 1);
  if (FW_NB == 3) load(b2, 2);
  after_issue();
  // No control flow inside the ring: with conditional chunks the wait-count pass has to assume at the loop head that the chunk
  // it is about to multiply was the LAST one issued, i.e. it drains every load in flight once per round.  Whole rounds of FW_NB
  // chunks run in the loop (loads past the end read always-masked table entries: no memory traffic), the rest after it.
  const int full = nch / FW_NB;
  int c = 0;
  for (int k = 0; k < full; ++k, c += FW_NB) {
    __builtin_amdgcn_sched_barrier(0);
    mma(b0, c);
    __builtin_amdgcn_sched_barrier(0);
    load(b0, c + FW_NB);
    __builtin_amdgcn_sched_barrier(0);
    mma(b1, c + 1);
    __builtin_amdgcn_sched_barrier(0);
    load(b1, c + FW_NB + 1);
    if (FW_NB == 3) {
      __builtin_amdgcn_sched_barrier(0);
      mma(b2, c + 2);
      __builtin_amdgcn_sched_barrier(0);
      load(b2, c + 5);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if (c < nch) mma(b0, c);
  if (FW_NB == 3 && c + 1 < nch) mma(b1, c + 1);
  if (want_s) {
    sacc += __shfl_xor(sacc, 16);
    ssum = sacc + __shfl_xor(sacc, 32);
  }
}

// 32-node tile: X in the gather channel map (see gather_tile)
template <bool want_s, class F>
__device__ __forceinline__ void fw_walk32(Frag& X, const float* cm, const uint2* tab, int n, __amdgpu_buffer_rsrc_t rsrc, int lane,
                                          float& ssum, F after_issue) {
  const int h = lane >> 5, j = lane & 31;
  float sacc = 0.0f;
#pragma unroll
  for (int R = 0; R < 32; ++R) FRAG_AT(X, R) = 0.0f;
  const unsigned lane_off = 8u * (unsigned)j;
  const int nch = (n + 7) / 8;
  auto load = [&](FwChunk32& c, int ch) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint2 e = tab[2 * (4 * ch + u) + h];
      c.v[u] = buf_load2(rsrc, (e.x == BUF_OOB || (FUSED_ABL & 1)) ? BUF_OOB : e.x + lane_off);
      c.cr[u] = e.y;
    }
  };
  auto mma = [&](const FwChunk32& c, int ch) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float b = cm[c.cr[u] + j];
      if (want_s) sacc += 2 * (4 * ch + u) + h < n ? b : 0.0f;
      if (FUSED_ABL & 2) {
        X.t[0][u] += c.v[u].x * b;
        X.t[1][u] += c.v[u].y * b;
      } else {
        X.t[0] = mfma32(c.v[u].x, b, X.t[0]);
        X.t[1] = mfma32(c.v[u].y, b, X.t[1]);
      }
    }
  };
  FwChunk32 b0, b1, b2;
  load(b0, 0);
  load(b1, 1);
  if (FW_NB == 3) load(b2, 2);
  after_issue();
  // No control flow inside the ring: with conditional chunks the wait-count pass has to assume at the loop head that the chunk
  // it is about to multiply was the LAST one issued, i.e. it drains every load in flight once per round.  Whole rounds of FW_NB
  // chunks run in the loop (loads past the end read always-masked table entries: no memory traffic), the rest after it.
  const int full = nch / FW_NB;
  int c = 0;
  for (int k = 0; k < full; ++k, c += FW_NB) {
    __builtin_amdgcn_sched_barrier(0);
    mma(b0, c);
    __builtin_amdgcn_sched_barrier(0);
    load(b0, c + FW_NB);
    __builtin_amdgcn_sched_barrier(0);
    mma(b1, c + 1);
    __builtin_amdgcn_sched_barrier(0);
    load(b1, c + FW_NB + 1);
    if (FW_NB == 3) {
      __builtin_amdgcn_sched_barrier(0);
      mma(b2, c + 2);
      __builtin_amdgcn_sched_barrier(0);
      load(b2, c + 5);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  if (c < nch) mma(b0, c);
  if (FW_NB == 3 && c + 1 < nch) mma(b1, c + 1);
  if (want_s) ssum = sacc + __shfl_xor(sacc, 32);
}

// the folded node update on a full or partly filled pending tile (see node_update_loop; lds: PackUpdF3 image, POST block behind it)
template <bool POST>
__device__ __forceinline__ void fused_chain(const FArgs& a, const float* lds, Frag& P, int gc_amb, float r0, float r1, float sw, bool valid, int lane) {
  const int h = lane >> 5;
  const int gc = gc_amb & 0x7fffffff;              // bit 31: the node is ambiguous (beta > 0: it has a cached P' row)
  Frag H, H2;
  frag_bias(H, lds + PackUpdF3::BA, h);
  {
    const float x[1] = {(h ? r1 : r0) * sw};            // + s.(r0 Wa0.bp + r1 Wa1.bp): the bias of the source rows' deferred projection
    gemm_small<1>(lds + PackUpdF3::VAW, lane, H, x);
  }
  gemm_w64_bf3<1>(lds + PackUpdF3::WAS3, lane, H, [&](int s) { return FRAG_AT(P, s) * r0; });
  const bool amb = valid && gc_amb < 0;
  if (__any(amb)) {
    const float dr = valid ? r1 - r0 : 0.0f;
    gemm_w64_bf3<1>(lds + PackUpdF3::WA1S3, lane, H, [&](int s) { return FRAG_AT(P, s) * dr; });
  }
  // P' of an ambiguous node: its cached row (k_pre); of every other node: the bias row
  frag_load_rowptr(H2, amb ? a.u.P + (long)gc * 64 : a.u.pack + PackUpd::BCBROW, h);
  frag_relu(H);
  gemm_w64_bf3<1>(lds + PackUpdF3::WCB3, lane, H2, [&](int s) { return FRAG_AT(H, s); });
  frag_relu(H2);
  if (valid) {
    if (frag_has_nan(H2)) atomicOr(a.u.status, 1);
    if (a.u.mu) frag_store_rows(H2, a.u.mu, gc, h);
  }
  if (POST) {
#pragma unroll
    for (int R = 0; R < 32; ++R) FRAG_AT(H, R) = 0.0f;
    gemm_w64_bf3<1>(lds + PackUpdF3::FLOATS, lane, H, [&](int s) { return FRAG_AT(H2, s); });
    if (valid) frag_store_rows(H, a.u.post, gc, h);
  }
}

// LANES: dst nodes per gather tile (16: forward edges on the 16x16x4 MFMA, 32: 32x32x2).  SRC: 0 dense source rows, 1 sparse walk
// (the source is a ReLU layer), 2 round-0 embedding computed in the gather.  POST: see UpdArgs.
template <int LANES, int SRC, bool POST, int NIT = 1>
__global__ __launch_bounds__(FUSED_WAVES * 64, FUSED_WAVES / 4) void k_gather_update(FArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr bool SPARSE = SRC == 1, EMBED = SRC == 2;
  FT_DECL;
  float* gbase = lds + PackUpdF3::FLOATS + (POST ? 6144 : 0);
  const GatherLds gl = gather_lds(gbase, a.g.g, a.g.tm.TPS);
  stage_gather(gl.cm, gl.ko, gl.tt, gl.kvo, a.g.g, a.g.tm.TPS);
  copy_to_lds(lds + PackUpdF3::BA, a.u.pack + PackUpd::BA, 64);
  copy_to_lds(lds + PackUpdF3::BCB, a.u.pack + PackUpd::BCB, 64 + 64 + 128);          // BCB, BCBROW, VAW
  copy_to_lds(lds + PackUpdF3::WAS3, a.u.pack + (LANES == 16 ? (int)PackUpd::WAS3_G16 : (int)PackUpd::WAS3_G32), 6144);
  copy_to_lds(lds + PackUpdF3::WCB3, a.u.pack + PackUpd::WCB3, 6144);
  copy_to_lds(lds + PackUpdF3::WA1S3, a.u.pack + (LANES == 16 ? (int)PackUpd::WA1S3_G16 : (int)PackUpd::WA1S3_G32), 6144);
  if (POST) copy_to_lds(lds + PackUpdF3::FLOATS, a.u.wp, 6144);
  __syncthreads();
  FT_MARK(0);      // staging
  const int lane0 = threadIdx.x & 63, wave = threadIdx.x >> 6;
  char* wbase = reinterpret_cast<char*>(gl.kvo + ((gather_slots(a.g.g.K2, LANES) + 3) & ~3)) + (size_t)wave * fused_wave_bytes(a.g.g.K2, LANES, SPARSE);
  int* perm = reinterpret_cast<int*>(wbase);
  uint2* tab = reinterpret_cast<uint2*>(wbase + 128);      // (the embed variant's gather does not use it)

  const EmbedLane el = embed_lane<EMBED && LANES == 32>(a.g, lane0 & 31);
  float ew[4][3] = {}, eb[4] = {};
  if (EMBED && LANES == 16 && !EMBED_MFMA) {
    const int jn = lane0 & 15;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      eb[c] = a.g.es.wb[192 + 4 * jn + c];
#pragma unroll
      for (int q = 0; q < 3; ++q) ew[c][q] = a.g.es.wb[(4 * jn + c) * 3 + q];
    }
  }

  Frag P;                               // pending chain tile: slots [0, cnt) hold live nodes waiting for their update
#pragma unroll
  for (int R = 0; R < 32; ++R) FRAG_AT(P, R) = 0.0f;
  int cnt = 0;                          // wave-uniform
  int gcP = 0;
  float r0P = 0.0f, r1P = 0.0f, swP = 0.0f;
  auto chain_sw = [&]() {
    const int g = gcP & 0x7fffffff;
    return a.sw_from_gather ? swP : a.u.sarr[a.u.smod > 0 ? g % a.u.smod : g];
  };

  // Rounds of FUSED_WAVES tiles (one per wave) are dealt round-robin over the workgroups in XCD-grouped order (see k_gather16)
  int wg = blockIdx.x;
  const int nwg = gridDim.x;
  if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);
  const long nrounds = (a.g.ntiles + FUSED_WAVES - 1) / FUSED_WAVES;
  const int nslots = (LANES == 16 ? 4 : 2) * a.g.g.K2;
  FwNext<NIT> nx;
  nx.has = false;
  if (!EMBED && wg < nrounds) fw_prefetch<LANES, SPARSE, NIT>(a.g, gl, (long)wg * FUSED_WAVES + wave, lane0, LANES == 16 ? (lane0 & 15) : (lane0 & 31), nx);
  for (long r = wg; r < nrounds; r += nwg) {
    const long tile = r * FUSED_WAVES + wave;
    if (tile >= a.g.ntiles) break;
    // The lane id is made opaque once per tile: everything derived from it (LDS addresses, masks, offsets -- some 25 registers)
    // is then recomputed per tile instead of being hoisted out of the loop and spilled to scratch at three waves per SIMD.
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const int h = lane >> 5;
    const int jn = LANES == 16 ? (lane & 15) : (lane & 31);      // this lane's node inside a gather tile
    const int jd = lane & 31;                                    // this lane's slot inside the pending chain tile
    const long tile_next = r + nwg < nrounds ? (r + nwg) * FUSED_WAVES + wave : a.g.ntiles;
    const int sample = __builtin_amdgcn_readfirstlane((int)(tile / a.g.tm.TPS));
    const int t = __builtin_amdgcn_readfirstlane((int)(tile - (long)sample * a.g.tm.TPS));
    const TileCtx tc = block_decode(a.g.tm, gl.tt, sample, t, jn);
    const int gc = (int)(tc.sample * a.g.tm.N + tc.n);
    float lb, ub;
    FwNext<NIT> cu;
    if (EMBED) {
      lb = a.g.lb[gc];
      ub = a.g.ub[gc];
    } else {
      cu = nx;
      lb = cu.lb;
      ub = cu.ub;
    }
    const bool need = tc.valid && node_is_live(lb, ub);
    if (!__any(need)) {
      if (!EMBED) fw_prefetch<LANES, SPARSE, NIT>(a.g, gl, tile_next, lane, jn, nx);
      FT_MARK(1);
      continue;
    }
    FT_MARK(1);      // decode + wait for the prefetched bounds
    const Ratio rt = compute_ratio(lb, ub);
    float ssum = 0.0f;
    Frag X;                             // LANES == 32: the aggregate in the gather channel map
    f32x4 acc[4];                       // LANES == 16
    if (EMBED) {
      if (LANES == 32) gather_compute_tile<true, false>(a.g, tc, sample, gl.cm, gl.ko, gl.kvo, tab, el, lane, X, ssum);
      else gather_compute_tile16<true, false>(a.g, tc, sample, gl.cm, gl.ko, gl.kvo, tab, ew, eb, lane, acc, ssum);
    } else {
      const int n = fw_build_tab<LANES, SPARSE, NIT>(cu, tab, nslots, lane);
      FT_MARK(2);    // slot table
      const float* sbase = a.g.mu_src + (long)sample * a.g.g.Ns * 64;
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)sbase, 0, a.g.g.Ns * 256, 0x00020000);
      const float* cmt = gl.cm + tc.cg * a.g.g.K2 * 64;
      auto after = [&]() { fw_prefetch<LANES, SPARSE, NIT>(a.g, gl, tile_next, lane, jn, nx); };
      if (LANES == 32) {
        fw_walk32<SPARSE>(X, cmt, tab, n, rsrc, lane, ssum, after);
        if (a.g.g.normalise) {
          const int wy0 = tc.by * a.g.g.ystep + a.g.g.ybase, wx0 = tc.bx * a.g.g.xstep + a.g.g.xbase;
          const int ny = tap_count(tc.y, wy0, a.g.g.WY, a.g.g.Hs, a.g.g.kh, a.g.g.stride, a.g.g.pad);
          const int nxx = tap_count(tc.x, wx0, a.g.g.WX, a.g.g.Ws, a.g.g.kw, a.g.g.stride, a.g.g.pad);
          const int f = tc.valid ? ny * nxx : 1;
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
      } else {
        fw_walk16<SPARSE>(acc, cmt, tab, n, rsrc, lane, ssum, after);
      }
    }

    FT_MARK(3);      // walk (+ next tile's prefetch issue, + tap-count division)
    // ---- the live nodes of the tile: into the pending tile ----
    const unsigned long long bal = __ballot(need) & (LANES == 16 ? 0xffffull : 0xffffffffull);
    const int n = __popcll(bal);
    if (need && lane < LANES) perm[__popcll(bal & ((1ull << lane) - 1ull))] = jn;
    __builtin_amdgcn_wave_barrier();
    // slots [d0, d0 + count) of P <- the nodes of rank [rank0, rank0 + count) of this tile.  merge: the other slots keep what they
    // hold; else (P was just consumed by the chain, d0 = 0) they are don't-cares, so nothing of the old P stays alive across the chain
    auto fill = [&](int d0, int rank0, int count, bool merge) {
      const int d = jd - d0;
      const bool take = d >= 0 && d < count;
      const bool keep = merge && !take;
      const int src = perm[take ? rank0 + d : 0];
      if (FUSED_ABL & 8) {
#pragma unroll
        for (int R = 0; R < 32; ++R) FRAG_AT(P, R) += (LANES == 32 ? FRAG_AT(X, R) : acc[R & 3][(R >> 2) & 3]);
      } else if (LANES == 32) {
        // (8 pulls in flight, then their 8 selects: one pull + wait + select at a time serialises 35 crossbar round trips)
        const int addr = (src + 32 * h) * 4;
#pragma unroll
        for (int R0 = 0; R0 < 32; R0 += 8) {
          float v[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) v[q] = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(FRAG_AT(X, R0 + q))));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 8; ++q) FRAG_AT(P, R0 + q) = keep ? FRAG_AT(P, R0 + q) : v[q];
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int addr = (src + 16 * (2 * h + q)) * 4;
#pragma unroll
          for (int r4 = 0; r4 < 4; r4 += 2) {
            float v[8];
#pragma unroll
            for (int w = 0; w < 8; ++w) v[w] = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(acc[w & 3][r4 + (w >> 2)])));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int w = 0; w < 8; ++w) {
              const int R = 16 * q + 4 * (r4 + (w >> 2)) + (w & 3);
              FRAG_AT(P, R) = keep ? FRAG_AT(P, R) : v[w];
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      const int a0 = src * 4;           // per-node scalars sit on the lanes of group / half 0
      const int g2 = __builtin_amdgcn_ds_bpermute(a0, gc | (rt.amb != 0.0f ? (int)0x80000000 : 0));
      const float r2 = __int_as_float(__builtin_amdgcn_ds_bpermute(a0, __float_as_int(rt.r0)));
      const float q2 = __int_as_float(__builtin_amdgcn_ds_bpermute(a0, __float_as_int(rt.r1)));
      const float s2 = __int_as_float(__builtin_amdgcn_ds_bpermute(a0, __float_as_int(ssum)));
      gcP = keep ? gcP : g2;
      r0P = keep ? r0P : r2;
      r1P = keep ? r1P : q2;
      swP = keep ? swP : s2;
    };
    const int take1 = n < 32 - cnt ? n : 32 - cnt;
    fill(cnt, 0, take1, true);
    cnt += take1;
    FT_MARK(4);      // ambiguous rows + compaction
    if (cnt == 32) {
      const float sw = chain_sw();
      if (FUSED_ABL & 4) {
        if (sw > 1e30f) frag_store_rows(P, a.u.mu, gcP & 0x7fffffff, h);
      } else fused_chain<POST>(a, lds, P, gcP, r0P, r1P, sw, true, lane);
      FT_MARK(5);    // chain
      cnt = n - take1;
      if (cnt > 0) fill(0, take1, cnt, false);
      FT_MARK(4);
    }
  }

  // ---- what is left in P: one partly filled chain tile ----
  if (cnt > 0) {
    const int lane = lane0;
    const bool valid = (lane & 31) < cnt;
    const float sw = valid ? chain_sw() : 0.0f;
    fused_chain<POST>(a, lds, P, valid ? gcP : 0, valid ? r0P : 0.0f, valid ? r1P : 0.0f, sw, valid, lane);
  }
  FT_MARK(6);        // last, partly filled tile
  if (FUSED_TIMING_ON) FT_FLUSH();
}
