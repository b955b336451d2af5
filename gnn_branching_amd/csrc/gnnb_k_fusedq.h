// gnnb_k_fusedq.h -- part of libgnnb.so, included by gnnb.hip (one translation unit; see its header comment).
// k_gather_update_q: one conv half-pass (gather of the neighbour aggregate + folded node update, reference graph_conv.py:110-181
// forward, :299-349 backward) in ONE kernel with the waves of a workgroup SPECIALISED: QG_WAVES gather waves, the rest chain waves, and an
// LDS row queue between them -- the aggregate rows never reach HBM.
//
// Why this shape (measured): the gather wants many light waves (72-104 registers; 16 per CU: 76 us, 8: 85 us),
// the node update 150 registers per wave.  The first fused form -- every wave gathers a tile, compacts its live nodes into a
// pending chain tile IN REGISTERS (ds_bpermute) and runs the chain when that is full -- had to run 8 waves of 240 registers and
// serialised the two latency chains in each wave: parity-green, 33 % less HBM traffic, and 5-20 % SLOWER than the two kernels (12
// waves spilled 30-90 registers and were slower still; a software-pipelined table walk for its gather bought nothing).  Here the
// gather waves run the stand-alone gathers' code and only stop storing rows to HBM: the live nodes of a tile are ranked with a
// ballot and written to consecutive rows of a ring of 4 x 32 rows in LDS.  A chain wave claims the next tile, waits for its 32
// rows, copies them to registers, hands the ring slot back and runs the bf16x3 chain on them (the whole kernel fits 128 registers
// = 4 waves per SIMD), stores the rows of mu.
//
// Queue protocol (all in LDS; `q` points at QHDR_INTS ints):
//   reserve            rows handed out so far (atomic add by the gather waves: rows of a tile are consecutive)
//   filled[s]          rows written into ring slot s for its current tile
//   free_id[s]         the tile id slot s accepts (starts at s; the chain wave sets T + QTILES when it is done with tile T)
//   done               gather waves that have finished
//   staged             chain waves whose share of the node-update weights is in LDS (the chain waves wait for all of them)
// A gather wave writes its rows of tile T only after free_id[T % QTILES] == T, then adds their number to filled[.] (LDS executes
// one wave's operations in order, so the count lands after the rows).  Tile T's rows were reserved before tile T + QTILES's, and a
// wave publishes its part of T before it waits for T + 1: no cycle.  Chain wave c takes tiles c, c + 4, ...; it leaves when all
// gather waves are done and `reserve` says there is no tile T (a last, partly filled tile runs with its missing lanes masked).
// Every poll loop has an iteration cap that raises status bit 1 (value 2) and leaves: a protocol bug must never hang the GPU.
// Ambiguous nodes ride in the same tiles: every node goes through H = WAS.(r0 x) + Wa[:, 64:].((r1 - r0) x) (PackUpdL3), the second
// block -- exact zeros for r0 == r1 -- only runs when the tile holds such a node.  k_node_update does the same arithmetic per node, so
// the fused and the two-kernel half-pass are bit-identical and which one runs is a pure scheduling decision.
#pragma once

// The bf16 x 3 image of a node-update pack (PackUpd -> PackUpdL3: BA | BCB, BCBROW, VAW | WAS3, WCB3, WA1S3, contiguous in both; plus the
// POST block `wp` behind it) into LDS by threads ct of cn, EVERY load requested before the first LDS write: one memory round trip for the
// 75-98 KB instead of one per piece (five or six copy_to_lds_part calls in a row cost a chain wave 17.7 k cycles = 9 us at the start
// of k_gather_update_q, tools/fusedq_timing.py).  Batches of UPDL3_BATCH 16-byte pieces per lane (512 threads: one batch).
#define UPDL3_BATCH 13
__device__ __forceinline__ void stage_updl3(float* lds, const float* pack, const float* wp, int ct, int cn) {
  constexpr int NSMALL = 16 + 64, NBIG = 3 * 6144 / 4;
  const int n4 = NSMALL + NBIG + (wp ? 6144 / 4 : 0);
  for (int i0 = ct; i0 < n4; i0 += UPDL3_BATCH * cn) {
    f32x4 v[UPDL3_BATCH];
#pragma unroll
    for (int u = 0; u < UPDL3_BATCH; ++u) {
      const int i = i0 + u * cn;
      const float* src = pack + PackUpd::BA + 4 * i;                                              // i < 16
      if (i >= 16) src = pack + PackUpd::BCB + 4 * (i - 16);
      if (i >= NSMALL) src = pack + PackUpd::WAS3 + 4 * (i - NSMALL);
      if (i >= NSMALL + NBIG) src = wp + 4 * (i - NSMALL - NBIG);
      v[u] = *reinterpret_cast<const f32x4*>(i < n4 ? src : pack);
    }
#pragma unroll
    for (int u = 0; u < UPDL3_BATCH; ++u) {
      const int i = i0 + u * cn;
      float* dst = lds + PackUpdL3::BA + 4 * i;
      if (i >= 16) dst = lds + PackUpdL3::BCB + 4 * (i - 16);
      if (i >= NSMALL) dst = lds + PackUpdL3::WAS3 + 4 * (i - NSMALL);
      if (i >= NSMALL + NBIG) dst = lds + PackUpdL3::FLOATS + 4 * (i - NSMALL - NBIG);
      if (i < n4) *reinterpret_cast<f32x4*>(dst) = v[u];
    }
  }
}
static_assert(PackUpdL3::BCB == PackUpdL3::BA + 64 && PackUpdL3::WAS3 == PackUpdL3::BCB + 256 && PackUpdL3::FLOATS == PackUpdL3::WAS3 + 3 * 6144,
              "stage_updl3: layout of the LDS image");

struct FArgs {
  GArgs g;             // the gather (k_gather / k_gather16 arguments; g.nb is unused; g.sout != null: the sparse walk computes the bias sums)
  UpdArgs u;           // the node update (k_node_update arguments; list0 / list1 / cnt0 / cnt1, nb are unused)
  int sw_from_gather;  // 1: the bias-sum scalar of a node comes out of its (sparse) gather; 0: u.sarr holds it (table / k_livesum)
  int qtiles;          // ring slots (2..QTILES): as many as fit beside the weights and the gather's tables in LDS
};

#ifndef QG_WAVES
#define QG_WAVES 8            // gather waves (8 + 8 chain waves measured best on base B = 256; 10 + 6 and 12 + 4 are 2-8 % slower)
#endif
#ifndef Q_TOTAL_WAVES
#define Q_TOTAL_WAVES 16      // waves per workgroup = per CU (128 VGPRs each: the whole register file).  dev A/B: 8 / 12 with QG_WAVES 4 / 6 = one / one and a
#endif                        // half wave pairs per SIMD instead of two -- how the half-pass time scales with resident waves (profiles/r06_wave_scaling_ab.txt)
#define QC_WAVES (Q_TOTAL_WAVES - QG_WAVES)      // chain waves
#define QTILES 4              // ring slots of 32 rows at most; FArgs.qtiles (2..4) says how many this launch has room for
#define QROW 72               // floats per ring row: 64 channels, {node id | ambiguous << 31, r0, r1, s}, 4 pad (16-B rows, <= 2-way bank conflicts)
#define QHDR_INTS 16
#define Q_POLL_CAP (1 << 22)
__host__ __device__ constexpr size_t fusedq_queue_floats(int qtiles) { return (size_t)qtiles * 32 * QROW + QHDR_INTS; }

struct QHdr { int reserve, done, claim, staged, filled[QTILES], free_id[QTILES], pad2[4]; };
static_assert(sizeof(QHdr) == QHDR_INTS * 4, "queue header");

// (acquire / release at workgroup scope: LDS needs no cache maintenance, this only keeps compiler and wait counters honest)
__device__ __forceinline__ int q_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// The folded node update (see node_update_loop) of one tile whose aggregate rows are in registers: X = fragment (lane (j, h):
// node j, register 4 q + c = feature 8 q + 4 h + c), gc = node id, (r0, r1, amb) its ratios, sw its bias-sum scalar, valid = the lane
// holds a node.  `lds`: the PackUpdL3 image.  Callers: k_gather_update_q / k_scored_tail (rows out of the LDS ring) and k_top (rows
// out of the transposed Linear edge's accumulators) -- the same arithmetic per node as k_node_update, whatever tile a node rides in.
// keep != nullptr: the rows E are also handed back in *keep (k_scored_tail feeds them to the score head without reading them back)
template <bool POST, bool PIPE = (GEMM_BF3_PIPE != 0)>
__device__ __forceinline__ void upd_chain_frag(const UpdArgs& u, const float* lds, const Frag& X, int gc, float r0, float r1, bool amb, float sw,
                                               bool valid, int lane, Frag* keep = nullptr) {
  const int h = lane >> 5;
  Frag H, H2;
  frag_bias(H, lds + PackUpdL3::BA, h);
  {
    const float x[1] = {(h ? r1 : r0) * sw};            // + s.(r0 Wa0.bp + r1 Wa1.bp): the bias of the source rows' deferred projection
    gemm_small<1>(lds + PackUpdL3::VAW, lane, H, x);
  }
  gemm_w64_bf3<1, PIPE>(lds + PackUpdL3::WAS3, lane, H, [&](int s) { return FRAG_AT(X, s) * r0; });
#ifdef Q_ABL_NOAMB      // dev, timing only (wrong results): what the chain would cost if no tile held an ambiguous node
  if (false) {
#else
  if (__any(amb)) {
#endif
    const float dr = r1 - r0;
    gemm_w64_bf3<1, PIPE>(lds + PackUpdL3::WA1S3, lane, H, [&](int s) { return FRAG_AT(X, s) * dr; });
  }
  // P' of an ambiguous node: its cached row (k_pre); of every other node: the bias row
  frag_load_rowptr(H2, amb ? u.P + (long)gc * 64 : u.pack + PackUpd::BCBROW, h);
  frag_relu(H);
  gemm_w64_bf3<1, PIPE>(lds + PackUpdL3::WCB3, lane, H2, [&](int s) { return FRAG_AT(H, s); });
  frag_relu(H2);
  if (valid) {
    if (frag_has_nan(H2)) atomicOr(u.status, 1);
    if (u.mu) frag_store_rows(H2, u.mu, gc, h);
  }
  if (keep) *keep = H2;
  if (POST) {
#pragma unroll
    for (int R = 0; R < 32; ++R) FRAG_AT(H, R) = 0.0f;
    gemm_w64_bf3<1, PIPE>(lds + PackUpdL3::FLOATS, lane, H, [&](int s) { return FRAG_AT(H2, s); });
    if (valid) {
      if (u.post3) frag_store_rows3(H, u.post3, gc, h);
      else frag_store_rows(H, u.post, gc, h);
    }
  }
}

// the same on ring slot `ring` (32 rows of QROW floats); `release` runs once the rows are in registers
template <bool POST, bool PIPE = (GEMM_BF3_PIPE != 0), class Release>
__device__ __forceinline__ void q_chain(const FArgs& a, const float* lds, const float* ring, int nvalid, int lane, Release release, Frag* keep = nullptr) {
  const int h = lane >> 5, j = lane & 31;
  const bool valid = j < nvalid;
  const float* row = ring + j * QROW;
  const f32x4 sc = *reinterpret_cast<const f32x4*>(row + 64);
  const int gc_amb = __float_as_int(sc[0]);
  const int gc = valid ? (gc_amb & 0x7fffffff) : 0;
  const float r0 = valid ? sc[1] : 0.0f, r1 = valid ? sc[2] : 0.0f;
  const bool amb = valid && gc_amb < 0;
  // the bias-sum scalar: out of the sparse gather itself (rides in the row), or looked up in u.sarr -- by the index the gather wave left in its place: the
  // node's index inside its sample for the per-sample table of edge 1 (smod > 0; no modulo of the flat index here), else the flat index
  const float sw = !valid ? 0.0f : (a.sw_from_gather ? sc[3] : a.u.sarr[__float_as_int(sc[3])]);
  Frag X;                                    // fragment order: register 4 q + c = feature 8 q + 4 h + c
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + 8 * q + 4 * h);
#pragma unroll
    for (int c = 0; c < 4; ++c) FRAG_AT(X, 4 * q + c) = v[c];
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): the rows and scalars are in registers
  release();
  upd_chain_frag<POST, PIPE>(a.u, lds, X, gc, r0, r1, amb, sw, valid, lane, keep);
}

#if defined(FUSED_TIMING) && FUSED_TIMING == 6      // dev: per-phase cycle sums of ONE k_gather_update_q template (Q_TIME_LANES / _SRC / _POST); slots 0-4 gather waves, 5-9 chain waves
#ifndef Q_TIME_LANES
#define Q_TIME_LANES 32
#define Q_TIME_SRC 1
#define Q_TIME_POST true
#endif
#define QT_ON (LANES == Q_TIME_LANES && SRC == Q_TIME_SRC && POST == Q_TIME_POST)
__device__ unsigned long long g_qt_wall[2 * 16 * 1024];
#define QT_DECL unsigned long long qt_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long qt_last = __builtin_readcyclecounter(); const unsigned long long qt_w0 = wall_clock64()
#define QT_MARK(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); qt_[i] += n_ - qt_last; qt_last = n_; } while (0)
#define QT_FLUSH() do { if (QT_ON && (threadIdx.x & 63) == 0) { const int w_ = blockIdx.x * 16 + (threadIdx.x >> 6); g_qt_wall[2 * w_] = qt_w0; g_qt_wall[2 * w_ + 1] = wall_clock64(); }      /* start / end of EVERY wave, 100 MHz chip-wide clock */ \
  if (QT_ON && (threadIdx.x & 63) == 0 && (blockIdx.x & 31) == 5) {      /* (a sample of the workgroups: 40 k atomics on ten words cost 0.7 ms) */ for (int i_ = 0; i_ < 10; ++i_) atomicAdd(&g_fused_t[i_], qt_[i_]); atomicAdd(&g_fused_t[chain_role ? 14 : 15], 1ull); } } while (0)
#else
#define QT_DECL
#define QT_MARK(i)
#define QT_FLUSH()
#endif
// LANES: dst nodes per gather tile (16: forward edges, 32: transposed edges).  SRC: 0 dense source rows, 1 sparse walk (the source
// is a ReLU layer), 2 round-0 embedding computed in the gather (16-node tiles only).  POST: see UpdArgs.
// fusedq_body: one half-pass by this workgroup (all 16 waves call it; it starts with workgroup barriers; k_scored_tail shares q_chain).  Returns false when a wait hit its
// iteration cap (status bit 1 is raised; the caller must leave the kernel), true when this wave's part of the half-pass is done.
template <int LANES, int SRC, bool POST>
__device__ __forceinline__ bool fusedq_body(const FArgs& a, float* lds) {
  constexpr bool SPARSE = SRC == 1, EMBED = SRC == 2;
  QT_DECL;
  float* qbase = lds + PackUpdL3::FLOATS + (POST ? 6144 : 0);
  const int NQ = a.qtiles;
  QHdr* q = reinterpret_cast<QHdr*>(qbase + (size_t)NQ * 32 * QROW);
  float* gbase = qbase + fusedq_queue_floats(NQ);
  const GatherLds gl = gather_lds(gbase, a.g.g, a.g.tm.TPS);
  stage_gather(gl.cm, gl.ko, gl.tt, gl.kvo, a.g.g, a.g.tm.TPS);
  if (threadIdx.x < QHDR_INTS) reinterpret_cast<int*>(q)[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x < QTILES) q->free_id[threadIdx.x] = threadIdx.x;
  __syncthreads();
  // (the thread index goes through an opaque asm so that hipcc derives the lane addresses here, behind the barriers, not in the prologue: the
  // form the round-5 measurements were taken with)
  int tidx = threadIdx.x;
  asm volatile("" : "+v"(tidx));
  const int lane = tidx & 63, hwave = __builtin_amdgcn_readfirstlane(tidx >> 6);
  // Q_ROLE_BY_SIMD (dev): the waves w, w + 4, w + 8, w + 12 of a workgroup share a SIMD; gather waves = those with (w & 3) < 2, so that
  // two SIMDs run only gathers (fp32 MFMAs: they hold the SIMD's vector issue for their whole duration, tools/micro/mfma_valu_overlap.hip)
  // and two only chains
#if defined(Q_ROLE_BY_SIMD) && QG_WAVES == 8
  const bool chain_role = (hwave & 3) >= 2;
  const int wave = chain_role ? QG_WAVES + ((hwave >> 2) * 2 + (hwave & 1)) : ((hwave >> 2) * 2 + (hwave & 1));
#else
  const int wave = hwave;
  const bool chain_role = wave >= QG_WAVES;
#endif

  QT_MARK(chain_role ? 5 : 0);                      // gather tables staged, queue header zeroed (both roles)
#if defined(Q_PRIO) && Q_PRIO == 1                  // dev A/B: static priority for one role (MI355X_MICROARCH.md, two waves per SIMD, item 4)
  if (chain_role) __builtin_amdgcn_s_setprio(1);
#elif defined(Q_PRIO) && Q_PRIO == 2
  if (!chain_role) __builtin_amdgcn_s_setprio(1);
#endif
  if (chain_role) {
    // ---------------- chain wave ----------------
    // The node-update weights (74 KB + 24 KB of POST) are staged by the chain waves alone while the gather waves, which only need the
    // gather's tables, already walk their first tiles; `staged` counts the chain waves whose part is in LDS.
    {
      const int ct = (wave - QG_WAVES) * 64 + lane, cn = QC_WAVES * 64;
      stage_updl3(lds, a.u.pack, POST ? a.u.wp : nullptr, ct, cn);
      __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): this wave's part is in LDS before it is counted
      if (lane == 0) __hip_atomic_fetch_add(&q->staged, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      bool ok = false;
      for (int it = 0; it < Q_POLL_CAP; ++it) {
        if (q_ld(&q->staged) == QC_WAVES) { ok = true; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      if (!ok) { if (lane == 0) atomicOr(a.u.status, 2); return false; }
    }
    QT_MARK(6);                                     // node-update weights staged (all chain waves)
    // claims the next tile, copies it out of its ring slot, releases the slot, runs the chain
    for (;;) {
      int T = 0;
      if (lane == 0) T = atomicAdd(&q->claim, 1);
      T = __builtin_amdgcn_readfirstlane(T);
      const int s = T % NQ;
      const float* ring = qbase + (size_t)s * 32 * QROW;
      int nvalid = -1;
      for (int it = 0; it < Q_POLL_CAP; ++it) {
        if (q_ld(&q->free_id[s]) == T && q_ld(&q->filled[s]) == 32) { nvalid = 32; break; }
        if (q_ld(&q->done) == QG_WAVES) {            // every row there will ever be has been published
          const int rem = q_ld(&q->reserve) - 32 * T;
          nvalid = rem <= 0 ? 0 : (rem < 32 ? rem : 32);
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
      if (nvalid < 0) { if (lane == 0) atomicOr(a.u.status, 2); return false; }
      QT_MARK(7);                                   // waiting for a tile of rows
      if (nvalid == 0) { QT_FLUSH(); return true; }
      q_chain<POST>(a, lds, ring, nvalid, lane, [&]() {
        // (called once the rows are in registers) hand the slot back before the chain runs: the ring only has to cover the
        // time a tile takes to fill and to be copied out, not the ~10 us of its chain
        if (nvalid == 32 && lane == 0) {
          __hip_atomic_store(&q->filled[s], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          __hip_atomic_store(&q->free_id[s], T + NQ, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      });
      QT_MARK(8);                                   // rows -> registers, chain, row stores
      if (nvalid < 32) { QT_FLUSH(); return true; }      // the last, partly filled tile
    }
  }

  // ---------------- gather wave ----------------
  const int jn = LANES == 16 ? (lane & 15) : (lane & 31);
  const int h = lane >> 5;
  char* wsc = reinterpret_cast<char*>(gl.kvo + ((gather_slots(a.g.g.K2, LANES) + 3) & ~3));
  uint2* tab = reinterpret_cast<uint2*>(wsc) + (size_t)wave * ((LANES == 16 ? 4 : 2) * a.g.g.K2 + 32);      // SPARSE: live window slots of this wave's tile
  const EmbedLane el{};
  float ew[4][3] = {}, eb[4] = {};
  if (EMBED && !EMBED_MFMA) {
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      eb[cc] = a.g.es.wb[192 + 4 * jn + cc];
#pragma unroll
      for (int qq = 0; qq < 3; ++qq) ew[cc][qq] = a.g.es.wb[(4 * jn + cc) * 3 + qq];
    }
  }
  int wg = blockIdx.x;
  const int nwg = gridDim.x;
  if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);      // workgroups b, b + 8, ... share an XCD: give them neighbouring rounds of tiles
  const long ntl = a.g.ntiles;
  const long nrounds = (ntl + QG_WAVES - 1) / QG_WAVES;
  const long r0 = wg, rstep = nwg;
  bool stuck = false;
  // The bounds of a tile's dst nodes are fetched one tile ahead (they decide which lanes are live: the tile cannot start without
  // them, and a gather wave spent 14 % of its time waiting for them).
  struct Next { TileCtx tc; int sample, gc; bool in; float lb, ub; } nx;
  auto fetch = [&](long r) {
    const long tile = r * QG_WAVES + wave;
    nx.in = r < nrounds && tile < ntl;
    const long tl = nx.in ? tile : 0;
    const int si = tile_sample(a.g.tm, tl);
    nx.sample = si;
    const int t = __builtin_amdgcn_readfirstlane((int)(tl - (long)si * a.g.tm.TPS));
    nx.tc = block_decode(a.g.tm, gl.tt, nx.sample, t, jn);
    nx.gc = (int)(nx.tc.sample * a.g.tm.N + nx.tc.n);
    nx.lb = a.g.lb[nx.gc];
    nx.ub = a.g.ub[nx.gc];
  };
  fetch(r0);
  for (long r = r0; r < nrounds && !stuck; r += rstep) {
    if (!nx.in) break;
    const TileCtx tc = nx.tc;
    const int sample = nx.sample, gc = nx.gc;
    const float lb = nx.lb, ub = nx.ub;
    fetch(r + rstep);
    const bool need = tc.valid && node_is_live(lb, ub);
    QT_MARK(1);                                     // tile decode, its bounds (requested one tile ahead)
    if (!__any(need)) continue;
    const Ratio rt = compute_ratio(lb, ub);
    float ssum = 0.0f;
    Frag X;
    f32x4 acc[4];
    if (LANES == 32) gather_compute_tile<false, SPARSE>(a.g, tc, sample, gl.cm, gl.ko, gl.kvo, tab, el, lane, X, ssum);
    else gather_compute_tile16<EMBED, SPARSE>(a.g, tc, sample, gl.cm, gl.ko, gl.kvo, tab, ew, eb, lane, acc, ssum);
    if (!a.sw_from_gather) ssum = __int_as_float(a.u.smod > 0 ? tc.n : gc);      // (q_chain: the index of the node's entry in u.sarr)

    QT_MARK(2);                                     // table build + walk
    // ---- the live nodes of the tile -> consecutive rows of the ring ----
    const unsigned long long bal = __ballot(need) & (LANES == 16 ? 0xffffull : 0xffffffffull);
    const int n = __popcll(bal);
    int base = 0;
    if (lane == 0) base = atomicAdd(&q->reserve, n);
    base = __builtin_amdgcn_readfirstlane(base);
    const int pos = base + __popcll(bal & ((1ull << jn) - 1ull));        // this lane's node (if needed) goes to row `pos`
    const int T0 = base >> 5, T1 = (base + n - 1) >> 5;
    for (int T = T0; T <= T1; ++T) {
      const int s = T % NQ;
      bool ok = false;
      for (int it = 0; it < Q_POLL_CAP; ++it) {
        if (q_ld(&q->free_id[s]) == T) { ok = true; break; }
        __builtin_amdgcn_s_sleep(4);
      }
      if (!ok) { if (lane == 0) atomicOr(a.u.status, 2); stuck = true; break; }
      QT_MARK(3);                                   // waiting for the ring slot
      const bool mine = need && (pos >> 5) == T;
      if (mine) {
        float* row = qbase + ((size_t)s * 32 + (pos & 31)) * QROW;
        if (LANES == 32) {
          float* d = row + 8 * h;                   // gather channel map: x.t[0][r], x.t[1][r] = channels 2k, 2k + 1, k = (r & 3) + 8 (r >> 2) + 4 h
#pragma unroll
          for (int rr = 0; rr < 16; ++rr)
            *reinterpret_cast<float2*>(d + 2 * (rr & 3) + 16 * (rr >> 2)) = make_float2(X.t[0][rr], X.t[1][rr]);
          if (h == 0)
            *reinterpret_cast<f32x4*>(row + 64) = f32x4{__int_as_float(gc | (rt.amb != 0.0f ? (int)0x80000000 : 0)), rt.r0, rt.r1, ssum};
        } else {
          const int gq = lane >> 4;                 // lane (j, g'): channels 16 g' + 4 r + t
          f32x4* d = reinterpret_cast<f32x4*>(row + 16 * gq);
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) d[rr] = f32x4{acc[0][rr], acc[1][rr], acc[2][rr], acc[3][rr]};
          if (gq == 0)
            *reinterpret_cast<f32x4*>(row + 64) = f32x4{__int_as_float(gc | (rt.amb != 0.0f ? (int)0x80000000 : 0)), rt.r0, rt.r1, ssum};
        }
      }
      const int lo = T * 32 > base ? T * 32 : base, hi = (T + 1) * 32 < base + n ? (T + 1) * 32 : base + n;
      __builtin_amdgcn_s_waitcnt(0xc07f);            // lgkmcnt(0): the rows are in LDS before their count is
      if (lane == 0) __hip_atomic_fetch_add(&q->filled[s], hi - lo, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      QT_MARK(4);                                   // rows -> ring
    }
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  if (lane == 0) __hip_atomic_fetch_add(&q->done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  QT_FLUSH();
 
  return !stuck;
}

template <int LANES, int SRC, bool POST>
__global__ __launch_bounds__((QG_WAVES + QC_WAVES) * 64, 4) void k_gather_update_q(FArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  fusedq_body<LANES, SRC, POST>(a, lds);
}
