// gnnb_k_fused.h -- part of libgnnb.so, included by gnnb.hip (one translation unit; see its header comment).
// k_gather_update: one half-pass over a conv edge in ONE kernel -- the MFMA gather of the neighbour aggregate and the folded node
// update of the nodes it feeds, with the aggregate rows never written to HBM (reference graph_conv.py:110-181 forward,
// :299-349 backward: aggregate -> update of one layer).
//
// A gather tile is a block of 16 / 32 dst nodes of one sample, of which only the live ones (~55 %) are updated, and the node
// MLP wants 32 nodes on its lanes.  So every wave keeps ONE pending chain tile P in registers and compacts the live, not
// ambiguous nodes of each gathered tile into it, register for register, with ds_bpermute (the LDS crossbar, no LDS memory):
// the destination lane of pending slot d pulls the accumulator registers of the source lane that holds the (d - cnt)-th live
// node of the tile (a 32-entry per-wave table in LDS, written by the live lanes at their rank, turns ranks into lanes).  When P
// is full the short chain (r0 == r1: two bf16x3 blocks, + the input update's map for POST) runs on it and writes the rows of
// mu; nodes of the tile that did not fit start the next P.  The channel order of P is whatever the gather's accumulators
// hold -- the first layer's weights are packed for that order (PackUpd::WA_G.. / WAS3_G..).
// Ambiguous nodes (3-19 % of the live ones) need the general chain (128-wide first layer, cached P' row), and every wave ends
// with a partly filled P: those rows go to HBM after all (nb, by node id, in P's channel order; ids appended to two per-workgroup
// lists), and after a workgroup barrier the workgroup runs the ordinary node-update loop over its two lists -- a few tiles.
#pragma once

struct FArgs {
  GArgs g;             // the gather (k_gather / k_gather16 arguments; g.nb receives the tail rows, g.sout their bias sums)
  UpdArgs u;           // the node update (k_node_update arguments; list0 / list1 / cnt0 / cnt1 are unused)
  int* tail;           // per-workgroup list segments: [wg][2][seg] node ids (0: live nodes left in a partly filled P, 1: ambiguous)
  int seg;
  int sw_from_gather;  // 1: the bias-sum scalar of a node comes out of its (sparse) gather; 0: u.sarr holds it (table / k_livesum)
};

#define FUSED_WAVES 12
#define FUSED_MAX_LDS (160 * 1024 - 256)      // dynamic LDS limit: the kernel also holds a few bytes of static LDS (its tail counters)

// per-wave LDS scratch of the fused kernel: rank -> lane table, then the sparse walk's slot table
__host__ __device__ inline size_t fused_wave_bytes(int K2, int lanes, bool sparse) {
  return 32 * 4 + (sparse ? (size_t)((lanes == 16 ? 4 : 2) * K2 + 32) * 8 : 0);
}

// the short chain on a full or partly filled pending tile (see node_update_loop, kind 0)
template <bool POST>
__device__ __forceinline__ void fused_chain(const FArgs& a, const float* lds, Frag& P, int gc, float r0, float sw, bool valid, int lane) {
  const int h = lane >> 5;
  Frag H, H2;
  frag_bias(H, lds + PackUpdL3::BA, h);
  {
    const float x[1] = {r0 * sw};                       // + s.(r0 Wa0.bp + r1 Wa1.bp), r0 == r1
    gemm_small<1>(lds + PackUpdL3::VAW, lane, H, x);
  }
  gemm_w64_bf3<1>(lds + PackUpdL3::WAS3, lane, H, [&](int s) { return FRAG_AT(P, s) * r0; });
  frag_bias(H2, lds + PackUpdL3::BCB, h);
  frag_relu(H);
  gemm_w64_bf3<1>(lds + PackUpdL3::WCB3, lane, H2, [&](int s) { return FRAG_AT(H, s); });
  frag_relu(H2);
  if (valid) {
    if (frag_has_nan(H2)) atomicOr(a.u.status, 1);
    if (a.u.mu) frag_store_rows(H2, a.u.mu, gc, h);
  }
  if (POST) {
#pragma unroll
    for (int R = 0; R < 32; ++R) FRAG_AT(H, R) = 0.0f;
    gemm_w64_bf3<1>(lds + PackUpdL3::FLOATS, lane, H, [&](int s) { return FRAG_AT(H2, s); });
    if (valid) frag_store_rows(H, a.u.post, gc, h);
  }
}

// LANES: dst nodes per gather tile (16: forward edges on the 16x16x4 MFMA, 32: 32x32x2).  SRC: 0 dense source rows, 1 sparse walk
// (the source is a ReLU layer), 2 round-0 embedding computed in the gather.  POST: see UpdArgs.
template <int LANES, int SRC, bool POST>
__global__ __launch_bounds__(FUSED_WAVES * 64, FUSED_WAVES / 4) void k_gather_update(FArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int tail_cnt[2];
  constexpr bool SPARSE = SRC == 1, EMBED = SRC == 2;
  float* gbase = lds + PackUpdL3::FLOATS + (POST ? 6144 : 0);
  const GatherLds gl = gather_lds(gbase, a.g.g, a.g.tm.TPS);
  stage_gather(gl.cm, gl.ko, gl.tt, gl.kvo, a.g.g, a.g.tm.TPS);
  copy_to_lds(lds + PackUpdL3::WA, a.u.pack + (LANES == 16 ? (int)PackUpd::WA_G16 : (int)PackUpd::WA_G32), 8192);
  copy_to_lds(lds + PackUpdL3::BA, a.u.pack + PackUpd::BA, 64);
  copy_to_lds(lds + PackUpdL3::BCB, a.u.pack + PackUpd::BCB, 64 + 64 + 128);          // BCB, BCBROW, VAW
  copy_to_lds(lds + PackUpdL3::WAS3, a.u.pack + (LANES == 16 ? (int)PackUpd::WAS3_G16 : (int)PackUpd::WAS3_G32), 6144);
  copy_to_lds(lds + PackUpdL3::WCB3, a.u.pack + PackUpd::WCB3, 6144);
  if (POST) copy_to_lds(lds + PackUpdL3::FLOATS, a.u.wp, 6144);
  if (threadIdx.x < 2) tail_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
  const int jn = LANES == 16 ? (lane & 15) : (lane & 31);      // this lane's node inside a gather tile
  const int jd = lane & 31;                                    // this lane's slot inside the pending chain tile
  char* wbase = reinterpret_cast<char*>(gl.kvo + ((gather_slots(a.g.g.K2, LANES) + 3) & ~3)) + (size_t)wave * fused_wave_bytes(a.g.g.K2, LANES, SPARSE);
  int* perm = reinterpret_cast<int*>(wbase);
  uint2* tab = reinterpret_cast<uint2*>(wbase + 128);
  int* list0 = a.tail + (size_t)blockIdx.x * 2 * a.seg;
  int* list1 = list0 + a.seg;

  const EmbedLane el = embed_lane<EMBED && LANES == 32>(a.g, jn);
  float ew[4][3] = {}, eb[4] = {};
  if (EMBED && LANES == 16 && !EMBED_MFMA) {
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
  float r0P = 0.0f, swP = 0.0f;

  // append node ids of the lanes flagged `f` (flag replicated over the lane groups of a node; group 0 writes) to tail list `which`
  auto append = [&](bool f, int which, int gc) {
    const unsigned long long bal = __ballot(f) & (LANES == 16 ? 0xffffull : 0xffffffffull);
    const int n = __popcll(bal);
    if (n == 0) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(&tail_cnt[which], n);
    base = __builtin_amdgcn_readfirstlane(base);
    if (f && lane < LANES) (which ? list1 : list0)[base + __popcll(bal & ((1ull << lane) - 1ull))] = gc;
  };

  // Rounds of FUSED_WAVES tiles (one per wave) are dealt round-robin over the workgroups in XCD-grouped order (see k_gather16)
  int wg = blockIdx.x;
  const int nwg = gridDim.x;
  if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);
  const long nrounds = (a.g.ntiles + FUSED_WAVES - 1) / FUSED_WAVES;
  for (long r = wg; r < nrounds; r += nwg) {
    const long tile = r * FUSED_WAVES + wave;
    if (tile >= a.g.ntiles) break;
    const int sample = __builtin_amdgcn_readfirstlane((int)(tile / a.g.tm.TPS));
    const int t = __builtin_amdgcn_readfirstlane((int)(tile - (long)sample * a.g.tm.TPS));
    const TileCtx tc = block_decode(a.g.tm, gl.tt, sample, t, jn);
    const int gc = (int)(tc.sample * a.g.tm.N + tc.n);
    const float lb = a.g.lb[gc], ub = a.g.ub[gc];
    const bool need = tc.valid && node_is_live(lb, ub);
    if (!__any(need)) continue;
    const Ratio rt = compute_ratio(lb, ub);
    const bool amb = need && rt.amb != 0.0f;
    const bool k0 = need && !amb;
    float ssum = 0.0f;
    Frag X;                             // LANES == 32: the aggregate in the gather channel map
    f32x4 acc[4];                       // LANES == 16
    if (LANES == 32) gather_compute_tile<EMBED, SPARSE>(a.g, tc, sample, gl.cm, gl.ko, gl.kvo, tab, el, lane, X, ssum);
    else gather_compute_tile16<EMBED, SPARSE>(a.g, tc, sample, gl.cm, gl.ko, gl.kvo, tab, ew, eb, lane, acc, ssum);

    // ---- ambiguous nodes: row to HBM in P's channel order (what frag_load_rows of the tail will put into register R) ----
    if (__any(amb)) {
      if (amb) {
        if (LANES == 32) frag_store_rows(X, a.g.nb, gc, h);
        else {
          const int gq = lane >> 4, hh = gq >> 1, q = gq & 1;
          f32x4* p = reinterpret_cast<f32x4*>(a.g.nb + (long)gc * 64 + 4 * hh);
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) p[2 * (4 * q + r4)] = f32x4{acc[0][r4], acc[1][r4], acc[2][r4], acc[3][r4]};
        }
        if (a.sw_from_gather && lane < LANES) a.g.sout[gc] = ssum;
      }
      append(amb, 1, gc);
    }

    // ---- live, not ambiguous nodes: into the pending tile ----
    const unsigned long long bal = __ballot(k0) & (LANES == 16 ? 0xffffull : 0xffffffffull);
    const int n = __popcll(bal);
    if (n == 0) continue;
    if (k0 && lane < LANES) perm[__popcll(bal & ((1ull << lane) - 1ull))] = jn;
    __builtin_amdgcn_wave_barrier();
    // slots [d0, d0 + count) of P <- the nodes of rank [rank0, rank0 + count) of this tile
    auto fill = [&](int d0, int rank0, int count) {
      const int d = jd - d0;
      const bool take = d >= 0 && d < count;
      const int src = perm[take ? rank0 + d : 0];
      if (LANES == 32) {
        const int addr = (src + 32 * h) * 4;
#pragma unroll
        for (int R = 0; R < 32; ++R) {
          const float v = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(FRAG_AT(X, R))));
          FRAG_AT(P, R) = take ? v : FRAG_AT(P, R);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int addr = (src + 16 * (2 * h + q)) * 4;
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
              const float v = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(acc[tt][r4])));
              FRAG_AT(P, 16 * q + 4 * r4 + tt) = take ? v : FRAG_AT(P, 16 * q + 4 * r4 + tt);
            }
        }
      }
      const int a0 = src * 4;           // per-node scalars sit on the lanes of group / half 0
      const int g2 = __builtin_amdgcn_ds_bpermute(a0, gc);
      const float r2 = __int_as_float(__builtin_amdgcn_ds_bpermute(a0, __float_as_int(rt.r0)));
      const float s2 = __int_as_float(__builtin_amdgcn_ds_bpermute(a0, __float_as_int(ssum)));
      gcP = take ? g2 : gcP;
      r0P = take ? r2 : r0P;
      swP = take ? s2 : swP;
    };
    const int take1 = n < 32 - cnt ? n : 32 - cnt;
    fill(cnt, 0, take1);
    cnt += take1;
    if (cnt == 32) {
      const float sw = a.sw_from_gather ? swP : a.u.sarr[a.u.smod > 0 ? gcP % a.u.smod : gcP];
      fused_chain<POST>(a, lds, P, gcP, r0P, sw, true, lane);
      cnt = n - take1;
      if (cnt > 0) fill(0, take1, cnt);
    }
  }

  // ---- what is left in P: rows to HBM, ids to list 0 ----
  {
    const bool left = jd < cnt;
    if (left) {
      frag_store_rows(P, a.g.nb, gcP, h);
      if (a.sw_from_gather && h == 0) a.g.sout[gcP] = swP;
    }
    const int n = cnt;
    if (n > 0) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&tail_cnt[0], n);
      base = __builtin_amdgcn_readfirstlane(base);
      if (left && h == 0) list0[base + jd] = gcP;
    }
  }
  // The tail rows were written by this workgroup's own waves through this CU's write-through L1 and are read back by it below:
  // every storing wave drains its stores, then the workgroup barrier; no other CU is involved.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  UpdArgs u = a.u;
  u.list0 = list0;
  u.list1 = list1;
  node_update_loop<true, POST, true, false>(u, lds, tail_cnt[0], tail_cnt[1], wave, FUSED_WAVES, lane);
}
