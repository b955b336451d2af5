// gnnb_dev.h -- part of libgnnb.so, included by gnnb.hip (one translation unit; see its header comment).
// device helpers of the MFMA kernels: fragments, the 64-wide GEMM blocks (fp32 MFMA and three-piece bf16), row loads / stores, tile maps.
#pragma once

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
struct Frag {
  f32x16 t[2];  // 64 features x 32 nodes; register R = 16*it + r <-> feature 8*(R>>2) + 4*h + (R&3)
};

#define FRAG_AT(x, R) ((x).t[(R) >> 4][(R)&15])

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// acc += W(64 x 2*KSTEPS, operand order in LDS) * in, where getB(s) yields the B operand of k-step s
template <int KSTEPS, class GetB>
__device__ __forceinline__ void gemm_w64(const float* wl, int lane, Frag& acc, GetB getB) {
  // A operands are prefetched one 4-k-step block ahead; the sched_barrier keeps hipcc from hoisting
  // every LDS read of the fully unrolled chain to the top (which spills the 256-VGPR budget).
  const f32x4* w4 = reinterpret_cast<const f32x4*>(wl) + lane;
  f32x4 a0 = w4[0], a1 = w4[64];
#pragma unroll
  for (int s4 = 0; s4 < KSTEPS / 4; ++s4) {
    f32x4 n0 = a0, n1 = a1;
    if (s4 + 1 < KSTEPS / 4) {
      n0 = w4[((s4 + 1) * 2 + 0) * 64];
      n1 = w4[((s4 + 1) * 2 + 1) * 64];
    }
    __builtin_amdgcn_sched_barrier(0);      // reads of the next block issue BEFORE this block's MFMAs, not after them
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float b = getB(s4 * 4 + c);
      acc.t[0] = mfma32(a0[c], b, acc.t[0]);
      acc.t[1] = mfma32(a1[c], b, acc.t[1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    a0 = n0;
    a1 = n1;
  }
}

// first layers on scalar node features: x[s] = input feature 2*s + h of this lane's node
template <int KSTEPS>
__device__ __forceinline__ void gemm_small(const float* wl, int lane, Frag& acc, const float (&x)[KSTEPS]) {
#pragma unroll
  for (int s = 0; s < KSTEPS; ++s) {
    const float a0 = wl[(s * 2 + 0) * 64 + lane];
    const float a1 = wl[(s * 2 + 1) * 64 + lane];
    acc.t[0] = mfma32(a0, x[s], acc.t[0]);
    acc.t[1] = mfma32(a1, x[s], acc.t[1]);
  }
}

// ---- the 64x64 block on v_mfma_f32_32x32x16_bf16 with both operands in three bf16 pieces (gnnb_pack.h pack_w64_bf3):
// acc += W.x with the six products w1x1 + w1x2 + w2x1 + w1x3 + w2x2 + w3x1, smallest first.  48 MFMAs of 32 cycles per
// 64 inputs instead of 64 of 64 cycles; the price is the VALU work of splitting the activations (cvt_pk + subtract per piece).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pk_bf16(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{a, b}, bf16x2v)); }
__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// three-piece split of the 8 B-operand values of k-step fk (registers 8 fk .. 8 fk + 7 of the input fragment)
__device__ __forceinline__ void split_pair_bf3(float a, float b, unsigned& u1, unsigned& u2, unsigned& u3);
template <class GetB>
__device__ __forceinline__ void split_bf3(GetB& getB, int fk, u32x4& p1, u32x4& p2, u32x4& p3) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned u1, u2, u3;
    split_pair_bf3(getB(8 * fk + 2 * q), getB(8 * fk + 2 * q + 1), u1, u2, u3);
    p1[q] = u1; p2[q] = u2; p3[q] = u3;
  }
}

// ---- rows in three bf16 pieces ("rows3"): what a conv edge's aggregate reads when it runs on the bf16 matrix rate -------------------
// v_mfma_f32_32x32x2_f32 / _16x16x4_f32 run on the VECTOR pipe: while a wave's fp32 MFMAs execute, no other wave of the SIMD issues a
// vector instruction (tools/micro/mfma_valu_overlap.hip: a partner wave's v_fma stream makes NO progress beside them and 89 % of its
// progress beside v_mfma_f32_32x32x16_bf16).  So the fp32 tap MFMAs of the gathers and the operand splits of the chains add up on one
// pipe.  A gather on the bf16 rate needs its source rows as bf16 pieces WITHOUT splitting them per read (a row is read by 2-4 tiles):
// the producer of the rows writes the pieces once.  Layout: a row of 64 values = 32 pairs x 3 pieces, pair u (values 2u, 2u + 1) =
// 12 bytes at 12 u: {p1 | p2 | p3}, each a dword with the even value in its low half.  x = p1 + p2 + p3 to 24 bits (round to nearest
// each time, as the activations of gemm_w64_bf3).
#define ROW3_BYTES 384
#define ROW3_FLOATS 96
__device__ __forceinline__ void split_pair_bf3(float a, float b, unsigned& u1, unsigned& u2, unsigned& u3) {
  u1 = pk_bf16(a, b);
  const float ra = a - __uint_as_float(u1 << 16), rb = b - __uint_as_float(u1 & 0xffff0000u);
  u2 = pk_bf16(ra, rb);
  const float sa = ra - __uint_as_float(u2 << 16), sb = rb - __uint_as_float(u2 & 0xffff0000u);
  u3 = pk_bf16(sa, sb);
}
// fragment -> rows3: lane (j, h) owns values [8q + 4h, 8q + 4h + 4) of its row = pairs 4q + 2h, 4q + 2h + 1: 24 contiguous bytes per q
__device__ __forceinline__ void frag_store_rows3(const Frag& x, void* base, long row, int h);

// GEMM_BF3_PIPE (default 1): software-pipelined form -- the split of k-step s + 1 (~56 vector instructions) is issued BETWEEN the 12
// MFMAs of k-step s (sched_group_barrier: one MFMA, then up to five vector instructions, twelve times), so a wave keeps the matrix
// pipe fed while it splits (an MFMA holds the vector issue for 8 of its 32 cycles: five 4-cycle instructions fit each gap).  The
// sequential form (0) split a k-step, then issued its 12 MFMAs back to back: ~240 + 384 cycles per k-step with the matrix pipe idle
// during every split unless another wave of the SIMD filled it.  Same operations on the same values in the same order per
// accumulator: results are bit-identical.
#ifndef GEMM_BF3_PIPE
#define GEMM_BF3_PIPE 1
#endif
template <int NFRAG, bool PIPE = (GEMM_BF3_PIPE != 0), class GetB>
__device__ __forceinline__ void gemm_w64_bf3(const float* wl, int lane, Frag& acc, GetB getB) {
  const u32x4* w = reinterpret_cast<const u32x4*>(wl) + lane;
  if constexpr (PIPE) {
  u32x4 p1, p2, p3;
  split_bf3(getB, 0, p1, p2, p3);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int fk = 0; fk < 4 * NFRAG; ++fk) {
    const bf16x8 x1 = __builtin_bit_cast(bf16x8, p1), x2 = __builtin_bit_cast(bf16x8, p2), x3 = __builtin_bit_cast(bf16x8, p3);
    const bf16x8 w1a = __builtin_bit_cast(bf16x8, w[((fk * 2 + 0) * 3 + 0) * 64]), w1b = __builtin_bit_cast(bf16x8, w[((fk * 2 + 1) * 3 + 0) * 64]);
    const bf16x8 w2a = __builtin_bit_cast(bf16x8, w[((fk * 2 + 0) * 3 + 1) * 64]), w2b = __builtin_bit_cast(bf16x8, w[((fk * 2 + 1) * 3 + 1) * 64]);
    const bf16x8 w3a = __builtin_bit_cast(bf16x8, w[((fk * 2 + 0) * 3 + 2) * 64]), w3b = __builtin_bit_cast(bf16x8, w[((fk * 2 + 1) * 3 + 2) * 64]);
    acc.t[0] = mfma_bf16(w3a, x1, acc.t[0]);
    acc.t[1] = mfma_bf16(w3b, x1, acc.t[1]);
    acc.t[0] = mfma_bf16(w2a, x2, acc.t[0]);
    acc.t[1] = mfma_bf16(w2b, x2, acc.t[1]);
    acc.t[0] = mfma_bf16(w1a, x3, acc.t[0]);
    acc.t[1] = mfma_bf16(w1b, x3, acc.t[1]);
    acc.t[0] = mfma_bf16(w2a, x1, acc.t[0]);
    acc.t[1] = mfma_bf16(w2b, x1, acc.t[1]);
    acc.t[0] = mfma_bf16(w1a, x2, acc.t[0]);
    acc.t[1] = mfma_bf16(w1b, x2, acc.t[1]);
    acc.t[0] = mfma_bf16(w1a, x1, acc.t[0]);
    acc.t[1] = mfma_bf16(w1b, x1, acc.t[1]);
    if (fk + 1 < 4 * NFRAG) split_bf3(getB, fk + 1, p1, p2, p3);
    // the 6 weight reads first (their MFMAs wait on them), then MFMA / vector interleaved
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  } else {
#pragma unroll
  for (int fk = 0; fk < 4 * NFRAG; ++fk) {         // fragment fk / 4, k-step fk % 4: registers 8 (fk % 4) .. + 7
    u32x4 p1, p2, p3;
    split_bf3(getB, fk, p1, p2, p3);
    const bf16x8 x1 = __builtin_bit_cast(bf16x8, p1), x2 = __builtin_bit_cast(bf16x8, p2), x3 = __builtin_bit_cast(bf16x8, p3);
    // the two output tiles' product chains alternate: an MFMA's accumulator was written two instructions earlier, not one
    // (back to back, each of the six waits out its predecessor's result latency); each accumulator still sees its six
    // products in the same order
    const bf16x8 w1a = __builtin_bit_cast(bf16x8, w[((fk * 2 + 0) * 3 + 0) * 64]), w1b = __builtin_bit_cast(bf16x8, w[((fk * 2 + 1) * 3 + 0) * 64]);
    const bf16x8 w2a = __builtin_bit_cast(bf16x8, w[((fk * 2 + 0) * 3 + 1) * 64]), w2b = __builtin_bit_cast(bf16x8, w[((fk * 2 + 1) * 3 + 1) * 64]);
    const bf16x8 w3a = __builtin_bit_cast(bf16x8, w[((fk * 2 + 0) * 3 + 2) * 64]), w3b = __builtin_bit_cast(bf16x8, w[((fk * 2 + 1) * 3 + 2) * 64]);
    acc.t[0] = mfma_bf16(w3a, x1, acc.t[0]);
    acc.t[1] = mfma_bf16(w3b, x1, acc.t[1]);
    acc.t[0] = mfma_bf16(w2a, x2, acc.t[0]);
    acc.t[1] = mfma_bf16(w2b, x2, acc.t[1]);
    acc.t[0] = mfma_bf16(w1a, x3, acc.t[0]);
    acc.t[1] = mfma_bf16(w1b, x3, acc.t[1]);
    acc.t[0] = mfma_bf16(w2a, x1, acc.t[0]);
    acc.t[1] = mfma_bf16(w2b, x1, acc.t[1]);
    acc.t[0] = mfma_bf16(w1a, x2, acc.t[0]);
    acc.t[1] = mfma_bf16(w1b, x2, acc.t[1]);
    acc.t[0] = mfma_bf16(w1a, x1, acc.t[0]);
    acc.t[1] = mfma_bf16(w1b, x1, acc.t[1]);
    __builtin_amdgcn_sched_barrier(0);      // one k-step's pieces and weight fragments at a time (else hipcc hoists them all and spills)
  }
  }
}

__device__ __forceinline__ void frag_bias(Frag& a, const float* bl, int h) {
  const f32x4* b4 = reinterpret_cast<const f32x4*>(bl + h * 32);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 v = b4[q];
#pragma unroll
    for (int c = 0; c < 4; ++c) FRAG_AT(a, 4 * q + c) = v[c];
  }
}

// torch's relu propagates NaN (fmaxf would swallow it and hide a 0/0 of compute_ratio)
__device__ __forceinline__ float relu_nan(float x) { return x < 0.0f ? 0.0f : x; }
// the same in ONE vector instruction: IEEE-754-2019 maximum propagates NaN (v_maximum3_f32 on gfx950; compare + select costs two).  Differs from
// relu_nan only in the sign of a zero it returns for -0.0 (+0.0 here); used where 32 values per lane go through it (round 6: -64 vector
// instructions per relu of a fragment, on kernels whose vector issue is 87 % busy, profiles/r06_input_update_ablation.txt)
__device__ __forceinline__ float relu_max(float x) { return __builtin_elementwise_maximum(x, 0.0f); }

__device__ __forceinline__ void frag_relu(Frag& a) {
#pragma unroll
  for (int R = 0; R < 32; ++R) FRAG_AT(a, R) = relu_max(FRAG_AT(a, R));
}

__device__ __forceinline__ void frag_scale(Frag& a, float s) {
#pragma unroll
  for (int R = 0; R < 32; ++R) FRAG_AT(a, R) *= s;
}

// row-major (G, 64) <-> fragment: lane (j, h) owns features [8q+4h, 8q+4h+4) of row `row`, q = 0..7
__device__ __forceinline__ void frag_load_rows(Frag& x, const float* base, long row, int h) {
  const f32x4* p = reinterpret_cast<const f32x4*>(base + row * 64 + 4 * h);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 v = p[2 * q];
#pragma unroll
    for (int c = 0; c < 4; ++c) FRAG_AT(x, 4 * q + c) = v[c];
  }
}
__device__ __forceinline__ void frag_load_rowptr(Frag& x, const float* rowptr, int h) {
  const f32x4* p = reinterpret_cast<const f32x4*>(rowptr + 4 * h);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 v = p[2 * q];
#pragma unroll
    for (int c = 0; c < 4; ++c) FRAG_AT(x, 4 * q + c) = v[c];
  }
}
__device__ __forceinline__ void frag_store_rows(const Frag& x, float* base, long row, int h) {
  f32x4* p = reinterpret_cast<f32x4*>(base + row * 64 + 4 * h);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    f32x4 v;
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = FRAG_AT(x, 4 * q + c);
    p[2 * q] = v;
  }
}
// tile-major scratch layout for the cached P vectors: float4 index (tile*8 + q)*64 + lane (1 KiB per wave-instruction)
__device__ __forceinline__ void frag_load_tiled(Frag& x, const float* base, long tile, int lane) {
  const f32x4* p = reinterpret_cast<const f32x4*>(base) + tile * 512 + lane;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const f32x4 v = p[q * 64];
#pragma unroll
    for (int c = 0; c < 4; ++c) FRAG_AT(x, 4 * q + c) = v[c];
  }
}
__device__ __forceinline__ void frag_store_tiled(const Frag& x, float* base, long tile, int lane) {
  f32x4* p = reinterpret_cast<f32x4*>(base) + tile * 512 + lane;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    f32x4 v;
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = FRAG_AT(x, 4 * q + c);
    p[q * 64] = v;
  }
}

__device__ __forceinline__ void frag_store_rows3(const Frag& x, void* base, long row, int h) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  u32x2* p = reinterpret_cast<u32x2*>(reinterpret_cast<char*>(base) + row * ROW3_BYTES + 24 * h);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    unsigned a1, a2, a3, b1, b2, b3;
    split_pair_bf3(FRAG_AT(x, 4 * q), FRAG_AT(x, 4 * q + 1), a1, a2, a3);
    split_pair_bf3(FRAG_AT(x, 4 * q + 2), FRAG_AT(x, 4 * q + 3), b1, b2, b3);
    p[6 * q + 0] = u32x2{a1, a2};
    p[6 * q + 1] = u32x2{a3, b1};
    p[6 * q + 2] = u32x2{b2, b3};
  }
}

// any NaN among the 32 values of a lane: a NaN-propagating maximum over them (v_maximum3_f32: two values per instruction) and ONE compare,
// instead of 32 compares and 32 scalar ORs
__device__ __forceinline__ bool frag_has_nan(const Frag& x) {
  float m = FRAG_AT(x, 0);
#pragma unroll
  for (int R = 1; R + 1 < 32; R += 2) m = __builtin_elementwise_maximum(__builtin_elementwise_maximum(m, FRAG_AT(x, R)), FRAG_AT(x, R + 1));
  m = __builtin_elementwise_maximum(m, FRAG_AT(x, 31));
  return m != m;
}

// compute_ratio (graph_conv.py:499-514), op for op
struct Ratio { float r0, r1, beta, amb, live; };
__device__ __forceinline__ Ratio compute_ratio(float lb, float ub) {
  Ratio r;
  const float lower_temp = lb - relu_nan(lb);
  const float upper_temp = relu_nan(ub);
  r.r0 = upper_temp / (upper_temp - lower_temp);
  r.beta = -1.0f * lower_temp * r.r0;
  r.amb = r.beta > 0.0f ? 1.0f : 0.0f;
  r.r1 = (1.0f - 2.0f * (r.r0 * r.amb)) * r.amb + r.r0;
  r.live = (r.r0 != 0.0f) ? 1.0f : 0.0f;   // (ratio_0 != 0), :178 / :347 (NaN != 0 is true)
  return r;
}

// global -> LDS copy of a weight pack.  Loads are issued 8 at a time before their LDS stores: a plain copy loop keeps
// one 16-B load in flight per thread and serialises ~10 L2 round trips per workgroup at the start of every launch.
__device__ __forceinline__ void copy_to_lds(float* lds, const float* src, int nfloats) {
  const f32x4* g = reinterpret_cast<const f32x4*>(src);
  f32x4* l = reinterpret_cast<f32x4*>(lds);
  const int n4 = nfloats / 4, stride = blockDim.x;
  for (int i0 = threadIdx.x; i0 < n4; i0 += 8 * stride) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * stride;
      v[u] = g[i < n4 ? i : i0];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * stride;
      if (i < n4) l[i] = v[u];
    }
  }
}
// the same by threads tid = 0 .. nthr - 1 of a subset of the workgroup (the caller passes its index in the subset)
__device__ __forceinline__ void copy_to_lds_part(float* lds, const float* src, int nfloats, int tid, int nthr) {
  const f32x4* g = reinterpret_cast<const f32x4*>(src);
  f32x4* l = reinterpret_cast<f32x4*>(lds);
  const int n4 = nfloats / 4;
  for (int i0 = tid; i0 < n4; i0 += 8 * nthr) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * nthr;
      v[u] = g[i < n4 ? i : i0];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * nthr;
      if (i < n4) l[i] = v[u];
    }
  }
}

__device__ __forceinline__ void stage_pack(float* lds, const float* pack, int nfloats) {
  copy_to_lds(lds, pack, nfloats);
  __syncthreads();
}

// ---- tile -> node mapping (gnnb_pack.h TileMap) ----
struct DTileMap { int mode, N, C, H, W, CT, PY, PX, ay, ax, NBY, NBX, NCG, TPS, lpy, lpx; unsigned tps_magic; };
// sample = tile / TPS for a wave-uniform tile index: the loops carry it as a `long`, and a 64-bit signed division by a run-time divisor is ~60 vector
// instructions per tile (an eighth of what a tile of the input update issues, profiles/r06_vector_trims_ab.txt).  tps_magic = floor(2^32 / TPS) + 1 (host,
// to_dtm): the high word of tile * magic is the quotient or one above it for every tile < 2^32, the compare puts it right -- three scalar instructions.
__device__ __forceinline__ int tile_sample(const DTileMap& tm, long tile) {
  const unsigned n = (unsigned)__builtin_amdgcn_readfirstlane((int)tile), d = (unsigned)tm.TPS;
  if (d <= 1u) return (int)n;
  unsigned q = __umulhi(n, tm.tps_magic);      // floor(n / d) or one above it (magic > 2^32 / d)
  if (q * d > n) --q;
  return (int)q;
}
struct TileCtx { long sample; int n, cg, by, bx, y, x; bool valid; };

// lane j of tile `tile`: which node of which sample.  mode 0: 32 consecutive rows of the flat (B*N) layer;
// mode 1: CT channels x (PY x PX) pixel block of one sample (the blocks an MFMA gather works on).
__device__ __forceinline__ TileCtx tile_decode(const DTileMap& tm, long tile, int j, long total_rows) {
  TileCtx c;
  if (tm.mode == 0) {
    const long g = tile * 32 + j;
    c.valid = g < total_rows;
    const long gc = c.valid ? g : total_rows - 1;
    c.sample = gc / tm.N;
    c.n = (int)(gc - c.sample * tm.N);
    c.cg = c.by = c.bx = c.y = c.x = 0;
  } else {
    c.sample = tile / tm.TPS;
    const int t = (int)(tile - c.sample * tm.TPS);
    const int nb = tm.NBY * tm.NBX;
    c.cg = t / nb;
    const int rem = t - c.cg * nb;
    c.by = rem / tm.NBX;
    c.bx = rem - c.by * tm.NBX;
    const int pp = tm.PY * tm.PX;
    const int cl = j / pp;
    const int r2 = j - cl * pp;
    const int py = r2 / tm.PX, px = r2 - py * tm.PX;
    c.y = c.by * tm.PY + tm.ay + py;
    c.x = c.bx * tm.PX + tm.ax + px;
    c.valid = cl < tm.CT && (unsigned)c.y < (unsigned)tm.H && (unsigned)c.x < (unsigned)tm.W;
    c.n = c.valid ? ((c.cg * tm.CT + cl) * tm.H + c.y) * tm.W + c.x : 0;
  }
  return c;
}

// block tiles without integer divisions: ttab[t] = cg | by << 8 | bx << 20 for tile t of a sample (built on the host),
// PY and PX are powers of two.
__device__ __forceinline__ TileCtx block_decode(const DTileMap& tm, const int* ttab, long sample, int t, int j) {
  TileCtx c;
  c.sample = sample;
  const int e = ttab[t];
  c.cg = e & 0xff;
  c.by = (e >> 8) & 0xfff;
  c.bx = (e >> 20) & 0xfff;
  const int cl = j >> (tm.lpy + tm.lpx);
  const int py = (j >> tm.lpx) & (tm.PY - 1), px = j & (tm.PX - 1);
  c.y = c.by * tm.PY + tm.ay + py;
  c.x = c.bx * tm.PX + tm.ax + px;
  c.valid = cl < tm.CT && (unsigned)c.y < (unsigned)tm.H && (unsigned)c.x < (unsigned)tm.W;
  c.n = c.valid ? ((c.cg * tm.CT + cl) * tm.H + c.y) * tm.W + c.x : 0;
  return c;
}

// persistent tile loop: workgroup -> contiguous chunk of tiles, chunks dealt so that the workgroups of one
// XCD (blockIdx % 8 labels the XCD group) own neighbouring chunks: the samples they gather from stay in that L2.
__device__ __forceinline__ void tile_range(long ntiles, int waves, long& begin, long& end) {
  int wg = blockIdx.x;
  const int nwg = gridDim.x;
  if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);
  // chunks differ by at most one tile (rounding every chunk up to whole rounds of `waves` tiles left up to a sixth of the
  // workgroups without work on the 81-tiles-per-sample edge)
  (void)waves;
  const long base = ntiles / nwg, rem = ntiles - base * nwg;
  begin = (long)wg * base + (wg < rem ? wg : rem);
  end = begin + base + (wg < rem ? 1 : 0);
}

// dev instrumentation (-DFUSED_TIMING): cycle sums per kernel phase, read back through gnnb_debug_read (tools/gather_timing.py)
#ifdef FUSED_TIMING      // dev: per-phase cycle sums over all waves (tools/gather_timing.py reads them through gnnb_debug_read)
__device__ unsigned long long g_fused_t[16];
#define FUSED_TIMING_ON 1
#define FT_DECL unsigned long long ft_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long ft_last = __builtin_readcyclecounter()
#define FT_MARK(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); ft_[i] += n_ - ft_last; ft_last = n_; } while (0)
#define FT_FLUSH() do { if ((threadIdx.x & 63) == 0) { for (int i_ = 0; i_ < 10; ++i_) atomicAdd(&g_fused_t[i_], ft_[i_]); atomicAdd(&g_fused_t[15], 1ull); } } while (0)
#else
#define FUSED_TIMING_ON 0
#define FT_DECL
#define FT_MARK(i)
#define FT_FLUSH()
#endif

#define WG_MLP 512       // 8 waves: 2 per SIMD share one LDS copy of the weights
#define WAVES_MLP 8
