// dev micro-benchmark (GPU box), round 6: the operand split of the bf16x3 blocks on the MATRIX pipe (gemm_w64_bf3_ms below: residuals
// x - bf16(x) of a whole accumulator tile as one MFMA with minus a selection matrix) against the vector-instruction split (gnnb_dev.h gemm_w64_bf3).
// MEASURED AND NOT ADOPTED (profiles/r06_msplit_ab.txt): alone on the chip the block is within +-3 % either way (the pipelined vector split already hides
// under the 48 product MFMAs); in the product (-DGEMM_BF3_MSPLIT=1 build of round 6) every kernel that uses it got slower: step 0.766 -> 0.811 ms.
//   1. bit-exactness of a chain of blocks on random data, incl. zeros, -0, tiny and huge values (any difference is printed and counted);
//   2. cycles and wall time per 64x64 block and wave at 1 / 2 / 3 / 4 waves per SIMD, alone on the chip.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gnn_branching_amd/csrc -o tools/micro/mfma_split tools/micro/mfma_split.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#include "gnnb_dev.h"

// ---- the operand split on the MATRIX pipe (round 6) ------------------------------------------------------------------------------------
// split_bf3 costs 11 vector instructions per pair of values (cvt_pk, two unpacks, two subtractions -- twice -- and a last cvt_pk): ~176 per
// 64x64 block, half of a chain tile's vector issue, and the fused half-passes are bound by exactly that issue (profiles/r06_wave_scaling_ab.txt,
// DESIGN.md 5.1).  The residual x - bf16(x) of a whole 32 x 32 accumulator tile is ONE matrix operation: with S = minus a selection matrix
// (S[m][k] = -1 where accumulator row m and operand slot k hold the same value of a lane, 0 elsewhere), D = C + S.P computes r = x - p for all
// 16 registers of the tile at once -- every sum has a single non-zero product (-1 . p, exact) and x - p is representable, so r is EXACTLY what
// v_sub_f32 gives: the same pieces, the same bits downstream.  A tile of 16 values per lane = two k-steps of 8: two MFMAs per residual level,
// two levels: 8 v_mfma_f32_32x32x16_bf16 (256 matrix-pipe cycles, 64 of vector issue) + 48 v_cvt_pk per block instead of 176 vector instructions.
// Operand slot k = 8 kg + i of lane (n, kg) holds that lane's fragment register i of the k-step; accumulator register r of lane (n, hh) is row
// m = (r & 3) + 8 (r >> 2) + 4 hh.  First k-step of a tile = registers 0..7 = rows m < 16, second = registers 8..15 = rows 16 + ...: lane
// (m, kg) of the A operand holds S[m][8 kg + i], i < 8 -- at most one -1.0 (bf16 0xBF80).
struct SelOps { u32x4 a, b; };
__device__ __forceinline__ SelOps make_sel(int lane) {
  const int m = lane & 31, kg = lane >> 5;
  SelOps s;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int d = m - 16 * half - 4 * kg;                    // (i & 3) + 8 (i >> 2) for the slot i that maps to row m, if any
    const bool has = d >= 0 && (d & ~11) == 0;               // d in {0..3, 8..11}
    const int i = (d & 3) + 4 * ((d >> 3) & 1);
    u32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (has && (i >> 1) == q) ? (0xBF80u << (16 * (i & 1))) : 0u;
    if (half == 0) s.a = v; else s.b = v;
  }
  return s;
}
// the three pieces of the 16 values v (one accumulator tile = two k-steps) -> pa[level] (k-step 0), pb[level] (k-step 1)
__device__ __forceinline__ void split_tile_ms(f32x16 v, const SelOps& sel, u32x4 (&pa)[3], u32x4 (&pb)[3]) {
  const bf16x8 sa = __builtin_bit_cast(bf16x8, sel.a), sb = __builtin_bit_cast(bf16x8, sel.b);
#pragma unroll
  for (int lvl = 0; lvl < 3; ++lvl) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      pa[lvl][q] = pk_bf16(v[2 * q], v[2 * q + 1]);
      pb[lvl][q] = pk_bf16(v[8 + 2 * q], v[8 + 2 * q + 1]);
    }
    if (lvl < 2) {
      v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sa, __builtin_bit_cast(bf16x8, pa[lvl]), v, 0, 0, 0);
      v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sb, __builtin_bit_cast(bf16x8, pb[lvl]), v, 0, 0, 0);
    }
  }
}
// gemm_w64_bf3 with that split: the products, their order and every accumulator's sequence are those of the vector-split forms
template <int NFRAG, class GetB>
__device__ __forceinline__ void gemm_w64_bf3_ms(const float* wl, int lane, Frag& acc, GetB getB) {
  const u32x4* w = reinterpret_cast<const u32x4*>(wl) + lane;
  const SelOps sel = make_sel(lane);
#pragma unroll
  for (int t = 0; t < 2 * NFRAG; ++t) {                // tile t of the input: fragment registers 16 t .. 16 t + 15 = k-steps 2 t, 2 t + 1
    f32x16 v;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = getB(16 * t + r);
    u32x4 pa[3], pb[3];
    split_tile_ms(v, sel, pa, pb);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int fk = 2 * t + half;
      const bf16x8 x1 = __builtin_bit_cast(bf16x8, half ? pb[0] : pa[0]), x2 = __builtin_bit_cast(bf16x8, half ? pb[1] : pa[1]),
                   x3 = __builtin_bit_cast(bf16x8, half ? pb[2] : pa[2]);
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
      __builtin_amdgcn_sched_barrier(0);      // one k-step's weight fragments at a time (else hipcc hoists them all and spills)
    }
  }
}


// MODE 0: vector split, sequential; 1: vector split, software-pipelined; 2: matrix-pipe split
template <int MODE>
__device__ __forceinline__ void block(const float* wl, int lane, Frag& acc, const Frag& X, float scale) {
  auto getB = [&](int s) { return FRAG_AT(X, s) * scale; };
  if constexpr (MODE == 2) gemm_w64_bf3_ms<1>(wl, lane, acc, getB);
  else gemm_w64_bf3<1, MODE == 1>(wl, lane, acc, getB);
}

template <int MODE>
__global__ __launch_bounds__(1024) void k_exact(const float* wsrc, const float* xin, float* out, int nblocks) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 3 * 6144; i += blockDim.x) lds[i] = wsrc[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  Frag X, H;
#pragma unroll
  for (int R = 0; R < 32; ++R) FRAG_AT(X, R) = xin[((long)wv * 64 + lane) * 32 + R];
  for (int b = 0; b < nblocks; ++b) {
#pragma unroll
    for (int R = 0; R < 32; ++R) FRAG_AT(H, R) = 0.25f;
    block<MODE>(lds + 6144 * (b % 3), lane, H, X, 0.75f);
    if (b + 1 < nblocks) {
#pragma unroll
      for (int R = 0; R < 32; ++R) FRAG_AT(X, R) = relu_nan(FRAG_AT(H, R)) - 0.125f * FRAG_AT(X, R);
    }
  }
#pragma unroll
  for (int R = 0; R < 32; ++R) out[((long)wv * 64 + lane) * 32 + R] = FRAG_AT(H, R);
}

template <int MODE>
__global__ __launch_bounds__(1024) void k_time(const float* wsrc, float* out, unsigned long long* cyc, int nblocks) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 3 * 6144; i += blockDim.x) lds[i] = wsrc[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  Frag X, H;
#pragma unroll
  for (int R = 0; R < 32; ++R) { FRAG_AT(X, R) = 0.001f * (float)(lane + R) + 0.37f; FRAG_AT(H, R) = 0.0f; }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int b = 0; b < nblocks; ++b) {
    block<MODE>(lds + 6144 * (b % 3), lane, H, X, 1.0f);
    frag_relu(H);                                     // chained like the product's node update: the next block's input is this block's output
#pragma unroll
    for (int R = 0; R < 32; ++R) { FRAG_AT(X, R) = FRAG_AT(H, R) * 1e-3f + 0.11f; FRAG_AT(H, R) = 0.0f; }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
  float s = 0.0f;
#pragma unroll
  for (int R = 0; R < 32; ++R) s += FRAG_AT(H, R) + FRAG_AT(X, R);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void timeit(const char* name, int waves, const float* w, float* out, unsigned long long* cyc, int nblocks) {
  hipFuncSetAttribute((const void*)k_time<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 6144 * 4);
  const int nwg = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_time<MODE><<<nwg, waves * 64, 3 * 6144 * 4>>>(w, out, cyc, nblocks);
  hipEventRecord(e0);
  k_time<MODE><<<nwg, waves * 64, 3 * 6144 * 4>>>(w, out, cyc, nblocks);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nwg * waves);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (auto v : h) sum += (double)v;
  printf("%-28s %2d waves / CU (%d per SIMD): %8.0f ticks per block and wave; kernel %8.1f us = %6.0f ns per block and SIMD\n", name, waves, waves / 4,
         sum / h.size() / nblocks, 1e3 * ms, 1e6 * ms / nblocks / (waves / 4));
}

int main() {
  // weights: a bf16x3 image of random values in MFMA operand order is not needed for exactness of the SPLIT -- any bf16 pieces do: random bf16 words
  std::vector<float> w(3 * 6144);
  srand(7);
  for (auto& v : w) {
    unsigned hi = ((rand() & 0xff) - 128 + 0x3f00) & 0xffff, lo = ((rand() & 0xff) - 128 + 0x3b00) & 0xffff;      // bf16 pairs around 0.5 / 0.002, either sign
    if (rand() & 1) hi |= 0x8000;
    if (rand() & 1) lo |= 0x8000;
    unsigned u = (hi << 16) | lo;
    memcpy(&v, &u, 4);
  }
  const int nwg = 64, waves = 8, nw = nwg * waves;
  std::vector<float> x((size_t)nw * 64 * 32);
  for (size_t i = 0; i < x.size(); ++i) {
    const int k = rand() % 100;
    float v = (float)rand() / RAND_MAX * 2.0f - 1.0f;
    if (k < 5) v = 0.0f; else if (k < 7) v = -0.0f; else if (k < 12) v *= 1e-30f; else if (k < 17) v *= 1e20f; else if (k < 30) v *= 37.5f; else if (k < 34) v *= 1e-41f * 1e3f;
    x[i] = v;
  }
  float *dw, *dx, *dout[3]; unsigned long long* dc;
  hipMalloc(&dw, w.size() * 4); hipMalloc(&dx, x.size() * 4); hipMalloc(&dc, 256 * 16 * 8);
  for (auto& p : dout) hipMalloc(&p, std::max(x.size(), (size_t)256 * 1024) * 4);
  hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k_exact<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 6144 * 4);
  hipFuncSetAttribute((const void*)k_exact<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 6144 * 4);
  hipFuncSetAttribute((const void*)k_exact<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 6144 * 4);
  for (int nblocks : {1, 4}) {
    k_exact<0><<<nwg, waves * 64, 3 * 6144 * 4>>>(dw, dx, dout[0], nblocks);
    k_exact<1><<<nwg, waves * 64, 3 * 6144 * 4>>>(dw, dx, dout[1], nblocks);
    k_exact<2><<<nwg, waves * 64, 3 * 6144 * 4>>>(dw, dx, dout[2], nblocks);
    hipDeviceSynchronize();
    std::vector<float> o[3];
    for (int m = 0; m < 3; ++m) { o[m].resize(x.size()); hipMemcpy(o[m].data(), dout[m], x.size() * 4, hipMemcpyDeviceToHost); }
    size_t d01 = 0, d02 = 0, nan0 = 0, nan2 = 0, shown = 0;
    for (size_t i = 0; i < x.size(); ++i) {
      unsigned a, b, c; memcpy(&a, &o[0][i], 4); memcpy(&b, &o[1][i], 4); memcpy(&c, &o[2][i], 4);
      const bool n0 = std::isnan(o[0][i]), n2 = std::isnan(o[2][i]);
      nan0 += n0; nan2 += n2;
      if (a != b && !(n0 && std::isnan(o[1][i]))) ++d01;
      if (a != c && !(n0 && n2)) { ++d02; if (shown++ < 8) printf("  diff at %zu: vector %.9g (%08x)  matrix %.9g (%08x)\n", i, o[0][i], a, o[2][i], c); }
    }
    printf("chain of %d block(s), %zu outputs: sequential vs pipelined vector split differ at %zu; vector vs MATRIX split differ at %zu (NaN outputs %zu / %zu)\n",
           nblocks, x.size(), d01, d02, nan0, nan2);
  }
  const int nb = 2000;
  for (int wv : {4, 8, 12, 16}) {
    timeit<1>("vector split (pipelined)", wv, dw, dout[0], dc, nb);
    timeit<2>("matrix-pipe split", wv, dw, dout[0], dc, nb);
  }
  return 0;
}
