// dev micro-benchmark (GPU box): cycles of ONE 64x64 bf16x3 block (gemm_w64_bf3, the product's code: 48 v_mfma_f32_32x32x16_bf16 + the
// operand splits, weights out of LDS) per wave, as a function of the waves per SIMD -- against its 48 x 32 = 1536 cycles of MFMA issue.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gnn_branching_amd/csrc -o tools/micro/chain_block_cycles tools/micro/chain_block_cycles.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#include "gnnb_dev.h"

template <bool PIPE, bool RELU>
__global__ __launch_bounds__(1024) void k_blocks(const float* wsrc, float* out, unsigned long long* cyc, int nblocks) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 3 * 6144; i += blockDim.x) lds[i] = wsrc[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  Frag X, H;
#pragma unroll
  for (int R = 0; R < 32; ++R) { FRAG_AT(X, R) = 0.001f * (float)(lane + R); FRAG_AT(H, R) = 0.0f; }
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int b = 0; b < nblocks; ++b) {
    gemm_w64_bf3<1, PIPE>(lds + 6144 * (b % 3), lane, H, [&](int s) { return FRAG_AT(X, s); });
    if (RELU) {
      frag_relu(H);
#pragma unroll
      for (int R = 0; R < 32; ++R) { FRAG_AT(X, R) = FRAG_AT(H, R) * 1e-3f; FRAG_AT(H, R) = 0.0f; }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
  float s = 0.0f;
#pragma unroll
  for (int R = 0; R < 32; ++R) s += FRAG_AT(H, R) + FRAG_AT(X, R);
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the same block on v_mfma_f32_16x16x32_bf16 (timing only: operand order not validated): 32 nodes = 2 node tiles of 16, 4 feature tiles,
// 2 k-steps of 32, six products -> 96 MFMAs of half the flops; same LDS reads (24 x 16 B per lane) and the same 16 pair splits
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <bool RELU>
__global__ __launch_bounds__(1024) void k_blocks16(const float* wsrc, float* out, unsigned long long* cyc, int nblocks) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int i = threadIdx.x; i < 3 * 6144; i += blockDim.x) lds[i] = wsrc[i];
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float X[32];
  f32x4v H[2][4];
#pragma unroll
  for (int R = 0; R < 32; ++R) X[R] = 0.001f * (float)(lane + R);
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) H[nt][mt] = f32x4v{0.f, 0.f, 0.f, 0.f};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int b = 0; b < nblocks; ++b) {
    const u32x4* w = reinterpret_cast<const u32x4*>(lds + 6144 * (b % 3)) + lane;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x4 p[2][3];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          unsigned u1, u2, u3;
          split_pair_bf3(X[16 * nt + 8 * ks + 2 * q], X[16 * nt + 8 * ks + 2 * q + 1], u1, u2, u3);
          p[nt][0][q] = u1; p[nt][1][q] = u2; p[nt][2][q] = u3;
        }
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const bf16x8 w1 = __builtin_bit_cast(bf16x8, w[((ks * 4 + mt) * 3 + 0) * 64]), w2 = __builtin_bit_cast(bf16x8, w[((ks * 4 + mt) * 3 + 1) * 64]),
                     w3 = __builtin_bit_cast(bf16x8, w[((ks * 4 + mt) * 3 + 2) * 64]);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          const bf16x8 x1 = __builtin_bit_cast(bf16x8, p[nt][0]), x2 = __builtin_bit_cast(bf16x8, p[nt][1]), x3 = __builtin_bit_cast(bf16x8, p[nt][2]);
          f32x4v a = H[nt][mt];
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3, x1, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, x2, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, x3, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2, x1, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, x2, a, 0, 0, 0);
          a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, x1, a, 0, 0, 0);
          H[nt][mt] = a;
        }
      }
    }
    if (RELU) {
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            X[16 * nt + 4 * mt + r] = relu_nan(H[nt][mt][r]) * 1e-3f;
            H[nt][mt][r] = 0.0f;
          }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
  float sacc = 0.0f;
#pragma unroll
  for (int R = 0; R < 32; ++R) sacc += X[R];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) sacc += H[nt][mt][0] + H[nt][mt][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sacc;
}

template <bool RELU>
static void run16(const char* name, int waves, const float* w, float* out, unsigned long long* cyc, int nblocks) {
  hipFuncSetAttribute((const void*)k_blocks16<RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 6144 * 4);
  const int nwg = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_blocks16<RELU><<<nwg, waves * 64, 3 * 6144 * 4>>>(w, out, cyc, nblocks);
  hipEventRecord(e0);
  k_blocks16<RELU><<<nwg, waves * 64, 3 * 6144 * 4>>>(w, out, cyc, nblocks);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nwg * waves);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (auto v : h) sum += (double)v;
  printf("%-34s %2d waves / CU (%d per SIMD): %8.0f ticks per block and wave; kernel %.1f us = %.0f ns per block and SIMD; ticks / wall = %.2f GHz\n", name, waves,
         waves / 4, sum / h.size() / nblocks, 1e3 * ms, 1e6 * ms / nblocks / (waves / 4), sum / h.size() / (1e6 * ms));
}

template <bool PIPE, bool RELU>
static void run(const char* name, int waves, const float* w, float* out, unsigned long long* cyc, int nblocks) {
  hipFuncSetAttribute((const void*)k_blocks<PIPE, RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 6144 * 4);
  const int nwg = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_blocks<PIPE, RELU><<<nwg, waves * 64, 3 * 6144 * 4>>>(w, out, cyc, nblocks);
  hipEventRecord(e0);
  k_blocks<PIPE, RELU><<<nwg, waves * 64, 3 * 6144 * 4>>>(w, out, cyc, nblocks);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(nwg * waves);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (auto v : h) sum += (double)v;
  printf("%-34s %2d waves / CU (%d per SIMD): %8.0f ticks per block and wave; kernel %.1f us = %.0f ns per block and SIMD; ticks / wall = %.2f GHz\n", name, waves,
         waves / 4, sum / h.size() / nblocks, 1e3 * ms, 1e6 * ms / nblocks / (waves / 4), sum / h.size() / (1e6 * ms));
}

int main() {
  float *w, *out;
  unsigned long long* cyc;
  std::vector<float> hw(3 * 6144);
  // random bf16 pieces (two per word) with magnitudes ~0.1: all-zero operands let the chip clock higher
  srand(1);
  for (size_t i = 0; i < hw.size(); ++i) {
    auto piece = [] { float v = ((rand() % 2001) - 1000) * 1e-4f; unsigned u; memcpy(&u, &v, 4); return u >> 16; };
    const unsigned w = piece() | (piece() << 16);
    memcpy(&hw[i], &w, 4);
  }
  hipMalloc(&w, hw.size() * 4); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
  hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  const int nb = 3000;
  for (int waves : {4, 8, 12, 16}) {
    run<true, false>("pipelined, independent blocks", waves, w, out, cyc, nb);
    run<true, true>("pipelined, relu + chained blocks", waves, w, out, cyc, nb);
    run<false, true>("sequential form, chained blocks", waves, w, out, cyc, nb);
    run16<false>("16x16x32, independent blocks", waves, w, out, cyc, nb);
    run16<true>("16x16x32, relu + chained blocks", waves, w, out, cyc, nb);
  }
  return 0;
}
