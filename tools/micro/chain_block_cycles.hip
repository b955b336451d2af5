// dev micro-benchmark (GPU box): cycles of ONE 64x64 bf16x3 block (gemm_w64_bf3, the product's code: 48 v_mfma_f32_32x32x16_bf16 + the
// operand splits, weights out of LDS) per wave, as a function of the waves per SIMD -- against its 48 x 32 = 1536 cycles of MFMA issue.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I gnn_branching_amd/csrc -o tools/micro/chain_block_cycles tools/micro/chain_block_cycles.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
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

template <bool PIPE, bool RELU>
static void run(const char* name, int waves, const float* w, float* out, unsigned long long* cyc, int nblocks) {
  hipFuncSetAttribute((const void*)k_blocks<PIPE, RELU>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 6144 * 4);
  const int nwg = 256;
  for (int rep = 0; rep < 2; ++rep) k_blocks<PIPE, RELU><<<nwg, waves * 64, 3 * 6144 * 4>>>(w, out, cyc, nblocks);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(nwg * waves);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (auto v : h) sum += (double)v;
  printf("%-34s %2d waves / CU (%d per SIMD): %8.0f cycles per block and wave (MFMA issue alone: 1536)\n", name, waves, waves / 4, sum / h.size() / nblocks);
}

int main() {
  float *w, *out;
  unsigned long long* cyc;
  std::vector<float> hw(3 * 6144);
  for (size_t i = 0; i < hw.size(); ++i) hw[i] = 0.0f;      // (bf16 pieces of zero weights: the timing does not depend on the values)
  hipMalloc(&w, hw.size() * 4); hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 16 * 8);
  hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  const int nb = 300;
  for (int waves : {4, 8, 12, 16}) {
    run<true, false>("pipelined, independent blocks", waves, w, out, cyc, nb);
    run<true, true>("pipelined, relu + chained blocks", waves, w, out, cyc, nb);
    run<false, true>("sequential form, chained blocks", waves, w, out, cyc, nb);
  }
  return 0;
}
