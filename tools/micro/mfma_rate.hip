// dev microbenchmark: issue rate of v_mfma_f32_32x32x2_f32 vs number of independent accumulators / waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, bool LDSA>
__global__ void k(float* out, int iters) {
  __shared__ float w[64 * 64];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) w[i] = i * 1e-6f;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float b = lane * 0.001f, av = lane * 0.002f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      float aa = LDSA ? w[((it + u) & 63) * 64 + lane] : av;
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa, b, acc[a], 0, 0, 0);
    }
  }
  float s = 0;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, bool LDSA>
void run(const char* name, int wg_threads) {
  float* out; hipMalloc(&out, 256 * 1024 * 4 * 4);
  const int iters = 2000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<NACC, LDSA>), dim3(256), dim3(wg_threads), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k<NACC, LDSA>), dim3(256), dim3(wg_threads), 0, 0, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double mfma_per_simd = (double)iters * 16 * NACC * (wg_threads / 64) / 4.0;
  double cyc = ms * 1e-3 * 2.4e9;
  printf("%-28s wg %4d: %.3f ms  -> %.1f cycles/MFMA/SIMD @2.4GHz, %.1f TF\n", name, wg_threads, ms, cyc / mfma_per_simd,
         256.0 * (wg_threads / 64) * iters * 16 * NACC * 4096 / (ms * 1e-3) / 1e12);
  hipFree(out);
}
int main() {
  for (int wg : {256, 512, 1024}) {
    run<1, false>("1 acc, A in VGPR", wg);
    run<2, false>("2 acc, A in VGPR", wg);
    run<4, false>("4 acc, A in VGPR", wg);
    run<2, true>("2 acc, A from LDS", wg);
    run<4, true>("4 acc, A from LDS", wg);
  }
  return 0;
}
