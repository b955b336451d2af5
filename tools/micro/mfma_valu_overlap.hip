// dev micro-benchmark (GPU box): do a wave's MFMAs and a co-resident wave's vector instructions on the SAME SIMD overlap?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
// One workgroup of 512 threads per CU: waves 0-3 (one per SIMD) run an MFMA loop, waves 4-7 (their SIMD partners) a v_fma_f32 loop.
// Reported: cycles of each loop alone and of both together, for v_mfma_f32_32x32x2_f32, v_mfma_f32_16x16x4_f32 and
// v_mfma_f32_32x32x16_bf16.  together ~ max(alone) -> the pipes overlap; together ~ sum -> the MFMA holds the SIMD's issue.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(512) void k(int n_mfma, int n_valu, int mode, unsigned long long* out, float* sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool do_mfma = wave < 4 && (mode & 1), do_valu = wave >= 4 && (mode & 2);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  float r = 0.f;
  if (do_mfma) {
    if (KIND == 0) {
      f32x16 a0 = {0}, a1 = {0};
      const float x = lane * 0.001f, y = 1.0f;
      for (int i = 0; i < n_mfma; i += 2) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
      }
      r = a0[0] + a1[3];
    } else if (KIND == 1) {
      f32x4 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
      const float x = lane * 0.001f, y = 1.0f;
      for (int i = 0; i < n_mfma; i += 4) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
      }
      r = a0[0] + a1[1] + a2[2] + a3[3];
    } else {
      f32x16 a0 = {0}, a1 = {0};
      bf16x8 x, y;
      for (int q = 0; q < 8; ++q) { x[q] = (__bf16)(lane * 0.01f + q); y[q] = (__bf16)1.0f; }
      for (int i = 0; i < n_mfma; i += 2) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y, x, a1, 0, 0, 0);
      }
      r = a0[0] + a1[3];
    }
  }
  if (do_valu) {
    float v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3, v4 = lane + 4, v5 = lane + 5, v6 = lane + 6, v7 = lane + 7;
    const float c = 1.0001f, d = 0.5f;
    for (int i = 0; i < n_valu; i += 8) {
      v0 = __builtin_fmaf(v0, c, d); v1 = __builtin_fmaf(v1, c, d); v2 = __builtin_fmaf(v2, c, d); v3 = __builtin_fmaf(v3, c, d);
      v4 = __builtin_fmaf(v4, c, d); v5 = __builtin_fmaf(v5, c, d); v6 = __builtin_fmaf(v6, c, d); v7 = __builtin_fmaf(v7, c, d);
      asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
    }
    r = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (r == 12345.678f) sink[0] = r;
  if (lane == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
}

template <int KIND>
void run(const char* name, int n_mfma, int n_valu) {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 64); hipMalloc(&sink, 4);
  unsigned long long h[8];
  printf("%s: %d MFMAs per wave (waves 0-3), %d v_fma_f32 per wave (waves 4-7)\n", name, n_mfma, n_valu);
  for (int mode = 1; mode <= 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, n_mfma, n_valu, mode, out, sink);
    hipDeviceSynchronize();
    hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    printf("  %-22s mfma wave %8llu cycles (%.1f / MFMA)   valu wave %8llu cycles (%.2f / instr)\n",
           mode == 1 ? "MFMA waves alone" : mode == 2 ? "vector waves alone" : "both together", h[0], (double)h[0] / n_mfma, h[4], (double)h[4] / n_valu);
  }
  hipFree(out); hipFree(sink);
}

int main() {
  run<0>("v_mfma_f32_32x32x2_f32", 2048, 32768);
  run<1>("v_mfma_f32_16x16x4_f32", 4096, 32768);
  run<2>("v_mfma_f32_32x32x16_bf16", 4096, 32768);
  return 0;
}
