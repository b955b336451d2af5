// dev micro-test (GPU box): the residual of a bf16 split, x - bf16(x), computed by v_dot2c_f32_bf16 (one instruction, no unpack) against
// the shift / and / subtract form -- are the three pieces the same bits?   hipcc --offload-arch=gfx950 -O3 -o split_dot2 split_dot2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{a, b}, bf16x2v)); }
// (the constant goes through an SGPR the compiler cannot see into: as an immediate it is folded to the inline constant "-1.0", which the
// hardware reads as 0xbf800000 = {0, -1} -- the HIGH half)
__device__ __forceinline__ float sub_lo(float x, unsigned u) {
  unsigned c = 0x0000bf80u;
  asm volatile("" : "+s"(c));
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v, u), __builtin_bit_cast(bf16x2v, c), x, false);
}
__device__ __forceinline__ float sub_hi(float x, unsigned u) { return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2v, u), __builtin_bit_cast(bf16x2v, 0xbf800000u), x, false); }
__global__ void k(const float* in, unsigned* o_old, unsigned* o_new, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  float a = in[2 * i], b = in[2 * i + 1];
  {
    unsigned u1 = pk_bf16(a, b);
    float ra = a - __uint_as_float(u1 << 16), rb = b - __uint_as_float(u1 & 0xffff0000u);
    unsigned u2 = pk_bf16(ra, rb);
    float sa = ra - __uint_as_float(u2 << 16), sb = rb - __uint_as_float(u2 & 0xffff0000u);
    o_old[3 * i] = u1; o_old[3 * i + 1] = u2; o_old[3 * i + 2] = pk_bf16(sa, sb);
  }
  {
    unsigned u1 = pk_bf16(a, b);
    float ra = sub_lo(a, u1), rb = sub_hi(b, u1);
    unsigned u2 = pk_bf16(ra, rb);
    float sa = sub_lo(ra, u2), sb = sub_hi(rb, u2);
    o_new[3 * i] = u1; o_new[3 * i + 1] = u2; o_new[3 * i + 2] = pk_bf16(sa, sb);
  }
}
int main() {
  const int n = 1 << 22;
  float* h = (float*)malloc(n * 4);
  srand(1);
  for (int i = 0; i < n; ++i) {
    unsigned r = ((unsigned)rand() << 16) ^ (unsigned)rand();
    int mode = i & 7;
    float v;
    if (mode < 4) { v = (float)((double)r / 4294967296.0 * 2 - 1) * powf(2.0f, (float)((int)(r % 40) - 20)); }
    else if (mode < 6) { memcpy(&v, &r, 4); if (!std::isfinite(v)) v = 1.0f; }   // any bit pattern incl. denormals, huge
    else if (mode == 6) v = 0.0f;
    else { unsigned d = r & 0x807fffffu; memcpy(&v, &d, 4); }                     // denormals
    h[i] = v;
  }
  float* d; unsigned *o1, *o2;
  hipMalloc(&d, n * 4); hipMalloc(&o1, n * 6); hipMalloc(&o2, n * 6);
  hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
  k<<<n / 2 / 256, 256>>>(d, o1, o2, n);
  unsigned* a = (unsigned*)malloc(n * 6), *b = (unsigned*)malloc(n * 6);
  hipMemcpy(a, o1, n * 6, hipMemcpyDeviceToHost); hipMemcpy(b, o2, n * 6, hipMemcpyDeviceToHost);
  long diff = 0; int shown = 0;
  for (int i = 0; i < n / 2 * 3; ++i) if (a[i] != b[i]) { ++diff; if (shown++ < 10) printf("pair %d piece %d: old %08x new %08x  in %g %g\n", i / 3, i % 3, a[i], b[i], h[2 * (i / 3)], h[2 * (i / 3) + 1]); }
  printf("differences: %ld of %d words\n", diff, n / 2 * 3);
  return 0;
}
