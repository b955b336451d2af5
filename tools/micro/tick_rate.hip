// dev micro (GPU box): what does __builtin_readcyclecounter() (s_memtime) count on gfx950?  A wave spins for N ticks; HIP events give the
// wall time.  Also s_memrealtime (100 MHz) and wall_clock64 for comparison.   hipcc --offload-arch=gfx950 -O3 -o tick_rate tick_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long n, unsigned long long* out) {
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  while (__builtin_readcyclecounter() - t0 < n) {}
  out[0] = __builtin_readcyclecounter() - t0;
  out[1] = wall_clock64() - r0;
}
int main() {
  unsigned long long* d; hipMalloc(&d, 16);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (unsigned long long n : {100000000ull, 400000000ull}) {
    hipEventRecord(a); spin<<<1, 64>>>(n, d); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    printf("%llu ticks in %.3f ms -> %.1f MHz; wall_clock64 %llu (rate attribute %d kHz)\n", h[0], ms, h[0] / ms / 1e3, h[1], rate);
  }
  return 0;
}
