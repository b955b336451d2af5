"""dev helper: build tools/ablate/qends.so = libgnnb where every workgroup of k_gather_update_q records when its gather waves and its chain
waves finished (wall clock, 10 ns ticks, relative to the earliest start in the launch) -> how uneven the static tile deal leaves the
workgroups.  Two stamps per wave: the timing is the shipped one.  gnnb_dev_qends(out[4][3][512], reset) reads them out."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _flat import flat_source
src = flat_source()

def rep(old, new, count=1):
    global src
    assert src.count(old) == count, (src.count(old), old)
    src = src.replace(old, new)

rep('struct FArgs {', '__device__ unsigned long long g_qe[4][3][512];      // [variant][start | gather end | chain end][workgroup]: max over its waves / min for start\nstruct FArgs {')
rep('''  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr bool SPARSE = SRC == 1, EMBED = SRC == 2;
  float* qbase = lds + PackUpdL3::FLOATS + (POST ? 6144 : 0);''', '''  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr bool SPARSE = SRC == 1, EMBED = SRC == 2;
  constexpr int QV = LANES == 32 ? 3 : SRC;
  const long long t_begin = wall_clock64();
  float* qbase = lds + PackUpdL3::FLOATS + (POST ? 6144 : 0);''')
# chain wave exits: three return sites after staging
rep('''      if (nvalid == 0) return;
      q_chain<POST>(a, lds, ring, nvalid, lane, [&]() {''', '''      if (nvalid == 0) { if (lane == 0 && blockIdx.x < 512) atomicMax(&g_qe[QV][2][blockIdx.x], (unsigned long long)(wall_clock64() - t_begin)); return; }
      q_chain<POST>(a, lds, ring, nvalid, lane, [&]() {''')
rep('''      if (nvalid < 32) return;                       // the last, partly filled tile''', '''      if (nvalid < 32) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (lane == 0 && blockIdx.x < 512) atomicMax(&g_qe[QV][2][blockIdx.x], (unsigned long long)(wall_clock64() - t_begin)); return; }''')
rep('''  __builtin_amdgcn_s_waitcnt(0xc07f);
  if (lane == 0) __hip_atomic_fetch_add(&q->done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}''', '''  __builtin_amdgcn_s_waitcnt(0xc07f);
  if (lane == 0) __hip_atomic_fetch_add(&q->done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (lane == 0 && blockIdx.x < 512) atomicMax(&g_qe[QV][1][blockIdx.x], (unsigned long long)(wall_clock64() - t_begin));
}''')
src += '''
extern "C" int gnnb_dev_qends(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_qe), sizeof(unsigned long long) * 4 * 3 * 512) != hipSuccess) return -1;
  if (reset) {
    static unsigned long long z[4][3][512];
    for (int v = 0; v < 4; ++v) for (int w = 0; w < 512; ++w) { z[v][0][w] = 0; z[v][1][w] = 0; z[v][2][w] = 0; }
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_qe), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
'''
os.makedirs('/root/repo/tools/ablate', exist_ok=True)
open('/tmp/gnnb_qends.hip', 'w').write(src)
subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-pthread', '-o', '/root/repo/tools/ablate/qends.so', '/tmp/gnnb_qends.hip'])
print("built tools/ablate/qends.so")
