"""dev helper: k_gather variant whose prefetch loop interleaves one load group with one MFMA pair (tools/ablate/interleave.so)."""
import subprocess
s = open('/root/repo/gnn_branching_amd/csrc/gnnb.hip').read()
s = s.replace('"../../include/gnnb.h"', '"/root/repo/include/gnnb.h"').replace('"gnnb_pack.h"', '"/root/repo/gnn_branching_amd/csrc/gnnb_pack.h"')
a = s.index('template <bool INTERIOR>\n__device__ __forceinline__ void gather_tile(')
b = s.index('// `sbase` = first row of this sample')
t = s[a:b]
old_loop = t[t.index('  load(cur, 0);\n  const int npairs = K2 / (2 * GATHER_CH);'):]
new_loop = '''  auto load1 = [&](float2 (&dst)[GATHER_CH], int s0, int u) {
    if (INTERIOR) {
      const unsigned vo = kvo[2 * (s0 + u) + h] + lane_off;
      const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, vo, soff, 0);
      dst[u] = make_float2(__uint_as_float(v[0]), __uint_as_float(v[1]));
    } else {
      const unsigned long long ev = reinterpret_cast<const unsigned long long*>(ko)[2 * (s0 + u) + h];
      const int ex = (int)(unsigned)ev, ey = (int)(unsigned)(ev >> 32);
      const int wy = wy0 + (ey & 0xffff), wx = wx0 + (ey >> 16);
      unsigned o = (unsigned)(origin + ex) * 256u + lane_off;
      o = ((unsigned)wy < (unsigned)Hs && (unsigned)wx < (unsigned)Ws) ? o : BUF_OOB;
      dst[u] = buf_load2(rsrc, o);
    }
  };
  auto mma1 = [&](const float2 (&v)[GATHER_CH], int s0, int u) {
    const float b = cm[(s0 + u) * 64 + lane];
    X.t[0] = mfma32(v[u].x, b, X.t[0]);
    X.t[1] = mfma32(v[u].y, b, X.t[1]);
  };
  load(cur, 0);
  const int npairs = K2 / (2 * GATHER_CH);
  int s0 = 0;
  for (int pr = 0; pr < npairs; ++pr, s0 += 2 * GATHER_CH) {
#pragma unroll
    for (int u = 0; u < GATHER_CH; ++u) {
      load1(nxt, s0 + GATHER_CH, u);
      mma1(cur, s0, u);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = 0; u < GATHER_CH; ++u) {
      load1(cur, s0 + 2 * GATHER_CH, u);
      mma1(nxt, s0 + GATHER_CH, u);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (K2 & GATHER_CH) mma(cur, s0);
}

'''
t = t.replace(old_loop, new_loop)
open('/tmp/il.hip', 'w').write(s[:a] + t + s[b:])
subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-o', '/root/repo/tools/ablate/interleave.so', '/tmp/il.hip'])
print('built')
