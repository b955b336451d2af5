"""dev helper: build tools/ablate/toptime.so = libgnnb with wall-clock stamps at the phase boundaries of k_top (printf)."""
import subprocess
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _flat import flat_source
src = flat_source()
a = src.index('template <int TS>\n__device__ __forceinline__ void top_sample(')
b = src.index('// TS = 4: one workgroup per sample; TS = 2 / 1')
body = src[a:b]
marks = ["  // ---- F1: this workgroup's rows of C", '  // per-lane node of the update phases', '  // ---- F2: forward node update', '  // ---- F3: property node',
         '  // ---- B1: backward node update', '  // ---- B2: aggregate rows']
body = body.replace('  const int N = a.N;', '  const int N = a.N;\n  long long tt[8]; int ti = 0;\n  tt[ti++] = wall_clock64();')
for m in marks:
    assert m in body, m
    body = body.replace(m, '  __syncthreads(); tt[ti++] = wall_clock64();\n' + m)
i = body.rindex('}')
body = body[:i] + ('  __syncthreads(); tt[ti++] = wall_clock64();\n  if (blockIdx.x == 0 && threadIdx.x == 0) printf("k_top phases (10ns ticks): stage %lld F1 %lld setup %lld '
                   'F2 %lld F3 %lld B1 %lld B2 %lld\\n", tt[1]-tt[0], tt[2]-tt[1], tt[3]-tt[2], tt[4]-tt[3], tt[5]-tt[4], tt[6]-tt[5], tt[7]-tt[6]);\n}\n\n')
open('/tmp/gnnb_t.hip', 'w').write(src[:a] + body + src[b:])
subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-o', '/root/repo/tools/ablate/toptime.so', '/tmp/gnnb_t.hip'])
