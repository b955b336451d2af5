"""dev helper: ablation builds of k_node_update into tools/ablate/nu_<name>.so (what bounds it?)
   nomfma : loads + stores, MFMAs replaced by one VALU op each      noload : no aggregate-row loads      nostore : no row stores
"""
import subprocess, sys
base = open('/root/repo/gnn_branching_amd/csrc/gnnb.hip').read()
base = base.replace('"../../include/gnnb.h"', '"/root/repo/include/gnnb.h"').replace('"gnnb_pack.h"', '"/root/repo/gnn_branching_amd/csrc/gnnb_pack.h"')
a = base.index('template <bool DEFERRED>\n__device__ __forceinline__ void node_update_loop(')
b = base.index('template <int WAVES, bool DEFERRED>\n__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k_node_update')
body = base[a:b]

def variant(name):
    t = body
    if name == 'nomfma':
        # a private gemm that burns no MFMA: shadow gemm_w64 inside the loop function via a macro
        t = '#define gemm_w64 gemm_w64_fake\ntemplate <int KSTEPS, class GetB>\n__device__ __forceinline__ void gemm_w64_fake(const float* wl, int lane, Frag& acc, GetB getB) {\n  const f32x4* w4 = reinterpret_cast<const f32x4*>(wl) + lane;\n#pragma unroll\n  for (int s4 = 0; s4 < KSTEPS / 4; ++s4) { const f32x4 a0 = w4[(s4 * 2) * 64]; const f32x4 a1 = w4[(s4 * 2 + 1) * 64];\n#pragma unroll\n    for (int c = 0; c < 4; ++c) { const float b = getB(s4 * 4 + c); acc.t[0][c] += a0[c] * b; acc.t[1][c] += a1[c] * b; } }\n}\n' + t + '#undef gemm_w64\n'
    if name == 'noload':
        t = t.replace('    frag_load_rows(x_, a.nb, g_, h);', '#pragma unroll\n    for (int R = 0; R < 32; ++R) FRAG_AT(x_, R) = l_ + R;')
    if name == 'nostore':
        t = t.replace('      frag_store_rows(H2, a.mu, gc, h);', '      if (FRAG_AT(H2, 0) == 123.456f) frag_store_rows(H2, a.mu, gc, h);')
    s = base[:a] + t + base[b:]
    open('/tmp/abl_nu.hip', 'w').write(s)
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-o', f'/root/repo/tools/ablate/nu_{name}.so', '/tmp/abl_nu.hip'])

for n in sys.argv[1:]:
    variant(n)
    print('built', n)
