"""dev helper: ablation builds of k_node_update (bf16 x 3 form) into tools/ablate/nu_<name>.so (what bounds it?)
   nomfma  : no bf16 MFMAs (one VALU op instead)         nosplit : operands not split (the first piece used three times)
   noload  : no aggregate-row loads                      nostore : no row stores
   run:  KPAT=k_node_update tools/kstats.sh "X=1" "GNNB_LIB=$PWD/tools/ablate/nu_<name>.so" """
import subprocess, sys
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _flat import flat_source
base = flat_source()

def one(s, x, y):
    assert s.count(x) == 1, x[:60]
    return s.replace(x, y)

def variant(name):
    s = base
    if name == 'nomfma':
        s = one(s, '__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }',
                '__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) { c[0] += (float)a[0] * (float)b[0]; return c; }')
    if name == 'nosplit':
        s = one(s, '      p1[q] = u1; p2[q] = u2; p3[q] = pk_bf16(sa, sb);', '      p1[q] = u1; p2[q] = u1; p3[q] = u1;')
    if name == 'noload':
        s = one(s, '    frag_load_rows(x_, a.nb, g_, h);', '#pragma unroll\n    for (int R = 0; R < 32; ++R) FRAG_AT(x_, R) = l_ + R;')
    if name == 'nostore':
        s = one(s, '      if (a.mu) frag_store_rows(H2, a.mu, gc, h);', '      if (a.mu && FRAG_AT(H2, 0) == 123.456f) frag_store_rows(H2, a.mu, gc, h);')
    open('/tmp/abl_nu.hip', 'w').write(s)
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-o', f'/root/repo/tools/ablate/nu_{name}.so', '/tmp/abl_nu.hip'])

for n in sys.argv[1:]:
    variant(n)
    print('built', n)
