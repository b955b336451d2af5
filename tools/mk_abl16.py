"""dev helper: ablation builds of the plain 16-node gather (gather_tile16) into tools/ablate/g16_<name>.so
   noload : MFMAs + stores, no source loads      nomfma : loads + stores, no MFMAs      nostore : loads + MFMAs, no stores
   none   : skeleton only"""
import subprocess, sys
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _flat import flat_source
base = flat_source()
a = base.index('template <bool INTERIOR>\n__device__ __forceinline__ void gather_tile16(')
b = base.index('#ifndef GATHER_CHS16')
tile = base[a:b]

def variant(name):
    t, s = tile, base
    if name in ('noload', 'none'):
        x = '        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, soff, 0);\n        dst[u] = f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};'
        assert t.count(x) == 1
        t = t.replace(x, '        dst[u] = f32x4{__uint_as_float(vo), 1.0f, 1.0f, 1.0f};')
        x = '        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0);\n        dst[u] = f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};'
        assert t.count(x) == 1
        t = t.replace(x, '        dst[u] = f32x4{__uint_as_float(o), 1.0f, 1.0f, 1.0f};')
    if name in ('nomfma', 'none'):
        x = '      for (int t = 0; t < 4; ++t) acc[t] = mfma16(v[u][t], b, acc[t]);'
        assert t.count(x) == 1
        t = t.replace(x, '      for (int t = 0; t < 4; ++t) acc[t][u] += v[u][t] * b;')
    s = s[:a] + t + s[b:]
    if name in ('nostore', 'none'):
        x = '  if (need) {                                  // lane (j, g\'): channels 16g\' + 4r + t of its node'
        assert s.count(x) == 1
        s = s.replace(x, '  if (need && (SPARSE || EMBED || acc[0][0] == 123.456f)) {')
    open('/tmp/abl16.hip', 'w').write(s)
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-o', f'/root/repo/tools/ablate/g16_{name}.so', '/tmp/abl16.hip'])

for n in sys.argv[1:]:
    variant(n)
    print('built', n)
