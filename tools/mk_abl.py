"""dev helper: ablation builds of k_gather into tools/ablate/abl_<name>.so
   none     : no source loads, no MFMAs, no stores (skeleton only)
   noflag   : skeleton, and the per-tile need check reads nothing (need = valid)
   nonorm   : skeleton without the tap-count division
   nomfma   : loads + stores, no MFMAs          noload : MFMAs + stores, no loads
"""
import subprocess, sys
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _flat import flat_source
base = flat_source()
a = base.index('template <bool INTERIOR>\n__device__ __forceinline__ void gather_tile(')
b = base.index('// `sbase` = first row of this sample')
tile = base[a:b]

def variant(name):
    t = tile
    s = base
    if name in ('none', 'noflag', 'nonorm', 'noload', 'nostage', 'notiles'):
        t = t.replace('        const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, vo, soff, 0);\n        dst[u] = make_float2(__uint_as_float(v[0]), __uint_as_float(v[1]));',
                      '        dst[u] = make_float2(__uint_as_float(vo), 1.0f);')
        t = t.replace('        dst[u] = buf_load2(rsrc, o);', '        dst[u] = make_float2(__uint_as_float(o), 1.0f);')
    if name in ('none', 'noflag', 'nonorm', 'nomfma', 'nostage', 'notiles'):
        t = t.replace('      X.t[0] = mfma32(v[u].x, b, X.t[0]);\n      X.t[1] = mfma32(v[u].y, b, X.t[1]);',
                      '      X.t[0][u] += v[u].x * b;\n      X.t[1][u] += v[u].y * b;')
    s = s[:a] + t + s[b:]
    if name in ('none', 'noflag', 'nonorm', 'nostage', 'notiles'):
        s = s.replace('    if (need) frag_store_rows_gathered(X, a.nb, gc, h);\n  }\n}', '    if (need && X.t[0][0] == 123.456f) frag_store_rows_gathered(X, a.nb, gc, h);\n  }\n}')
    if name == 'nostage':
        s = s.replace('  stage_gather(gl.cm, gl.ko, gl.tt, gl.kvo, a.g, a.tm.TPS);\n  __syncthreads();\n  const int lane = threadIdx.x & 63, j = lane & 31, wave = threadIdx.x >> 6;\n  const EmbedLane el', '  const int lane = threadIdx.x & 63, j = lane & 31, wave = threadIdx.x >> 6;\n  const EmbedLane el', 1)
    if name == 'notiles':
        s = s.replace('  long tile = t0 + wave;\n  if (tile >= t1) return;\n  // tile, sample and t are wave-uniform by construction; make them provably so (scalar registers, scalar base address)\n  int sample = __builtin_amdgcn_readfirstlane((int)(tile / a.tm.TPS));\n  int t = __builtin_amdgcn_readfirstlane((int)(tile - (long)sample * a.tm.TPS));\n  for (; tile < t1; tile += WAVES_MLP, t += WAVES_MLP) {\n    while (t >= a.tm.TPS) { t -= a.tm.TPS; ++sample; }\n    const TileCtx tc = block_decode(a.tm, gl.tt, sample, t, j);\n    gather_process_tile<EMBED>', '  long tile = t0 + wave;\n  if (tile >= t1 || a.ntiles > 0) return;\n  int sample = __builtin_amdgcn_readfirstlane((int)(tile / a.tm.TPS));\n  int t = __builtin_amdgcn_readfirstlane((int)(tile - (long)sample * a.tm.TPS));\n  for (; tile < t1; tile += WAVES_MLP, t += WAVES_MLP) {\n    while (t >= a.tm.TPS) { t -= a.tm.TPS; ++sample; }\n    const TileCtx tc = block_decode(a.tm, gl.tt, sample, t, j);\n    gather_process_tile<EMBED>', 1)
    if name == 'noflag':
        s = s.replace('    else need = tc.valid && node_is_live(a.lb[gc], a.ub[gc]);', '    else need = tc.valid;')
    if name == 'nonorm':
        s = s.replace('    if (a.g.normalise) {\n      const int ny = tap_count(tc.y, wy0', '    if (a.g.normalise && a.g.K2 == 12345) {\n      const int ny = tap_count(tc.y, wy0', 1)
    open('/tmp/abl.hip', 'w').write(s)
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-o', f'/root/repo/tools/ablate/abl_{name}.so', '/tmp/abl.hip'])

for n in sys.argv[1:]:
    variant(n)
    print('built', n)
