"""dev helper: ablation builds of the gather side of the fused conv half-pass (wrong results, timing only) -> tools/ablate/
  nomfma : the gathers' fp32 MFMAs removed (operands still loaded and waited for)
  nogather: a tile's gather skipped altogether (decode, bounds, ratio, ring push and the chain stay)
  nogather_nochain: + the chain waves drop their rows: per-tile overhead + queue alone"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _flat import flat_source
base = flat_source()
os.makedirs('/root/repo/tools/ablate', exist_ok=True)

def build(name, src):
    open(f'/tmp/gnnb_{name}.hip', 'w').write(src)
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-pthread', '-o', f'/root/repo/tools/ablate/{name}.so', f'/tmp/gnnb_{name}.hip'])

def rep(src, old, new):
    assert src.count(old) == 1, (src.count(old), old)
    return src.replace(old, new)

M16 = "__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }"
s = rep(base, M16, '__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { asm volatile("" :: "v"(a), "v"(b)); return c; }')
s = s.replace("__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {",
              '__device__ __forceinline__ f32x16 mfma32_off(float a, float b, f32x16 c) { asm volatile("" :: "v"(a), "v"(b)); return c; }\n'
              "__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {", 1)
for ops in ["v[u].x, b, X.t[0]", "v[u].y, b, X.t[1]", "c.v[u].x, b, X.t[0]", "c.v[u].y, b, X.t[1]", "e0, b, X.t[0]", "e1, b, X.t[1]"]:
    assert s.count("mfma32(" + ops + ")") == 1, ops
    s = s.replace("mfma32(" + ops + ")", "mfma32_off(" + ops + ")")
build('nomfma', s)

G = """    if (LANES == 32) gather_compute_tile<false, SPARSE>(a.g, tc, sample, gl.cm, gl.ko, gl.kvo, tab, el, lane, X, ssum);
    else gather_compute_tile16<EMBED, SPARSE>(a.g, tc, sample, gl.cm, gl.ko, gl.kvo, tab, ew, eb, lane, acc, ssum);
"""
GN = """    if (LANES == 32) { for (int R = 0; R < 32; ++R) FRAG_AT(X, R) = lb; }
    else { for (int t = 0; t < 4; ++t) acc[t] = f32x4{lb, ub, lb, ub}; }
"""
ng = rep(base, G, GN)
build('nogather', ng)
CH_OLD = "  release();\n  upd_chain_frag<POST>(a.u, lds, X, gc, r0, r1, amb, sw, valid, lane, keep);"
CH_NEW = "  release();\n  if (nvalid >= 0 && !keep) return;\n  upd_chain_frag<POST>(a.u, lds, X, gc, r0, r1, amb, sw, valid, lane, keep);"
build('nogather_nochain', rep(ng, CH_OLD, CH_NEW))
