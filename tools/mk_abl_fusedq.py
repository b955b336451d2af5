"""dev helper: ablation builds of the fused conv half-pass (wrong results, timing only) -> tools/ablate/{nochain,noload}.so
  nochain: chain waves take a tile's rows out of the ring and drop them (no GEMMs, no row stores): what the gather side alone costs
  noload : every row load of the gathers goes out of range (returns 0 without touching memory): everything but the memory system"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _flat import flat_source
base = flat_source()
os.makedirs('/root/repo/tools/ablate', exist_ok=True)

def build(name, src):
    open(f'/tmp/gnnb_{name}.hip', 'w').write(src)
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-o', f'/root/repo/tools/ablate/{name}.so', f'/tmp/gnnb_{name}.hip'])

CH_OLD = "  release();\n  upd_chain_frag<POST>(a.u, lds, X, gc, r0, r1, amb, sw, valid, lane, keep);"
CH_NEW = "  release();\n  if (nvalid >= 0 && !keep) return;\n  upd_chain_frag<POST>(a.u, lds, X, gc, r0, r1, amb, sw, valid, lane, keep);"
assert base.count(CH_OLD) == 1
build('nochain', base.replace(CH_OLD, CH_NEW))

src = base
n = 0
for old, new in [
    ("const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, soff, 0);", "const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo | BUF_OOB, soff, 0);"),
    ("const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0);", "const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o | BUF_OOB, 0, 0);"),
    ("const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, vo, soff, 0);", "const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, vo | BUF_OOB, soff, 0);"),
    ("c.v[u] = buf_load2(rsrc, e.x == BUF_OOB ? BUF_OOB : e.x + lane_off);", "c.v[u] = buf_load2(rsrc, BUF_OOB | e.x);"),
    ("dst[u] = buf_load2(rsrc, o);", "dst[u] = buf_load2(rsrc, o | BUF_OOB);"),
]:
    k = src.count(old)
    assert k >= 1, old
    src = src.replace(old, new)
    n += k
print("noload: patched", n, "sites")
build('noload', src)

# ---- decomposition of the gather waves' time: no ring traffic at all (chain waves idle), then also no loads, then also half the MFMAs
old_push = "    const unsigned long long bal = __ballot(need) & (LANES == 16 ? 0xffffull : 0xffffffffull);\n    const int n = __popcll(bal);"
assert base.count(old_push) == 1
nopush = base.replace(CH_OLD, CH_NEW)
nopush = nopush.replace(old_push, "    if (LANES == 16 ? acc[0][0] == 12345.0f : X.t[0][0] == 12345.0f) a.u.status[0] = 1;\n    if (ssum < 1e30f) continue;\n" + old_push)
build('nopush', nopush)
LOADS = [
    ("const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo, soff, 0);", "const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, vo | BUF_OOB, soff, 0);"),
    ("const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o, 0, 0);", "const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o | BUF_OOB, 0, 0);"),
    ("const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, vo, soff, 0);", "const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, vo | BUF_OOB, soff, 0);"),
    ("c.v[u] = buf_load2(rsrc, e.x == BUF_OOB ? BUF_OOB : e.x + lane_off);", "c.v[u] = buf_load2(rsrc, BUF_OOB | e.x);"),
    ("dst[u] = buf_load2(rsrc, o);", "dst[u] = buf_load2(rsrc, o | BUF_OOB);"),
]
src2 = nopush
for o, nw in LOADS:
    assert src2.count(o) >= 1
    src2 = src2.replace(o, nw)
build('nopush_noload', src2)
src3 = src2
for o, nw in [
    ("      X.t[0] = mfma32(v[u].x, b, X.t[0]);\n      X.t[1] = mfma32(v[u].y, b, X.t[1]);", "      X.t[0] = mfma32(v[u].x + v[u].y, b, X.t[0]);"),
    ("      X.t[0] = mfma32(c.v[u].x, b, X.t[0]);\n      X.t[1] = mfma32(c.v[u].y, b, X.t[1]);", "      X.t[0] = mfma32(c.v[u].x + c.v[u].y, b, X.t[0]);"),
    ("      for (int t = 0; t < 4; ++t) acc[t] = mfma16(v[u][t], b, acc[t]);", "      for (int t = 0; t < 2; ++t) acc[t] = mfma16(v[u][t] + v[u][t + 2], b, acc[t]);"),
    ("      for (int t = 0; t < 4; ++t) acc[t] = mfma16(c.v[u][t], b, acc[t]);", "      for (int t = 0; t < 2; ++t) acc[t] = mfma16(c.v[u][t] + c.v[u][t + 2], b, acc[t]);"),
]:
    assert src3.count(o) >= 1
    src3 = src3.replace(o, nw)
build('nopush_noload_half', src3)
