"""dev helper: tools/ablate/halfgather.so = libgnnb with HALF of every gather's MFMAs removed (wrong results, timing only): how much
of the conv half-pass kernels' time follows the gathers' matrix-pipe cycles (the question behind pre-split rows)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _flat import flat_source
src = flat_source()
n = 0
for old, new in [
    ("      X.t[0] = mfma32(v[u].x, b, X.t[0]);\n      X.t[1] = mfma32(v[u].y, b, X.t[1]);", "      X.t[0] = mfma32(v[u].x + v[u].y, b, X.t[0]);"),
    ("      X.t[0] = mfma32(c.v[u].x, b, X.t[0]);\n      X.t[1] = mfma32(c.v[u].y, b, X.t[1]);", "      X.t[0] = mfma32(c.v[u].x + c.v[u].y, b, X.t[0]);"),
    ("      for (int t = 0; t < 4; ++t) acc[t] = mfma16(v[u][t], b, acc[t]);", "      for (int t = 0; t < 2; ++t) acc[t] = mfma16(v[u][t] + v[u][t + 2], b, acc[t]);"),
    ("      for (int t = 0; t < 4; ++t) acc[t] = mfma16(c.v[u][t], b, acc[t]);", "      for (int t = 0; t < 2; ++t) acc[t] = mfma16(c.v[u][t] + c.v[u][t + 2], b, acc[t]);"),
]:
    k = src.count(old)
    assert k >= 1, old
    src = src.replace(old, new)
    n += k
print("patched", n, "sites")
open('/tmp/gnnb_hg.hip', 'w').write(src)
os.makedirs('/root/repo/tools/ablate', exist_ok=True)
subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-o', '/root/repo/tools/ablate/halfgather.so', '/tmp/gnnb_hg.hip'])
