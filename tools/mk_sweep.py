"""dev helper: small constant sweeps of the fused conv half-pass as tools/ablate/sw_*.so (timing AND results valid: scheduling constants only)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _flat import flat_source
base = flat_source()
os.makedirs('/root/repo/tools/ablate', exist_ok=True)

def build(name, src):
    open(f'/tmp/gnnb_{name}.hip', 'w').write(src)
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-pthread', '-o', f'/root/repo/tools/ablate/{name}.so', f'/tmp/gnnb_{name}.hip'])

def rep(src, old, new, count=1):
    assert src.count(old) == count, (src.count(old), old)
    return src.replace(old, new)

CH = "        __builtin_amdgcn_s_sleep(8);\n      }\n      if (nvalid < 0)"
GA = "        __builtin_amdgcn_s_sleep(4);\n      }\n      if (!ok) { if (lane == 0) atomicOr(a.u.status, 2); stuck = true; break; }"
build('sw_chain2', rep(base, CH, CH.replace("s_sleep(8)", "s_sleep(2)")))
build('sw_gath1', rep(base, GA, GA.replace("s_sleep(4)", "s_sleep(1)")))
s = rep(base, CH, CH.replace("s_sleep(8)", "s_sleep(2)"))
build('sw_both', rep(s, GA, GA.replace("s_sleep(4)", "s_sleep(1)")))
