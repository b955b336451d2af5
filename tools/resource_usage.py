#!/usr/bin/env python
"""dev helper: table of hipcc's -Rpass-analysis=kernel-resource-usage remarks (VGPRs, spills, scratch, occupancy) per kernel.
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Rpass-analysis=kernel-resource-usage -o /tmp/x.so gnnb.hip 2> ru.txt
   python tools/resource_usage.py ru.txt [name filter]"""
import re
import subprocess
import sys

t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: [^\n]*Function Name: ", t)[1:]


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip() or n
    except OSError:
        return n


KEYS = [("vgpr", r"VGPRs"), ("agpr", r"AGPRs"), ("spillV", r"VGPRs Spill"), ("spillS", r"SGPRs Spill"),
        ("scratch", r"ScratchSize \[bytes/lane\]"), ("occ", r"Occupancy \[waves/SIMD\]"), ("sgpr", r"SGPRs"),
        ("lds", r"LDS Size \[bytes/block\]")]
print(f"{'kernel':70s} " + " ".join(f"{k:>7s}" for k, _ in KEYS))
for b in blocks:
    name = demangle(b.split("\n")[0].split(" [")[0].strip())
    name = re.sub(r"^void ", "", name).split("(")[0]
    if flt and flt not in name:
        continue
    vals = []
    for _, pat in KEYS:
        m = re.search(r"remark: [^\n]*\s" + pat + r": (\d+)", b)
        vals.append(int(m.group(1)) if m else -1)
    print(f"{name[:70]:70s} " + " ".join(f"{v:7d}" for v in vals))
