#!/usr/bin/env python3
"""dev helper: where the SGPR spills of a kernel execute.  hipcc spills scalar registers into lanes of a reserved vector register
(v_writelane_b32 / v_readlane_b32: vector-issue slots).  This script reads the kernel's ISA (hipcc --save-temps: *.gfx950.s) and counts
those instructions per loop depth, using the compiler's own basic-block annotations ("in Loop: Header=BB.. Depth=N").
   python3 tools/sgpr_spill_sites.py <file.s> <mangled kernel name substring> [...]"""
import re
import sys

t = open(sys.argv[1]).read()
for pat in sys.argv[2:]:
    for m in re.finditer(r"\n(_Z\w*" + re.escape(pat) + r"\w*):[^\n]*\n", t):
        name = m.group(1)
        body = t[m.end():t.index(".Lfunc_end", m.end())].split("\n")
        depth, by = 0, {}
        mf = {}
        for l in body:
            b = re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)\s*(;.*)?", l)      # a labelled block, or a fall-through block (comment line)
            if b:
                d = re.search(r"Depth=(\d+)", l)
                depth = int(d.group(1)) if d else 0
            for key in ("v_writelane_b32", "v_readlane_b32", "scratch_store", "scratch_load"):
                if key in l:
                    by.setdefault(key, {}).setdefault(depth, 0)
                    by[key][depth] += 1
            if "v_mfma" in l:
                mf[depth] = mf.get(depth, 0) + 1
        print(name)
        for key, d in by.items():
            print(f"   {key:18s} " + "  ".join(f"depth {k}: {v}" for k, v in sorted(d.items())))
        print("   (v_mfma by depth:    " + "  ".join(f"depth {k}: {v}" for k, v in sorted(mf.items())) + ")")
