#!/usr/bin/env python
"""Print the per-kernel ms/step table of a bench.py JSON line (dev helper)."""
import json
import sys

for path in sys.argv[1:]:
    line = [l for l in open(path) if l.startswith("{")][-1]
    d = json.loads(line)
    st = d["steps"]
    print(f"== {path}: {d['value']/1e6:.2f} M scores/s, {d['ms_per_step']:.3f} ms/step; roofline {d['roofline']['kernel']} "
          f"alg {d['roofline']['achieved']} TF issued {d['roofline'].get('issued_mfma_tflops')} TF")
    for k, v in sorted(d["kernels"].items(), key=lambda kv: -kv[1]["ms_total"]):
        print(f"   {k:26s} {v['ms_total']/st:7.3f} ms/step  {v['launches']//st:3d} launches  {v['avg_us']:8.1f} us avg")
