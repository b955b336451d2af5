#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (dev helper): per kernel, mean counter values over the last forward's dispatches."""
import csv, glob, sys, collections
tag = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'gpurun_out/pmc_{tag}/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        vals[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted({c for k in vals for c in vals[k]})
for k in sorted(vals):
    if not k.startswith(('k_', 'void k_')): continue
    print(k)
    for c in names:
        v = vals[k].get(c)
        if v: print(f"    {c:28s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
