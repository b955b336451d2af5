"""dev helper: print every dispatch of the LAST forward found in a rocprofv3 --kernel-trace CSV (start offset, duration, gap)."""
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
# the last forward starts at the last k_setup
idx = max(i for i, r in enumerate(rows) if "k_setup" in r[2] or "k_classify" in r[2])
step = rows[idx:]
t0 = step[0][0]
prev_end = t0
total = 0
for s, e, name in step:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  {name[:70]}")
    prev_end = e
    total += e - s
print(f"span {(step[-1][1] - t0) / 1e3:.1f} us, sum of kernels {total / 1e3:.1f} us, {len(step)} dispatches")
