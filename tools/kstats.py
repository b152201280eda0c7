#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 kernel_stats.csv (name shortened, calls, total ms)."""
import csv, glob, re, sys
pat = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/anim_kt_50/*/*kernel_stats.csv"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
import os
f = max(glob.glob(pat), key=os.path.getmtime)
tot = 0.0
rows = list(csv.DictReader(open(f)))
for r in rows:
    tot += int(r["TotalDurationNs"]) / 1e6
for r in rows[:n]:
    m = re.search(r"(\w+(?:<[^>]*>)?)\(", r["Name"].replace("(anonymous namespace)::", ""))
    print(f"{(m.group(1) if m else r['Name'][:40]):34s} {r['Calls']:>5s} {int(r['TotalDurationNs'])/1e6:9.3f} ms")
print(f"{'total':34s}       {tot:9.3f} ms")
