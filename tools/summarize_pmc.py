#!/usr/bin/env python3
"""rocprofv3 --pmc counter_collection.csv -> one line per kernel: launches and, per counter, the SUM over the launches (and the
mean per launch).  Kernel names are shortened to their function name.   Usage: summarize_pmc.py <rocprof output dir> <out.csv>"""
import collections
import csv
import re
import sys
from pathlib import Path

src, dst = Path(sys.argv[1]), Path(sys.argv[2])
hits = sorted(src.rglob("*counter_collection.csv"))
if not hits:
    sys.exit(f"no counter_collection.csv under {src}")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(set)
counters = []
for path in hits:
    with open(path) as fh:
        for r in csv.DictReader(fh):
            m = re.search(r"((?:anim|tetra|anib)_\w+?_kernel)(<[^>]*>)?", r["Kernel_Name"])
            k = (m.group(1) + (m.group(2) or "")) if m else r["Kernel_Name"][:60]
            c = r["Counter_Name"]
            if c not in counters:
                counters.append(c)
            acc[k][c] += float(r["Counter_Value"])
            launches[k].add(r.get("Dispatch_Id") or r.get("Correlation_Id"))
with open(dst, "w", newline="") as fh:      # (kernel names hold commas: tetra_count_kernel<3, 0, 512>)
    wr = csv.writer(fh)
    wr.writerow(["kernel", "launches"] + [x for c in counters for x in (f"{c}_sum", f"{c}_per_launch")])
    for k in sorted(acc, key=lambda k: -max(acc[k].values())):
        n = max(1, len(launches[k]))
        wr.writerow([k, n] + [x for c in counters for x in (f"{acc[k][c]:.0f}", f"{acc[k][c] / n:.1f}")])
print(open(dst).read()[:6000])
