#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of tools/gpu_profile.sh from gpurun_out/prof (scratch) into profiles/ (tracked) and
derive profiles/pmc_tetra_count.json (HBM bytes per launch of the count kernel, with the gfx950 FETCH_SIZE x2
correction of /opt/skills/guides/MI355X_MICROARCH.md §HBM), which bench.py reports as roofline.traffic."""
import collections
import csv
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "gpurun_out" / "prof"
DST = ROOT / "profiles"
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def counter_avgs(path, kernel):
    acc = collections.defaultdict(list)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if kernel in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


shutil.copyfile(SRC / "kt" / "kt_kernel_stats.csv", DST / f"{tag}_rocprofv3_kernel_stats.csv")
rows = []
for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
    for kern in ("tetra_count_kernel", "tetra_finalize_kernel", "tetra_pairs_kernel"):
        for name, (avg, n) in sorted(counter_avgs(SRC / sub / "pmc_counter_collection.csv", kern).items()):
            rows.append((sub, kern, name, avg, n))
with open(DST / f"{tag}_rocprofv3_pmc_summary.csv", "w") as fh:
    fh.write("pass,kernel,counter,avg_per_launch,launches\n")
    for r in rows:
        fh.write(",".join(str(x) for x in r) + "\n")
f = counter_avgs(SRC / "pmc_fetch" / "pmc_counter_collection.csv", "tetra_count_kernel")["FETCH_SIZE"]
w = counter_avgs(SRC / "pmc_write" / "pmc_counter_collection.csv", "tetra_count_kernel")["WRITE_SIZE"]
out = {
    "kernel": "tetra_count_kernel", "workload": "C2 (200 x 5 Mb synthetic genomes, seed 20250228)", "round": tag,
    "FETCH_SIZE_KiB_avg": f[0], "WRITE_SIZE_KiB_avg": w[0], "launches_sampled": f[1],
    "correction": "gfx950 rocprofv3 tallies 128-B read requests at 64 B: FETCH_SIZE reads exactly half the bytes of a wide "
                  "coalesced stream (MI355X_MICROARCH.md §HBM) -> doubled. WRITE_SIZE taken as reported (uncalibrated, ~1 MB).",
    "hbm_bytes_per_launch": int(2 * f[0] * 1024 + w[0] * 1024),
}
(DST / "pmc_tetra_count.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out, indent=1))
