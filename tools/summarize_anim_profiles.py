#!/usr/bin/env python3
"""rocprofv3 output of tools/gpu_r02_session.sh (gpurun_out/r02, scratch) -> profiles/ (tracked):
  profiles/<tag>_anim_C4_rocprofv3_kernel_stats.csv   the --kernel-trace --stats summary of the bench command
  profiles/<tag>_anim_C4_pmc_summary.csv              per kernel: launches, avg FETCH_SIZE / WRITE_SIZE (KiB) per launch
  profiles/pmc_anim.json                              HBM bytes per launch per stage kernel (FETCH_SIZE x 2 on gfx950, as
                                                      /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes, + WRITE_SIZE),
                                                      keyed by the stage names bench.py reports (roofline.traffic)"""
import collections
import csv
import json
import re
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
SRC = ROOT / "gpurun_out" / tag
DST = ROOT / "profiles"


def find(sub, suffix):
    hits = sorted((SRC / sub).rglob(f"*{suffix}"))
    return hits[0] if hits else None


def counter_by_kernel(path, counter):
    acc = collections.defaultdict(list)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] == counter:
                m = re.search(r"(anim_\w+?_kernel|tetra_\w+?_kernel|anib_\w+?_kernel)", r["Kernel_Name"])
                acc[m.group(1) if m else r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    return acc


ks = find("kt", "kernel_stats.csv")
if ks:
    shutil.copyfile(ks, DST / f"{tag}_anim_C4_rocprofv3_kernel_stats.csv")
    print(open(ks).read()[:4000])
f, w = find("pmc_fetch", "counter_collection.csv"), find("pmc_write", "counter_collection.csv")
if f and w:
    fetch, write = counter_by_kernel(f, "FETCH_SIZE"), counter_by_kernel(w, "WRITE_SIZE")
    rows = []
    for k in sorted(set(fetch) | set(write)):
        fv, wv = fetch.get(k, []), write.get(k, [])
        rows.append((k, len(fv), sum(fv) / max(len(fv), 1), sum(fv), len(wv), sum(wv) / max(len(wv), 1), sum(wv)))
    with open(DST / f"{tag}_anim_C4_pmc_summary.csv", "w") as fh:
        fh.write("kernel,fetch_launches,FETCH_SIZE_KiB_avg,FETCH_SIZE_KiB_total,write_launches,WRITE_SIZE_KiB_avg,WRITE_SIZE_KiB_total\n")
        for r in rows:
            fh.write(",".join(str(x) for x in r) + "\n")
    stage_of = {"anim_seed_kernel": "anim_seed_kernel", "anim_cluster_wave_kernel": "anim_cluster_wave_kernel",
                "anim_postnuc_kernel": "anim_postnuc_kernel|anim_extend_kernels",
                "anim_postnuc_forced_kernel": "anim_postnuc_forced_kernel|anim_extdp_lane_kernel",
                "anim_postnuc_gap_kernel": "anim_postnuc_gap_kernels|anim_gap_kernels", "anim_finish_kernel": "anim_finish_kernel",
                "anim_postnuc_fwd_kernel": "anim_postnuc_fwd_kernel", "anim_postnuc_bwd_kernel": "anim_postnuc_rehearse_kernel+anim_postnuc_bwd_kernel"}
    out = {"round": tag, "command": "python bench.py --gpus 1 --steps 1 --warmup 1 --no-cpu-baseline --no-tetra (C4, a step = a tenth of the grid; two workers: launches of two streams overlap)",
           "correction": "FETCH_SIZE doubled (gfx950 rocprofv3 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md §HBM; exact for wide "
                         "coalesced streams, an upper bound for narrower accesses); WRITE_SIZE as reported (uncalibrated)"}
    for k, stage in stage_of.items():
        fv, wv = fetch.get(k, []), write.get(k, [])
        if fv:
            out[stage] = {"launches": len(fv), "FETCH_SIZE_KiB_avg": sum(fv) / len(fv), "WRITE_SIZE_KiB_avg": sum(wv) / max(len(wv), 1),
                          "hbm_bytes_per_launch": int(2 * 1024 * sum(fv) / len(fv) + 1024 * sum(wv) / max(len(wv), 1))}
    (DST / "pmc_anim.json").write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out, indent=1))
