#!/usr/bin/env python3
"""Print the duration of every launch of the kernels whose name contains one of the given substrings (rocprofv3 kernel_trace.csv)."""
import csv, glob, os, sys
pat = sys.argv[1]
subs = sys.argv[2:] or ["extdp_lane", "anim_extend"]
f = max(glob.glob(pat), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    if any(x in r["Kernel_Name"] for x in subs):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        print(f"  {name:28s} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:9.3f} ms")
