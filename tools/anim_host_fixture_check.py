#!/usr/bin/env python3
"""Development aid (CPU only): run the HOST build of the ANIm core (tools/anim_debug) on every fixture pair that has both
FASTA files and real MUMmer output, and count how many of MUMmer's .delta alignment records (coordinates + error count)
the core reproduces exactly.  Used to judge changes to the extension rules before spending GPU time on them.
Usage: python tools/anim_host_fixture_check.py [-j N] [--only substring]"""
import argparse
import gzip
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import anim_oracle  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("-j", type=int, default=8)
ap.add_argument("--only", default="")
ap.add_argument("--diff", action="store_true", help="print the records that differ")
args = ap.parse_args()
exe = ROOT / "tools/anim_debug/anim_debug"
subprocess.run(["g++", "-O2", "-std=c++17", f"-I{ROOT}/pyani_amd/csrc", str(exe) + ".cpp", "-o", str(exe)], check=True)
tmp = Path(tempfile.mkdtemp())
paths = {}
for grp in ("blochmannia", "caulobacter"):
    for gz in sorted((ROOT / "tests/golden/genomes" / grp).glob("*.fna.gz")):
        dst = tmp / gz.name[:-3]
        with gzip.open(gz, "rb") as fi, open(dst, "wb") as fo:
            shutil.copyfileobj(fi, fo)
        paths[dst.stem] = dst
jobs = []
for grp in ("blochmannia", "caulobacter"):
    for f in sorted((ROOT / "tests/golden/anim" / grp).glob("*.delta.gz")):
        a, b = f.name[:-len(".delta.gz")].split("_vs_")
        if a in paths and b in paths and args.only in f.name:
            jobs.append((f, a, b))


def run(job):
    f, a, b = job
    r = subprocess.run([str(exe), str(paths[a]), str(paths[b]), "--dump"], capture_output=True, text=True)
    got = set()
    for line in r.stdout.splitlines():
        if line.startswith("ALN "):
            t = line.split()
            got.add((t[1], t[2], int(t[3]), int(t[4]), int(t[5]), int(t[6]), int(t[7])))
    want = {(x.ref_id, x.qry_id, x.rs, x.re, x.qs, x.qe, x.errors) for x in anim_oracle.read_delta(f)[0]}
    coords = {w[:6] for w in want} & {g[:6] for g in got}
    if args.diff:
        for w in sorted(want - got): print("  MUMMER", w, flush=True)
        for g in sorted(got - want): print("  OURS  ", g, flush=True)
    return f.name, len(want), len(want & got), len(coords), len(got), r.stdout.splitlines()[0] if r.stdout else ""


tot_w = tot_e = tot_c = 0
with ThreadPoolExecutor(args.j) as ex:
    for name, nw, ne, nc, ng, first in ex.map(run, jobs):
        tot_w += nw; tot_e += ne; tot_c += nc
        print(f"{name[:70]:70s} mummer {nw:4d}  exact {ne:4d}  coords-only {nc:4d}  ours {ng:4d}", flush=True)
print(f"TOTAL records {tot_w}  exact {tot_e}  same coordinates {tot_c}")
