#!/usr/bin/env python3
"""Development aid (CPU only): run the HOST build of the ANIm core (tools/anim_debug) on every fixture pair that has both
FASTA files and real MUMmer output, and count how many of MUMmer's .delta alignment records (coordinates + error count)
the core reproduces exactly.  Used to judge changes to the extension rules before spending GPU time on them.
Usage: python tools/anim_host_fixture_check.py [-j N] [--only substring]"""
import argparse
import gzip
import hashlib
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
ap.add_argument("--json", default="", help="write the per-pair report (records + parse_delta tuples) here")
ap.add_argument("--oracle", action="store_true", help="run oracle/nucmer_oracle.cpp (the restatement of MUMmer's own algorithm) instead of the engine's host build")
ap.add_argument("--groups", default="blochmannia,caulobacter,group2,jspecies")
ap.add_argument("--delta", action="store_true", help="also compare every alignment's indel offset list with the .delta file's (host build: needs ANIM_EXACT=1)")
args = ap.parse_args()
import os
import shlex
exe = ROOT / "tools/anim_debug/anim_debug"
src = str(exe) + ".cpp"
if os.environ.get("ANIM_CXXFLAGS"):      # parameter sweeps: -DPGA_...=value builds get their own binary
    exe = Path(str(exe) + "_" + hashlib.sha1(os.environ["ANIM_CXXFLAGS"].encode()).hexdigest()[:8])
if args.oracle:
    exe = ROOT / "oracle/_build/nucmer_oracle"
    exe.parent.mkdir(exist_ok=True)
    src = str(ROOT / "oracle/nucmer_oracle.cpp")
subprocess.run(["g++", "-O2", "-std=c++17", f"-I{ROOT}/pyani_amd/csrc", *shlex.split(os.environ.get("ANIM_CXXFLAGS", "")), src, "-o", str(exe)], check=True)
tmp = Path(tempfile.mkdtemp())
paths = {}
for grp in ("blochmannia", "caulobacter", "group2"):
    for gz in sorted((ROOT / "tests/golden/genomes" / grp).glob("*.fna.gz")):
        dst = tmp / gz.name[:-3]
        with gzip.open(gz, "rb") as fi, open(dst, "wb") as fo:
            shutil.copyfileobj(fi, fo)
        paths[dst.stem] = dst
# JSpecies ran nucmer on the single-record NC_002696 (= the two records of the fixture file joined)
joined = tmp / "js" / "NC_002696.fna"
joined.parent.mkdir()
body = "".join(l.strip() for l in open(paths["NC_002696"]) if not l.startswith(">"))
joined.write_text(">gi|16124256|ref|NC_002696.2| joined\n" + "\n".join(body[i:i + 70] for i in range(0, len(body), 70)) + "\n")
jobs = []
for grp in args.groups.split(","):
    for f in sorted((ROOT / "tests/golden/anim" / grp).glob("*.delta.gz")):
        a, b = f.name[:-len(".delta.gz")].split("_vs_")
        if a in paths and b in paths and args.only in f"{grp}/{f.name}":
            jobs.append((f, a, b, grp))


def run(job):
    f, a, b, grp = job
    pa, pb = (joined if grp == "jspecies" and s == "NC_002696" else paths[s] for s in (a, b))
    r = subprocess.run([str(exe), str(pa), str(pb), "--dump"] + (["--delta"] if args.delta else []), capture_output=True, text=True)
    if r.returncode != 0:
        print("FAILED", f.name, r.stderr[-300:], flush=True)
    got, kept, lists, cur, visits, seq = set(), [], {}, None, {}, []
    for line in r.stdout.splitlines():
        if args.delta and cur is not None and not line.startswith("ALN "):
            t = line.split()
            if len(t) == 1 and t[0] != "0":
                lists[cur].append(int(t[0]))
            elif t and t[0] == "VISIT":
                visits[cur] = (int(t[1]), int(t[2]))
            continue
        if line.startswith("ALN "):
            t = line.split()
            got.add((t[1], t[2], int(t[3]), int(t[4]), int(t[5]), int(t[6]), int(t[7])))
            cur = (t[1], t[2], int(t[3]), int(t[4]), int(t[5]), int(t[6]), int(t[7]))
            lists[cur] = []
            seq.append(cur)
            if len(t) > 8 and t[8] == "keep=3":
                kept.append(anim_oracle.Aln(t[1], t[2], int(t[3]), int(t[4]), int(t[5]), int(t[6]), int(t[7]), int(t[7]), 0, ()))
    want = {(x.ref_id, x.qry_id, x.rs, x.re, x.qs, x.qe, x.errors) for x in anim_oracle.read_delta(f)[0]}
    coords = {w[:6] for w in want} & {g[:6] for g in got}
    if args.diff:
        for w in sorted(want - got): print("  MUMMER", w, flush=True)
        for g in sorted(got - want): print("  OURS  ", g, flush=True)
    rep = {"mummer_records": len(want), "exact": len(want & got), "same_coordinates": len(coords), "ours": len(got)}
    if args.delta:
        wl = {(x.ref_id, x.qry_id, x.rs, x.re, x.qs, x.qe, x.errors): list(x.indels) for x in anim_oracle.read_delta(f)[0]}
        rep["indel_lists"] = len(wl)
        rep["indel_lists_equal"] = sum(1 for k, v in wl.items() if lists.get(k) == v)
        DELTA_TOTALS[0] += rep["indel_lists"]; DELTA_TOTALS[1] += rep["indel_lists_equal"]
        # MUMmer's print order inside each (reference record, query record) block: by the visit the alignment was created in, forward strand first
        file_seq = [(x.ref_id, x.qry_id, x.rs, x.re, x.qs, x.qe, x.errors) for x in anim_oracle.read_delta(f)[0]]
        blocks_file, blocks_ours = {}, {}
        for k in file_seq:
            blocks_file.setdefault(k[:2], []).append(k)
        for i, k in enumerate(seq):
            blocks_ours.setdefault(k[:2], []).append((visits.get(k, (0, 0)), i, k))
        same = sum(1 for b, v in blocks_file.items() if [k for _, _, k in sorted(blocks_ours.get(b, []))] == v)
        rep["blocks"] = len(blocks_file); rep["blocks_in_mummer_order"] = same
        rep["block_sequence_equal"] = list(dict.fromkeys(k[:2] for k in file_seq)) == list(dict.fromkeys(k[:2] for k in seq))
        DELTA_TOTALS[2] += len(blocks_file); DELTA_TOTALS[3] += same; DELTA_TOTALS[4] += int(rep["block_sequence_equal"]); DELTA_TOTALS[5] += 1
    allrecs = [anim_oracle.Aln(*g, g[6], 0, ()) for g in got]
    rep["delta_tuple_mummer"] = list(anim_oracle.parse_delta(f))
    rep["delta_tuple_ours"] = list(anim_oracle.parse_delta_records(allrecs)) if allrecs else None
    flt = Path(str(f).replace(".delta.gz", ".filter.gz"))
    if flt.exists():
        rep["filter_tuple_mummer"] = list(anim_oracle.parse_delta(flt))
        rep["filter_tuple_ours"] = list(anim_oracle.parse_delta_records(kept)) if kept else None
    for k in ("delta", "filter"):
        m, o = rep.get(f"{k}_tuple_mummer"), rep.get(f"{k}_tuple_ours")
        if m and o:
            rep[f"{k}_identity_abs_diff"] = abs(m[2] - o[2])
            rep[f"{k}_ref_aln_len_rel_diff"] = abs(m[0] - o[0]) / m[0]
            rep[f"{k}_qry_aln_len_rel_diff"] = abs(m[1] - o[1]) / m[1]
    report[f"{grp}/{f.name[:-len('.delta.gz')]}"] = rep
    return f"{grp}/{f.name}", len(want), len(want & got), len(coords), len(got), r.stdout.splitlines()[0] if r.stdout else ""


report = {}
DELTA_TOTALS = [0, 0, 0, 0, 0, 0]
tot_w = tot_e = tot_c = 0
with ThreadPoolExecutor(args.j) as ex:
    for name, nw, ne, nc, ng, first in ex.map(run, jobs):
        tot_w += nw; tot_e += ne; tot_c += nc
        print(f"{name[:70]:70s} mummer {nw:4d}  exact {ne:4d}  coords-only {nc:4d}  ours {ng:4d}", flush=True)
print(f"TOTAL records {tot_w}  exact {tot_e}  same coordinates {tot_c}")
if args.delta:
    print(f"TOTAL indel lists {DELTA_TOTALS[0]}  equal {DELTA_TOTALS[1]};  blocks {DELTA_TOTALS[2]}, alignments in MUMmer's order inside {DELTA_TOTALS[3]} of them;  files whose block sequence is ours (by record ordinal) {DELTA_TOTALS[4]} of {DELTA_TOTALS[5]}")
if args.json:
    import json
    worst = {k: max((r.get(k, 0.0) for r in report.values()), default=0.0) for k in
             ("filter_identity_abs_diff", "filter_ref_aln_len_rel_diff", "filter_qry_aln_len_rel_diff", "delta_identity_abs_diff",
              "delta_ref_aln_len_rel_diff", "delta_qry_aln_len_rel_diff")}
    Path(args.json).write_text(json.dumps({"total_records": tot_w, "exact": tot_e, "same_coordinates": tot_c, "worst": worst,
                                           "pairs": report}, indent=1, sort_keys=True))
