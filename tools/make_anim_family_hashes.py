#!/usr/bin/env python3
"""Whole FAMILIES of the C4 benchmark (BASELINE.json configs[3]: 1000 x 5 Mb, seed 20250301; a family = the 25 descendants of one
ancestor, 600 ordered pairs, identity 0.72 ... 0.999, 1 - 3 records per genome, some with N runs) through the INDEPENDENT nucmer oracle
(oracle/nucmer_oracle.cpp) and the pure-Python 1-to-1 filter + parse_delta (oracle/anim_oracle.py), at full size.  600 pairs of
records would be megabytes, so per ordered pair only a digest is kept: the number of records, the SHA-1 of the sorted records, the
SHA-1 of the sorted records with their keep / drop decision, and the filtered tuple (identity as float.hex).
tests/test_anim_oracle_family_gpu.py recomputes the same digests from pg_anim_alignments_batch / pg_anim_pairs on the GPU.
Output: tests/golden/anim_oracle_family_digests.json.gz.   Usage: python tools/make_anim_family_hashes.py [--families 5,17] [--threads 8]"""
import argparse
import gzip
import hashlib
import json
import subprocess
import sys
import tempfile
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
from pyani_amd import synth  # noqa: E402
from tests.stress_genomes import expected_filtered  # noqa: E402

N, L, SEED = 1000, 5_000_000, 20250301
K = (N + 24) // 25


def digest(records, keep):
    """records: (rrec, qrec, rs, re, qs, qe, err) tuples; keep: parallel booleans"""
    rows = sorted(zip(records, keep))
    h1 = hashlib.sha1("\n".join(",".join(map(str, r)) for r, _ in rows).encode()).hexdigest()
    h2 = hashlib.sha1("\n".join(",".join(map(str, r)) + ("+" if k else "-") for r, k in rows).encode()).hexdigest()
    return h1, h2


EXE = ROOT / "oracle" / "_build" / "nucmer_oracle"


def run(job):
    f, a, b, pa, pb = job
    out = subprocess.run([str(EXE), str(pa), str(pb)], capture_output=True, text=True, check=True).stdout
    named = [(t[1], t[2]) + tuple(int(x) for x in t[3:8]) for t in (ln.split() for ln in out.splitlines()) if t and t[0] == "ALN"]
    keep, tup = expected_filtered(named)                    # (the oracle's output order: delta-filter's ties look at it)
    recs = [(int(r[0].rsplit("_r", 1)[1]), int(r[1].rsplit("_r", 1)[1])) + r[2:] for r in named]
    h1, h2 = digest(recs, keep)
    t = None if tup is None else [tup[0], tup[1], float(tup[2]).hex(), tup[3], tup[4]]
    return [a, b, len(recs), sum(keep), h1, h2, t]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--families", default="5,17")
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    EXE.parent.mkdir(exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", str(ROOT / "oracle" / "nucmer_oracle.cpp"), "-o", str(EXE)], check=True)
    tmp = Path(tempfile.mkdtemp())
    fams = [int(x) for x in args.families.split(",")]
    jobs = []
    for f in fams:
        members = [g for g in range(N) if g % K == f]
        paths = {}
        for g in members:
            seq, off = synth.genome(SEED, N, g, L)
            paths[g] = tmp / f"{synth.genome_name(g)}.fna"
            synth.write_fasta(paths[g], seq, off, synth.genome_name(g))
        jobs += [(f, a, b, paths[a], paths[b]) for a in members for b in members if a != b]

    with ProcessPoolExecutor(args.threads) as ex:          # (processes: the filter restatement is pure Python)
        rows = list(ex.map(run, jobs, chunksize=4))
    out = {"n": N, "L": L, "seed": SEED, "families": fams, "pairs": rows,
           "digest": "sha1 of the sorted 'rrec,qrec,rs,re,qs,qe,errors' lines; second digest: the same lines with '+' (kept by delta-filter -1) or '-'"}
    dst = ROOT / "tests" / "golden" / "anim_oracle_family_digests.json.gz"
    with gzip.open(dst, "wt", compresslevel=9) as fh:
        json.dump(out, fh, separators=(",", ":"))
    n_rec = sum(r[2] for r in rows)
    print(f"{len(rows)} ordered pairs, {n_rec} records, {n_rec - sum(r[3] for r in rows)} dropped by the filter; wrote {dst} ({dst.stat().st_size} bytes)")


if __name__ == "__main__":
    main()
