#!/usr/bin/env python3
"""Host ingest rate: read + parse + 2-bit pack of FASTA files with pg_add_fasta_batch on 1 .. all host threads."""
import argparse
import json
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pyani_amd import synth  # noqa: E402
from pyani_amd.engine import Engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100)
ap.add_argument("--length", type=int, default=5_000_000)
args = ap.parse_args()
tmp = Path(tempfile.mkdtemp(dir="/tmp"))


def make(g):
    seq, off = synth.genome(20250228, args.n, g, args.length)
    p = tmp / f"{synth.genome_name(g)}.fna"
    synth.write_fasta(p, seq, off, synth.genome_name(g))
    return p


with ThreadPoolExecutor(16) as ex:
    files = list(ex.map(make, range(args.n)))
nbytes = sum(f.stat().st_size for f in files)
eng = Engine(0)
out = {"files": args.n, "fasta_bytes": nbytes, "host_threads": os.cpu_count(), "runs": []}
for threads in (1, 8, 32, 0):
    eng.clear_genomes()
    t0 = time.perf_counter()
    eng.add_fasta_batch(files, threads=threads)
    t1 = time.perf_counter()
    eng.upload()
    t2 = time.perf_counter()
    out["runs"].append({"threads": threads or os.cpu_count(), "parse_pack_s": t1 - t0, "upload_s": t2 - t1,
                        "fasta_GB_per_s": nbytes / (t1 - t0) / 1e9})
print(json.dumps(out))
for f in files:
    f.unlink()
