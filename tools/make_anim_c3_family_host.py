#!/usr/bin/env python3
"""Fixture for the full-size ANIm GPU test: one whole C3 family (C3 = 200 synthetic 5 Mb genomes, seed
20250228, K = 8 ancestors; family f = the 25 genomes g with g % 8 == f) — all 600 related ordered pairs — plus 200
unrelated ordered pairs, through the CPU statement of the search (oracle/anim_cpu.cpp, exhaustive seeding, scalar core).
The GPU pipeline must reproduce every tuple (tests/test_anim_c3_gpu.py).  ~15 CPU-s per pair.
Usage: python tools/make_anim_c3_family_host.py [--family 3] [--threads 8]"""
import argparse
import hashlib
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
import anim_cpu  # noqa: E402
from pyani_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--family", type=int, default=3)
ap.add_argument("--threads", type=int, default=0)
args = ap.parse_args()
N, L, SEED, K = 200, 5_000_000, 20250228, 8
fam = [g for g in range(N) if g % K == args.family]
related = [(a, b) for a in fam for b in fam if a != b]
rng = np.random.RandomState(20250228)
unrelated = []
while len(unrelated) < 200:
    a, b = int(fam[rng.randint(len(fam))]), int(rng.randint(N))
    if b % K != args.family and (a, b) not in unrelated and (b, a) not in unrelated:
        unrelated.append((a, b) if len(unrelated) % 2 == 0 else (b, a))
pairs = related + unrelated
used = sorted({g for p in pairs for g in p})
genomes = [synth.genome(SEED, N, g, L) if g in used else None for g in range(N)]
res, secs = anim_cpu.anim_cpu_pairs(genomes, [a for a, _ in pairs], [b for _, b in pairs], threads=args.threads)
out = {"n": N, "length": L, "seed": SEED, "family": args.family, "n_related": len(related), "cpu_seconds": float(secs.sum()),
       "pairs": [[a, b, int(r["ref_aln_len"]), int(r["qry_aln_len"]), float(r["identity"]).hex(), int(r["sim_errors"]),
                  int(r["n_alignments"]), int(r["status"])] for (a, b), r in zip(pairs, res)]}
out["sha1"] = hashlib.sha1(json.dumps(out["pairs"]).encode()).hexdigest()
(ROOT / "tests" / "golden" / "anim_c3_family_host.json").write_text(json.dumps(out, separators=(",", ":")))
print("wrote", len(pairs), "pairs,", out["cpu_seconds"], "CPU-s, sha1", out["sha1"])
