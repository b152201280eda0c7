#!/usr/bin/env python3
"""Fixture for the full-size ANIm GPU test: one whole C3 family (C3 = 200 synthetic 5 Mb genomes, seed
20250228, K = 8 ancestors; family f = the 25 genomes g with g % 8 == f) — all 600 related ordered pairs — plus 200
unrelated ordered pairs, through the CPU statement of the search (oracle/anim_cpu.cpp, exhaustive seeding, scalar core).
The GPU test (tests/test_anim_c3_gpu.py) runs all 800 pairs in one call; the CPU statement (~15 CPU-s per pair) is computed
for the `--subset` first genomes of the family (8 -> 56 related ordered pairs) and the first 24 unrelated pairs, which the GPU
must reproduce tuple for tuple; the other pairs are listed without expectation (null) and are checked through
size-independent properties and a pinned hash of the GPU's own results.
Usage: python tools/make_anim_c3_family_host.py [--family 3] [--subset 8] [--threads 8]"""
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
ap.add_argument("--subset", type=int, default=8)
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
sub = set(fam[: args.subset])
checked = [k for k, (a, b) in enumerate(pairs) if (k < len(related) and a in sub and b in sub) or len(related) <= k < len(related) + 24]
used = sorted({g for k in checked for g in pairs[k]})
genomes = [synth.genome(SEED, N, g, L) if g in used else None for g in range(N)]
res, secs = anim_cpu.anim_cpu_pairs(genomes, [pairs[k][0] for k in checked], [pairs[k][1] for k in checked], threads=args.threads)
cpu = {k: [int(r["ref_aln_len"]), int(r["qry_aln_len"]), float(r["identity"]).hex(), int(r["sim_errors"]), int(r["n_alignments"]),
           int(r["status"])] for k, r in zip(checked, res)}
out = {"n": N, "length": L, "seed": SEED, "family": args.family, "n_related": len(related), "cpu_seconds": float(secs.sum()),
       "pairs": [[a, b, cpu.get(k)] for k, (a, b) in enumerate(pairs)]}
(ROOT / "tests" / "golden" / "anim_c3_family_host.json").write_text(json.dumps(out, separators=(",", ":")))
print("wrote", len(pairs), "pairs,", len(checked), "of them with the CPU statement,", out["cpu_seconds"], "CPU-s")
