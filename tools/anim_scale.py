#!/usr/bin/env python3
"""Throughput probe of the GPU ANIm engine on synthetic RELATED genomes (SURVEY.md §8(d) generator): n genomes of L bases
from ceil(n/25) ancestors, all ordered pairs.  Prints pairs/s and a summary of identities / statuses."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from pyani_amd import synth  # noqa: E402
from pyani_amd.engine import Engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=25)
ap.add_argument("--length", type=int, default=1_000_000)
ap.add_argument("--seed", type=int, default=20250228)
ap.add_argument("--out", default="")
ap.add_argument("--only", choices=["all", "related", "unrelated"], default="all")
args = ap.parse_args()
eng = Engine(0)
t0 = time.time()
ids = [eng.add_genome(*synth.genome(args.seed, args.n, g, args.length)) for g in range(args.n)]
eng.upload()
t_prep = time.time() - t0
K0 = (args.n + 24) // 25
pairs = [(a, b) for a in range(args.n) for b in range(args.n) if a != b and
         (args.only == "all" or ((a % K0) == (b % K0)) == (args.only == "related"))]
t0 = time.time()
res = eng.anim_pairs([ids[a] for a, _ in pairs], [ids[b] for _, b in pairs])
dt = time.time() - t0
ok = res["status"] == 0
K = (args.n + 24) // 25
related = np.array([(a % K) == (b % K) for a, b in pairs])
out = {
    "n": args.n, "length": args.length, "pairs": len(pairs), "seconds": dt, "pairs_per_s": len(pairs) / dt, "prep_s": t_prep,
    "status_counts": {str(int(s)): int((res["status"] == s).sum()) for s in np.unique(res["status"])},
    "related_pairs": int(related.sum()), "related_ok": int((ok & related).sum()),
    "only": args.only,
    "identity_related_min_med_max": [float(np.min(res["identity"][ok & related])), float(np.median(res["identity"][ok & related])),
                                     float(np.max(res["identity"][ok & related]))] if (ok & related).any() else None,
    "coverage_related_median": float(np.median(res["ref_aln_len"][ok & related])) / args.length if (ok & related).any() else None,
    "unrelated_with_alignment": int((ok & ~related).sum()),
    "results_sha1": __import__("hashlib").sha1(res.tobytes()).hexdigest(),
}
print(json.dumps(out))
if args.out:
    Path(args.out).write_text(json.dumps(out, indent=1))
