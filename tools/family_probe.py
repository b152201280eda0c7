#!/usr/bin/env python3
"""Development aid (needs an MI355X): ONE family-sized ANIm call (25 genomes of one ancestor of the C4 generator, 600 ordered pairs, all
related) with the engines' own counters on — where a small call's time goes, which forced runs are its tail.
  PYANI_DEV_KNOBS=1 PYANI_PN_STATS=1 PYANI_ANIM_WORKERS=1 python tools/family_probe.py     ([pn-stats] lines on stderr, stage times on stdout)"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from pyani_amd import _lib  # noqa: E402
from pyani_amd.engine import Engine  # noqa: E402

n, L, seed = 1000, 5_000_000, 20250301
K = (n + 24) // 25
fam = [g for g in range(n) if g % K == 1][:25]
with Engine(0) as eng:
    ids = [eng.add_genome(*bench.synth_genomes(seed, n, L, g, g + 1, 1)[0]) for g in fam]
    eng.upload()
    pairs = [(a, b) for a in ids for b in ids if a != b]
    r, q = [a for a, _ in pairs], [b for _, b in pairs]
    eng.anim_pairs(r, q)
    print("---- warm-up done ----", file=sys.stderr, flush=True)
    stages = [_lib.K_ANIM_SEED, _lib.K_ANIM_HIT, _lib.K_ANIM_CLUSTER, _lib.K_ANIM_GAPS, _lib.K_ANIM_FWD, _lib.K_ANIM_BWD, _lib.K_ANIM_EXTEND, _lib.K_ANIM_EXTLANE, _lib.K_ANIM_FINISH]
    eng.profile_reset()
    eng.profile_config(kernel_mask=sum(1 << s for s in stages), every_n=1)
    eng.profile_enable(True)
    t0 = time.perf_counter()
    eng.anim_pairs(r, q)
    dt = time.perf_counter() - t0
    eng.profile_enable(False)
    print(f"{len(pairs)} pairs in {dt:.3f} s;", {eng.kernel_name(s): round(eng.profile_get(s)[0], 1) for s in stages})
