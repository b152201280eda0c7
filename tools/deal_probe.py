#!/usr/bin/env python3
"""Development aid (needs an MI355X): how the N > 1 dealing of bench.py / pyani_amd.parallel would fare on C4, measured on ONE GPU.
(1) the 8 static shards (parallel.anim_row_shard) of the two 800-row steps a job of 8 ranks runs: one call each, timed — max / mean is
the imbalance a static deal would have; (2) calls of 100, 80, 50, 25, 12, 6, 3, 2 scrambled rows: what a call of that size costs per row,
i.e. what every extra chunk of a dynamic deal pays in launch tails.  Writes gpurun_out/r05/deal_probe.json."""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from pyani_amd import parallel  # noqa: E402
from pyani_amd.engine import Engine  # noqa: E402

n, L, seed = 1000, 5_000_000, 20250301
data = bench.synth_genomes(seed, n, L, 0, n, 1)
out = {"static": [], "sizes": []}
with Engine(0) as eng:
    ids = np.asarray([eng.add_genome(s, o) for s, o in data], dtype=np.int32)
    eng.upload()

    def call(rows):
        pairs = parallel.anim_pair_array(n, rows, symmetric=True)
        t0 = time.perf_counter()
        eng.anim_pairs(ids[pairs[:, 0]], ids[pairs[:, 1]])
        return time.perf_counter() - t0, len(pairs)

    call(list(range(900, 1000)))                       # warm-up: seed lists of every genome, scratch
    call(list(range(0, 100)))
    for step in (0, 1):
        rows = [(step * 800 + i) % n for i in range(800)]
        for world in (8, 4, 2):
            ts = [call(parallel.anim_row_shard(rows, r, world))[0] for r in range(world)]
            out["static"].append({"step": step, "world": world, "seconds": ts, "max_over_mean": max(ts) / (sum(ts) / len(ts))})
            print(out["static"][-1], flush=True)
    order = sorted(range(800), key=lambda q: ((q * 0x9E3779B1) & 0xFFFFFFFF, q))
    for size in (100, 80, 50, 25, 12, 6, 3, 2):
        reps = max(1, min(6, 200 // size))
        ts = []
        for k in range(reps):
            t, m = call(order[300 + k * size:300 + (k + 1) * size])
            ts.append(t)
        out["sizes"].append({"rows": size, "calls": reps, "seconds_per_call": sum(ts) / len(ts), "seconds_per_row": sum(ts) / len(ts) / size})
        print(out["sizes"][-1], flush=True)
dst = ROOT / "gpurun_out" / "r05" / "deal_probe.json"
dst.parent.mkdir(parents=True, exist_ok=True)
dst.write_text(json.dumps(out, indent=1))
