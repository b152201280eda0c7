#!/usr/bin/env python3
"""GPU fuzz campaign of fragment mode against the INDEPENDENT blastn oracle (needs an MI355X; oracle/blastn_oracle.cpp runs on the host).

  python tools/anib_fuzz_gpu.py [--seeds 12] [--length 200000] [--out gpurun_out/r06/anib_fuzz_gpu.json]

Per seed (none of them used by the test suite): a family of 6 descendants of one ancestor of the synthetic generator (pyani_amd.synth:
0.1 ... 15 % divergence per genome, 1 - 3 records, inversions), all 30 ordered pairs through pg_anib_pair_rows (the HIP path, C ABI)
and through the oracle; the rows parse_blast_tab uses (first row of a fragment with coverage > 70 %, identity > 30 %: pyani/anib.py:
641-649) compared one by one.  The report buckets the pairs by the oracle's mean identity: used rows, identical rows, rows on one side
only, and the largest tuple differences in the bucket."""
import argparse
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "oracle", ROOT / "tools"):
    sys.path.insert(0, str(p))

import blastn_oracle  # noqa: E402
import blastn_oracle_agreement as agreement  # noqa: E402
from anib_product_vs_oracle import side_by_side, tuples  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=12)
    ap.add_argument("--first-seed", type=int, default=7_310_001)
    ap.add_argument("--length", type=int, default=200_000)
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "r06" / "anib_fuzz_gpu.json"))
    a = ap.parse_args()
    from pyani_amd import synth
    from pyani_amd.engine import Engine
    edges = [(99.0, 101.0), (95.0, 99.0), (90.0, 95.0), (85.0, 90.0), (80.0, 85.0), (0.0, 80.0)]
    buckets = {f"{lo:g}-{hi:g}": {"pairs": 0, "used_rows_oracle": 0, "used_rows_gpu": 0, "identical": 0, "same_extent_and_gaps_mismatch_count_differs": 0,
                                   "only_gpu": 0, "only_oracle": 0, "max_abs_identity_pp_diff": 0.0, "max_abs_aln_length_rel_diff": 0.0,
                                   "max_abs_sim_errors_rel_diff": 0.0, "pairs_with_every_used_row_identical": 0} for lo, hi in edges}
    worst = []
    t0 = time.time()
    with Engine(0) as eng:
        for k in range(a.seeds):
            seed, n = a.first_seed + 977 * k, 6
            data = [synth.genome(seed, n, g, a.length + 1013 * k) for g in range(n)]
            eng.clear_genomes()
            ids = [eng.add_genome(*d) for d in data]
            eng.upload()
            for x in range(n):
                for y in range(n):
                    if x == y:
                        continue
                    up = agreement.used_rows(tuples(eng.anib_pair_rows(ids[x], ids[y])))
                    uo = agreement.used_rows(tuples(blastn_oracle.blastn_pair(data[x], data[y], threads=a.threads)))
                    if not uo and not up:
                        continue
                    rep = side_by_side(up, uo)
                    pid = rep["tuple_other"][2]
                    key = next(f"{lo:g}-{hi:g}" for lo, hi in edges if lo <= pid < hi)
                    b = buckets[key]
                    b["pairs"] += 1
                    b["used_rows_oracle"] += rep["used_rows_other"]
                    b["used_rows_gpu"] += rep["used_rows_product"]
                    b["identical"] += rep["identical"]
                    b["same_extent_and_gaps_mismatch_count_differs"] += rep["same_extent_and_gaps_mismatch_count_differs"]
                    b["only_gpu"] += rep["only_product"]
                    b["only_oracle"] += rep["only_other"]
                    b["pairs_with_every_used_row_identical"] += int(rep["identical"] == rep["used_rows_other"] == rep["used_rows_product"])
                    for f, src in (("max_abs_identity_pp_diff", "identity_pp_diff"), ("max_abs_aln_length_rel_diff", "aln_length_rel_diff"),
                                   ("max_abs_sim_errors_rel_diff", "sim_errors_rel_diff")):
                        b[f] = max(b[f], abs(rep[src]))
                    if rep["identical"] != rep["used_rows_other"]:
                        worst.append({"seed": seed, "query": x, "subject": y, "oracle_identity": pid, **{f: rep[f] for f in ("used_rows_other", "identical", "only_product", "only_other", "identity_pp_diff")}})
            print(f"seed {seed}: done ({time.time() - t0:.0f} s)", flush=True)
    for b in buckets.values():
        b["identical_fraction"] = b["identical"] / max(1, b["used_rows_oracle"])
    tot = sum(b["used_rows_oracle"] for b in buckets.values())
    same = sum(b["identical"] for b in buckets.values())
    worst.sort(key=lambda w: w["identical"] / max(1, w["used_rows_other"]))
    out = {"what": "pg_anib_pair_rows (HIP, through the C ABI) vs oracle/blastn_oracle.cpp on synthetic families; rows = those parse_blast_tab uses",
           "seeds": a.seeds, "first_seed": a.first_seed, "length": a.length, "pairs": sum(b["pairs"] for b in buckets.values()),
           "used_rows_oracle": tot, "identical": same, "identical_fraction": same / max(1, tot), "by_oracle_identity_percent": buckets,
           "pairs_with_a_difference_worst_first": worst[:40], "seconds": time.time() - t0}
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    Path(a.out).write_text(json.dumps(out, indent=1))
    print(json.dumps({k: v for k, v in out.items() if k not in ("pairs_with_a_difference_worst_first",)}, indent=1))


if __name__ == "__main__":
    main()
