#!/usr/bin/env python3
"""Development aid (needs an MI355X): the GPU twin of tools/anim_fuzz_stress.py.  N seeded trials of each generator of
tests/stress_genomes.py / tests/fuzz_genomes.py (seeds the test suite does not use), both directions of every pair, `--mum` and
`--maxmatch`: the engine through the C ABI — pg_anim_alignments_batch's records and keep flags, pg_anim_pairs' filtered tuple — against
the independent nucmer oracle (oracle/nucmer_oracle.cpp, run on the host threads) + oracle/anim_oracle.py's 1-to-1 filter and
parse_delta.  Writes a JSON summary (default gpurun_out/r05/fuzz_gpu.json) and exits 1 on any difference.
Usage: python tools/anim_fuzz_stress_gpu.py [--trials N] [--seed S] [--out FILE]"""
import argparse
import json
import os
import random
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
from tests.fuzz_genomes import make_pair, write_fasta  # noqa: E402
from tests.stress_genomes import expected_filtered, make_rearranged_pair, make_tandem_pair, make_two_strand_repeat_pair  # noqa: E402

KINDS = ["rearranged", "tandem", "twostrand", "multirecord"]


def genomes(kind, rng):
    if kind == "rearranged":
        return make_rearranged_pair(rng)
    if kind == "tandem":
        a, b = make_tandem_pair(rng)
        return [a], [b]
    if kind == "twostrand":
        return make_two_strand_repeat_pair(rng)
    return make_pair(rng, 4)


def oracle_records(exe, pa, pb, extra):
    out = subprocess.run([str(exe), str(pa), str(pb), *extra], capture_output=True, text=True, check=True).stdout
    return [(t[1], t[2], int(t[3]), int(t[4]), int(t[5]), int(t[6]), int(t[7])) for t in (ln.split() for ln in out.splitlines()) if t and t[0] == "ALN"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=150)
    ap.add_argument("--seed", type=int, default=910001)
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "r05" / "fuzz_gpu.json"))
    args = ap.parse_args()
    from pyani_amd import anim
    from pyani_amd.engine import Engine
    exe = ROOT / "oracle" / "_build" / "nucmer_oracle"
    exe.parent.mkdir(exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", str(ROOT / "oracle" / "nucmer_oracle.cpp"), "-o", str(exe)], check=True)
    tmp = Path(tempfile.mkdtemp())
    trials = []
    for ki, kind in enumerate(KINDS):
        for i in range(args.trials):
            seed = args.seed * 7919 + 104729 * i + ki
            ref, qry = genomes(kind, random.Random(seed))
            pa, pb = tmp / f"{kind}_{i}_r.fna", tmp / f"{kind}_{i}_q.fna"
            write_fasta(pa, f"r{ki}_{i}_", ref)
            write_fasta(pb, f"q{ki}_{i}_", qry)
            trials.append((kind, seed, pa, pb))
    # the oracle on the host threads: (trial, direction, maxmatch) -> records in its output order
    jobs = [(k, d, mm) for k in range(len(trials)) for d in (0, 1) for mm in (False, True)]
    t0 = time.perf_counter()

    def run(job):
        k, d, mm = job
        pa, pb = trials[k][2:] if d == 0 else trials[k][2:][::-1]
        return oracle_records(exe, pa, pb, ["--maxmatch"] if mm else [])
    with ThreadPoolExecutor(min(128, os.cpu_count() or 8)) as ex:
        want = dict(zip(jobs, ex.map(run, jobs)))
    t_oracle = time.perf_counter() - t0
    names = {}
    problems, n_records, n_dropped = [], 0, 0
    t0 = time.perf_counter()
    with Engine(0) as eng:
        ids = []
        for _, _, pa, pb in trials:
            ids.append((eng.add_fasta(pa)[0], eng.add_fasta(pb)[0]))
            for p in (pa, pb):
                names[p] = {n[0]: j for j, n in enumerate(anim.fasta_records(p))}
        for mm in (False, True):
            r = [i[0] for i in ids] + [i[1] for i in ids]
            q = [i[1] for i in ids] + [i[0] for i in ids]
            res = eng.anim_pairs(r, q, maxmatch=mm)                     # filter ON
            off, recs, _, _ = eng.anim_alignments_batch(r, q, maxmatch=mm)
            for j in range(2 * len(trials)):
                k, d = j % len(trials), j // len(trials)
                kind, seed, pa, pb = trials[k]
                a, b = (pa, pb) if d == 0 else (pb, pa)
                orc = want[(k, d, mm)]
                keep, tup = expected_filtered(orc)
                w = [(names[a][x[0]], names[b][x[1]]) + x[2:] for x in orc]
                got = {(int(x["ref_rec"]), int(x["qry_rec"]), int(x["rs"]), int(x["re"]), int(x["qs"]), int(x["qe"]), int(x["errors"])): int(x["kept"])
                       for x in recs[int(off[j]):int(off[j + 1])]}
                tag = f"{kind} seed {seed} dir {d} {'--maxmatch' if mm else '--mum'}"
                n_records += len(w)
                n_dropped += len(keep) - sum(keep)
                if set(got) != set(w):
                    problems.append(f"{tag}: records differ ({len(w)} vs {len(got)}): {sorted(set(got) ^ set(w))[:3]}")
                    continue
                bad = [(x, kp, got[x]) for x, kp in zip(w, keep) if (got[x] == 3) != kp]
                if bad:
                    problems.append(f"{tag}: {len(bad)} keep/drop decisions differ: {bad[:2]}")
                    continue
                t = res[j]
                if tup is None:
                    if not (int(t["n_alignments"]) == 0 and int(t["status"]) == 1):
                        problems.append(f"{tag}: expected no alignment, got {t}")
                elif (int(t["ref_aln_len"]), int(t["qry_aln_len"]), float(t["identity"]), int(t["sim_errors"]), int(t["n_alignments"])) != tup:
                    problems.append(f"{tag}: tuple {t} vs {tup}")
    t_gpu = time.perf_counter() - t0
    out = {"trials_per_kind": args.trials, "kinds": KINDS, "seed": args.seed, "comparisons": len(jobs), "records": n_records,
           "records_dropped_by_filter": n_dropped, "differences": len(problems), "first_differences": problems[:20],
           "oracle_seconds_on_host_threads": round(t_oracle, 1), "engine_seconds": round(t_gpu, 1),
           "what": "pg_anim_alignments_batch records + keep flags and pg_anim_pairs tuple (filter on) vs oracle/nucmer_oracle.cpp + "
                   "oracle/anim_oracle.py, both directions, --mum and --maxmatch"}
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(out, indent=1))
    print(json.dumps(out, indent=1))
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
