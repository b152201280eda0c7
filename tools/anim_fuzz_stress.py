#!/usr/bin/env python3
"""Development aid (CPU only): a fuzz CAMPAIGN of the product's host statement (tools/anim_debug: the engine's own headers) against
the independent nucmer oracle (oracle/nucmer_oracle.cpp) + the pure-Python 1-to-1 filter / parse_delta (oracle/anim_oracle.py), on
the generators of tests/stress_genomes.py and tests/fuzz_genomes.py with seeds the test suite does not use.  Every trial compares,
in both directions of the pair: the record set (filter off), every keep / drop decision and the printed tuple (filter on), under
`--mum` and `--maxmatch`.
Usage: python tools/anim_fuzz_stress.py [--kind rearranged|tandem|twostrand|multirecord|all] [--trials N] [--seed S] [--jobs J]"""
import argparse
import random
import subprocess
import sys
import tempfile
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
from tests.fuzz_genomes import make_pair, write_fasta  # noqa: E402
from tests.stress_genomes import expected_filtered, make_rearranged_pair, make_tandem_pair, make_two_strand_repeat_pair  # noqa: E402

ORACLE = ROOT / "oracle" / "_build" / "nucmer_oracle"
STMT = ROOT / "tools" / "anim_debug" / "anim_debug"


def _rec(t):
    return (t[1], t[2], int(t[3]), int(t[4]), int(t[5]), int(t[6]), int(t[7]))


def oracle_records(pa, pb, extra):
    out = subprocess.run([str(ORACLE), str(pa), str(pb), *extra], capture_output=True, text=True, check=True).stdout
    return [_rec(t) for t in (ln.split() for ln in out.splitlines()) if t and t[0] == "ALN"]


def statement(pa, pb, extra):
    r = subprocess.run([str(STMT), str(pa), str(pb), "--dump", *extra], capture_output=True, text=True)
    if r.returncode not in (0, 4):
        return None, None, f"exit {r.returncode}: {r.stderr[-200:]}"
    recs, tup = {}, None
    for ln in r.stdout.splitlines():
        t = ln.split()
        if t and t[0] == "ALN":
            recs[_rec(t)] = int(t[8].split("=")[1])
        elif len(t) == 5 and tup is None:
            tup = t
    return recs, tup, None


def genomes(kind, rng):
    if kind == "rearranged":
        return make_rearranged_pair(rng)
    if kind == "tandem":
        a, b = make_tandem_pair(rng)
        return [a], [b]
    if kind == "twostrand":
        return make_two_strand_repeat_pair(rng)
    return make_pair(rng, 4)


def trial(job):
    kind, seed = job
    rng = random.Random(seed)
    ref, qry = genomes(kind, rng)
    problems = []
    with tempfile.TemporaryDirectory() as d:
        pa, pb = Path(d) / "r.fna", Path(d) / "q.fna"
        write_fasta(pa, "r", ref)
        write_fasta(pb, "q", qry)
        for a, b in ((pa, pb), (pb, pa)):
            for extra in ([], ["--maxmatch"]):
                want = oracle_records(a, b, extra)
                keep, tup = expected_filtered(want)
                got, printed, err = statement(a, b, extra)
                tag = f"{kind} seed {seed} {a.name}->{b.name} {' '.join(extra) or '--mum'}"
                if err:
                    problems.append(f"{tag}: {err}")
                    continue
                if set(got) != set(want):
                    problems.append(f"{tag}: records differ ({len(want)} vs {len(got)}): {sorted(set(got) ^ set(want))[:3]}")
                    continue
                bad = [(r, k, got[r]) for r, k in zip(want, keep) if (got[r] == 3) != k]
                if bad:
                    problems.append(f"{tag}: {len(bad)} keep/drop decisions differ: {bad[:2]}")
                elif tup is not None and (int(printed[0]), int(printed[1]), float(printed[2]), int(printed[3]), int(printed[4])) != tup:
                    problems.append(f"{tag}: tuple {printed} vs {tup}")
    return problems


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="all")
    ap.add_argument("--trials", type=int, default=100)
    ap.add_argument("--seed", type=int, default=900001)
    ap.add_argument("--jobs", type=int, default=4)
    args = ap.parse_args()
    ORACLE.parent.mkdir(exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", str(ROOT / "oracle/nucmer_oracle.cpp"), "-o", str(ORACLE)], check=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", f"-I{ROOT}/pyani_amd/csrc", str(STMT) + ".cpp", "-o", str(STMT)], check=True)
    kinds = ["rearranged", "tandem", "twostrand", "multirecord"] if args.kind == "all" else [args.kind]
    jobs = [(k, args.seed * 7919 + 104729 * i + kinds.index(k)) for k in kinds for i in range(args.trials)]
    bad = 0
    with ProcessPoolExecutor(args.jobs) as ex:
        for job, problems in zip(jobs, ex.map(trial, jobs, chunksize=1)):
            for p in problems:
                print(p, flush=True)
            bad += bool(problems)
    print(f"{len(jobs) - bad} of {len(jobs)} trials identical ({', '.join(kinds)}; 2 directions x --mum / --maxmatch each)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
