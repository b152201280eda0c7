#!/usr/bin/env python3
"""Development aid (CPU only): fuzz the product's host statement (tools/anim_debug --exact: the engine's own MUM filter, mgaps and
postnuc code) against the independent nucmer oracle (oracle/nucmer_oracle.cpp) on small synthetic genomes whose RECORDS share
content (repeat elements spread over contigs, as rRNA operons / IS elements are in draft assemblies).  MUMmer tests query-side
uniqueness per query record (`mummer` streams one query sequence at a time), reference-side uniqueness over all reference records.
Usage: python tools/anim_fuzz_multirecord.py [--trials N] [--seed S] [--records-max R] [--keep DIR]"""
import argparse
import random
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tests.fuzz_genomes import make_pair, write_fasta  # noqa: E402


def records(stdout):
    return sorted(tuple(l.split()[1:8]) for l in stdout.splitlines() if l.startswith("ALN "))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--records-max", type=int, default=4)
    ap.add_argument("--maxmatch", action="store_true")
    ap.add_argument("--keep", default="")
    args = ap.parse_args()
    oracle = ROOT / "oracle/_build/nucmer_oracle"
    oracle.parent.mkdir(exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", str(ROOT / "oracle/nucmer_oracle.cpp"), "-o", str(oracle)], check=True)
    stmt = ROOT / "tools/anim_debug/anim_debug"
    subprocess.run(["g++", "-O2", "-std=c++17", f"-I{ROOT}/pyani_amd/csrc", str(stmt) + ".cpp", "-o", str(stmt)], check=True)
    tmp = Path(args.keep or tempfile.mkdtemp())
    tmp.mkdir(exist_ok=True)
    bad = 0
    extra = ["--maxmatch"] if args.maxmatch else []
    for t in range(args.trials):
        rng = random.Random(args.seed * 1000003 + t)
        ref, qry = make_pair(rng, args.records_max)
        pa, pb = tmp / f"r{t}.fna", tmp / f"q{t}.fna"
        write_fasta(pa, "r", ref)
        write_fasta(pb, "q", qry)
        o = subprocess.run([str(oracle), str(pa), str(pb)] + extra, capture_output=True, text=True)
        s = subprocess.run([str(stmt), str(pa), str(pb), "--dump", "--exact"] + extra, capture_output=True, text=True)
        if o.returncode or s.returncode:
            print(f"trial {t}: exit {o.returncode} / {s.returncode}: {o.stderr[-200:]} {s.stderr[-200:]}")
            bad += 1
            continue
        ro, rs = records(o.stdout), records(s.stdout)
        if ro != rs:
            bad += 1
            print(f"trial {t}: {len(ref)} ref / {len(qry)} qry records: oracle {len(ro)} vs statement {len(rs)} records; "
                  f"only oracle {sorted(set(ro) - set(rs))[:3]} only statement {sorted(set(rs) - set(ro))[:3]}")
        if not args.keep:
            pa.unlink()
            pb.unlink()
    print(f"{args.trials - bad} of {args.trials} trials identical")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
