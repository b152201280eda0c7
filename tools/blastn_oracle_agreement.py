"""Row-level agreement of oracle/blastn_oracle.cpp with the BLAST+ tables the reference's tests hold
(tests/golden/anib/*.blast_tab.gz = /root/reference/tests/fixtures/anib/blastn/*.blast_tab, copied as data).

  python tools/blastn_oracle_agreement.py [--pairs Q_vs_S ...] [--frags N] [--out profiles/r06_blastn_oracle_vs_blastplus.json]

Per table: rows of BLAST+, rows of the oracle, rows identical in (fragment, length, mismatch, gaps, qstart, qend, sstart, send),
rows on one side only; the same for the rows parse_blast_tab actually uses (the first row of a fragment that passes its filters,
pyani/anib.py:641-649); and parse_blast_tab's tuple on both sides.  TEST INFRASTRUCTURE: runs on the CPU, nothing of the product.
"""
import argparse
import gzip
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))

import anib_oracle  # noqa: E402
import blastn_oracle  # noqa: E402

GOLD = ROOT / "tests" / "golden"


def blast_rows(path, srec_names):
    """(frag0, length, mismatch, gaps, qstart, qend, sstart, send, srec) of every row of a 15-column BLAST+ table."""
    rows = []
    with gzip.open(path, "rt") as fh:
        for line in fh:
            f = line.rstrip("\n").split("\t")
            rows.append((int(f[0][4:]) - 1, int(f[2]), int(f[3]), int(f[14]), int(f[8]), int(f[9]), int(f[10]), int(f[11]),
                         srec_names.index(f[1]), int(f[6]), float(f[4])))
    return rows


def used_rows(rows):
    """parse_blast_tab's choice: first row per fragment with coverage > 0.7 and identity > 0.3 (rows: (frag, length, mismatch, gaps, ..., qlen at [9]))."""
    seen = {}
    for r in rows:
        alnlen = r[1] - r[3]
        if alnlen / r[9] > 0.7 and (alnlen - r[2]) / r[9] > 0.3 and r[0] not in seen:
            seen[r[0]] = r
    return seen


def reduce_used(used):
    if not used:
        return [0, 0, 0.0]
    ks = sorted(used)
    aln = sum(used[k][1] - used[k][3] for k in ks)
    err = sum(used[k][2] + used[k][3] for k in ks)
    pid = sum(used[k][10] for k in ks) / len(ks)
    return [aln, err, pid]


def read_fasta_gz(path):
    """gzipped FASTA -> (uint8 concatenated sequence, uint64 record offsets)."""
    import numpy as np
    parts, off, cur = [], [0], []
    with gzip.open(path, "rt") as fh:
        for line in fh:
            if line.startswith(">"):
                if cur or len(off) > 1 or parts:
                    parts.append("".join(cur)); off.append(off[-1] + len(parts[-1])); cur = []
                started = True
            else:
                cur.append(line.strip())
    parts.append("".join(cur)); off.append(off[-1] + len(parts[-1]))
    data = "".join(parts).encode("latin-1")
    return np.frombuffer(data, dtype=np.uint8).copy(), np.array(off, dtype=np.uint64)


def record_names(fasta_gz):
    names = []
    with gzip.open(fasta_gz, "rt") as fh:
        for line in fh:
            if line.startswith(">"):
                names.append(line[1:].split()[0])
    return names


def compare(q, s, max_frags=None, genomes=None):
    gdir = GOLD / "genomes" / "caulobacter"
    Q = genomes[q] if genomes else read_fasta_gz(gdir / f"{q}.fna.gz")
    S = genomes[s] if genomes else read_fasta_gz(gdir / f"{s}.fna.gz")
    if max_frags:
        Q = (Q[0][:int(min(len(Q[0]), max_frags * 1020))], Q[1].copy())
        Q[1][-1] = len(Q[0])
        Q = (Q[0], Q[1][Q[1] <= len(Q[0])])
    names = record_names(gdir / f"{s}.fna.gz")
    t0 = time.time()
    got = blastn_oracle.blastn_pair(Q, S)
    dt = time.time() - t0
    ours = [(int(r["frag"]), int(r["length"]), int(r["mismatch"]), int(r["gaps"]), int(r["qstart"]), int(r["qend"]), int(r["sstart"]),
             int(r["send"]), int(r["srec"]), int(r["qlen"]), float("%.3f" % (100.0 * int(r["nident"]) / int(r["length"])))) for r in got]
    blast = blast_rows(GOLD / "anib" / f"{q}_vs_{s}.blast_tab.gz", names)
    if max_frags:
        blast = [r for r in blast if r[0] < max_frags]
    key = lambda r: r[:9]      # noqa: E731
    sb, so = set(map(key, blast)), set(map(key, ours))
    ub, uo = used_rows(blast), used_rows(ours)
    same_used = sum(1 for k in ub if k in uo and key(ub[k]) == key(uo[k]))
    rb, ro = reduce_used(ub), reduce_used(uo)
    # a looser class: same fragment, strand and subject interval within 3 bases at both ends
    rep = {
        "rows_blast": len(blast), "rows_oracle": len(ours), "rows_identical": len(sb & so),
        "rows_only_blast": len(sb - so), "rows_only_oracle": len(so - sb),
        "identical_fraction_of_blast": len(sb & so) / max(1, len(sb)),
        "used_rows_blast": len(ub), "used_rows_oracle": len(uo), "used_rows_identical": same_used,
        "used_only_blast": len(set(ub) - set(uo)), "used_only_oracle": len(set(uo) - set(ub)),
        "used_identical_fraction_of_blast": same_used / max(1, len(ub)),
        "parse_blast_tab_blast": rb, "parse_blast_tab_oracle": ro,
        "identity_pp_diff": ro[2] - rb[2], "aln_length_rel_diff": (ro[0] - rb[0]) / max(1, rb[0]),
        "sim_errors_rel_diff": (ro[1] - rb[1]) / max(1, rb[1]), "oracle_seconds": round(dt, 1),
    }
    return rep, blast, ours


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", nargs="*")
    ap.add_argument("--frags", type=int, default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--show", type=int, default=0, help="print this many differing used rows")
    a = ap.parse_args()
    stems = ["NC_002696", "NC_010338", "NC_011916", "NC_014100"]
    pairs = a.pairs or [f"{q}_vs_{s}" for q in stems for s in stems if q != s]
    report = {}
    for p in pairs:
        q, s = p.split("_vs_")
        rep, blast, ours = compare(q, s, a.frags)
        report[p] = rep
        print(p, json.dumps(rep), flush=True)
        if a.show:
            ub, uo = used_rows(blast), used_rows(ours)
            n = 0
            for k in sorted(ub):
                if k not in uo or ub[k][:9] != uo[k][:9]:
                    print("  B", ub[k][:10], "\n  O", uo.get(k, ())[:10] if k in uo else None)
                    ob = [r[:9] for r in ours if r[0] == k]
                    print("     all oracle rows of this fragment:", ob[:6])
                    n += 1
                    if n >= a.show:
                        break
    if a.out:
        Path(a.out).write_text(json.dumps(report, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
