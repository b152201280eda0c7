"""oracle/anib_oracle.py — CPU restatement of pyani's ANIb fragmenting rule and BLAST-tab reduction.  TEST INFRASTRUCTURE ONLY.

  fragment_lengths(record_lengths, fragsize)   pyani/anib.py:164-203  (consecutive pieces, last one shorter, ids
                                               frag%05d numbered across all records of the file)
  parse_blast_tab(path)                        pyani/anib.py:569-667, mode "ANIb" (BLAST+ 15-column format):
        ani_alnlen = length - gaps ; ani_alnids = ani_alnlen - mismatch ; coverage = ani_alnlen/qlen ; pid = ani_alnids/qlen
        keep rows with coverage > 0.7 and pid > 0.3, then the FIRST kept row of every fragment;
        returns (sum ani_alnlen, sum mismatch + sum gaps, mean blast pident)      (0 hits -> pident 0)

Pinned by the reference's known answer for NC_002696_vs_NC_011916 (tests/test_anib.py:387-391:
4 016 551, 93, 99.997 693 577 050 029) and by dataframes/blastn_result.csv (identity = 0.01 * mean pident, 6 d.p.).
The BLAST search itself is external (BLAST+), absent here: only these files pin it.
"""
import gzip
import math
from typing import Dict, List, Tuple


def fragment_lengths(record_lengths: List[int], fragsize: int = 1020) -> Dict[str, int]:
    out, count = {}, 0
    for n in record_lengths:
        idx = 0
        while idx < n:
            count += 1
            out["frag%05d" % count] = min(fragsize, n - idx)
            idx += fragsize
    return out


def read_blast_tab(path):
    """Rows of a 15-column BLAST+ table: (qseqid, length, mismatch, pident, qlen, gaps)."""
    opener = gzip.open if str(path).endswith(".gz") else open
    rows = []
    with opener(path, "rt") as fh:
        for line in fh:
            f = line.rstrip("\n").split("\t")
            if len(f) < 15:
                continue
            rows.append((f[0], int(f[2]), int(f[3]), float(f[4]), int(f[6]), int(f[14])))
    return rows


def parse_blast_rows(rows) -> Tuple[int, int, float]:
    seen = {}
    for qid, length, mismatch, pident, qlen, gaps in rows:
        alnlen = length - gaps
        alnids = alnlen - mismatch
        if alnlen / qlen > 0.7 and alnids / qlen > 0.3 and qid not in seen:
            seen[qid] = (alnlen, mismatch, gaps, pident)
    if not seen:
        return 0, 0, 0.0
    kept = [seen[k] for k in sorted(seen)]      # groupby(index) sorts the fragment ids
    aln = sum(k[0] for k in kept)
    err = sum(k[1] for k in kept) + sum(k[2] for k in kept)
    pid = math.fsum(k[3] for k in kept) / len(kept)   # exactly rounded mean; pandas' pairwise sum agrees to ~1e-13
    return aln, err, pid


def parse_blast_tab(path) -> Tuple[int, int, float]:
    return parse_blast_rows(read_blast_tab(path))
