"""oracle/tetra_port.py — pure-Python port of pyani's TETRA path.   TEST INFRASTRUCTURE ONLY.

Why it exists: pyani's TETRA is pure Python (string slicing + dict increments, one core, no Pool:
scripts/average_nucleotide_identity.py:606-608), and the reference's files cannot travel to the GPU box.
This port does the same amount of the same kind of work per base, so bench.py's ``cpu_baseline`` leg
(kind "port") times it on the GPU box's host cores beside the HIP path.  tests/ also use it on tiny inputs.
Nothing under pyani_amd/ imports it.

Follows (read-only reference): pyani/tetra.py:78-139 (counts + Z), :143-153 (clean test), :158-194 (Pearson).
Semantics are those of CPython 3.10 (plain left-to-right ``sum``); pinned by tests/test_oracle_tetra.py
against vectors generated from the reference itself (tools/make_goldens.py).
"""
from collections import defaultdict
from math import sqrt

_ACGT = frozenset("ACGT")
# Biopython's DNA complement (Bio.Seq: ambiguous_dna_complement, plus U->A); other symbols unchanged
_COMPLEMENT = str.maketrans("ACGTUMRWSYKVHDBNacgtumrwsykvhdbn", "TGCAAKYWSRMBDHVNtgcaakywsrmbdhvn")


def read_fasta(path):
    """Yield (title, sequence) per FASTA record (whitespace inside the sequence removed)."""
    title, parts = None, []
    with open(path) as fh:
        for line in fh:
            if line[:1] == ">":
                if title is not None:
                    yield title, "".join(parts)
                title, parts = line[1:].rstrip(), []
            elif title is not None:
                parts.append(line.strip().replace(" ", "").replace("\r", ""))
    if title is not None:
        yield title, "".join(parts)


def is_clean(kmer):
    """True iff every symbol is one of A, C, G, T (tetra.py:143-153)."""
    return not (set(kmer) - _ACGT)


def _scan_strand(s, di, tri, tet):
    # windows 0 .. len-5 get all three k-mers; the tail only di/tri (tetra.py:106-116)
    for i in range(len(s[:-4])):
        di[s[i:i + 2]] += 1
        tri[s[i:i + 3]] += 1
        tet[s[i:i + 4]] += 1
    tri[s[-4:-1]] += 1
    tri[s[-3:]] += 1
    di[s[-4:-2]] += 1
    di[s[-3:-1]] += 1
    di[s[-2:]] += 1


def count_kmers(records):
    """records: iterable of sequence strings -> (di, tri, tet) dicts over both strands."""
    di, tri, tet = defaultdict(int), defaultdict(int), defaultdict(int)
    for seq in records:
        fwd = seq.upper()
        rev = seq.translate(_COMPLEMENT)[::-1].upper()
        _scan_strand(fwd, di, tri, tet)
        _scan_strand(rev, di, tri, tet)
    return di, tri, tet


def zscores_from_counts(di, tri, tet):
    """Teeling (2004) Z-score per observed clean tetranucleotide, in first-observation order."""
    z = {}
    for t in [k for k in tet if is_clean(k)]:
        left, right, mid = tri[t[:3]], tri[t[1:]], di[t[1:3]]
        expected = 1.0 * left * right / mid
        sd = sqrt(expected * (mid - left) * (mid - right) / (mid * mid))
        try:
            z[t] = (tet[t] - expected) / sd
        except ZeroDivisionError:
            z[t] = 1 / (mid * mid)
    return z


def tetra_zscore_file(path):
    """calculate_tetra_zscore(Path) equivalent."""
    return zscores_from_counts(*count_kmers(seq for _, seq in read_fasta(path)))


def correlations(tetra_z):
    """calculate_correlations equivalent; returns (sorted labels, dict-of-dicts matrix)."""
    orgs = sorted(tetra_z)
    out = {a: {b: 1.0 for b in orgs} for a in orgs}
    for n, a in enumerate(orgs[:-1]):
        for b in orgs[n + 1:]:
            keys = sorted(tetra_z[a])
            if keys != sorted(tetra_z[b]):
                raise AssertionError()
            za = [tetra_z[a][k] for k in keys]
            zb = [tetra_z[b][k] for k in keys]
            ma, mb = sum(za) / len(za), sum(zb) / len(zb)
            da = [v - ma for v in za]
            db = [v - mb for v in zb]
            num = sum([da[i] * db[i] for i in range(len(da))])
            ssa, ssb = sum([v * v for v in da]), sum([v * v for v in db])
            out[a][b] = out[b][a] = num / sqrt(ssa * ssb)
    return orgs, out
