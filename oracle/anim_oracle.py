"""oracle/anim_oracle.py — CPU restatement of pyani's ANIm *reduction*.   TEST INFRASTRUCTURE ONLY.

Covers what pyani itself computes from MUMmer output (the alignment search is MUMmer 3.23, third-party, not under
/root/reference — for that part parity is pinned only by the committed `.delta/.filter` fixtures):

  read_delta(path)            MUMmer delta grammar (pyani/nucmer.py:292-351, pyani/anim.py:355-394)
  parse_delta_records(recs)   (ref_aln_len, qry_aln_len, identity, sim_errors)        pyani/anim.py:292-411
  anim_matrices(...)          legacy matrix assembly of process_deltadir               pyani/anim.py:415-497,
                              pyani/pyani_tools.py:85-172 (ANIResults.add_*), incl. its overwrite order

Pinned by the reference's own known answers: tests/fixtures/anim/test.delta -> (4016947, 4017751,
0.9994621994447228, 2191) (tests/test_anim.py:96-100, tests/test_parsing.py:52-64) and
tests/fixtures/anim/dataframes/deltadir_result.csv (6 d.p.).  Nothing under pyani_amd/ imports this file.
"""
import gzip
from collections import defaultdict
from typing import Dict, List, NamedTuple, Tuple


class Aln(NamedTuple):
    ref_id: str
    qry_id: str
    rs: int
    re: int
    qs: int
    qe: int
    errors: int
    sim_errors: int
    stops: int
    indels: Tuple[int, ...]


def _open(path):
    return gzip.open(path, "rt") if str(path).endswith(".gz") else open(path, "r")


def read_delta(path) -> Tuple[List[Aln], Dict[str, int], Dict[str, int]]:
    """Return (alignments, ref sequence lengths, qry sequence lengths) of a .delta/.filter file."""
    alns: List[Aln] = []
    rlen: Dict[str, int] = {}
    qlen: Dict[str, int] = {}
    cur_ref = cur_qry = None
    header = None
    indels: List[int] = []
    with _open(path) as fh:
        for line in fh:
            f = line.strip().split()
            if not f or f[0] == "NUCMER":
                continue
            if f[0].startswith(">"):
                cur_ref, cur_qry = f[0][1:], f[1]
                rlen[cur_ref], qlen[cur_qry] = int(f[2]), int(f[3])
            elif len(f) == 7:
                header = [int(x) for x in f]
                indels = []
            elif len(f) == 1 and header is not None:
                v = int(f[0])
                if v == 0:
                    alns.append(Aln(cur_ref, cur_qry, *header, tuple(indels)))
                    header = None
                else:
                    indels.append(v)
    return alns, rlen, qlen


def _union_length(intervals: List[Tuple[int, int]]) -> int:
    """IntervalTree.from_tuples + merge_overlaps(strict=False) + sum(end - begin + 1)   (anim.py:399-409)."""
    total = 0
    cur_b = cur_e = None
    for b, e in sorted(intervals):
        if cur_b is None:
            cur_b, cur_e = b, e
        elif b <= cur_e:          # overlapping or touching half-open intervals are merged
            cur_e = max(cur_e, e)
        else:
            total += cur_e - cur_b + 1
            cur_b, cur_e = b, e
    if cur_b is not None:
        total += cur_e - cur_b + 1
    return total


def parse_delta_records(alns) -> Tuple[int, int, float, int]:
    """The reduction of pyani.anim.parse_delta over already-parsed alignment headers."""
    regions_ref, regions_qry = defaultdict(list), defaultdict(list)
    aligned, weighted, sim_error = 0, 0, 0
    for a in alns:
        regions_ref[a.ref_id].append(tuple(sorted((a.rs, a.re))))
        regions_qry[a.qry_id].append(tuple(sorted((a.qs, a.qe))))
        rl, ql = abs(a.re - a.rs) + 1, abs(a.qe - a.qs) + 1
        aligned += rl + ql
        sim_error += a.errors
        weighted += rl + ql - 2 * a.errors
    identity = weighted / aligned          # ZeroDivisionError when there are no alignments (anim.py:396)
    qaln = sum(_union_length(v) for v in regions_qry.values())
    raln = sum(_union_length(v) for v in regions_ref.values())
    return raln, qaln, identity, sim_error


def _ordinal(seq_id):
    """a sequence's place for tie-breaking: the number at the end of its id when there is one (the tests name records r0, r1, ... in
    file order; the goldens store ordinals), else the id itself"""
    import re
    m = re.search(r"(\d+)$", str(seq_id))
    return (0, int(m.group(1)), "") if m else (1, 0, str(seq_id))


def delta_filter_1to1(alns) -> List[bool]:
    """`delta-filter -1` (what pyani's delta_filter_wrapper.py:80-90 runs on every nucmer output): MUMmer 3.23's published 1-to-1
    mapping — an alignment survives iff it lies on the best weighted chain of its REFERENCE sequence and on the best weighted
    chain of its QUERY sequence (DeltaGraph_t::flagRLIS / flagQLIS + ScoreLocal).  Per sequence: alignments by start (equal
    starts: the higher own score first, then input order); score_i = max(own_i, max over earlier j of score_j + gain(i, j)) with
    own_i = trunc(len_i * idy_i^2), gain = trunc((len_i - overlap) * idy_i^2), integer scores, first best wins; the chain ending in
    the first maximal score is kept.  idy = 1 - 2 errors / (ref length + query length).  Independent of the engine's C++ (which
    runs the same rule on the GPU); pinned on every .delta / .filter pair the reference's tests hold (tests/test_anim_cpu.py)."""
    n = len(alns)
    keep = [0] * n
    for side in (0, 1):
        groups = defaultdict(list)
        for i, a in enumerate(alns):
            lo, hi = (min(a.rs, a.re), max(a.rs, a.re)) if side == 0 else (min(a.qs, a.qe), max(a.qs, a.qe))
            tot = (abs(a.re - a.rs) + 1) + (abs(a.qe - a.qs) + 1)
            idy = 1.0 - 2.0 * a.errors / tot if tot > 0 else 0.0
            length = hi - lo + 1
            own = int(length * (idy * idy))
            # equal start and equal score (two copies of a duplicated region): delta-filter's std::sort leaves them in an order MUMmer
            # does not define — here by the other sequence (its ordinal among the file's sequences: first appearance), the start there,
            # the strand, and only then the input order: a function of the alignments, not of how they were listed
            other = a.qry_id if side == 0 else a.ref_id
            o_lo = min(a.qs, a.qe) if side == 0 else min(a.rs, a.re)
            groups[a.ref_id if side == 0 else a.qry_id].append((lo, -own, i, hi, length, idy, own, (_ordinal(other), o_lo, 1 if a.qs > a.qe else 0)))
        for items in groups.values():
            items.sort(key=lambda t: (t[0], t[1], t[7], t[2]))
            score, frm = [], []
            best = -1
            for k, (lo, _, i, hi, length, idy, own, _) in enumerate(items):
                sc, fr = own, -1
                for kk in range(k):
                    lo_j, _, _, hi_j, len_j, _, _, _ = items[kk]
                    olap = max(0, hi_j - lo + 1)
                    if olap > 0 and (olap / length * 100.0 > 100.0 or olap / len_j * 100.0 > 100.0):
                        continue
                    cand = score[kk] + int((length - olap) * (idy * idy))
                    if cand > sc:
                        sc, fr = cand, kk
                score.append(sc); frm.append(fr)
                if best < 0 or sc > score[best]:
                    best = k
            k = best
            while k >= 0:
                keep[items[k][2]] |= 1 << side
                k = frm[k]
    return [f == 3 for f in keep]


def parse_delta(path) -> Tuple[int, int, float, int]:
    return parse_delta_records(read_delta(path)[0])


def anim_matrices(results: Dict[Tuple[str, str], Tuple[int, int, float, int]], org_lengths: Dict[str, int]):
    """process_deltadir's five matrices as dict-of-dicts [query][subject] (anim.py:438-497).  `results` maps
    (qname, sname) -> parse_delta tuple; files are visited in sorted path order "<q>/<q>_vs_<s>.filter", later
    files overwrite the mirrored cells written by earlier ones (pyani_tools.py:108-167)."""
    names = list(org_lengths)
    nan = float("nan")
    lengths = {a: {b: nan for b in names} for a in names}
    errors = {a: {b: 0.0 for b in names} for a in names}
    pid = {a: {b: 1.0 for b in names} for a in names}
    cov = {a: {b: 1.0 for b in names} for a in names}
    for org, length in org_lengths.items():
        lengths[org][org] = float(length)
    # sorted(Path) order (anim.py:438): Paths compare component by component, i.e. (directory, file name)
    for (q, s) in sorted(results, key=lambda k: (k[0], f"{k[0]}_vs_{k[1]}.filter")):
        raln, qaln, ident, err = results[(q, s)]
        qcov, scov = float(raln) / org_lengths[q], float(qaln) / org_lengths[s]
        lengths[q][s] = float(raln)
        if qaln:
            lengths[s][q] = float(qaln)
        errors[q][s] = errors[s][q] = float(err)
        pid[q][s] = ident
        cov[q][s] = qcov
        if scov:
            cov[s][q] = scov
    had = {a: {b: pid[a][b] * cov[a][b] for b in names} for a in names}
    return {"alignment_lengths": lengths, "similarity_errors": errors, "percentage_identity": pid,
            "alignment_coverage": cov, "hadamard": had}
