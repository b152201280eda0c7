"""CPU: the sketch mode's host side (pyani_amd/fastani.py, the mirror of pyani/fastani.py:59-66, 193-270) and its numpy oracle.

  * the result-file format is pinned on the reference's own fixture (tests/fixtures/fastani/ecoli_vs_shiga.fastani, a DATA file kept
    under tests/golden/fastani/): parse_fastani_file must return the reference test's ComparisonResult values
    (tests/test_fastani.py:45-51 of the reference), an empty file must raise, and write_fastani_file must round-trip;
  * the oracle (oracle/sketch_oracle.py) behaves as the definition says on hand-checkable inputs."""
import sys

import numpy as np
import pytest

from tests.conftest import GOLD, ROOT

sys.path.insert(0, str(ROOT / "oracle"))


def test_parse_fastani_file_reads_the_reference_fixture(tmp_path):
    from pyani_amd import fastani
    got = fastani.parse_fastani_file(GOLD / "fastani" / "ecoli_vs_shiga.fastani")
    assert got == fastani.ComparisonResult("ecoli.fna", "shiga.fna", 0.9766400000000001, 1322, 1547)      # the reference's expected tuple
    empty = tmp_path / "a_vs_b.fastani"
    empty.write_text("")
    with pytest.raises(fastani.PyaniFastANIException):
        fastani.parse_fastani_file(empty)
    res = fastani.ComparisonResult("q.fna", "r.fna", 0.9876, 12, 20)      # in-process order = parse order (query file first: the reference's quirk)
    p = fastani.write_fastani_file(tmp_path / "q_vs_r.fastani", "q.fna", "r.fna", res)
    back = fastani.parse_fastani_file(p)
    assert back[:2] == res[:2] and (back.matches, back.fragments) == (12, 20) and abs(back.ani - 0.9876) < 1e-12
    # the Comparison row of the reference's driver (subcmd_fastani.py:437-474), also for an empty result file
    row = fastani.comparison_row(back, "q.fna", "r.fna", 3000, 90_000)
    assert (row["query"], row["subject"], row["aln_length"], row["sim_errs"]) == ("q.fna", "r.fna", 36_000, 24_000) and abs(row["cov_query"] - 0.4) < 1e-12
    none = fastani.comparison_row(None, "q.fna", "r.fna", 3000, 90_000)
    assert (none["aln_length"], none["sim_errs"], none["identity"], none["cov_query"]) == (0, 0, 0.0, 0.0)
    assert fastani.write_fastani_file(tmp_path / "none.fastani", "q.fna", "r.fna", None).read_text() == ""


def test_result_matrices_keep_their_own_columns():
    from pyani_amd import fastani
    r = {("a", "b"): fastani.ComparisonResult("b.fna", "a.fna", 0.97, 30, 40), ("b", "a"): None, ("a", "a"): fastani.ComparisonResult("a.fna", "a.fna", 1.0, 40, 40)}
    m = fastani.result_matrices(["a", "b"], r)
    assert set(m) == {"identity", "matches", "fragments", "coverage"}
    assert m["identity"]["a"]["b"] == 0.97 and np.isnan(m["identity"]["b"]["a"]) and m["coverage"]["a"]["b"] == 0.75
    with pytest.raises(fastani.PyaniFastANIException):
        fastani.calculate_fastani_pairs(None, [0], [1], kmerSize=21)


def test_oracle_follows_the_definition():
    import sketch_oracle as so
    from pyani_amd import synth
    a, b, far = synth.genome(7, 4, 0, 60_000), synth.genome(7, 4, 2, 60_000), synth.genome(99, 4, 0, 60_000)
    sa, sb, sf = (so.genome_sketch(*g) for g in (a, b, far))
    assert sa[2] == sum((int(a[1][r + 1]) - int(a[1][r])) // 3000 for r in range(len(a[1]) - 1))      # fragments: whole 3 000-base pieces per record
    ani, matches, frags, status = so.sketch_pair(sa, sa)
    assert (ani, matches, frags, status) == (1.0, sa[2], sa[2], 0)                                      # a genome against itself
    ani, matches, frags, status = so.sketch_pair(sa, sb)                                                  # 0.1 % vs 2 % descendants of one ancestor
    assert status == 0 and matches == frags and 0.97 < ani < 0.985
    assert so.sketch_pair(sa, sf) == (0.0, 0, sa[2], 1)                                                   # unrelated: no result
    # an ambiguity symbol removes the 16 windows that contain it; lower case counts as upper case
    seq = np.frombuffer(b"ACGTTGCAAGCTTAGCCATG" * 200, dtype=np.uint8).copy()
    pos, km = so.record_kmers(seq)
    seq2 = seq.copy(); seq2[100] = ord("N")
    pos2, _ = so.record_kmers(seq2)
    assert len(pos) - len(pos2) == 16 and not set(range(85, 101)) & set(pos2.tolist())
    assert (so.record_kmers(np.frombuffer(bytes(seq).lower(), dtype=np.uint8))[1] == km).all()
    # canonical: a sequence and its reverse complement have the same k-mer set
    comp = bytes(seq)[::-1].translate(bytes.maketrans(b"ACGT", b"TGCA"))
    assert set(so.record_kmers(np.frombuffer(comp, dtype=np.uint8))[1].tolist()) == set(km.tolist())
