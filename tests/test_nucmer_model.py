"""pyani_amd.nucmer — the reference's .delta object model (pyani/nucmer.py:47-351) — on the real MUMmer files the reference's tests
hold: parsed files render back to the text they came from, equality follows the reference's rules (order-insensitive inside a
comparison, coordinates only), `identical()` also holds error counts and indel lists, and anim.read_delta(with_indels=True) gives
the lists the independent reader of the tests (oracle/anim_oracle.py) gives."""
import gzip
import io
import sys

from tests.conftest import GOLD, ROOT

sys.path.insert(0, str(ROOT / "oracle"))
import anim_oracle  # noqa: E402


def _files():
    return sorted((GOLD / "anim").glob("*/*.delta.gz"))[:8] + sorted((GOLD / "anim").glob("*/*.filter.gz"))[:4]


def test_delta_files_round_trip_through_the_object_model():
    from pyani_amd.nucmer import DeltaData
    for f in _files():
        text = gzip.open(f, "rt").read()
        d = DeltaData.from_file(f)
        assert d.program == "NUCMER" and len(d) >= 1
        assert str(d).split() == text.split(), f.name                      # same tokens in the same order
        again = DeltaData(f.name, io.StringIO(str(d)))
        assert again == d and again.identical(d, ordered=True)
        n_aln = sum(len(c) for c in d.comparisons)
        assert n_aln == len(anim_oracle.read_delta(f)[0])


def test_equality_rules_of_the_reference_and_the_strict_form():
    from pyani_amd.nucmer import DeltaData
    f = sorted((GOLD / "anim" / "blochmannia").glob("*.delta.gz"))[0]
    a, b = DeltaData.from_file(f), DeltaData.from_file(f)
    b.comparisons[0].alignments.reverse()                  # another MUMmer build's order
    assert a == b and a.identical(b)
    b.comparisons[0].alignments[0].errs += 1               # the reference's == does not look at error counts ...
    assert a == b and not a.identical(b)                   # ... identical() does
    b = DeltaData.from_file(f)
    b.comparisons[0].alignments[0].refend += 1
    assert a != b
    c = DeltaData.from_file(f)
    aln = next(x for comp in c.comparisons for x in comp.alignments if x.indel_offsets)
    aln.indels[0] = str(int(aln.indels[0]) + 1)
    assert a == c and not a.identical(c)


def test_read_delta_with_indel_lists_equals_the_independent_reader():
    from pyani_amd import anim
    for f in _files()[:6]:
        recs, lists = anim.read_delta(f, with_indels=True)
        want = anim_oracle.read_delta(f)[0]
        assert [r[2:] for r in recs] == [(a.rs, a.re, a.qs, a.qe, a.errors) for a in want]
        assert lists == [list(a.indels) for a in want]
        assert recs == anim.read_delta(f)


def test_str_has_no_trailing_newline_and_malformed_lines_raise():
    """As the reference's model (pyani/nucmer.py): str(DeltaData) is the joined lines without a final line separator, an alignment
    header that does not have its 7 fields raises (the reference unpacks them), and so does an indel line before any alignment."""
    import pytest
    from pyani_amd.nucmer import DeltaData
    good = "/a.fna /b.fna\nNUCMER\n>r q 100 90\n1 50 1 50 0 0 0\n5\n-3\n0\n"
    d = DeltaData("x", io.StringIO(good))
    assert str(d) == good.rstrip("\n").replace("\n", __import__("os").linesep) and not str(d).endswith("\n")
    for bad in ("/a /b\nNUCMER\n>r q 100 90\n1 50 1 50 0 0\n0\n",           # six fields
                "/a /b\nNUCMER\n>r q 100 90\n7\n1 50 1 50 0 0 0\n0\n",       # an indel offset before any alignment
                "/a /b\nNUCMER\n1 50 1 50 0 0 0\n0\n"):                      # an alignment before any header
        with pytest.raises(ValueError):
            DeltaData("x", io.StringIO(bad))
