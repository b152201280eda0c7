"""GPU: the reference's concordance tests (tests/test_concordance.py) — the three genomes of tests/fixtures/concordance
against JSpecies' published tables (tests/golden/ref_targets/jspecies_output.tab), at the reference's own tolerances:

  ANIm   test_anim_concordance   (:168-203)  six cells of percentage_identity x 100 within 0.1
  TETRA  test_tetra_concordance  (:290-302)  correlation matrix within 0.1
  ANIb   test_anib_concordance   (:207-254)  cells >= 90 within 0.2, cells below within 5

These genomes are the engine's HOLD-OUT: no constant of the search was chosen with them.  The ANIm tuples below
are those of the scalar host statement of MUMmer's algorithm (tools/anim_debug/anim_debug --exact), which the GPU has to
reproduce exactly (round 4: the two 98 % cells moved by ONE error each — 49912 -> 49911, 49840 -> 49839 — when the chain extraction
took mgaps' tie order, the earliest of equally scoring predecessors; the unfiltered records of both directions now equal the
independent nucmer oracle's, tests/test_nucmer_oracle.py::test_holdout_concordance_pair_equals_the_oracle); against JSpecies the 98 % pair gives 98.2580 / 98.2605 vs 98.19 (0.068 / 0.071 inside the 0.1).  Round 2's
banded64 extender gave 98.2907 / 98.2967 there — outside the tolerance, and 3.3e-4 away from what MUMmer's own algorithm
computes: the reason the extension stage was rebuilt as a restatement of postnuc instead of being fitted further."""
import csv

import pytest

from tests.conftest import GOLD

pytestmark = pytest.mark.gpu

TOLERANCE_ANIM, TOLERANCE_TETRA = 0.1, 0.1                                   # tests/test_concordance.py:156-165
TOLERANCE_ANIB_HI, TOLERANCE_ANIB_LO, THRESHOLD_ANIB_LO_HI = 0.2, 5, 90     # tests/test_concordance.py:118-153
A, B, C = ("GCF_000011325.1_ASM1132v1_genomic", "GCF_000227605.2_ASM22760v2_genomic", "GCF_002243555.1_ASM224355v1_genomic")
HOST_STATEMENT = {     # (nucmer reference, nucmer query) -> parse_delta tuple of the host statement
    (A, B): (37213, 37174, 0.8410206084396468, 5913),
    (B, A): (37174, 37213, 0.8410206084396468, 5913),
    (A, C): (2862738, 2864818, 0.9825808238292069, 49911),
    (C, A): (2864687, 2862679, 0.9826053754447123, 49839),
    (B, C): (38970, 39016, 0.8453825045520991, 6029),
    (C, B): (39016, 38970, 0.8453825045520991, 6029),
}


def _jspecies(block):
    rows = list(csv.reader(open(GOLD / "ref_targets" / "jspecies_output.tab"), delimiter="\t"))
    start = next(i for i, r in enumerate(rows) if r and r[0].strip() == block)
    names = [n[:-4] for n in rows[start + 1][1:] if n]
    want = {}
    for r in rows[start + 2: start + 2 + len(names)]:
        for s, v in zip(names, r[1:]):
            if v != "---":
                want[(r[0][:-4], s)] = float(v)
    return want


@pytest.fixture(scope="module")
def engine():
    from pyani_amd.engine import Engine
    eng = Engine(0)
    yield eng
    eng.close()


@pytest.fixture(scope="module")
def concordance_run(genome_dir, engine):
    from pyani_amd import anim
    return anim.calculate_anim_pairs(list(genome_dir["concordance"].values()), engine=engine)


def test_anim_concordance_tuples(concordance_run):
    res, lengths = concordance_run
    assert set(res) == set(HOST_STATEMENT)
    for pair, tup in HOST_STATEMENT.items():
        assert tuple(res[pair][:2]) == tup[:2] and res[pair][3] == tup[3], (pair, res[pair])
        assert res[pair][2] == tup[2], (pair, res[pair])


def test_anim_concordance_with_jspecies(concordance_run):
    """The reference's criterion, on the matrix the reference's test compares (legacy assembly: anim.py:415-497)."""
    from pyani_amd import anim
    res, lengths = concordance_run
    want = _jspecies("ANIm")
    assert len(want) == 6
    results = anim.assemble_legacy_results(res, lengths)
    for (q, s), pid in want.items():
        assert abs(100.0 * float(results.percentage_identity.loc[q, s]) - pid) <= TOLERANCE_ANIM, (q, s)
    # and every ordered comparison on its own (the v0.3 run matrices keep both directions)
    for (q, s), pid in want.items():
        assert abs(100.0 * res[(q, s)][2] - pid) <= TOLERANCE_ANIM, (q, s, res[(q, s)][2])


def test_tetra_concordance(genome_dir, engine):
    from pyani_amd import tetra
    files = list(genome_dir["concordance"].values())
    corr = tetra.calculate_correlations(tetra.calculate_tetra_zscores(files, engine=engine), engine=engine)
    want = _jspecies("Tetra")
    assert len(want) == 6
    for (q, s), r in want.items():
        assert abs(float(corr.loc[q, s]) - r) <= TOLERANCE_TETRA, (q, s, float(corr.loc[q, s]))


def test_anib_concordance(genome_dir, engine):
    from pyani_amd import anib
    res, lengths = anib.calculate_anib_pairs(list(genome_dir["concordance"].values()), engine=engine)
    pid = anib.process_blast_results(res, lengths)["percentage_identity"] * 100.0
    want = _jspecies("ANIb")
    assert len(want) == 6
    for (q, s), target in want.items():
        got = float(pid.loc[q, s])
        # the reference masks result and target by the threshold separately: a cell counts as "high" only where both are
        lo_r, hi_r = (got, 0.0) if got < THRESHOLD_ANIB_LO_HI else (0.0, got)
        lo_t, hi_t = (target, 0.0) if target < THRESHOLD_ANIB_LO_HI else (0.0, target)
        assert abs(lo_r - lo_t) <= TOLERANCE_ANIB_LO and abs(hi_r - hi_t) <= TOLERANCE_ANIB_HI, (q, s, got, target)
