"""GPU: the reference's ANIm concordance test (tests/test_concordance.py:168-203) — the three genomes of
tests/fixtures/concordance against JSpecies' published ANIm values, at the reference's tolerance of 0.1 percentage points.

Added after the round's GPU time was spent, so it has not run on a GPU yet (the file sorts last on purpose).  What it
expects was computed with the scalar HOST build of the same core (tools/anim_debug/anim_debug, exhaustive seeding), which
the GPU pipeline reproduces exactly on every fixture and synthetic set: 98.2803 / 98.2827 vs JSpecies 98.19, 84.1021 vs
84.11 / 84.09, 84.5383 vs 84.53 / 84.55."""
import csv

import pytest

from tests.conftest import GOLD

pytestmark = pytest.mark.gpu

TOLERANCE_ANIM = 0.1   # tests/test_concordance.py:157-159
HOST_STATEMENT = {     # identity per unordered pair from the host build; both directions lie within 3e-5 of these
    ("GCF_000011325.1_ASM1132v1_genomic", "GCF_002243555.1_ASM224355v1_genomic"): 0.98281,
    ("GCF_000011325.1_ASM1132v1_genomic", "GCF_000227605.2_ASM22760v2_genomic"): 0.84102,
    ("GCF_000227605.2_ASM22760v2_genomic", "GCF_002243555.1_ASM224355v1_genomic"): 0.84538,
}


def _jspecies_anim():
    rows = list(csv.reader(open(GOLD / "ref_targets" / "jspecies_output.tab"), delimiter="\t"))
    start = next(i for i, r in enumerate(rows) if r and r[0].strip() == "ANIm")
    names = [n[:-4] for n in rows[start + 1][1:] if n]
    want = {}
    for r in rows[start + 2: start + 2 + len(names)]:
        for s, v in zip(names, r[1:]):
            if v != "---":
                want[(r[0][:-4], s)] = float(v)
    return want


@pytest.mark.xfail(strict=False, reason="written after the round's GPU budget was spent: never run on a GPU yet; expected "
                                        "to pass (values from the host build of the same core) - drop this marker once seen")
def test_anim_concordance_with_jspecies(genome_dir):
    from pyani_amd import anim
    from pyani_amd.engine import Engine
    eng = Engine(0)
    try:
        res, lengths = anim.calculate_anim_pairs(list(genome_dir["concordance"].values()), engine=eng)
    finally:
        eng.close()
    want = _jspecies_anim()
    assert len(want) == 6 and set(res) == set(want)
    for pair, pid in want.items():
        got = res[pair][2]
        assert abs(100.0 * got - pid) <= TOLERANCE_ANIM, (pair, got, pid)
        assert abs(got - HOST_STATEMENT[tuple(sorted(pair))]) < 1e-4, (pair, got)
    results = anim.assemble_legacy_results(res, lengths)   # the matrix the reference test compares
    for (q, s), pid in want.items():
        assert abs(100.0 * float(results.percentage_identity.loc[q, s]) - pid) <= TOLERANCE_ANIM
