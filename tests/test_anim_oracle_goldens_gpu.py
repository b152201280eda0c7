"""The BENCHMARK workloads against the INDEPENDENT nucmer oracle, record for record.

tests/golden/anim_oracle_goldens.json.gz holds what oracle/nucmer_oracle.cpp — MUMmer 3.23's mummer -mum / mgaps / postnuc restated
with its own data structures, sharing no header with the product, pinned on the 43 nucmer runs the reference's tests hold — finds on
whole 5 Mb genomes of bench.py's own generator and seeds (tools/make_anim_oracle_goldens.py, run in the build container):
  c4_slice      BASELINE.json configs[3]: the 18 related ordered pairs of tests/test_config_slices_gpu.py
  c4_divergent  the same set's most divergent family members (substitution rates 0.05 ... 0.15 + 0.15), N runs, 3 records; three
                pairs with their .delta indel lists
  c3_family     BASELINE.json configs[2]: 8 ordered pairs of family 3
  c4_filter_stress  (round 5) C4's family 2 with ten OVERLAPPING rearrangements per genome (tests/stress_genomes.py), 20 ordered pairs:
                the set on which `delta-filter -1` has to choose (tests/test_anim_filter_oracle_gpu.py checks the filtered tuple)
The GPU (pg_anim_alignments_batch through the C ABI) must return exactly these records — reference record, query record, the four
coordinates, the error count — and, where listed, the indel offsets; pg_anim_pairs' tuple must equal pyani's parse_delta
(anim.py:292-411, restated in oracle/anim_oracle.py) of the oracle's records."""
import gzip
import json
import sys

import pytest

from tests.conftest import GOLD, ROOT

sys.path.insert(0, str(ROOT / "oracle"))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold():
    with gzip.open(GOLD / "anim_oracle_goldens.json.gz", "rt") as fh:
        return json.load(fh)


def golden_genome(S, g):
    """genome g of a golden set, regenerated from its seeds (sets marked `rearranged`: tests/stress_genomes.py on top of the generator)"""
    from pyani_amd import synth
    if S.get("rearranged"):
        from tests.stress_genomes import rearranged_benchmark_genome
        return rearranged_benchmark_genome(S["seed"], S["n"], g, S["L"])
    return synth.genome(S["seed"], S["n"], g, S["L"])


def _check_set(S, name):
    import anim_oracle
    from pyani_amd.engine import Engine
    pairs = S["pairs"]
    used = sorted({g for p in pairs for g in p[:2]})
    n_rec = 0
    with Engine(0) as eng:
        ids = {g: eng.add_genome(*golden_genome(S, g)) for g in used}
        q, s = [ids[p[0]] for p in pairs], [ids[p[1]] for p in pairs]
        off, recs, _, _ = eng.anim_alignments_batch(q, s)                       # the production path (pre-passes on)
        tup = eng.anim_pairs(q, s, filter_1to1=False)
        listed = [k for k, p in enumerate(pairs) if p[3] is not None]
        if listed:                                                               # the listing walk + GPU traceback
            off2, recs2, ioff, ind = eng.anim_alignments_batch([q[k] for k in listed], [s[k] for k in listed], with_indels=True)
    for k, (a, b, want, _) in enumerate(pairs):
        got = sorted([int(r["ref_rec"]), int(r["qry_rec"]), int(r["rs"]), int(r["re"]), int(r["qs"]), int(r["qe"]), int(r["errors"])]
                     for r in recs[int(off[k]):int(off[k + 1])])
        assert got == sorted(want), (name, a, b, len(got), len(want), [x for x in got if x not in want][:3], [x for x in want if x not in got][:3])
        n_rec += len(want)
        t = tup[k]
        if want:
            w = anim_oracle.parse_delta_records([anim_oracle.Aln(str(x[0]), str(x[1]), x[2], x[3], x[4], x[5], x[6], x[6], 0, ()) for x in want])
            assert (int(t["ref_aln_len"]), int(t["qry_aln_len"]), float(t["identity"]), int(t["sim_errors"]), int(t["n_alignments"])) == w + (len(want),), (name, a, b)
        else:
            assert int(t["n_alignments"]) == 0 and int(t["status"]) == 1, (name, a, b)      # no alignment: parse_delta's ZeroDivisionError
    for j, k in enumerate(listed):
        a, b, want, lists = pairs[k]
        got = {}
        for x in range(int(off2[j]), int(off2[j + 1])):
            r = recs2[x]
            got[(int(r["ref_rec"]), int(r["qry_rec"]), int(r["rs"]), int(r["re"]), int(r["qs"]), int(r["qe"]), int(r["errors"]))] = \
                [int(v) for v in ind[int(ioff[x]):int(ioff[x + 1])]]
        assert got == {tuple(w): l for w, l in zip(want, lists)}, (name, a, b, "indel lists")
    return n_rec


def test_c4_slice_records_equal_the_nucmer_oracle(gold):
    assert _check_set(gold["c4_slice"], "c4_slice") > 700


def test_c4_divergent_family_records_and_indel_lists_equal_the_nucmer_oracle(gold):
    assert _check_set(gold["c4_divergent"], "c4_divergent") > 900


def test_c3_family_records_equal_the_nucmer_oracle(gold):
    assert _check_set(gold["c3_family"], "c3_family") > 300


def test_c4_filter_stress_records_equal_the_nucmer_oracle(gold):
    assert _check_set(gold["c4_filter_stress"], "c4_filter_stress") > 500
