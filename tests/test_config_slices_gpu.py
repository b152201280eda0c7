"""GPU parity on the benchmark configurations' OWN workloads (BASELINE.json configs[3] and [4]) — slices small enough for the CPU
statements to finish in the test, large enough to be the real thing (whole 5 Mb / 1-2 Mb genomes of the bench's own generator and
seeds, related and unrelated pairs):

  C4  1000 synthetic ~5 Mb genomes, seed 20250301 (bench.py default): genomes {0, 40, 80, 120} of ancestor 0 and {1, 41, 81} of
      ancestor 1 -> 18 related + 24 unrelated ordered pairs through pg_anim_pairs == oracle/anim_cpu.cpp (the scalar statement of
      MUMmer's algorithm), tuple for tuple.
  C5  500 synthetic genomes of 1-12 Mb, seed 20250302, 1020-nt fragment mode (bench.py --workload anib): genomes {0, 20, 40} and
      {1, 21} -> all 20 ordered pairs through pg_anib_pairs == oracle/anib_cpu.cpp (fragment statement incl. the word tier)."""
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from tests.conftest import ROOT

sys.path.insert(0, str(ROOT / "oracle"))

pytestmark = pytest.mark.gpu


def test_c4_slice_equals_cpu_statement():
    import anim_cpu
    from pyani_amd import synth
    from pyani_amd.engine import Engine
    n, L, seed = 1000, 5_000_000, 20250301
    K = (n + 24) // 25
    assert K == 40
    pick = [0, 40, 80, 120, 1, 41, 81]
    data = {g: synth.genome(seed, n, g, L) for g in pick}
    pairs = [(a, b) for a in pick for b in pick if a != b]
    with Engine(0) as eng:
        ids = {g: eng.add_genome(*data[g]) for g in pick}
        got = eng.anim_pairs([ids[a] for a, _ in pairs], [ids[b] for _, b in pairs])
    genomes = [data.get(g) for g in range(max(pick) + 1)]
    want, _ = anim_cpu.anim_cpu_pairs(genomes, [a for a, _ in pairs], [b for _, b in pairs], threads=0)
    bad = []
    for (a, b), g, w in zip(pairs, got, want):
        gt = (int(g["ref_aln_len"]), int(g["qry_aln_len"]), float(g["identity"]).hex(), int(g["sim_errors"]), int(g["n_alignments"]), int(g["status"]))
        wt = (int(w["ref_aln_len"]), int(w["qry_aln_len"]), float(w["identity"]).hex(), int(w["sim_errors"]), int(w["n_alignments"]), int(w["status"]))
        if gt != wt:
            bad.append(((a, b), gt, wt))
        assert (int(g["status"]) == 0) == (a % K == b % K), (a, b, int(g["status"]))      # related pairs align, unrelated ones do not
    assert not bad, f"{len(bad)} of {len(pairs)} pairs differ, first: {bad[0]}"


def _c5_length(g):
    return 1_000_000 + (g * 22_045) % 11_000_001          # SURVEY.md §8(d) set C5 (bench.py: c5_length)


def test_c5_slice_equals_cpu_statement():
    import anib_cpu
    from pyani_amd import synth
    from pyani_amd.engine import Engine
    n, seed = 500, 20250302
    pick = [0, 20, 40, 1, 21]
    data = {g: synth.genome(seed, n, g, _c5_length(g)) for g in pick}
    pairs = [(a, b) for a in pick for b in pick if a != b]
    with Engine(0) as eng:
        ids = {g: eng.add_genome(*data[g]) for g in pick}
        got = eng.anib_pairs([ids[a] for a, _ in pairs], [ids[b] for _, b in pairs])
    with ThreadPoolExecutor(len(pairs)) as ex:
        want = list(ex.map(lambda p: anib_cpu.reduce_rows(anib_cpu.anib_cpu_pair(data[p[0]], data[p[1]])), pairs))
    related = 0
    for (a, b), g, (aln, err, pid, kept) in zip(pairs, got, want):
        assert (int(g["aln_length"]), int(g["sim_errors"]), int(g["n_kept"])) == (aln, err, len(kept)), (a, b)
        assert abs(float(g["pid"]) - pid) < 1e-9, (a, b)
        related += int(g["n_kept"]) > 0
    assert related >= 8       # the 6 + 2 pairs inside the two families


def test_c5_slice_against_the_independent_blastn_oracle():
    """A C5 slice (1 - 12 Mb genomes of BASELINE.json configs[4]; one family at 99.3 / 97.5 / 94 / 88 / 83.5 % identity) against
    oracle/blastn_oracle.cpp — code that shares nothing with the product (the C5 test used to compare only with the
    product's own header built for the host).  Every row parse_blast_tab uses must EQUAL the oracle's on the pairs down to 88 %,
    and all but a handful at 83.5 % (measured: 2 267 of 2 268); the pair tuples follow."""
    import sys
    from tests.conftest import ROOT
    sys.path.insert(0, str(ROOT / "oracle"))
    sys.path.insert(0, str(ROOT / "tools"))
    import blastn_oracle
    import blastn_oracle_agreement as agreement
    from anib_product_vs_oracle import side_by_side, tuples
    from pyani_amd import synth
    from pyani_amd.engine import Engine
    n, seed = 500, 20250302
    pick = [0, 20, 40, 60, 80, 1, 21]
    data = {g: synth.genome(seed, n, g, _c5_length(g)) for g in pick}
    with Engine(0) as eng:
        ids = {g: eng.add_genome(*data[g]) for g in pick}
        for a, b, exact in ((0, 20, True), (21, 1, True), (0, 40, True), (0, 60, True), (80, 0, True), (60, 80, False)):
            rows = eng.anib_pair_rows(ids[a], ids[b])
            rec = eng.anib_pairs([ids[a]], [ids[b]])[0]
            up = agreement.used_rows(tuples(rows))
            uo = agreement.used_rows(tuples(blastn_oracle.blastn_pair(data[a], data[b])))
            rep = side_by_side(up, uo)
            assert rep["used_rows_other"] > 900, (a, b, rep)
            if exact:
                assert rep["identical"] == rep["used_rows_other"] == rep["used_rows_product"], (a, b, rep)
                aln, err, pid = rep["tuple_other"]
                assert (int(rec["aln_length"]), int(rec["sim_errors"]), int(rec["n_kept"])) == (aln, err, len(uo)) and abs(float(rec["pid"]) - pid) < 1e-9, (a, b)
            else:
                assert rep["identical_fraction"] >= 0.995 and rep["only_product"] + rep["only_other"] <= 2 and abs(rep["identity_pp_diff"]) < 0.005, (a, b, rep)
