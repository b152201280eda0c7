"""CPU-only: pin the oracle (C restatement + pure-Python port) to the reference.

Sources of truth, all committed under tests/golden/ (made by tools/make_goldens.py from the real reference):
  * the reference's own test targets (tests/fixtures/targets/tetra/zscore.json, TETRA_correlations.tab …);
  * counts / Z-scores / correlations obtained by importing pyani/tetra.py itself on real genomes,
    hand-made edge cases and the seeded synthetic CI set.
Integer counts must be equal; floats must be BIT-equal (compared through float.hex()).
"""
import json

import numpy as np
import pytest

from tests import oracle_bind
from tests.conftest import GOLD

import tetra_port  # noqa: E402  (oracle/ is put on sys.path by oracle_bind)


def _check_genome(oracle, path, gold):
    seq, off = oracle_bind.read_fasta_arrays(path)
    c2, c3, c4 = oracle.counts(seq, off)
    assert c2.tolist() == gold["c2"]
    assert c3.tolist() == gold["c3"]
    assert c4.tolist() == gold["c4"]
    z, present = oracle.zscores(c2, c3, c4)
    got = oracle_bind.z_dict(z[0], present[0])
    assert set(got) == set(gold["z"])
    for k, v in gold["z"].items():
        assert got[k].hex() == v, (k, got[k], float.fromhex(v))
    return z[0], present[0]


@pytest.mark.parametrize("group", ["caulobacter", "blochmannia", "concordance", "edge"])
def test_c_oracle_counts_and_zscores_bit_exact(oracle, genome_dir, goldens, group):
    for stem, path in genome_dir[group].items():
        _check_genome(oracle, path, goldens[f"{group}/{stem}"])


def test_c_oracle_synthetic_ci(oracle, synth_ci_dir, goldens):
    for stem, path in synth_ci_dir.items():
        _check_genome(oracle, path, goldens[f"synthCI/{stem}"])


def _corr_check(oracle, paths, goldens, prefix, corr_key):
    gold = goldens[corr_key]
    zs, ps = [], []
    for label in gold["labels"]:
        z, p = _check_genome(oracle, paths[label], goldens[f"{prefix}/{label}"])
        zs.append(z), ps.append(p)
    rc, m = oracle.corr(np.array(zs), np.array(ps))
    assert rc == 0
    for i, row in enumerate(gold["matrix"]):
        for j, v in enumerate(row):
            assert float(m[i, j]).hex() == v, (i, j)
    return gold["labels"], m


@pytest.mark.parametrize("group", ["caulobacter", "blochmannia", "concordance"])
def test_c_oracle_correlation_bit_exact(oracle, genome_dir, goldens, group):
    _corr_check(oracle, genome_dir[group], goldens, group, f"{group}/__corr__")


def test_c_oracle_correlation_partial_keyset(oracle, genome_dir, goldens):
    labels, _ = _corr_check(oracle, genome_dir["edge"], goldens, "edge", "edge/__corr_acg__")
    assert len(goldens[f"edge/{labels[0]}"]["z"]) < 256  # really exercises the < 256 keys path


def test_c_oracle_correlation_synth(oracle, synth_ci_dir, goldens):
    _corr_check(oracle, synth_ci_dir, goldens, "synthCI", "synthCI/__corr__")


def test_reference_own_targets(oracle, genome_dir):
    """The reference's committed goldens: zscore.json (exact dict equality, tests/test_tetra.py:79-84) and the
    whole 4 x 4 tests/target_TETRA_output/TETRA_correlations.tab (NC_010338 / NC_014100 are missing blobs upstream as FASTA;
    recovered from the reference's JSpecies BLAST databases by tools/make_goldens.py — every cell equal to the last digit)."""
    with open(GOLD / "ref_targets" / "tetra_zscore_NC_002696.json") as fh:
        target = json.load(fh)
    zs, ps = [], []
    stems = ("NC_002696", "NC_010338", "NC_011916", "NC_014100")
    for stem in stems:
        seq, off = oracle_bind.read_fasta_arrays(genome_dir["caulobacter"][stem])
        z, p = oracle.zscores(*oracle.counts(seq, off))
        zs.append(z[0]), ps.append(p[0])
    assert oracle_bind.z_dict(zs[0], ps[0]) == target
    rc, m = oracle.corr(np.array(zs), np.array(ps))
    assert rc == 0
    lines = (GOLD / "ref_targets" / "TETRA_correlations_caulobacter_4x4.tab").read_text().splitlines()
    header = lines[0].split("\t")[1:]
    assert header == list(stems)
    for i, a in enumerate(stems):
        row = lines[1 + i].split("\t")
        assert row[0] == a and [repr(float(x)) for x in m[i]] == row[1:], a
    assert repr(float(m[0, 2])) == "0.9999899853711502"


def test_blochmannia_legacy_target_within_1ulp(oracle, genome_dir, goldens):
    """tests/test_targets/legacy_scripts/TETRA_mpl/TETRA_correlations.tab: produced by an older run of the
    reference; SURVEY §8c: reproducible to 1 ulp only (24/36 cells exact) — checked at 1e-15."""
    lines = (GOLD / "ref_targets" / "TETRA_correlations_blochmannia_6x6.tab").read_text().splitlines()
    labels = lines[0].split("\t")[1:]
    zs, ps = [], []
    for label in labels:
        seq, off = oracle_bind.read_fasta_arrays(genome_dir["blochmannia"][label])
        z, p = oracle.zscores(*oracle.counts(seq, off))
        zs.append(z[0]), ps.append(p[0])
    rc, m = oracle.corr(np.array(zs), np.array(ps))
    assert rc == 0
    for i, line in enumerate(lines[1:]):
        vals = [float(v) for v in line.split("\t")[1:]]
        assert np.allclose(m[i], vals, rtol=0, atol=1e-15)


def test_python_port_matches_goldens(genome_dir, synth_ci_dir, goldens):
    """The pure-Python port (cpu_baseline 'port') on the small inputs: same dict, same order, same bits."""
    cases = [(f"edge/{s}", p) for s, p in genome_dir["edge"].items()]
    cases += [(f"synthCI/{s}", p) for s, p in list(synth_ci_dir.items())[:3]]
    zs = {}
    for key, path in cases:
        z = tetra_port.tetra_zscore_file(path)
        gold = goldens[key]
        assert list(z.keys()) == gold["order"]
        assert {k: v.hex() for k, v in z.items()} == gold["z"]
        zs[key] = z
    labels, m = tetra_port.correlations({k.split("/")[1]: zs[k] for k in zs if k.startswith("synthCI/")})
    gold = goldens["synthCI/__corr__"]
    for a in labels:
        for b in labels:
            i, j = gold["labels"].index(a), gold["labels"].index(b)
            assert m[a][b].hex() == gold["matrix"][i][j]


def test_oracle_error_paths(oracle):
    z = np.zeros((2, 256))
    p = np.zeros((2, 256), dtype=np.uint8)
    p[0, 3] = 1
    assert oracle.corr(z, p)[0] == -2          # different key sets -> AssertionError in the reference
    p[:] = 0
    assert oracle.corr(z, p)[0] == -3          # empty key set -> ZeroDivisionError in the reference
    with pytest.raises(AssertionError):
        tetra_port.correlations({"a": {"AAAA": 1.0}, "b": {"CCCC": 1.0}})
