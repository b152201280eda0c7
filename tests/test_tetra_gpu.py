"""GPU parity tests for the TETRA path — everything goes through the C ABI (include/pyani_gpu.h).

Bars: integer counts EQUAL to the oracle / goldens; Z-scores and correlations BIT-equal (float.hex()).
"""
import json

import numpy as np
import pytest

from tests import oracle_bind
from tests.conftest import GOLD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from pyani_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _hexes(a):
    return [float(x).hex() for x in np.asarray(a).ravel()]


def _assert_bits(a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    assert a.shape == b.shape
    same = a.view(np.uint64) == b.view(np.uint64)
    assert same.all(), f"{(~same).sum()} of {same.size} doubles differ, first at {np.argwhere(~same)[0]}"


def _gold_arrays(gold):
    z = np.zeros(256)
    p = np.zeros(256, dtype=np.uint8)
    from pyani_amd.tetra import TETRAMERS
    for i, t in enumerate(TETRAMERS):
        if t in gold["z"]:
            z[i] = float.fromhex(gold["z"][t])
            p[i] = 1
    return z, p


@pytest.mark.parametrize("group", ["edge", "blochmannia", "caulobacter", "concordance"])
def test_counts_and_z_match_reference_goldens(eng, genome_dir, goldens, group):
    eng.clear_genomes()
    stems = list(genome_dir[group])
    ids = [eng.add_fasta(genome_dir[group][s])[0] for s in stems]
    c2, c3, c4 = eng.tetra_counts(ids)
    z, present, _ = eng.tetra_matrix(ids, want_corr=False)
    for k, s in enumerate(stems):
        gold = goldens[f"{group}/{s}"]
        assert c2[k].tolist() == gold["c2"], s
        assert c3[k].tolist() == gold["c3"], s
        assert c4[k].tolist() == gold["c4"], s
        gz, gp = _gold_arrays(gold)
        assert present[k].tolist() == gp.tolist(), s
        _assert_bits(z[k], gz)
    # the standalone Z entry point (host counts in) gives the same bits
    z2, p2 = eng.tetra_zscores_from_counts(c2, c3, c4)
    _assert_bits(z2, z)
    assert (p2 == present).all()


@pytest.mark.parametrize("group", ["blochmannia", "caulobacter", "concordance", "synthCI"])
def test_correlation_matrix_bit_exact(eng, genome_dir, synth_ci_dir, goldens, group):
    eng.clear_genomes()
    gold = goldens[f"{group}/__corr__"]
    paths = synth_ci_dir if group == "synthCI" else genome_dir[group]
    ids = [eng.add_fasta(paths[label])[0] for label in gold["labels"]]
    z, present, corr = eng.tetra_matrix(ids)
    want = np.array([[float.fromhex(v) for v in row] for row in gold["matrix"]])
    _assert_bits(corr, want)
    _assert_bits(eng.tetra_corr(z, present), want)   # host-Z entry point


def test_partial_keyset_and_errors(eng, genome_dir, goldens):
    from pyani_amd import _lib
    eng.clear_genomes()
    gold = goldens["edge/__corr_acg__"]
    ids = [eng.add_fasta(genome_dir["edge"][label])[0] for label in gold["labels"]]
    z, present, corr = eng.tetra_matrix(ids)
    assert 0 < present[0].sum() < 256
    _assert_bits(corr, np.array([[float.fromhex(v) for v in row] for row in gold["matrix"]]))
    # different key sets -> PG_E_KEYSET (AssertionError in the reference)
    other = eng.add_fasta(genome_dir["edge"]["e01_tiny_records"])[0]
    with pytest.raises(_lib.PyaniGpuError) as ei:
        eng.tetra_matrix([ids[0], other])
    assert ei.value.code == _lib.PG_E_KEYSET
    # empty key sets -> PG_E_EMPTY (ZeroDivisionError in the reference)
    with pytest.raises(_lib.PyaniGpuError) as ei:
        eng.tetra_corr(np.zeros((2, 256)), np.zeros((2, 256), dtype=np.uint8))
    assert ei.value.code == _lib.PG_E_EMPTY
    # RNA symbol is refused loudly
    with pytest.raises(_lib.PyaniGpuError) as ei:
        eng.add_genome(np.frombuffer(b"ACGUACGT", dtype=np.uint8), [0, 8])
    assert ei.value.code == _lib.PG_E_RNA


def test_reference_own_targets(eng, genome_dir):
    """zscore.json exact dict equality (reference tests/test_tetra.py:79-84) + the published correlation cell."""
    from pyani_amd import tetra
    with open(GOLD / "ref_targets" / "tetra_zscore_NC_002696.json") as fh:
        target = json.load(fh)
    eng.clear_genomes()
    assert tetra.calculate_tetra_zscore(genome_dir["caulobacter"]["NC_002696"], engine=eng) == target
    df = tetra.calculate_tetra(list(genome_dir["caulobacter"].values()), engine=eng)
    assert repr(float(df.loc["NC_002696", "NC_011916"])) == "0.9999899853711502"
    assert list(df.index) == ["NC_002696", "NC_010338", "NC_011916", "NC_014100"] and df.loc["NC_002696", "NC_002696"] == 1.0
    # the reference's whole committed 4 x 4 table (tests/target_TETRA_output/TETRA_correlations.tab), and the file itself
    out = GOLD.parent.parent / "gpurun_out" / "TETRA_correlations_caulobacter.tab"
    out.parent.mkdir(exist_ok=True)
    tetra.write_correlations_tab(df, out)
    assert out.read_text() == (GOLD / "ref_targets" / "TETRA_correlations_caulobacter_4x4.tab").read_text()


def test_module_api_mirrors_reference(eng, genome_dir, goldens, tmp_path):
    from pyani_amd import tetra
    eng.clear_genomes()
    files = list(genome_dir["blochmannia"].values())[:3]
    zs = tetra.calculate_tetra_zscores(files, engine=eng)
    assert sorted(zs) == sorted(f.stem for f in files)
    for f in files:
        assert {k: v.hex() for k, v in zs[f.stem].items()} == goldens[f"blochmannia/{f.stem}"]["z"]
    df = tetra.calculate_correlations(zs, engine=eng)
    gold = goldens["blochmannia/__corr__"]
    for a in df.index:
        for b in df.columns:
            i, j = gold["labels"].index(a), gold["labels"].index(b)
            assert float(df.loc[a, b]).hex() == gold["matrix"][i][j]
    with pytest.raises(AssertionError):
        tetra.calculate_correlations({"a": {"AAAA": 1.0}, "b": {"CCCC": 1.0}}, engine=eng)
    assert tetra.tetra_clean("ACGT") and not tetra.tetra_clean("ACGN") and not tetra.tetra_clean("acgt")
    out = tmp_path / "TETRA_correlations.tab"
    tetra.write_correlations_tab(df, out)
    first = out.read_text().splitlines()[0]
    assert first.startswith("\t") and first.split("\t")[1:] == list(df.columns)


def test_synthetic_vs_oracle_ragged_batch(eng, oracle):
    """Seeded synthetic genomes of ragged sizes (incl. N runs, multi-record) vs the C oracle, batch order shuffled."""
    from pyani_amd import synth
    eng.clear_genomes()
    specs = [(20250228, 40, g, L) for g, L in [(0, 70_000), (9, 131_072), (19, 65_535), (29, 65_536), (39, 300_001), (5, 64)]]
    data = [synth.genome(*s) for s in specs]
    ids = [eng.add_genome(seq, off) for seq, off in data]
    order = [3, 0, 5, 1, 4, 2, 0]          # a genome may appear twice in a batch
    c2, c3, c4 = eng.tetra_counts([ids[k] for k in order])
    for row, k in enumerate(order):
        o2, o3, o4 = oracle.counts(*data[k])
        assert (c2[row] == o2).all() and (c3[row] == o3).all() and (c4[row] == o4).all(), specs[k]
    z, present, corr = eng.tetra_matrix(ids[:5])
    cs = [oracle.counts(*d) for d in data[:5]]
    oz, op = oracle.zscores(np.array([c[0] for c in cs]), np.array([c[1] for c in cs]), np.array([c[2] for c in cs]))
    _assert_bits(z, oz)
    rc, ocorr = oracle.corr(oz, op)
    assert rc == 0
    _assert_bits(corr, ocorr)


def test_empty_and_degenerate_inputs(eng, oracle):
    eng.clear_genomes()
    g_empty = eng.add_genome(np.zeros(0, dtype=np.uint8), [0])              # no records at all
    g_norec = eng.add_genome(np.zeros(0, dtype=np.uint8), [0, 0])           # one empty record
    g_alln = eng.add_genome(np.frombuffer(b"N" * 1000, dtype=np.uint8), [0, 1000])
    c2, c3, c4 = eng.tetra_counts([g_empty, g_norec, g_alln])
    assert not c2.any() and not c3.any() and not c4.any()
    assert eng.genome_length(g_alln) == (1000, 1)
    z, present, _ = eng.tetra_matrix([g_alln], want_corr=False)
    assert not present.any()
    assert eng.tetra_counts([])[2].shape == (0, 256)


def test_full_size_properties(eng):
    """BASELINE config sizes (5 Mb genomes): size-independent properties instead of an oracle run.
    sum(c2) = 2*(clean dinucleotide windows), reverse-complement symmetry c_k[x] == c_k[rc(x)], c3/c2 marginals,
    and the Pearson matrix is symmetric with a unit diagonal and |r| <= 1."""
    from pyani_amd import synth
    eng.clear_genomes()
    n, L = 6, 5_000_000
    data = [synth.genome(20250228, 200, g, L) for g in (0, 9, 50, 101, 150, 199)]
    ids = [eng.add_genome(s, o) for s, o in data]
    c2, c3, c4 = eng.tetra_counts(ids)

    def rc(x, k):
        c, r = (4 ** k - 1) - x, 0
        for _ in range(k):
            r, c = r * 4 + (c & 3), c >> 2
        return r
    for k, c in ((2, c2), (3, c3)):
        perm = [rc(x, k) for x in range(4 ** k)]
        assert (c[:, perm] == c).all()
    # c4 is symmetric only up to the reference's quirk: <= 2 uncounted windows per record (tetra.py:106)
    perm4 = [rc(x, 4) for x in range(256)]
    asym = np.abs(c4.astype(np.int64) - c4[:, perm4].astype(np.int64)).sum(1)
    assert (asym <= 4 * np.array([len(o) - 1 for _, o in data])).all()
    assert (c4.reshape(n, 64, 4).sum(2) <= c3).all()
    for row, (seq, off) in enumerate(data):
        clean = np.isin(seq, np.frombuffer(b"ACGT", dtype=np.uint8))
        pair = clean[:-1] & clean[1:]
        for b in off[1:-1]:
            pair[int(b) - 1] = False          # windows never cross record boundaries
        assert int(c2[row].sum()) == 2 * int(pair.sum())
        assert (c3[row].reshape(16, 4).sum(1) <= c2[row]).all()
    z, present, corr = eng.tetra_matrix(ids)
    assert present.all()
    assert (corr == corr.T).all() and (np.diag(corr) == 1.0).all() and (np.abs(corr) <= 1.0).all()


def test_whole_c2_batch_equals_the_c_oracle(eng, oracle):
    """BASELINE.json configs[1] at FULL size (C2 ran only as a 6-genome slice under -m gpu): all 200 synthetic
    ~5 Mb genomes of set C2 in ONE pg_tetra_matrix call — k-mer counts, Z-scores and the 200 x 200 Pearson matrix bit for bit equal
    to oracle/tetra_oracle.c (the C restatement of pyani/tetra.py:78-194, pinned on the reference's own targets in
    tests/test_oracle_tetra.py).  The oracle counts on the host's threads (1 GB of sequence: seconds)."""
    from concurrent.futures import ThreadPoolExecutor
    from pyani_amd import synth
    cfg = synth.SETS["C2"]
    n = cfg["n"]
    assert n == 200 and cfg["L"] == 5_000_000
    eng.clear_genomes()
    c2o, c3o, c4o = (np.zeros((n, k), dtype=np.uint64) for k in (16, 64, 256))
    ids = []
    for lo in range(0, n, 25):                                   # 25 genomes (125 MB of text) on the host at a time
        part = [synth.genome(cfg["seed"], n, g, cfg["L"]) for g in range(lo, lo + 25)]
        ids += [eng.add_genome(s_, o_) for s_, o_ in part]
        with ThreadPoolExecutor(8) as ex:                        # (ctypes releases the GIL)
            for g, (a, b, c) in zip(range(lo, lo + 25), ex.map(lambda d: oracle.counts(*d), part)):
                c2o[g], c3o[g], c4o[g] = a, b, c
    c2, c3, c4 = eng.tetra_counts(ids)
    assert (c2 == c2o).all() and (c3 == c3o).all() and (c4 == c4o).all()
    z, present, corr = eng.tetra_matrix(ids)
    zo, po = oracle.zscores(c2o, c3o, c4o)
    assert (present == po).all() and (z.view(np.uint64) == zo.view(np.uint64)).all()
    rc, co = oracle.corr(zo, po)
    assert rc == 0 and corr.shape == (200, 200) and (corr.view(np.uint64) == co.view(np.uint64)).all()
    eng.clear_genomes()


def test_batch_ingest_equals_per_file_ingest(eng, genome_dir):
    """pg_add_fasta_batch (multithreaded read + parse + pack): same ids order, lengths, record counts and counts as the
    per-file path; a missing file fails the whole call and adds nothing."""
    from pyani_amd._lib import PyaniGpuError
    files = list(genome_dir["blochmannia"].values()) + list(genome_dir["edge"].values())
    eng.clear_genomes()
    one = [eng.add_fasta(f) for f in files]
    c_one = eng.tetra_counts([g for g, _, _ in one])
    eng.clear_genomes()
    many = eng.add_fasta_batch(files, threads=5)
    assert [(t, r) for _, t, r in many] == [(t, r) for _, t, r in one]
    assert [g for g, _, _ in many] == list(range(len(files)))
    c_many = eng.tetra_counts([g for g, _, _ in many])
    for a, b in zip(c_one, c_many):
        assert np.array_equal(a, b)
    n0 = eng.genome_count()
    with pytest.raises(PyaniGpuError):
        eng.add_fasta_batch([files[0], "/nonexistent/x.fna"])
    assert eng.genome_count() == n0
