"""GPU tests of the ANIm path (through the C ABI).

  * the REDUCTION (parse_delta / delta-filter bookkeeping) has a pinned oracle -> integers equal, identity bit-equal;
  * the ALIGNMENT SEARCH is MUMmer 3.23's own algorithm (pga_postnuc.inc / pg_nucmer_core.h).
    MUMmer is absent from the reference tree, so it is pinned on the real nucmer / delta-filter output files the reference's
    tests hold: every alignment record (coordinates + error count), every delta-filter decision and every parse_delta tuple,
    bit for bit, on all pairs whose genomes are available (tests/test_anim_oos_gpu.py has the pairs recovered in round 2).
"""
import json

import numpy as np
import pytest

from tests.conftest import GOLD, ROOT

import sys
sys.path.insert(0, str(ROOT / "oracle"))
import anim_oracle  # noqa: E402

pytestmark = pytest.mark.gpu


# the genomes recovered in round 2 (tools/make_goldens.py) have their own module (tests/test_anim_oos_gpu.py: all 26 runs)
OUT_OF_SAMPLE = {"NC_010338", "NC_014100"}


@pytest.fixture(scope="module")
def eng():
    from pyani_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def gold():
    return json.loads((GOLD / "anim_goldens.json").read_text())["parse_delta"]


def test_reduction_bit_exact_on_all_fixture_files(eng, gold):
    """pg_anim_reduce == pyani.anim.parse_delta on all 55 real MUMmer files (incl. the reference's known answer)."""
    from pyani_amd import anim
    rels = sorted(gold)
    recs = [anim.read_delta(GOLD / "anim" / (r + ".gz")) for r in rels]
    out = eng.anim_reduce(recs, apply_filter=False)
    for rel, r in zip(rels, out):
        want = gold[rel]
        assert [int(r["ref_aln_len"]), int(r["qry_aln_len"]), int(r["sim_errors"])] == [want[0], want[1], want[3]], rel
        assert float(r["identity"]).hex() == float(want[2]).hex(), rel
    assert anim.parse_delta(GOLD / "anim" / "test.delta.gz", engine=eng) == (4016947, 4017751, 0.9994621994447228, 2191)
    with pytest.raises(ZeroDivisionError):
        anim._tuple(eng.anim_reduce([[]])[0])       # empty file: parse_delta raises ZeroDivisionError (anim.py:396)


def test_delta_filter_then_reduce_equals_the_filter_files(eng, gold):
    """delta -> (GPU 1-to-1 filter) -> reduction == the parse_delta tuple of MUMmer's own .filter file, on all 27 pairs that have
    both: aligned lengths, errors, identity to the last bit (the filter reproduces every keep / drop decision of delta-filter -1)."""
    from pyani_amd import anim
    rels = sorted(r for r in gold if r.endswith(".delta") and r.replace(".delta", ".filter") in gold)
    assert len(rels) == 27
    out = eng.anim_reduce([anim.read_delta(GOLD / "anim" / (r + ".gz")) for r in rels], apply_filter=True)
    for rel, r in zip(rels, out):
        want = gold[rel.replace(".delta", ".filter")]
        assert [int(r["ref_aln_len"]), int(r["qry_aln_len"]), float(r["identity"]).hex(), int(r["sim_errors"])] == \
               [want[0], want[1], float(want[2]).hex(), want[3]], rel


@pytest.fixture(scope="module")
def fixture_runs(eng, genome_dir, gold):
    eng.clear_genomes()
    ids = {}
    for grp in ("blochmannia", "caulobacter"):
        for stem, p in genome_dir[grp].items():
            ids[stem] = eng.add_fasta(p)[0]
    pairs = []
    for rel in sorted(gold):
        if rel.endswith(".filter") and "/" in rel:
            a, b = rel.split("/")[1][:-7].split("_vs_")
            if a in ids and b in ids and not ({a, b} & OUT_OF_SAMPLE):
                pairs.append((rel, a, b))
    res = eng.anim_pairs([ids[a] for _, a, _ in pairs], [ids[b] for _, _, b in pairs])
    return [(rel, a, b, r, gold[rel]) for (rel, a, b), r in zip(pairs, res)]


def test_alignment_search_vs_real_mummer_output(fixture_runs):
    """17 ordered pairs with both genomes and real nucmer+delta-filter output (15 Blochmannia pairs at 83-98 % identity,
    the near-identical Caulobacter pair in both directions).  BASELINE.json's bar is identity and coverage within 1e-4;
    the engine reproduces every one of these parse_delta tuples EXACTLY: aligned lengths, error counts, and the
    identity to the last bit (tools/anim_host_fixture_check.py: 505 of 505 .delta alignment records)."""
    assert len(fixture_runs) == 17
    for rel, a, b, r, want in fixture_runs:
        assert int(r["status"]) == 0, rel
        got = [int(r["ref_aln_len"]), int(r["qry_aln_len"]), float(r["identity"]).hex(), int(r["sim_errors"])]
        assert got == [want[0], want[1], float(want[2]).hex(), want[3]], rel


def test_module_api_and_matrices(eng, genome_dir):
    """calculate_anim_pairs + both matrix assemblies on three small genomes; legacy assembly == the oracle's."""
    from pyani_amd import anim
    eng.clear_genomes()
    files = list(genome_dir["blochmannia"].values())[3:6]
    res, lengths = anim.calculate_anim_pairs(files, engine=eng)
    assert len(res) == 6 and set(lengths) == {f.stem for f in files}
    legacy = anim.assemble_legacy_results(res, lengths)
    want = anim_oracle.anim_matrices(res, lengths)
    for df, key in ((legacy.alignment_lengths, "alignment_lengths"), (legacy.percentage_identity, "percentage_identity"),
                    (legacy.alignment_coverage, "alignment_coverage"), (legacy.similarity_errors, "similarity_errors"),
                    (legacy.hadamard, "hadamard")):
        for q in lengths:
            for s in lengths:
                assert float(df.loc[q, s]) == want[key][q][s], (key, q, s)
    run = anim.assemble_run_matrices(res, lengths)
    q, s = sorted(lengths)[:2]
    assert run["identity"].loc[q, s] == res[(q, s)][2] and run["identity"].loc[q, q] == 1.0
    assert run["coverage"].loc[q, s] == res[(q, s)][0] / lengths[q]
    assert run["aln_lengths"].loc[q, q] == lengths[q] and run["sim_errors"].loc[q, q] == 0.0
    assert run["hadamard"].loc[q, s] == res[(q, s)][2] * (res[(q, s)][0] / lengths[q])
    assert [stem for _, stem in legacy.data][0] == "ANIm_alignment_lengths"


def test_unrelated_genomes_give_no_alignment(eng):
    """Random unrelated sequences: nucmer would write an empty .filter and parse_delta raises ZeroDivisionError."""
    from pyani_amd import anim, synth
    eng.clear_genomes()
    a = eng.add_genome(*synth.genome(1, 50, 0, 60_000))
    b = eng.add_genome(*synth.genome(1, 50, 1, 60_000))       # different ancestors (K = 2)
    r = eng.anim_pairs([a], [b])[0]
    assert int(r["status"]) == 1 and int(r["n_alignments"]) == 0
    with pytest.raises(ZeroDivisionError):
        anim._tuple(r)
    c = eng.add_genome(*synth.genome(1, 50, 2, 60_000))       # same ancestor as genome 0, 0.1 % divergence
    r2 = eng.anim_pairs([a], [c])[0]
    assert int(r2["status"]) == 0 and float(r2["identity"]) > 0.99 and int(r2["ref_aln_len"]) > 50_000


def test_gpu_pipeline_equals_scalar_host_statement_on_synthetic_pairs(eng):
    """Sampled-seed hashing + wave-cooperative clustering + the extension stage on the GPU give exactly what the
    exhaustive-seed scalar host build of the same statement gave (tools/make_anim_synth_host.py), for every divergence level of
    the synthetic generator, with and without the 1-to-1 filter (MUMmer's postnuc algorithm: wave engine == pgn::ScalarEngine)."""
    from pyani_amd import synth
    fx = json.loads((GOLD / "anim_synth_host.json").read_text())
    n, L, seed = fx["n"], fx["length"], fx["seed"]
    eng.clear_genomes()
    ids = [eng.add_genome(*synth.genome(seed, n, g, L)) for g in range(n)]
    eng.upload()
    pairs = [(a, b) for a in range(n) for b in range(n) if a != b]
    for mode, filt in (("filter", True), ("nofilter", False)):
        res = eng.anim_pairs([ids[a] for a, _ in pairs], [ids[b] for _, b in pairs], filter_1to1=filt)
        bad = []
        for (a, b), r in zip(pairs, res):
            want = fx["pairs"][f"{a},{b},{mode}"]
            got = [int(r["ref_aln_len"]), int(r["qry_aln_len"]), float(r["identity"]).hex(), int(r["sim_errors"]),
                   int(r["n_alignments"])]
            if got != want:
                bad.append((a, b, got, want))
        assert not bad, f"{mode}: {len(bad)} of {len(pairs)} pairs differ, first: {bad[0]}"


def _delta_data(path):
    """What pyani.nucmer.DeltaData equality looks at (nucmer.py:47-351): (program, [(header 4-tuple, sorted coordinate
    4-tuples)] in file order)."""
    import gzip
    opener = gzip.open if str(path).endswith(".gz") else open
    with opener(path, "rt") as fh:
        fh.readline()
        program = fh.readline().strip()
        comparisons = []
        for line in fh:
            t = line.split()
            if line.startswith(">"):
                comparisons.append(((t[0][1:], t[1], int(t[2]), int(t[3])), []))
            elif len(t) > 1:
                comparisons[-1][1].append(tuple(int(x) for x in t[:4]))
    return program, [(h, sorted(a)) for h, a in comparisons]


def test_alignment_records_equal_mummer_delta_and_filter_files(eng, genome_dir, gold, tmp_path):
    """Alignment level, on the GPU, through pg_anim_pair_alignments: for every fixture pair with both genomes the
    engine's records ARE the records of nucmer's .delta file (coordinates and error counts, 505 in all) and the ones
    flagged kept are exactly delta-filter -1's .filter file; the recovery file written from them gives pyani's own
    parse_delta tuple (checked with the oracle restatement of anim.py:292-411)."""
    from pyani_amd import anim
    eng.clear_genomes()
    ids, paths = {}, {}
    for grp in ("blochmannia", "caulobacter"):
        for stem, p in genome_dir[grp].items():
            ids[stem] = eng.add_fasta(p)[0]
            paths[stem] = p
    n_records = n_pairs = 0
    for grp in ("blochmannia", "caulobacter"):
        for f in sorted((GOLD / "anim" / grp).glob("*.delta.gz")):
            a, b = f.name[:-len(".delta.gz")].split("_vs_")
            if a not in ids or b not in ids or ({a, b} & OUT_OF_SAMPLE):
                continue
            rrec, qrec = anim.fasta_records(paths[a]), anim.fasta_records(paths[b])
            al = eng.anim_pair_alignments(ids[a], ids[b])
            mine = {(rrec[int(x["ref_rec"])][0], qrec[int(x["qry_rec"])][0], int(x["rs"]), int(x["re"]), int(x["qs"]), int(x["qe"]),
                     int(x["errors"])) for x in al}
            kept = {(rrec[int(x["ref_rec"])][0], qrec[int(x["qry_rec"])][0], int(x["rs"]), int(x["re"]), int(x["qs"]), int(x["qe"]),
                     int(x["errors"])) for x in al if int(x["kept"]) == 3}
            want = {(x.ref_id, x.qry_id, x.rs, x.re, x.qs, x.qe, x.errors) for x in anim_oracle.read_delta(f)[0]}
            want_f = {(x.ref_id, x.qry_id, x.rs, x.re, x.qs, x.qe, x.errors)
                      for x in anim_oracle.read_delta(str(f).replace(".delta.gz", ".filter.gz"))[0]}
            assert mine == want, (f.name, sorted(mine ^ want)[:4])
            assert kept == want_f, (f.name, sorted(kept ^ want_f)[:4])
            out = tmp_path / f"{a}_vs_{b}.filter"
            assert anim.write_delta(out, paths[a], paths[b], al, filtered=True) == len(want_f)
            assert list(anim_oracle.parse_delta(out)) == gold[f"{grp}/{a}_vs_{b}.filter"]
            # the reference's own file comparison (tests/tools.py:79-97 assertNucmerEqual = nucmer.DeltaData.__eq__,
            # nucmer.py:114-122, 154-160, 277-282): program line, per-comparison headers with the sequence lengths, and the
            # sorted alignment coordinates — our file against MUMmer's
            assert _delta_data(out) == _delta_data(str(f).replace(".delta.gz", ".filter.gz")), f.name
            n_records += len(want)
            n_pairs += 1
    assert (n_pairs, n_records) == (17, 505)


def test_batch_split_does_not_change_results_and_edge_inputs(eng):
    """The internal batching (pairs per launch, match budget) must be invisible: a call split into many tiny launches —
    including launches that stop early because the match budget is exhausted — returns the same records.  Plus the edge
    inputs: a genome against itself, genomes too short to seed, all-N and empty genomes, unknown ids; and --maxmatch."""
    from pyani_amd import synth
    from pyani_amd._lib import PyaniGpuError
    eng.clear_genomes()
    n, L = 6, 120_000
    ids = [eng.add_genome(*synth.genome(7, n, g, L)) for g in range(n)]
    tiny = eng.add_genome(np.frombuffer(b"ACGTACGTACGTAC", dtype=np.uint8), np.array([0, 14], dtype=np.uint64))
    alln = eng.add_genome(np.frombuffer(b"N" * 5000, dtype=np.uint8), np.array([0, 5000], dtype=np.uint64))
    empty = eng.add_genome(np.zeros(0, dtype=np.uint8), np.array([0, 0], dtype=np.uint64))
    eng.upload()
    pairs = [(a, b) for a in ids for b in ids if a != b] + [(ids[0], ids[0]), (ids[0], tiny), (tiny, ids[0]), (alln, ids[1]),
                                                             (ids[1], alln), (empty, ids[2]), (ids[2], empty), (tiny, alln)]
    ra, qa = [a for a, _ in pairs], [b for _, b in pairs]
    eng.anim_set_batch_budget(16384, 150 << 20)
    ref = eng.anim_pairs(ra, qa)
    for max_pairs, max_matches in ((3, 150 << 20), (16384, 4096), (2, 2048)):
        eng.anim_set_batch_budget(max_pairs, max_matches)
        assert eng.anim_pairs(ra, qa).tobytes() == ref.tobytes(), (max_pairs, max_matches)
    eng.anim_set_batch_budget(16384, 150 << 20)
    by = {p: r for p, r in zip(pairs, ref)}
    self_hit = by[(ids[0], ids[0])]
    assert int(self_hit["status"]) == 0 and float(self_hit["identity"]) == 1.0 and int(self_hit["sim_errors"]) == 0
    assert int(self_hit["ref_aln_len"]) == int(self_hit["qry_aln_len"]) >= L - 100
    for p in pairs[-7:]:
        assert int(by[p]["status"]) == 1 and int(by[p]["n_alignments"]) == 0, p    # parse_delta would raise ZeroDivisionError
    with pytest.raises(PyaniGpuError):
        eng.anim_pairs([ids[0]], [999])
    assert len(eng.anim_pairs([], [])) == 0
    # --maxmatch (every maximal match, not only unique ones): on repeat-free synthetic genomes it must agree with --mum
    mm = eng.anim_pairs(ra[:30], qa[:30], maxmatch=True)
    assert mm.tobytes() == ref[:30].tobytes()


def test_gap_forms_and_extenders_of_the_postnuc_stage_agree(monkeypatch):
    """The match-to-match alignments of a cluster run on one LANE each when their rectangle is small enough to rule out trimming
    and the break rule (pga_postnuc.inc, PN_SMALL) and on the wave engine otherwise: switching the lane form off
    (PYANI_ANIM_GAP_LANES=0, read when a context is created) must not change a single result."""
    from pyani_amd import synth
    from pyani_amd.engine import Engine
    n, L = 10, 250_000
    data = [synth.genome(5, n, g, L) for g in range(n)]
    out = {}
    for lanes in ("1", "0"):
        monkeypatch.setenv("PYANI_ANIM_GAP_LANES", lanes)
        with Engine(0) as e:
            ids = [e.add_genome(*d) for d in data]
            pairs = [(a, b) for a in ids for b in ids if a != b]
            out[lanes] = e.anim_pairs([a for a, _ in pairs], [b for _, b in pairs])
    assert (out["1"]["status"] == 0).sum() >= 40
    assert out["1"].tobytes() == out["0"].tobytes()
    # the backward searches run ahead of the units' walks (rehearsal + one wave per predicted search; results are taken only when
    # the walk repeats the predicted arguments) or inside them (PYANI_ANIM_BWD_AHEAD=0): not a single result may differ
    monkeypatch.setenv("PYANI_ANIM_GAP_LANES", "1")
    monkeypatch.setenv("PYANI_ANIM_BWD_AHEAD", "0")
    with Engine(0) as e:
        ids = [e.add_genome(*d) for d in data]
        pairs = [(a, b) for a in ids for b in ids if a != b]
        inside = e.anim_pairs([a for a, _ in pairs], [b for _, b in pairs])
    assert inside.tobytes() == out["1"].tobytes()
    # the forced re-alignments: every run on the window that just holds its certified band (default: one wave up to 2048 diagonals,
    # a group of four waves up to 8192, column strips beyond — a list that is empty on every workload seen) — with the single-wave
    # windows capped at 256 diagonals every wider run goes to the GROUP engine, and with the group switched off as well to the
    # STRIPS: three ways through the same certified-band loop, not a single result may differ
    monkeypatch.setenv("PYANI_ANIM_BWD_AHEAD", "1")
    for window_max, group_max in (("256", "8184"), ("256", "0"), ("128", "3064")):
        monkeypatch.setenv("PYANI_PN_WINDOW_MAX", window_max)
        monkeypatch.setenv("PYANI_PN_GROUP_MAX", group_max)
        with Engine(0) as e:
            ids = [e.add_genome(*d) for d in data]
            pairs = [(a, b) for a in ids for b in ids if a != b]
            got = e.anim_pairs([a for a, _ in pairs], [b for _, b in pairs])
        assert got.tobytes() == out["1"].tobytes(), (window_max, group_max)


def test_cluster_stage_forms_do_not_change_results(eng, monkeypatch):
    """The cluster stage finishes small units in one wave and cuts the chain extraction of big units (>= PYANI_ANIM_SPLIT_MIN
    matches, default 2048) into ranges of whole clusters, one wave each, merged in order (pga_cluster.inc:
    anim_chain_range_kernel / anim_chain_merge_kernel); the front half of big units runs in one wave or in a 16-wave
    workgroup.  None of that may show: never split, split everything into single ranges, into many short ranges, with either
    front half, --mum and --maxmatch, on genomes with repeats (clusters of hundreds of matches) — same records."""
    from pyani_amd import synth
    eng.clear_genomes()
    n, L = 4, 300_000
    ids = [eng.add_genome(*_with_repeats(*synth.genome(31, n, g, L), g)) for g in range(n)]
    ids += [eng.add_genome(*synth.genome(11, 3, g, 400_000)) for g in range(3)]
    eng.upload()
    pairs = [(a, b) for a in ids[:n] for b in ids[:n] if a != b] + [(a, b) for a in ids[n:] for b in ids[n:] if a != b]
    ra, qa = [a for a, _ in pairs], [b for _, b in pairs]
    for var in ("PYANI_ANIM_SPLIT_MIN", "PYANI_ANIM_RANGE_ENTRIES", "PYANI_ANIM_WAVE_PREP"):
        monkeypatch.delenv(var, raising=False)
    for mm in (False, True):
        monkeypatch.setenv("PYANI_ANIM_SPLIT_MIN", "0")          # every unit start to end in its one wave
        ref = eng.anim_pairs(ra, qa, maxmatch=mm)
        assert (ref["status"] == 0).sum() == len(pairs) and int(ref["n_alignments"].max()) > 20
        monkeypatch.delenv("PYANI_ANIM_SPLIT_MIN")
        assert eng.anim_pairs(ra, qa, maxmatch=mm).tobytes() == ref.tobytes(), ("default", mm)
        for split_min, entries, one_wave_front in (("1", "1000000", True), ("1", "16", True), ("1", "200", False), ("500", "64", False)):
            monkeypatch.setenv("PYANI_ANIM_SPLIT_MIN", split_min)
            monkeypatch.setenv("PYANI_ANIM_RANGE_ENTRIES", entries)
            if one_wave_front:
                monkeypatch.setenv("PYANI_ANIM_WAVE_PREP", "1")
            assert eng.anim_pairs(ra, qa, maxmatch=mm).tobytes() == ref.tobytes(), (split_min, entries, one_wave_front, mm)
            for var in ("PYANI_ANIM_SPLIT_MIN", "PYANI_ANIM_RANGE_ENTRIES", "PYANI_ANIM_WAVE_PREP"):
                monkeypatch.delenv(var, raising=False)


def _resplit(seq, step, salt):
    """Same cutting rule as tools/make_anim_synth_host.py (contig-shaped records, some too short to seed)."""
    cuts, p, k = [0], 0, 0
    while p < len(seq):
        p = min(len(seq), p + step + ((k * 7919 + salt) % step) - step // 2 + (12 if k % 9 == 4 else 0))
        if k % 11 == 5:
            p = min(len(seq), cuts[-1] + 15)
        cuts.append(p)
        k += 1
    return np.array(sorted(set(cuts)), dtype=np.uint64)


def test_shared_seeding_of_a_pair_and_its_reverse_does_not_change_results(eng, monkeypatch):
    """(A, B) and (B, A) have the same maximal exact matches, so a launch that holds both seeds one of them and hands the
    matches, transposed, to the other (pg_anim.hip "roles", pga_seed.inc anim_hit_kernel).  With repeats, reverse-complemented
    copies, many unequal records and ambiguity symbols in the genomes: the call with both directions == every pair seeded by
    itself (PYANI_ANIM_NO_MIRROR) == the directions computed in separate calls; pairs listed twice and self pairs included."""
    from pyani_amd import synth
    eng.clear_genomes()
    data = [_with_repeats(*synth.genome(31, 4, g, 300_000), g) for g in range(4)]
    for g in range(3):
        seq, _ = synth.genome(17, 3, g, 400_000)
        seq = seq.copy()
        seq[1000 * (g + 1): 1000 * (g + 1) + 7] = ord("N")        # ambiguity symbols: dirty positions inside would-be matches
        seq[200_000 + 13 * g] = ord("n")
        data.append((seq, _resplit(seq, 3000, g)))
    ids = [eng.add_genome(*d) for d in data]
    eng.upload()
    grid = [(a, b) for a in ids for b in ids if a != b]
    for mm in (False, True):
        monkeypatch.setenv("PYANI_ANIM_NO_MIRROR", "1")
        want = eng.anim_pairs([a for a, _ in grid], [b for _, b in grid], maxmatch=mm)
        monkeypatch.delenv("PYANI_ANIM_NO_MIRROR")
        assert (want["status"] == 0).sum() >= 18
        got = eng.anim_pairs([a for a, _ in grid], [b for _, b in grid], maxmatch=mm)
        assert got.tobytes() == want.tobytes(), mm
        upper = [k for k, (a, b) in enumerate(grid) if a < b]     # one direction only: nothing to share
        one = eng.anim_pairs([grid[k][0] for k in upper], [grid[k][1] for k in upper], maxmatch=mm)
        assert one.tobytes() == want[upper].tobytes(), mm
        odd = [grid[0], grid[0], grid[0][::-1], (ids[0], ids[0]), grid[5][::-1], grid[5], grid[0][::-1]]
        res = eng.anim_pairs([a for a, _ in odd], [b for _, b in odd], maxmatch=mm)
        index = {p: k for k, p in enumerate(grid)}
        for p, r in zip(odd, res):
            if p[0] != p[1]:
                assert r.tobytes() == want[index[p]].tobytes(), (p, mm)
        assert res[3].tobytes() == eng.anim_pairs([ids[0]], [ids[0]], maxmatch=mm)[0].tobytes()    # a genome against itself


def test_many_records_equal_scalar_host_statement(eng):
    """Draft-genome shape: ~150 records per genome of unequal length.  Alignments must stop at record ends and the
    per-sequence interval unions must hold: GPU == scalar host build of the same core (tools/make_anim_synth_host.py)."""
    from pyani_amd import synth
    fx = json.loads((GOLD / "anim_synth_host.json").read_text())
    n, L, seed, step = fx["n"], fx["length"], fx["seed"], fx["contigs"]["step"]
    eng.clear_genomes()
    ids = []
    for g in range(3):
        seq, _ = synth.genome(seed, n, g, L)
        off = _resplit(seq, step, g)
        assert len(off) > 100
        ids.append(eng.add_genome(seq, off))
    eng.upload()
    pairs = [(a, b) for a in range(3) for b in range(3) if a != b]
    res = eng.anim_pairs([ids[a] for a, _ in pairs], [ids[b] for _, b in pairs])
    for (a, b), r in zip(pairs, res):
        got = [int(r["ref_aln_len"]), int(r["qry_aln_len"]), float(r["identity"]).hex(), int(r["sim_errors"]), int(r["n_alignments"])]
        assert got == fx["contigs"]["pairs"][f"{a},{b}"], (a, b, got)


def test_run_anim_writes_recovery_files_and_recovers_from_them(eng, genome_dir, tmp_path):
    """pyani_amd.subcmd_anim.run_anim on the GPU: a run that writes its .filter files, then a --recovery run over the same
    output directory with one file removed: 5 comparisons come from the files, 1 is recomputed, every tuple / matrix cell is
    identical; and the tuples are the MUMmer goldens."""
    import numpy as np
    from pyani_amd import subcmd_anim
    eng.clear_genomes()
    gold = json.loads((GOLD / "anim_goldens.json").read_text())["parse_delta"]
    indir, outdir = tmp_path / "in", tmp_path / "out"
    indir.mkdir()
    stems = sorted(genome_dir["blochmannia"])[:3]
    for s in stems:
        (indir / f"{s}.fna").write_bytes(genome_dir["blochmannia"][s].read_bytes())
    first = subcmd_anim.run_anim(indir, outdir, write_output=True, engine=eng)
    assert len(first.written) == 6 and not first.recovered and eng.genome_count() == 0
    for (q, s), t in first.results.items():
        rel = f"blochmannia/{q}_vs_{s}.filter"
        if rel in gold:
            assert list(t) == gold[rel], rel
    # the files themselves (one batched call + the GPU traceback pass): MUMmer's own .filter files but for their first line (the paths)
    import gzip
    same = 0
    for f in first.written:
        g = GOLD / "anim" / "blochmannia" / (f.name + ".gz")
        if g.exists():
            assert f.read_text().splitlines()[1:] == gzip.open(g, "rt").read().splitlines()[1:], f.name
            same += 1
    assert same >= 3
    first.written[2].unlink()
    again = subcmd_anim.run_anim(indir, outdir, recovery=True, engine=eng)
    assert len(again.recovered) == 5 and again.results == first.results and again.comparisons and again.json == first.json
    for name in first.matrices:
        assert np.array_equal(first.matrices[name].values, again.matrices[name].values, equal_nan=True)
    assert list(first.matrices["identity"].index) == [1, 2, 3]


def _with_repeats(seq, off, salt):
    """Repeat-bearing variant of a synthetic genome (the generator's ancestors are i.i.d. and repeat-free): 3 copies of a 5 kb
    'rRNA operon' (one of them with 1 % substitutions) and 6 copies of a 1.4 kb 'IS element' (2 reverse-complemented) pasted
    over the first record; same length, same records."""
    rng = np.random.RandomState(1000 + salt)
    s = seq.copy()
    n = int(off[1])
    operon = np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.RandomState(77).randint(0, 4, size=5000)]
    ins = np.frombuffer(b"ACGT", dtype=np.uint8)[np.random.RandomState(78).randint(0, 4, size=1400)]
    comp = np.zeros(256, dtype=np.uint8)
    comp[list(b"ACGT")] = list(b"TGCA")
    slots = rng.permutation(np.arange(10_000, n - 10_000, 12_000))[:9]
    for k, pos in enumerate(slots[:3]):
        cp = operon.copy()
        if k == 2:
            hit = rng.rand(len(cp)) < 0.01
            cp[hit] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.randint(0, 4, size=int(hit.sum()))]
        s[pos:pos + 5000] = cp
    for k, pos in enumerate(slots[3:]):
        s[pos:pos + 1400] = comp[ins][::-1] if k % 3 == 0 else ins
    return s, off


@pytest.fixture(scope="module")
def repeat_runs(eng):
    """Genomes WITH repeats (operon and insertion-element copies, some diverged, some reverse-complemented) through the GPU
    pipeline and through the CPU statement of the same search (oracle/anim_cpu.cpp, exhaustive seeding), --mum and --maxmatch."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import anim_cpu
    from pyani_amd import synth
    eng.clear_genomes()
    n, L = 4, 300_000
    data = [_with_repeats(*synth.genome(31, n, g, L), g) for g in range(n)]
    ids = [eng.add_genome(*d) for d in data]
    eng.upload()
    pairs = [(a, b) for a in range(n) for b in range(n) if a != b]
    got, want_all = {}, {}
    for mm in (False, True):
        res = eng.anim_pairs([ids[a] for a, _ in pairs], [ids[b] for _, b in pairs], maxmatch=mm)
        want, _ = anim_cpu.anim_cpu_pairs(data, [a for a, _ in pairs], [b for _, b in pairs], maxmatch=mm)
        for p, r, w in zip(pairs, res, want):
            got[(mm, p)] = (int(r["ref_aln_len"]), int(r["qry_aln_len"]), int(r["sim_errors"]), int(r["n_alignments"]), float(r["identity"]).hex(),
                            int(r["status"]))
            want_all[(mm, p)] = (int(w["ref_aln_len"]), int(w["qry_aln_len"]), int(w["sim_errors"]), int(w["n_alignments"]),
                                 float(w["identity"]).hex(), int(w["status"]))
    nofilt = eng.anim_pairs([ids[0]], [ids[1]], filter_1to1=False)[0]
    return pairs, got, want_all, int(nofilt["n_alignments"])


def test_repeats_mum_mode_equals_the_cpu_statement(repeat_runs):
    """--mum (what pyani runs): with repeats the uniqueness filter and the 1-to-1 filter both have work to do; every tuple of
    the GPU pipeline equals the CPU statement's, and --maxmatch is a different search on these genomes."""
    pairs, got, want, n_unfiltered = repeat_runs
    bad = [(p, got[(False, p)], want[(False, p)]) for p in pairs if got[(False, p)] != want[(False, p)]]
    assert not bad, bad[:2]
    assert any(got[(False, p)] != got[(True, p)] for p in pairs)
    assert n_unfiltered > got[(False, (0, 1))][3]                     # delta-filter -1 had repeat alignments to drop


def test_repeats_maxmatch_mode_equals_the_cpu_statement(repeat_runs):
    """--maxmatch on the same genomes: several query copies anchor one reference copy, so chains START ON THE SAME REFERENCE
    BASE — their order is pga::chain_before's total order (start, then extraction order) in every form of the cluster stage
    (found in round 2: the scalar forms sorted such ties in an unspecified order and 3 of these 12 pairs differed)."""
    pairs, got, want, _ = repeat_runs
    bad = [(p, got[(True, p)], want[(True, p)]) for p in pairs if got[(True, p)] != want[(True, p)]]
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "anim_repeats_maxmatch_diff.json").write_text(json.dumps(bad))
    assert not bad, bad[:2]


def test_process_deltadir_on_the_gpu_equals_the_imported_reference(eng, tmp_path):
    """pyani_amd.anim.process_deltadir with the real engine (batched pg_anim_reduce) on the reference's deltadir fixture: all
    five matrices bit-equal to what the imported reference's process_deltadir produced (tools/make_goldens.py)."""
    import gzip
    from pyani_amd import anim
    fx = json.loads((GOLD / "ref_targets" / "anim_process_deltadir_cases.json").read_text())["caulobacter_deltadir"]
    lengths = dict(fx["lengths"])
    for gz in sorted((GOLD / "anim" / "caulobacter").glob("*.filter.gz")):
        q = gz.name.split("_vs_")[0]
        (tmp_path / q).mkdir(exist_ok=True)
        with gzip.open(gz, "rb") as fi:
            (tmp_path / q / gz.name[:-3]).write_bytes(fi.read())
    res = anim.process_deltadir(tmp_path, lengths, engine=eng)
    for df, stem in res.data:
        want = fx["matrices"][stem]
        assert list(df.index) == want["labels"]
        for a, row in zip(want["labels"], want["rows"]):
            for b, h in zip(want["labels"], row):
                w, g = float.fromhex(h), float(df.loc[a, b])
                assert g == w or (np.isnan(g) and np.isnan(w)), (stem, a, b, g, w)
