"""GPU (through the C ABI): what `value` of bench.py computes and what pyani's default job computes — the search, `delta-filter -1`
and parse_delta in one call, filter ON — against code that shares nothing with the product ("weak" 1 / "next" 1):

  * the oracle's goldens of the BENCHMARK workloads (tests/golden/anim_oracle_goldens.json.gz: C4 slice, C4's most divergent family,
    C3 family, and C4's family 2 with overlapping rearrangements per genome, the set built for this check — records of
    oracle/nucmer_oracle.cpp in its output order) through oracle/anim_oracle.py::delta_filter_1to1 and
    parse_delta_records (both pinned on the reference's own .delta / .filter files): pg_anim_pairs(filter_1to1 = 1) must return that
    tuple, and pg_anim_alignments_batch's `kept` flags must mark exactly the surviving records;
  * the same on pairs built to make the filter choose (tests/stress_genomes.py), with the oracle run here on the same FASTA files.
Reference semantics: pyani/anim.py:285-288 (`delta-filter -1`), :292-411 (parse_delta)."""
import gzip
import json
import random
import sys

import pytest

from tests.conftest import GOLD, ROOT

sys.path.insert(0, str(ROOT / "oracle"))

pytestmark = pytest.mark.gpu


def _tuple(t):
    return (int(t["ref_aln_len"]), int(t["qry_aln_len"]), float(t["identity"]), int(t["sim_errors"]), int(t["n_alignments"]))


def _check(name, k, want, keep, tup, recs, res):
    """want: the oracle's records (rrec, qrec, rs, re, qs, qe, errors) in its order; keep / tup: tests.stress_genomes.expected_filtered;
    recs: the engine's ALN records of the pair; res: its pg_anim_result with the filter on"""
    got = {(int(r["ref_rec"]), int(r["qry_rec"]), int(r["rs"]), int(r["re"]), int(r["qs"]), int(r["qe"]), int(r["errors"])): int(r["kept"]) for r in recs}
    assert set(got) == {tuple(w) for w in want}, (name, k, "records")
    bad = [(w, kp, got[tuple(w)]) for w, kp in zip(want, keep) if (got[tuple(w)] == 3) != kp]
    assert not bad, (name, k, "kept flags", bad[:3])
    if tup is None:
        assert int(res["n_alignments"]) == 0 and int(res["status"]) == 1, (name, k)
    else:
        assert _tuple(res) == tup, (name, k, _tuple(res), tup)
    return len(keep) - sum(keep)


@pytest.mark.parametrize("name", ["c4_filter_stress", "c4_slice", "c4_divergent", "c3_family"])
def test_benchmark_workloads_filtered_tuple_and_kept_flags_equal_the_independent_oracle(name):
    from pyani_amd.engine import Engine
    from tests.stress_genomes import expected_filtered
    from tests.test_anim_oracle_goldens_gpu import golden_genome
    with gzip.open(GOLD / "anim_oracle_goldens.json.gz", "rt") as fh:
        S = json.load(fh)[name]
    pairs = S["pairs"]
    used = sorted({g for p in pairs for g in p[:2]})
    with Engine(0) as eng:
        ids = {g: eng.add_genome(*golden_genome(S, g)) for g in used}
        q, s = [ids[p[0]] for p in pairs], [ids[p[1]] for p in pairs]
        res = eng.anim_pairs(q, s)                                    # filter ON: bench.py's call, pyani's default
        off, recs, _, _ = eng.anim_alignments_batch(q, s)
    dropped = 0
    for k, (a, b, want, _) in enumerate(pairs):
        keep, tup = expected_filtered([(str(w[0]), str(w[1])) + tuple(w[2:]) for w in want])
        dropped += _check(name, (a, b), want, keep, tup, recs[int(off[k]):int(off[k + 1])], res[k])
    # (the benchmark generator moves and inverts blocks cleanly: on its three sets delta-filter -1 drops nothing — 0 of 2 100 records —
    # which is WHY c4_filter_stress exists: the same family with overlapping rearrangements, where the filter has to choose)
    assert dropped > 20 if name == "c4_filter_stress" else dropped == 0, (name, dropped)


def test_rearranged_pairs_filtered_tuple_and_kept_flags_equal_the_independent_oracle(tmp_path):
    from pyani_amd import anim
    from pyani_amd.engine import Engine
    from tests.fuzz_genomes import write_fasta
    from tests.stress_genomes import expected_filtered, make_rearranged_pair
    from tests.test_anim_filter_oracle_cpu import oracle_records
    from tests.test_anim_multirecord_gpu import _oracle
    exe = _oracle()
    trials = []
    for t in range(40):
        rng = random.Random(3000017 + t)
        ref, qry = make_rearranged_pair(rng)
        pa, pb = tmp_path / f"r{t}.fna", tmp_path / f"q{t}.fna"
        write_fasta(pa, f"r{t}_", ref)
        write_fasta(pb, f"q{t}_", qry)
        trials.append((pa, pb))
    with Engine(0) as eng:
        ids = [(eng.add_fasta(pa)[0], eng.add_fasta(pb)[0]) for pa, pb in trials]
        q = [i[0] for i in ids] + [i[1] for i in ids]
        s = [i[1] for i in ids] + [i[0] for i in ids]
        res = eng.anim_pairs(q, s)
        off, recs, _, _ = eng.anim_alignments_batch(q, s)
    dropped = with_drops = 0
    for j in range(2 * len(trials)):
        fwd = j < len(trials)
        pa, pb = trials[j % len(trials)] if fwd else trials[j % len(trials)][::-1]
        na, nb = {n[0]: i for i, n in enumerate(anim.fasta_records(pa))}, {n[0]: i for i, n in enumerate(anim.fasta_records(pb))}
        orc = oracle_records(exe, pa, pb)
        keep, tup = expected_filtered(orc)
        want = [(na[r[0]], nb[r[1]]) + r[2:] for r in orc]
        d = _check("rearranged", j, want, keep, tup, recs[int(off[j]):int(off[j + 1])], res[j])
        dropped += d
        with_drops += d > 0
    assert dropped > 80 and with_drops >= 45, (dropped, with_drops)


def test_chain_dp_beyond_the_64_entry_window_equals_oracle_on_tandem_repeats(tmp_path):
    """DESIGN §4's former deviation (2), directed (CPU twin: tests/test_anim_filter_oracle_cpu.py): tandem repeats under --maxmatch give
    mgaps clusters of hundreds of overlapping matches whose best predecessors lie more than 64 entries back in query order.  The
    wave form of the chain DP (pga_cluster.inc, extract_chains: a 64-entry register window) now scans the entries outside the window
    whenever one of them could matter; every record must equal the oracle's (full scan, as mgaps) — 25 of the first 60 pairs of this
    generator did not in round 4."""
    from pyani_amd.engine import Engine
    from tests.fuzz_genomes import write_fasta
    from tests.stress_genomes import make_tandem_pair
    from tests.test_anim_filter_oracle_cpu import oracle_records
    from tests.test_anim_multirecord_gpu import _oracle
    exe = _oracle()
    trials = []
    for t in range(48):
        rng = random.Random(7000001 + t)
        ref, qry = make_tandem_pair(rng)
        pa, pb = tmp_path / f"tr{t}.fna", tmp_path / f"tq{t}.fna"
        write_fasta(pa, "r", [ref])
        write_fasta(pb, "q", [qry])
        trials.append((pa, pb))
    with Engine(0) as eng:
        ids = [(eng.add_fasta(pa)[0], eng.add_fasta(pb)[0]) for pa, pb in trials]
        q = [i[0] for i in ids] + [i[1] for i in ids]
        s = [i[1] for i in ids] + [i[0] for i in ids]
        off, recs, _, _ = eng.anim_alignments_batch(q, s, maxmatch=True)
    n = 0
    for j in range(2 * len(trials)):
        pa, pb = trials[j % len(trials)] if j < len(trials) else trials[j % len(trials)][::-1]
        want = {r[2:] for r in oracle_records(exe, pa, pb, ["--maxmatch"])}
        got = {(int(r["rs"]), int(r["re"]), int(r["qs"]), int(r["qe"]), int(r["errors"])) for r in recs[int(off[j]):int(off[j + 1])]}
        assert got == want, (j, sorted(got ^ want)[:4])
        n += len(want)
    assert n >= 96


def test_two_strand_walk_equals_oracle_where_a_walk_per_strand_would_not(tmp_path):
    """DESIGN §4's former deviation (1), directed (CPU twin: tests/test_anim_filter_oracle_cpu.py, which also proves that the
    inputs make the two rules differ): tandem arrays that match on both strands, --maxmatch and --mum, 1 - 2 records per genome.
    The walk kernel runs the two strands of a pair as the two waves of a workgroup that consult each other (pga_postnuc.inc,
    PnWavePrim / pgn::PnPairSync); every record must equal the oracle's, which walks both strands in one list as MUMmer does.  Also
    the canonical order of clusters that start on the same reference base (pga::chain_before) and mgaps' (query start, reference
    start) match order under --maxmatch: 34 of the first 120 pairs of this generator differed from the oracle before round 5."""
    from pyani_amd import anim
    from pyani_amd.engine import Engine
    from tests.fuzz_genomes import write_fasta
    from tests.stress_genomes import TWO_STRAND_TRIALS, make_two_strand_repeat_pair
    from tests.test_anim_filter_oracle_cpu import oracle_records
    from tests.test_anim_multirecord_gpu import _oracle
    exe = _oracle()
    trials = []
    for t in list(range(0, 40)) + TWO_STRAND_TRIALS:
        rng = random.Random(13000003 + t)
        ref, qry = make_two_strand_repeat_pair(rng)
        pa, pb = tmp_path / f"pr{t}.fna", tmp_path / f"pq{t}.fna"
        write_fasta(pa, f"r{t}_", ref)
        write_fasta(pb, f"q{t}_", qry)
        trials.append((pa, pb, t % 4 != 0))
    n = 0
    with Engine(0) as eng:
        ids = [(eng.add_fasta(pa)[0], eng.add_fasta(pb)[0]) for pa, pb, _ in trials]
        for mm in (True, False):
            sel = [k for k, tr in enumerate(trials) if tr[2] == mm]
            q = [ids[k][0] for k in sel] + [ids[k][1] for k in sel]
            s = [ids[k][1] for k in sel] + [ids[k][0] for k in sel]
            off, recs, _, _ = eng.anim_alignments_batch(q, s, maxmatch=mm)
            for j, k in enumerate(sel + sel):
                pa, pb = trials[k][:2] if j < len(sel) else trials[k][:2][::-1]
                na, nb = [x[0] for x in anim.fasta_records(pa)], [x[0] for x in anim.fasta_records(pb)]
                want = set(oracle_records(exe, pa, pb, ["--maxmatch"] if mm else []))
                got = {(na[int(r["ref_rec"])], nb[int(r["qry_rec"])], int(r["rs"]), int(r["re"]), int(r["qs"]), int(r["qe"]), int(r["errors"]))
                       for r in recs[int(off[j]):int(off[j + 1])]}
                assert got == want, (mm, k, j < len(sel), sorted(got ^ want)[:4])
                n += len(want)
    assert n > 3000
