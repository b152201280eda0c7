"""CPU: what pyani's DEFAULT job computes — nucmer, then `delta-filter -1`, then parse_delta (pyani/anim.py:280, 285-288, 292-411) —
checked against code that shares NOTHING with the product: the nucmer oracle's records
(oracle/nucmer_oracle.cpp) through the pure-Python restatement of the 1-to-1 filter and of parse_delta (oracle/anim_oracle.py; both
pinned on the reference's own .delta / .filter files, tests/test_anim_cpu.py).  The product's host statement
(tools/anim_debug/anim_debug: the engine's seeding / clustering / pgn::postnuc_unit / pga::lis_filter / pga::reduce_pair) must make
every keep / drop decision the same way and print the same tuple — on pairs built so that the filter has something to decide
(tests/stress_genomes.py: translocated blocks that carry their flanks, diverged duplicates, inversions with duplicated flanks, on
either side).  The GPU runs the same pairs through the C ABI in tests/test_anim_filter_oracle_gpu.py."""
import random
import subprocess
import sys

import pytest

from tests.conftest import ROOT
from tests.fuzz_genomes import write_fasta
from tests.stress_genomes import expected_filtered, make_rearranged_pair

sys.path.insert(0, str(ROOT / "oracle"))


@pytest.fixture(scope="module")
def programs():
    oracle = ROOT / "oracle" / "_build" / "nucmer_oracle"
    oracle.parent.mkdir(exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", str(ROOT / "oracle" / "nucmer_oracle.cpp"), "-o", str(oracle)], check=True)
    stmt = ROOT / "tools" / "anim_debug" / "anim_debug"
    subprocess.run(["g++", "-O2", "-std=c++17", f"-I{ROOT / 'pyani_amd' / 'csrc'}", str(stmt) + ".cpp", "-o", str(stmt)], check=True)
    return oracle, stmt


def oracle_records(exe, pa, pb, extra=()):
    """the oracle's ALN lines, in its output order (= MUMmer's: delta-filter's tie rule looks at input order)"""
    out = subprocess.run([str(exe), str(pa), str(pb), *extra], capture_output=True, text=True, check=True).stdout
    return [(t[1], t[2], int(t[3]), int(t[4]), int(t[5]), int(t[6]), int(t[7])) for t in (ln.split() for ln in out.splitlines()) if t and t[0] == "ALN"]


def statement_records(exe, pa, pb):
    """(record -> keep flag, printed tuple) of the host statement with the filter on"""
    r = subprocess.run([str(exe), str(pa), str(pb), "--dump"], capture_output=True, text=True)
    assert r.returncode in (0, 4), r.stderr[-300:]
    recs, tup = {}, None
    for ln in r.stdout.splitlines():
        t = ln.split()
        if t and t[0] == "ALN":
            recs[(t[1], t[2], int(t[3]), int(t[4]), int(t[5]), int(t[6]), int(t[7]))] = int(t[8].split("=")[1])
        elif len(t) == 5 and tup is None:
            tup = t
    return recs, tup


def test_statement_filter_decisions_and_tuple_equal_oracle_on_rearranged_pairs(programs, tmp_path):
    oracle, stmt = programs
    n_dropped = n_pairs_with_drops = n_records = 0
    for t in range(30):
        rng = random.Random(3000017 + t)
        ref, qry = make_rearranged_pair(rng)
        pa, pb = tmp_path / f"r{t}.fna", tmp_path / f"q{t}.fna"
        write_fasta(pa, "r", ref)
        write_fasta(pb, "q", qry)
        for a, b in ((pa, pb), (pb, pa)):
            want = oracle_records(oracle, a, b)
            keep, tup = expected_filtered(want)
            got, printed = statement_records(stmt, a, b)
            assert set(got) == set(want), (t, sorted(set(got) ^ set(want))[:4])
            bad = [(r, k, got[r]) for r, k in zip(want, keep) if (got[r] == 3) != k]
            assert not bad, (t, a.name, bad[:3])
            if tup is not None:
                assert (int(printed[0]), int(printed[1]), float(printed[2]), int(printed[3]), int(printed[4])) == tup, (t, printed, tup)
            n_records += len(want)
            n_dropped += len(keep) - sum(keep)
            n_pairs_with_drops += sum(keep) < len(keep)
    # the generator does what it is for: the filter drops records in most pairs
    assert n_records > 600 and n_dropped > 60 and n_pairs_with_drops >= 35, (n_records, n_dropped, n_pairs_with_drops)


def test_chain_dp_beyond_the_64_entry_window_equals_oracle_on_tandem_repeats(programs, tmp_path):
    """DESIGN §4's former deviation (2), directed: tandem repeats under --maxmatch put hundreds of overlapping matches into one mgaps
    cluster, where a match's best predecessor lies more than 64 entries back.  The statement's chain DP (pga::mgaps_strand) keeps a
    64-entry window and now goes on to the rest whenever an entry outside it could matter; it reports how often that happened and
    how often it CHANGED the predecessor — so this test proves it is exercised — and the records equal the oracle's full scan."""
    from tests.stress_genomes import make_tandem_pair
    oracle, stmt = programs
    changed = trials_changed = 0
    for t in range(24):
        rng = random.Random(7000001 + t)
        ref, qry = make_tandem_pair(rng)
        pa, pb = tmp_path / f"tr{t}.fna", tmp_path / f"tq{t}.fna"
        write_fasta(pa, "r", [ref])
        write_fasta(pb, "q", [qry])
        want = set(oracle_records(oracle, pa, pb, ["--maxmatch"]))
        r = subprocess.run([str(stmt), str(pa), str(pb), "--dump", "--maxmatch", "--nofilter"], capture_output=True, text=True)
        got = {(x[1], x[2], int(x[3]), int(x[4]), int(x[5]), int(x[6]), int(x[7])) for x in (ln.split() for ln in r.stdout.splitlines()) if x and x[0] == "ALN"}
        assert got == want, (t, sorted(got ^ want)[:4])
        for ln in r.stderr.splitlines():
            if ln.startswith("chain DP:"):
                n = int(ln.split(",")[1].split()[0])
                changed += n
                trials_changed += n > 0
    assert changed > 1000 and trials_changed >= 8, (changed, trials_changed)


def test_two_strand_walk_equals_oracle_where_a_walk_per_strand_would_not(programs, tmp_path):
    """DESIGN §4's former deviation (1), directed: MUMmer walks the clusters of BOTH strands of a record pair in one list and tests a
    cluster for being shadowed from the pair's CURRENT alignment backwards; the engine walks the strands side by side and lets the
    walks consult each other (pgn::PnPairSync).  On tandem arrays that match on both strands (tests/stress_genomes.py) the statement
    reports how many shadow tests needed the record pair's current alignment and how many of those a walk per strand — what rounds 3-4
    did — would have answered otherwise: some must, and every record equals the oracle's (which walks one list, as MUMmer does)."""
    import re
    from tests.stress_genomes import TWO_STRAND_TRIALS, make_two_strand_repeat_pair
    oracle, stmt = programs
    asked = differs = 0
    for t in TWO_STRAND_TRIALS:
        rng = random.Random(13000003 + t)
        ref, qry = make_two_strand_repeat_pair(rng)
        pa, pb = tmp_path / f"pr{t}.fna", tmp_path / f"pq{t}.fna"
        write_fasta(pa, "r", ref)
        write_fasta(pb, "q", qry)
        extra = ["--maxmatch"] if t % 4 != 0 else []
        want = set(oracle_records(oracle, pa, pb, extra))
        r = subprocess.run([str(stmt), str(pa), str(pb), "--dump", "--nofilter"] + extra, capture_output=True, text=True)
        got = {(x[1], x[2], int(x[3]), int(x[4]), int(x[5]), int(x[6]), int(x[7])) for x in (ln.split() for ln in r.stdout.splitlines()) if x and x[0] == "ALN"}
        assert got == want, (t, extra, sorted(got ^ want)[:4])
        asked += sum(int(x) for x in re.findall(r"current alignment: (\d+),", r.stderr))
        differs += sum(int(x) for x in re.findall(r"own current alignment would: (\d+)", r.stderr))
    assert asked > 5000 and differs >= 5, (asked, differs)
