"""Small genome pairs built to stress `delta-filter -1` (pyani/anim.py:285-288 runs it on every nucmer output): rearrangements whose
alignments OVERLAP on one side — a translocated block copied with its flanks (the flanks then align twice: the reference intervals of
two alignments overlap), diverged duplicates (one reference region, two query candidates: the weighted LIS has to choose), inversions
with duplicated flanks, the same on the reference side (overlaps in query coordinates).  The benchmark generator
(pyani_amd/csrc/synth.cpp) moves and inverts blocks cleanly, so its alignments abut and the filter has little to decide; here every
pair makes it choose.  Test DATA for tests/test_anim_filter_oracle_{cpu,gpu}.py.  Deterministic for a given Python version
(random.Random), as tests/fuzz_genomes.py."""
from tests.fuzz_genomes import COMP, mutate


def _rc(s):
    return s[::-1].translate(COMP)


def _rearrange(rng, s, n_ops, scale=1):
    """n_ops overlapping rearrangements of s (a str); scale: block and flank lengths are multiplied by it (whole genomes: 10)."""
    for _ in range(n_ops):
        n = len(s)
        op = rng.choice(["dup_div", "dup_div", "transloc_flank", "transloc_flank", "inv_flank", "inversion"])
        ln = rng.randint(700 * scale, 4000 * scale)
        a = rng.randint(1000 * scale, n - ln - 1000 * scale)
        b = a + ln
        if op == "dup_div":              # a diverged second copy somewhere else (sometimes reverse-complemented)
            cp = mutate(rng, s[a:b], rng.choice([0.0, 0.01, 0.03, 0.05]), 0.002)
            if rng.random() < 0.4:
                cp = _rc(cp)
            d = rng.randint(500, n - 500)
            s = s[:d] + cp + s[d:]
        elif op == "transloc_flank":     # the block moves, its copy carries flanks that also stay where they were
            f1, f2 = rng.randint(0, 600 * scale), rng.randint(0, 600 * scale)
            blk = s[max(0, a - f1):min(n, b + f2)]
            if rng.random() < 0.3:
                blk = _rc(blk)
            rest = s[:a] + s[b:]
            d = rng.randint(500, len(rest) - 500)
            s = rest[:d] + blk + rest[d:]
        elif op == "inv_flank":          # inversion whose inverted copy includes the flanks (kept in place as well)
            f1, f2 = rng.randint(100 * scale, 500 * scale), rng.randint(100 * scale, 500 * scale)
            s = s[:a] + _rc(s[a - f1:b + f2]) + s[b:]
        else:
            s = s[:a] + _rc(s[a:b]) + s[b:]
    return s


def make_rearranged_pair(rng, L=60000):
    """(reference records, query records): one ancestor, the query diverged (0.5 - 6 % substitutions) and rearranged, sometimes the
    reference rearranged too; 1 - 3 records each."""
    anc = "".join(rng.choice("ACGT") for _ in range(L))
    p = rng.choice([0.005, 0.01, 0.03, 0.06])
    qry = _rearrange(rng, mutate(rng, anc, p, p / 10), rng.randint(3, 7))
    ref = _rearrange(rng, anc, rng.randint(0, 3)) if rng.random() < 0.5 else anc

    def split(s, n):
        if n <= 1:
            return [s]
        cs = sorted(rng.sample(range(2000, len(s) - 2000), n - 1))
        return [s[x:y] for x, y in zip([0] + cs, cs + [len(s)])]

    return split(ref, rng.choice([1, 1, 2, 3])), split(qry, rng.choice([1, 1, 2, 3]))


def rearranged_benchmark_genome(seed, n, g, L, n_ops=10):
    """genome g of a benchmark set (pyani_amd.synth: bench.py's generator) with n_ops overlapping rearrangements at whole-genome scale
    (blocks of 7 - 40 kb, flanks up to 6 kb), as (uint8 sequence, record offsets): the `c4_filter_stress` set of the oracle goldens
    (tools/make_anim_oracle_goldens.py) and of the GPU test that regenerates its genomes from the seeds."""
    import random
    import numpy as np
    from pyani_amd import synth
    seq, off = synth.genome(seed, n, g, L)
    s = _rearrange(random.Random(seed * 1000 + g), bytes(seq).decode("ascii"), n_ops, scale=10)
    n_rec = len(off) - 1
    cuts = [0] + [int(len(s) * k / n_rec) for k in range(1, n_rec)] + [len(s)]
    return np.frombuffer(s.encode("ascii"), dtype=np.uint8).copy(), np.array(cuts, dtype=np.uint64)


def make_tandem_pair(rng):
    """(reference, query) single-record sequences of 3 - 8 kb carrying 1 - 3 TANDEM REPEATS (units of 3 ... 31 bases, 30 - 200 copies,
    lightly mutated).  Under --maxmatch (pyani's --maxmatch: every maximal match, not only unique ones) such a repeat yields hundreds of
    overlapping matches on neighbouring diagonals in ONE mgaps cluster, and the best predecessor of a match in the chain DP often lies
    more than 64 matches back in query order: the case the engine's 64-entry window used to miss (deviation (2) of
    rounds 3-4; 25 of the first 60 pairs of this generator gave other records than the oracle then)."""
    L = rng.randint(3000, 8000)
    base = "".join(rng.choice("ACGT") for _ in range(L))
    parts, pos = [], 0
    for _ in range(rng.randint(1, 3)):
        c = rng.randint(pos + 200, max(pos + 201, L - 200)) if pos + 201 < L - 200 else L
        parts.append(base[pos:c])
        pos = c
        unit = "".join(rng.choice("ACGT") for _ in range(rng.choice([3, 4, 5, 6, 7, 9, 11, 13, 17, 23, 31])))
        parts.append(mutate(rng, unit * rng.randint(30, 200), rng.choice([0.0, 0.005, 0.02]), 0.0))
    parts.append(base[pos:])
    ref = "".join(parts)
    return ref, mutate(rng, ref, rng.choice([0.0, 0.002, 0.01]), rng.choice([0.0, 0.001]))


def make_two_strand_repeat_pair(rng):
    """(reference records, query records) of 4 - 9 kb with tandem arrays that match on BOTH strands (palindromic arrays (U + rc U)^k,
    arrays followed by an inverted copy), the query lightly diverged, half of the time with an inverted segment, 1 - 2 records each.
    Under --maxmatch both strands of a record pair then carry dozens of clusters in the same reference region, clusters contained in
    alignments of their own strand (isShadowedCluster has something to find: hundreds of such tests per pair), backward searches that
    merge into older alignments — the configuration in which MUMmer's ONE walk over both strands of a record pair and a walk per
    strand see different "current" alignments (deviation (1) of rounds 3-4), and in which many clusters start on the same
    reference base (the canonical order of pga::chain_before)."""
    L = rng.randint(4000, 9000)
    base = "".join(rng.choice("ACGT") for _ in range(L))
    parts, pos = [], 0
    for _ in range(rng.randint(2, 4)):
        c = rng.randint(pos + 200, max(pos + 201, L - 200)) if pos + 201 < L - 200 else L
        parts.append(base[pos:c])
        pos = c
        unit = "".join(rng.choice("ACGT") for _ in range(rng.choice([5, 7, 9, 11, 13, 17, 23, 31, 47])))
        k = rng.randint(15, 80)
        kind = rng.random()
        if kind < 0.4:
            arr = (unit + _rc(unit)) * k
        elif kind < 0.7:
            arr = unit * k + _rc(unit * rng.randint(5, k))
        else:
            arr = unit * k
        parts.append(mutate(rng, arr, rng.choice([0.0, 0.005, 0.02]), 0.0))
    parts.append(base[pos:])
    ref = "".join(parts)
    qry = mutate(rng, ref, rng.choice([0.0, 0.002, 0.01, 0.03]), rng.choice([0.0, 0.001, 0.003]))
    if rng.random() < 0.5:
        a = rng.randint(100, len(qry) // 2)
        b = rng.randint(a + 200, len(qry) - 100)
        qry = qry[:a] + _rc(qry[a:b]) + qry[b:]

    def split(s, n):
        if n <= 1:
            return [s]
        cs = sorted(rng.sample(range(300, len(s) - 300), n - 1))
        return [s[x:y] for x, y in zip([0] + cs, cs + [len(s)])]

    return split(ref, rng.choice([1, 1, 2])), split(qry, rng.choice([1, 1, 2]))


TWO_STRAND_TRIALS = list(range(40, 62)) + list(range(150, 172))      # (seeds 13000003 + t; 51, 153, 158, 169: the walks' answers differ)


def expected_filtered(records):
    """oracle records [(ref id, qry id, rs, re, qs, qe, errors), ...] in the oracle's OUTPUT ORDER -> (keep flags, parse_delta tuple of
    the kept ones or None): delta-filter -1 and parse_delta as oracle/anim_oracle.py restates them (both pinned on the reference's
    files)."""
    import anim_oracle
    alns = [anim_oracle.Aln(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[6], 0, ()) for r in records]
    keep = anim_oracle.delta_filter_1to1(alns)
    kept = [a for a, k in zip(alns, keep) if k]
    return keep, (anim_oracle.parse_delta_records(kept) + (len(kept),) if kept else None)
