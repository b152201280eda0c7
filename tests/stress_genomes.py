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


def expected_filtered(records):
    """oracle records [(ref id, qry id, rs, re, qs, qe, errors), ...] in the oracle's OUTPUT ORDER -> (keep flags, parse_delta tuple of
    the kept ones or None): delta-filter -1 and parse_delta as oracle/anim_oracle.py restates them (both pinned on the reference's
    files)."""
    import anim_oracle
    alns = [anim_oracle.Aln(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[6], 0, ()) for r in records]
    keep = anim_oracle.delta_filter_1to1(alns)
    kept = [a for a, k in zip(alns, keep) if k]
    return keep, (anim_oracle.parse_delta_records(kept) + (len(kept),) if kept else None)
