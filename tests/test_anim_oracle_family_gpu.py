"""GPU (through the C ABI), at BASELINE.json's full size: WHOLE families of the C4 benchmark — 25 descendants of one ancestor, 600
ordered 5 Mb pairs each, identity 0.72 ... 0.999 — against the independent nucmer oracle + the pure-Python delta-filter -1 / parse_delta
(tools/make_anim_family_hashes.py ran oracle/nucmer_oracle.cpp and oracle/anim_oracle.py on every ordered pair and kept per pair the
number of records, digests of the sorted records without and with their keep / drop decision, and the filtered tuple:
tests/golden/anim_oracle_family_digests.json.gz).  pg_anim_alignments_batch must give the same digests and pg_anim_pairs (filter on:
what bench.py's `value` counts and pyani's default job computes, pyani/anim.py:240-289, 292-411) the same tuple, bit for bit."""
import gzip
import hashlib
import json

import numpy as np
import pytest

from tests.conftest import GOLD

pytestmark = pytest.mark.gpu


def _digests(recs):
    rows = sorted(((int(r["ref_rec"]), int(r["qry_rec"]), int(r["rs"]), int(r["re"]), int(r["qs"]), int(r["qe"]), int(r["errors"])), int(r["kept"]) == 3)
                  for r in recs)
    h1 = hashlib.sha1("\n".join(",".join(map(str, r)) for r, _ in rows).encode()).hexdigest()
    h2 = hashlib.sha1("\n".join(",".join(map(str, r)) + ("+" if k else "-") for r, k in rows).encode()).hexdigest()
    return h1, h2


def test_whole_c4_families_at_full_size_equal_the_independent_oracle():
    from pyani_amd import synth
    from pyani_amd.engine import Engine
    with gzip.open(GOLD / "anim_oracle_family_digests.json.gz", "rt") as fh:
        S = json.load(fh)
    pairs = S["pairs"]
    used = sorted({g for p in pairs for g in p[:2]})
    bad = []
    with Engine(0) as eng:
        ids = {g: eng.add_genome(*synth.genome(S["seed"], S["n"], g, S["L"])) for g in used}
        r, q = [ids[p[0]] for p in pairs], [ids[p[1]] for p in pairs]
        res = eng.anim_pairs(r, q)
        for lo in range(0, len(pairs), 200):                     # (record batches of 200 pairs)
            off, recs, _, _ = eng.anim_alignments_batch(r[lo:lo + 200], q[lo:lo + 200])
            for k, (a, b, n_rec, n_kept, h1, h2, tup) in enumerate(pairs[lo:lo + 200]):
                mine = recs[int(off[k]):int(off[k + 1])]
                g1, g2 = _digests(mine)
                t = res[lo + k]
                got_t = None if int(t["n_alignments"]) == 0 else [int(t["ref_aln_len"]), int(t["qry_aln_len"]), float(t["identity"]).hex(),
                                                                   int(t["sim_errors"]), int(t["n_alignments"])]
                if (len(mine), int((mine["kept"] == 3).sum()), g1, g2, got_t) != (n_rec, n_kept, h1, h2, tup):
                    bad.append((a, b, len(mine), n_rec, g1 == h1, g2 == h2, got_t, tup))
    assert not bad, (len(bad), bad[:3])
    assert len(pairs) >= 600 and sum(p[2] for p in pairs) > 30_000
    # what the set covers (so that a regenerated file cannot silently shrink): identities from < 0.80 to > 0.995
    ident = np.array([float.fromhex(p[6][2]) for p in pairs if p[6]])
    assert ident.min() < 0.80 and ident.max() > 0.995, (ident.min(), ident.max())
