"""oracle/sketch_oracle.py — numpy restatement of the SKETCH mode's definition (pyani_amd/csrc/pg_sketch_core.h).
TEST INFRASTRUCTURE ONLY: nothing under pyani_amd/ imports this file.

The sketch mode stands in for pyani's fastANI wrapper (pyani/fastani.py:193-270: `fastANI -q query -r ref --fragLen 3000 -k 16
--minFraction 0.2` -> one line: query, reference, ANI estimate, matching fragments, query fragments).  fastANI itself is third-party
and absent from /root/reference and from this image, and its estimator (MashMap mapping + Mash distance per fragment) is NOT what is
restated here: the product's sketch mode is an estimator of its own (FracMinHash containment per fragment, identity = C^(1/16)) in
fastANI's output shape — "parity unpinned" against fastANI by construction; what this file pins is that the GPU computes exactly the
estimator the header defines (bit for bit), and tests/test_sketch_gpu.py prices the estimate against the exact ANIm engine.
The only reference-held datum on this path is the fixture line of tests/fixtures/fastani/ecoli_vs_shiga.fastani, which pins the FILE
FORMAT (tests/test_sketch.py: parse_fastani_file reads it into the reference's ComparisonResult values)."""
import math

import numpy as np

K = 16
MIN_IDENTITY = 0.80
_CODE = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i
    _CODE[ord(chr(_c).lower())] = _i


def mix32(h):
    h = np.asarray(h, dtype=np.uint64)
    h ^= h >> np.uint64(16); h = (h * np.uint64(0x85ebca6b)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13); h = (h * np.uint64(0xc2b2ae35)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    return h


def record_kmers(seq_bytes):
    """(start positions, canonical 16-mers) of every window of 16 unambiguous bases of one record (numpy uint8 of ASCII)."""
    codes = _CODE[np.asarray(seq_bytes, dtype=np.uint8)]
    n = len(codes)
    if n < K:
        return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.uint64)
    ok = codes < 4
    c = np.where(ok, codes, 0).astype(np.uint64)
    fwd = np.zeros(n - K + 1, dtype=np.uint64)
    rc = np.zeros(n - K + 1, dtype=np.uint64)
    bad = np.zeros(n - K + 1, dtype=np.int64)
    for i in range(K):      # first base in the HIGH bits; rc: the complement of the LAST base in the high bits
        fwd |= c[i:n - K + 1 + i] << np.uint64(2 * (K - 1 - i))
        rc |= (np.uint64(3) - c[K - 1 - i:n - i]) << np.uint64(2 * (K - 1 - i))
        bad += (~ok[i:n - K + 1 + i]).astype(np.int64)
    pos = np.nonzero(bad == 0)[0]
    return pos, np.minimum(fwd, rc)[pos]


def genome_sketch(seq, rec_off, frag_len=3000, scale=16):
    """seq: uint8 ASCII of the records back to back, rec_off: record boundaries (as pyani_amd.synth.genome returns them).
    -> (set of sampled canonical k-mers, per-fragment arrays of sampled k-mer occurrences, number of fragments)"""
    kset, frags = set(), []
    for r in range(len(rec_off) - 1):
        rec = np.asarray(seq[int(rec_off[r]):int(rec_off[r + 1])])
        pos, km = record_kmers(rec)
        keep = (mix32(km) & np.uint64(scale - 1)) == 0
        pos, km = pos[keep], km[keep]
        kset.update(int(x) for x in km)
        n_full = len(rec) // frag_len
        j = pos // frag_len
        inside = (j < n_full) & ((pos - j * frag_len + K) <= frag_len)
        for f in range(n_full):
            frags.append(km[inside & (j == f)])
    return kset, frags, len(frags)


def sketch_pair(query_sketch, ref_sketch, min_fraction=0.2):
    """(ani fraction, matches, fragments, status) of one ordered pair: the definition, in its order (fragments ascending)."""
    _, frags, nf = query_sketch
    rset = ref_sketch[0]
    total, matches = 0.0, 0
    for occ in frags:
        n = len(occ)
        if n == 0:
            continue
        h = sum(1 for x in occ if int(x) in rset)
        if h < 2:
            continue
        ident = math.sqrt(math.sqrt(math.sqrt(math.sqrt(h / n))))
        if ident >= MIN_IDENTITY:
            total = total + ident
            matches += 1
    enough = matches > 0 and float(matches) >= min_fraction * float(nf)
    return (total / matches if enough else 0.0, matches, nf, 0 if enough else 1)
