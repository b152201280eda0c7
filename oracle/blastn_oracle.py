"""ctypes binding of oracle/libblastnoracle.so — the INDEPENDENT restatement of blastn for pyani's ANIb command line
(oracle/blastn_oracle.cpp; pyani/anib.py:451-471).  TEST / MEASUREMENT INFRASTRUCTURE ONLY."""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
import oracle_build as _obuild  # noqa: E402

ROW_DTYPE = np.dtype([("frag", "<i4"), ("length", "<i4"), ("mismatch", "<i4"), ("gaps", "<i4"), ("nident", "<i4"), ("qlen", "<i4"),
                      ("qstart", "<i4"), ("qend", "<i4"), ("sstart", "<i4"), ("send", "<i4"), ("srec", "<i4"), ("score", "<i4")])


def blastn_pair(query, subject, fragsize=1020, first_only=False, threads=None):
    """query / subject: (uint8 sequence array, uint64 record offsets).  The rows BLAST+ prints for the `fragsize`-nt fragments of
    `query` against a database of `subject` (every HSP of the reported subject record, BLAST's order; `first_only`: the top HSP)."""
    lib = ctypes.CDLL(str(_obuild.build_blastn_oracle()))
    lib.blastn_oracle_pair.restype = ctypes.c_int64
    qs, qo = np.ascontiguousarray(query[0], dtype=np.uint8), np.ascontiguousarray(query[1], dtype=np.uint64)
    ss, so = np.ascontiguousarray(subject[0], dtype=np.uint8), np.ascontiguousarray(subject[1], dtype=np.uint64)
    cap = 64 * (len(qs) // fragsize + len(qo) + 8)
    out = np.zeros(cap, dtype=ROW_DTYPE)
    threads = threads or min(16, os.cpu_count() or 1)
    n = lib.blastn_oracle_pair(ctypes.c_void_p(qs.ctypes.data), ctypes.c_void_p(qo.ctypes.data), ctypes.c_uint32(len(qo) - 1),
                               ctypes.c_void_p(ss.ctypes.data), ctypes.c_void_p(so.ctypes.data), ctypes.c_uint32(len(so) - 1),
                               ctypes.c_int32(fragsize), ctypes.c_void_p(out.ctypes.data), ctypes.c_uint64(cap),
                               ctypes.c_uint32(1 if first_only else 0), ctypes.c_int32(threads))
    assert 0 <= n <= cap, (n, cap)
    return out[:n]
