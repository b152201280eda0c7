"""ctypes binding of oracle/libanibcpu.so — the host statement of fragment mode (oracle/anib_cpu.cpp).
TEST / MEASUREMENT INFRASTRUCTURE ONLY."""
import ctypes
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
import oracle_build as _obuild  # noqa: E402

ROW_DTYPE = np.dtype([("frag", "<i4"), ("length", "<i4"), ("mismatch", "<i4"), ("gaps", "<i4"), ("nident", "<i4"), ("qlen", "<i4"),
                      ("qstart", "<i4"), ("qend", "<i4"), ("sstart", "<i4"), ("send", "<i4"), ("srec", "<i4"), ("score", "<i4")])


def anib_cpu_pair(query, subject, fragsize=1020):
    """query / subject: (uint8 sequence array, uint64 record offsets).  Rows of the BLAST-shaped table of the ordered pair
    (fragments of `query` against `subject`), best score first within a fragment."""
    lib = ctypes.CDLL(str(_obuild.build_anib_cpu()))
    lib.anib_cpu_pair.restype = ctypes.c_int64
    qs, qo = np.ascontiguousarray(query[0], dtype=np.uint8), np.ascontiguousarray(query[1], dtype=np.uint64)
    ss, so = np.ascontiguousarray(subject[0], dtype=np.uint8), np.ascontiguousarray(subject[1], dtype=np.uint64)
    cap = 4 * (len(qs) // fragsize + len(qo) + 8)
    out = np.zeros(cap, dtype=ROW_DTYPE)
    n = lib.anib_cpu_pair(ctypes.c_void_p(qs.ctypes.data), ctypes.c_void_p(qo.ctypes.data), ctypes.c_uint32(len(qo) - 1),
                          ctypes.c_void_p(ss.ctypes.data), ctypes.c_void_p(so.ctypes.data), ctypes.c_uint32(len(so) - 1),
                          ctypes.c_int32(fragsize), ctypes.c_void_p(out.ctypes.data), ctypes.c_uint64(cap))
    assert 0 <= n <= cap
    return out[:n]


def reduce_rows(rows):
    """parse_blast_tab's arithmetic (pyani/anib.py:641-665) over such rows: (aln_length, sim_errors, mean pident, kept rows).
    pident as BLAST prints it: 3 decimals."""
    aln = err = 0
    pids, kept, seen = [], [], set()
    for r in rows:
        f = int(r["frag"])
        if f in seen:
            continue
        alnlen = int(r["length"]) - int(r["gaps"])
        if alnlen / int(r["qlen"]) > 0.7 and (alnlen - int(r["mismatch"])) / int(r["qlen"]) > 0.3:
            seen.add(f)
            kept.append(r)
            aln += alnlen
            err += int(r["mismatch"]) + int(r["gaps"])
            pids.append(float("%.3f" % (100.0 * int(r["nident"]) / int(r["length"]))))
    return aln, err, (sum(pids) / len(pids) if pids else 0.0), kept
