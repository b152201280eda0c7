"""ctypes binding of oracle/libanimcpu.so — the host statement of the ANIm pair search (oracle/anim_cpu.cpp).
TEST / MEASUREMENT INFRASTRUCTURE ONLY: used by tests (GPU == scalar statement) and by bench.py's cpu_baseline leg."""
import ctypes
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
import oracle_build as _obuild  # noqa: E402

RESULT_DTYPE = np.dtype([("ref_aln_len", "<i8"), ("qry_aln_len", "<i8"), ("sim_errors", "<i8"), ("n_alignments", "<i8"),
                         ("identity", "<f8"), ("status", "<i4"), ("reserved", "<i4")])


def anim_cpu_pairs(genomes, ref_ids, qry_ids, maxmatch=False, filter_1to1=True, threads=0):
    """genomes: list of (uint8 sequence array, uint64 record offsets) as Engine.add_genome takes them (entries not named by
    any pair may be None).  Returns (structured result array like Engine.anim_pairs, per-pair CPU seconds)."""
    lib = ctypes.CDLL(str(_obuild.build_anim_cpu()))
    lib.anim_cpu_pairs.restype = ctypes.c_int
    n = len(genomes)
    keep = []
    seqs, offs = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
    nrec = np.zeros(n, dtype=np.uint32)
    for g, item in enumerate(genomes):
        if item is None:
            continue
        s = np.ascontiguousarray(item[0], dtype=np.uint8)
        o = np.ascontiguousarray(item[1], dtype=np.uint64)
        keep += [s, o]
        seqs[g], offs[g], nrec[g] = s.ctypes.data, o.ctypes.data, len(o) - 1
    r = np.ascontiguousarray(ref_ids, dtype=np.int32)
    q = np.ascontiguousarray(qry_ids, dtype=np.int32)
    out = np.zeros(len(r), dtype=RESULT_DTYPE)
    secs = np.zeros(len(r), dtype=np.float64)
    rc = lib.anim_cpu_pairs(seqs, offs, ctypes.c_void_p(nrec.ctypes.data), ctypes.c_uint32(n), ctypes.c_void_p(r.ctypes.data),
                            ctypes.c_void_p(q.ctypes.data), ctypes.c_uint32(len(r)), int(bool(maxmatch)), int(bool(filter_1to1)),
                            int(threads), ctypes.c_void_p(out.ctypes.data), ctypes.c_void_p(secs.ctypes.data))
    assert rc == 0
    return out, secs
