"""ctypes binding of oracle/liboracle.so (the CPU checker).  Test infrastructure only."""
import ctypes
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        vp, u32 = ctypes.c_void_p, ctypes.c_uint32
        lib.orc_tetra_counts.restype = ctypes.c_int
        lib.orc_tetra_counts.argtypes = [vp, vp, u32, vp, vp, vp]
        lib.orc_tetra_zscores.restype = None
        lib.orc_tetra_zscores.argtypes = [vp, vp, vp, u32, vp, vp]
        lib.orc_tetra_corr.restype = ctypes.c_int
        lib.orc_tetra_corr.argtypes = [vp, vp, u32, vp]

    def counts(self, seq: np.ndarray, rec_off: np.ndarray):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
        c2, c3, c4 = (np.zeros(n, dtype=np.uint64) for n in (16, 64, 256))
        rc = self.lib.orc_tetra_counts(seq.ctypes.data if len(seq) else None, rec_off.ctypes.data, len(rec_off) - 1,
                                       c2.ctypes.data, c3.ctypes.data, c4.ctypes.data)
        assert rc == 0
        return c2, c3, c4

    def zscores(self, c2, c3, c4):
        c2, c3, c4 = (np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, k) for a, k in ((c2, 16), (c3, 64), (c4, 256)))
        n = c4.shape[0]
        z = np.zeros((n, 256), dtype=np.float64)
        present = np.zeros((n, 256), dtype=np.uint8)
        self.lib.orc_tetra_zscores(c2.ctypes.data, c3.ctypes.data, c4.ctypes.data, n, z.ctypes.data, present.ctypes.data)
        return z, present

    def corr(self, z, present):
        z = np.ascontiguousarray(z, dtype=np.float64)
        present = np.ascontiguousarray(present, dtype=np.uint8)
        n = z.shape[0]
        out = np.zeros((n, n), dtype=np.float64)
        rc = self.lib.orc_tetra_corr(z.ctypes.data, present.ctypes.data, n, out.ctypes.data)
        return rc, out


def load() -> Oracle:
    import oracle_build   # oracle/oracle_build.py
    return Oracle(ctypes.CDLL(str(oracle_build.build_oracle())))


def read_fasta_arrays(path):
    """FASTA -> (uint8 concatenated sequence, uint64 record offsets) using the oracle's own reader."""
    import tetra_port
    parts, off = [], [0]
    for _, s in tetra_port.read_fasta(path):
        parts.append(s.encode("latin-1"))
        off.append(off[-1] + len(s))
    data = b"".join(parts)
    return np.frombuffer(data, dtype=np.uint8).copy(), np.array(off, dtype=np.uint64)


KMERS4 = ["".join(p) for p in __import__("itertools").product("ACGT", repeat=4)]


def z_dict(z_row, present_row):
    return {KMERS4[t]: float(z_row[t]) for t in range(256) if present_row[t]}
