"""Deterministic synthetic genome sets — thin ctypes wrapper over libpgsynth.so.

Test/bench DATA generation only; the generator is one C++ implementation (pyani_amd/csrc/synth.cpp) so
the same (seed, n, g, L) yields byte-identical genomes in tests, in bench.py and on the GPU box.
"""
import ctypes
from pathlib import Path
from typing import List, Tuple

import numpy as np

from . import build as _build

# named sets from SURVEY.md §8(d)
SETS = {
    "C2": dict(n=200, L=5_000_000, seed=20250228),
    "C4": dict(n=1000, L=5_000_000, seed=20250301),
    "CI": dict(n=8, L=50_000, seed=20250228),
}

_lib = None


def _load():
    global _lib
    if _lib is None:
        path = _build.SYNTH_LIB
        if not path.exists():
            _build.build_synth()
        lib = ctypes.CDLL(str(path))
        lib.pgs_genome.restype = ctypes.c_int64
        lib.pgs_genome.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64,
                                   ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint32,
                                   ctypes.POINTER(ctypes.c_uint32)]
        _lib = lib
    return _lib


def genome(seed: int, n: int, g: int, L: int) -> Tuple[np.ndarray, np.ndarray]:
    """Return (ascii uint8 array of the concatenated records, uint64 record offsets [n_rec+1])."""
    lib = _load()
    cap = L + L // 4 + 1024
    buf = np.empty(cap, dtype=np.uint8)
    off = np.zeros(8, dtype=np.uint64)
    nrec = ctypes.c_uint32(0)
    tot = lib.pgs_genome(seed, n, g, L, buf.ctypes.data, cap, off.ctypes.data, 7, ctypes.byref(nrec))
    if tot < 0:
        raise RuntimeError(f"pgs_genome failed ({tot})")
    return buf[:tot], off[: nrec.value + 1].copy()


def genome_name(g: int) -> str:
    return f"syn{g:05d}"


def write_fasta(path: Path, seq: np.ndarray, rec_off: np.ndarray, name: str, width: int = 70) -> None:
    """70-column FASTA, record ids <name>_r<k>."""
    with open(path, "wb") as fh:
        for r in range(len(rec_off) - 1):
            fh.write(f">{name}_r{r} synthetic\n".encode())
            s = seq[int(rec_off[r]): int(rec_off[r + 1])]
            full = (len(s) // width) * width
            if full:
                block = np.empty((full // width, width + 1), dtype=np.uint8)
                block[:, :width] = s[:full].reshape(-1, width)
                block[:, width] = 10
                fh.write(block.tobytes())
            if full < len(s):
                fh.write(s[full:].tobytes() + b"\n")


def write_set(outdir: Path, seed: int, n: int, L: int) -> List[Path]:
    outdir = Path(outdir)
    outdir.mkdir(parents=True, exist_ok=True)
    paths = []
    for g in range(n):
        seq, off = genome(seed, n, g, L)
        p = outdir / f"{genome_name(g)}.fna"
        write_fasta(p, seq, off, genome_name(g))
        paths.append(p)
    return paths
