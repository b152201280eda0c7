"""Engine — Python owner of one libpyani_gpu context (one per process and GPU).

Holds genomes resident in HBM (2-bit codes + 1-bit mask) and exposes the TETRA kernels.  numpy arrays are the
only currency across the ctypes boundary; nothing here computes on the CPU.
"""
import ctypes
from pathlib import Path
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data


class Engine:
    def __init__(self, device: int = 0):
        self.lib = _lib.load()
        h = ctypes.c_void_p()
        rc = self.lib.pg_create(ctypes.byref(h), device)
        if rc == _lib.PG_E_NODEVICE:
            raise _lib.PyaniGpuError(rc, "no HIP device visible: pyani_amd needs an MI355X (there is no CPU fallback)")
        if rc != 0:
            raise _lib.PyaniGpuError(rc, "pg_create failed")
        self._h = h
        self.device = device

    # -- plumbing ---------------------------------------------------------------------------------------------
    def _check(self, rc: int):
        if rc != 0:
            raise _lib.PyaniGpuError(rc, self.lib.pg_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self.lib.pg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def sync(self):
        self._check(self.lib.pg_sync(self._h))

    # -- genome store -------------------------------------------------------------------------------------------
    def add_genome(self, seq: np.ndarray, rec_off: Sequence[int]) -> int:
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        off = np.ascontiguousarray(rec_off, dtype=np.uint64)
        gid = ctypes.c_int32(-1)
        self._check(self.lib.pg_add_genome(self._h, seq.ctypes.data if seq.size else None, off.ctypes.data,
                                           max(len(off) - 1, 0), ctypes.byref(gid)))
        return gid.value

    def add_fasta(self, path) -> Tuple[int, int, int]:
        """Returns (genome id, total length = sum of record lengths, number of records)."""
        gid, tot, nrec = ctypes.c_int32(-1), ctypes.c_uint64(0), ctypes.c_uint32(0)
        self._check(self.lib.pg_add_fasta(self._h, str(path).encode(), ctypes.byref(gid), ctypes.byref(tot),
                                          ctypes.byref(nrec)))
        return gid.value, tot.value, nrec.value

    def add_fasta_batch(self, paths, threads: int = 0) -> List[Tuple[int, int, int]]:
        """Multithreaded ingest of many files; returns [(genome id, total length, number of records)] in input order."""
        paths = [str(p).encode() for p in paths]
        n = len(paths)
        arr = (ctypes.c_char_p * n)(*paths)
        ids = np.zeros(n, dtype=np.int32); tot = np.zeros(n, dtype=np.uint64); nrec = np.zeros(n, dtype=np.uint32)
        self._check(self.lib.pg_add_fasta_batch(self._h, ctypes.cast(arr, ctypes.c_void_p), n, int(threads), ids.ctypes.data,
                                                tot.ctypes.data, nrec.ctypes.data))
        return [(int(i), int(t), int(r)) for i, t, r in zip(ids, tot, nrec)]

    def genome_count(self) -> int:
        return self.lib.pg_genome_count(self._h)

    def genome_length(self, gid: int) -> Tuple[int, int]:
        tot, nrec = ctypes.c_uint64(0), ctypes.c_uint32(0)
        self._check(self.lib.pg_genome_length(self._h, gid, ctypes.byref(tot), ctypes.byref(nrec)))
        return tot.value, nrec.value

    def clear_genomes(self):
        self._check(self.lib.pg_clear_genomes(self._h))

    def upload(self):
        self._check(self.lib.pg_upload(self._h))

    def tetra_algorithmic_bytes(self, ids: Optional[Iterable[int]] = None) -> Tuple[int, int]:
        b, n = ctypes.c_uint64(0), ctypes.c_uint64(0)
        if ids is None:
            self._check(self.lib.pg_tetra_algorithmic_bytes(self._h, None, 0, ctypes.byref(b), ctypes.byref(n)))
        else:
            a = np.ascontiguousarray(list(ids), dtype=np.int32)
            self._check(self.lib.pg_tetra_algorithmic_bytes(self._h, a.ctypes.data, len(a), ctypes.byref(b), ctypes.byref(n)))
        return b.value, n.value

    # -- TETRA ------------------------------------------------------------------------------------------------------
    @staticmethod
    def _ids(ids) -> np.ndarray:
        return np.ascontiguousarray(list(ids), dtype=np.int32)

    def tetra_counts(self, ids) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        a = self._ids(ids)
        n = len(a)
        c2, c3, c4 = (np.zeros((n, k), dtype=np.uint64) for k in (16, 64, 256))
        self._check(self.lib.pg_tetra_counts(self._h, a.ctypes.data, n, c2.ctypes.data, c3.ctypes.data, c4.ctypes.data))
        return c2, c3, c4

    def tetra_zscores_from_counts(self, c2, c3, c4) -> Tuple[np.ndarray, np.ndarray]:
        c2, c3, c4 = (np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, k) for x, k in ((c2, 16), (c3, 64), (c4, 256)))
        n = c4.shape[0]
        z = np.zeros((n, 256), dtype=np.float64)
        present = np.zeros((n, 256), dtype=np.uint8)
        self._check(self.lib.pg_tetra_zscores(self._h, c2.ctypes.data, c3.ctypes.data, c4.ctypes.data, n, z.ctypes.data,
                                              present.ctypes.data))
        return z, present

    def tetra_corr(self, z: np.ndarray, present: np.ndarray) -> np.ndarray:
        z = np.ascontiguousarray(z, dtype=np.float64).reshape(-1, 256)
        present = np.ascontiguousarray(present, dtype=np.uint8).reshape(-1, 256)
        n = z.shape[0]
        out = np.zeros((n, n), dtype=np.float64)
        self._check(self.lib.pg_tetra_corr(self._h, z.ctypes.data, present.ctypes.data, n, out.ctypes.data))
        return out

    def tetra_matrix(self, ids, want_corr: bool = True) -> Tuple[np.ndarray, np.ndarray, Optional[np.ndarray]]:
        """Fused counts -> Z -> Pearson on the device.  Returns (z, present, corr or None)."""
        a = self._ids(ids)
        n = len(a)
        z = np.zeros((n, 256), dtype=np.float64)
        present = np.zeros((n, 256), dtype=np.uint8)
        corr = np.zeros((n, n), dtype=np.float64) if want_corr else None
        self._check(self.lib.pg_tetra_matrix(self._h, a.ctypes.data, n, z.ctypes.data, present.ctypes.data, _ptr(corr)))
        return z, present, corr

    def tetra_matrix_enqueue(self, ids_array: np.ndarray, fetch_z: bool = True):
        self._check(self.lib.pg_tetra_matrix_enqueue(self._h, ids_array.ctypes.data, len(ids_array), int(fetch_z)))

    def tetra_matrix_fetch(self, n: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        z = np.zeros((n, 256), dtype=np.float64)
        present = np.zeros((n, 256), dtype=np.uint8)
        corr = np.zeros((n, n), dtype=np.float64)
        self._check(self.lib.pg_tetra_matrix_fetch(self._h, n, z.ctypes.data, present.ctypes.data, corr.ctypes.data))
        return z, present, corr

    def tetra_zscores_dev(self, ids, d_z_ptr: int, d_present_ptr: int):
        a = self._ids(ids)
        self._check(self.lib.pg_tetra_zscores_dev(self._h, a.ctypes.data, len(a), d_z_ptr, d_present_ptr))

    def tetra_corr_rows_dev(self, d_z_ptr: int, d_present_ptr: int, n: int, row0: int, nrows: int, d_out_ptr: int):
        self._check(self.lib.pg_tetra_corr_rows_dev(self._h, d_z_ptr, d_present_ptr, n, row0, nrows, d_out_ptr))

    # -- ANIm -------------------------------------------------------------------------------------------------------
    ANIM_DTYPE = np.dtype([("ref_aln_len", "<i8"), ("qry_aln_len", "<i8"), ("sim_errors", "<i8"), ("n_alignments", "<i8"),
                           ("identity", "<f8"), ("status", "<i4"), ("reserved", "<i4")])

    def anim_pairs(self, ref_ids, qry_ids, filter_1to1: bool = True, maxmatch: bool = False) -> np.ndarray:
        """One record per ORDERED pair: ref = nucmer's reference (pyani's query genome), qry = nucmer's query."""
        r, q = self._ids(ref_ids), self._ids(qry_ids)
        if len(r) != len(q):
            raise ValueError("ref_ids and qry_ids must have the same length")
        out = np.zeros(len(r), dtype=self.ANIM_DTYPE)
        self._check(self.lib.pg_anim_pairs(self._h, r.ctypes.data, q.ctypes.data, len(r), int(maxmatch), int(filter_1to1),
                                           out.ctypes.data))
        return out

    def anim_pairs_enqueue(self, ref_ids, qry_ids, filter_1to1: bool = True, maxmatch: bool = False):
        """Non-blocking anim_pairs (pg_anim_pairs_enqueue): returns a ticket at once; at most two calls are in flight per engine, the
        tail of one overlapping the front of the next (pyani's pool keeps its cores busy across job boundaries,
        run_multiprocessing.py:130-144).  Every ticket must be given to anim_pairs_fetch."""
        r, q = self._ids(ref_ids), self._ids(qry_ids)
        if len(r) != len(q):
            raise ValueError("ref_ids and qry_ids must have the same length")
        t = ctypes.c_uint64(0)
        self._check(self.lib.pg_anim_pairs_enqueue(self._h, r.ctypes.data, q.ctypes.data, len(r), int(maxmatch), int(filter_1to1), ctypes.byref(t)))
        return (int(t.value), len(r))

    def anim_pairs_fetch(self, ticket) -> np.ndarray:
        """Waits for the enqueued call and returns its records (the caller's pair order), exactly anim_pairs' result."""
        tid, n = ticket
        out = np.zeros(n, dtype=self.ANIM_DTYPE)
        self._check(self.lib.pg_anim_pairs_fetch(self._h, ctypes.c_uint64(tid), out.ctypes.data, n))
        return out

    def anim_set_workers(self, workers: int = 2) -> None:
        """Host worker threads (streams) sharing one anim_pairs / anib_pairs call: 1 ... 4 (pyani's --workers inside one device)."""
        self._check(self.lib.pg_anim_set_workers(self._h, int(workers)))

    def anim_counters(self, reset: bool = False) -> np.ndarray:
        """The nucmer extender's engine counters since the last reset (include/pyani_gpu.h, pg_anim_counters): uint64[64]."""
        out = np.zeros(64, dtype=np.uint64)
        self._check(self.lib.pg_anim_counters(self._h, out.ctypes.data, int(bool(reset))))
        return out

    def anim_set_batch_budget(self, max_pairs: int, max_matches: int) -> None:
        self._check(self.lib.pg_anim_set_batch_budget(self._h, int(max_pairs), int(max_matches)))

    ALN_DTYPE = np.dtype([("ref_rec", "<i4"), ("qry_rec", "<i4"), ("rs", "<i4"), ("re", "<i4"), ("qs", "<i4"), ("qe", "<i4"),
                          ("errors", "<i4"), ("kept", "<i4")])

    def anim_pair_alignments(self, ref_id: int, qry_id: int) -> np.ndarray:
        """Alignment records of one ordered pair (what nucmer's .delta would hold; kept == 3: survives delta-filter -1)."""
        n = ctypes.c_uint32(0)
        out = np.zeros(4096, dtype=self.ALN_DTYPE)
        self._check(self.lib.pg_anim_pair_alignments(self._h, int(ref_id), int(qry_id), out.ctypes.data, len(out), ctypes.byref(n)))
        if n.value > len(out):
            out = np.zeros(n.value, dtype=self.ALN_DTYPE)
            self._check(self.lib.pg_anim_pair_alignments(self._h, int(ref_id), int(qry_id), out.ctypes.data, len(out), ctypes.byref(n)))
        return out[:n.value].copy()

    def anim_alignments_batch(self, ref_ids, qry_ids, maxmatch: bool = False, with_indels: bool = False):
        """Alignment records of MANY ordered pairs in one call (pg_anim_alignments_batch): what pyani's nucmer jobs leave in
        their .delta files for a whole run (anim.py:240-289).  Returns (offsets, records, indel_offsets, indels): pair i owns
        records[offsets[i]:offsets[i + 1]] (ALN_DTYPE; kept == 3: survives delta-filter -1); with_indels=True adds the GPU
        traceback pass: record k's .delta indel offset list is indels[indel_offsets[k]:indel_offsets[k + 1]] (without the
        terminating 0) and the records of a pair come in MUMmer's own output order; else the last two are None."""
        r = np.ascontiguousarray(list(ref_ids), dtype=np.int32)
        q = np.ascontiguousarray(list(qry_ids), dtype=np.int32)
        if len(r) != len(q):
            raise ValueError("ref_ids and qry_ids must have the same length")
        offsets = np.zeros(len(r) + 1, dtype=np.uint64)
        n_ind = ctypes.c_uint64(0)
        self._check(self.lib.pg_anim_alignments_batch(self._h, r.ctypes.data, q.ctypes.data, len(r), int(bool(maxmatch)), int(bool(with_indels)),
                                                      offsets.ctypes.data, ctypes.byref(n_ind)))
        recs = np.zeros(int(offsets[-1]), dtype=self.ALN_DTYPE)
        ioff = np.zeros(len(recs) + 1, dtype=np.uint64) if with_indels else None
        ind = np.zeros(int(n_ind.value), dtype=np.int64) if with_indels else None
        self._check(self.lib.pg_anim_alignments_read(self._h, recs.ctypes.data if len(recs) else None, ioff.ctypes.data if with_indels else None,
                                                     ind.ctypes.data if with_indels and len(ind) else None))
        return offsets, recs, ioff, ind

    def anim_reduce(self, pairs, apply_filter: bool = False) -> np.ndarray:
        """pairs: list of per-pair record lists [(rseq, qseq, rs, re, qs, qe, errors), ...] in MUMmer coordinates
        (1-based closed, qs > qe on the reverse strand; rseq/qseq = sequence ordinals within the pair)."""
        offsets = np.zeros(len(pairs) + 1, dtype=np.uint64)
        for k, recs in enumerate(pairs):
            offsets[k + 1] = offsets[k] + len(recs)
        flat = np.array([r for recs in pairs for r in recs], dtype=np.int32).reshape(-1, 7)
        cols = [np.ascontiguousarray(flat[:, c]) for c in range(7)]
        out = np.zeros(len(pairs), dtype=self.ANIM_DTYPE)
        self._check(self.lib.pg_anim_reduce(self._h, len(pairs), offsets.ctypes.data, *(c.ctypes.data for c in cols),
                                            int(apply_filter), out.ctypes.data))
        return out

    def anib_reduce(self, pairs):
        """pairs: list of (n_frags, rows) with rows = [(frag ordinal, length, mismatch, gaps, qlen, pident), ...] in
        file order.  Returns (aln_length int64[], sim_errors int64[], mean pident float64[])."""
        n = len(pairs)
        offsets = np.zeros(n + 1, dtype=np.uint64)
        nfr = np.zeros(max(n, 1), dtype=np.uint32)
        for k, (nf, rows) in enumerate(pairs):
            offsets[k + 1] = offsets[k] + len(rows)
            nfr[k] = nf
        flat = [r for _, rows in pairs for r in rows]
        ints = np.array([r[:5] for r in flat], dtype=np.int32).reshape(-1, 5)
        cols = [np.ascontiguousarray(ints[:, c]) for c in range(5)]
        pid = np.ascontiguousarray([r[5] for r in flat], dtype=np.float64)
        aln, err, out = np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.int64), np.zeros(n, dtype=np.float64)
        self._check(self.lib.pg_anib_reduce(self._h, n, offsets.ctypes.data, nfr.ctypes.data, cols[0].ctypes.data,
                                            cols[1].ctypes.data, cols[2].ctypes.data, cols[3].ctypes.data, cols[4].ctypes.data,
                                            pid.ctypes.data, aln.ctypes.data, err.ctypes.data, out.ctypes.data))
        return aln, err, out

    # -- ANIb fragment mode ---------------------------------------------------------------------------------------
    ANIB_DTYPE = np.dtype([("aln_length", "<i8"), ("sim_errors", "<i8"), ("pid", "<f8"), ("n_frags", "<i4"), ("n_kept", "<i4"),
                           ("status", "<i4"), ("reserved", "<i4")])
    ANIB_ROW_DTYPE = np.dtype([("frag", "<i4"), ("length", "<i4"), ("mismatch", "<i4"), ("gaps", "<i4"), ("nident", "<i4"),
                               ("qlen", "<i4"), ("qstart", "<i4"), ("qend", "<i4"), ("sstart", "<i4"), ("send", "<i4"), ("srec", "<i4"),
                               ("score", "<i4")])

    def anib_pairs(self, qry_ids, sbj_ids, fragsize: int = 1020) -> np.ndarray:
        """One record per ORDERED pair: the fragments of genome qry against genome sbj (pyani's blastn job + parse_blast_tab)."""
        q, s = self._ids(qry_ids), self._ids(sbj_ids)
        if len(q) != len(s):
            raise ValueError("qry_ids and sbj_ids must have the same length")
        out = np.zeros(len(q), dtype=self.ANIB_DTYPE)
        self._check(self.lib.pg_anib_pairs(self._h, q.ctypes.data, s.ctypes.data, len(q), int(fragsize), out.ctypes.data))
        return out

    def anib_pair_rows(self, qry_id: int, sbj_id: int, fragsize: int = 1020) -> np.ndarray:
        """The BLAST-shaped table of one ordered pair (<= 4 rows per fragment, best score first)."""
        n = ctypes.c_uint32(0)
        nfr, _ = self.genome_length(qry_id)
        out = np.zeros(4 * (nfr // max(1, fragsize) + 4096), dtype=self.ANIB_ROW_DTYPE)
        self._check(self.lib.pg_anib_pair_rows(self._h, int(qry_id), int(sbj_id), int(fragsize), out.ctypes.data, len(out), ctypes.byref(n)))
        if n.value > len(out):
            out = np.zeros(n.value, dtype=self.ANIB_ROW_DTYPE)
            self._check(self.lib.pg_anib_pair_rows(self._h, int(qry_id), int(sbj_id), int(fragsize), out.ctypes.data, len(out), ctypes.byref(n)))
        return out[:n.value].copy()

    # -- measurement ----------------------------------------------------------------------------------------------
    # -- sketch mode (fastANI-shaped estimate; never mixed into the exact results) ----------------------------------------------
    SKETCH_DTYPE = np.dtype([("ani", "<f8"), ("matches", "<i4"), ("fragments", "<i4"), ("status", "<i4"), ("reserved", "<i4")])

    def sketch_pairs(self, qry_ids, ref_ids, frag_len: int = 3000, scale: int = 16, min_fraction: float = 0.2) -> np.ndarray:
        """pg_sketch_pairs: one record per ORDERED pair (query fragmented, reference as a k-mer set): ani (a fraction), matches,
        fragments, status (0 / 1 = fewer than min_fraction of the fragments matched: fastANI writes no line then)."""
        q, r = self._ids(qry_ids), self._ids(ref_ids)
        if len(q) != len(r):
            raise ValueError("qry_ids and ref_ids must have the same length")
        out = np.zeros(len(q), dtype=self.SKETCH_DTYPE)
        self._check(self.lib.pg_sketch_pairs(self._h, q.ctypes.data, r.ctypes.data, len(q), int(frag_len), int(scale), float(min_fraction),
                                             out.ctypes.data))
        return out

    def profile_enable(self, on: bool = True):
        self._check(self.lib.pg_profile_enable(self._h, int(on)))

    def profile_config(self, kernel_mask: int = 0xFFFFFFFF, every_n: int = 1):
        self._check(self.lib.pg_profile_config(self._h, kernel_mask, every_n))

    def profile_reset(self):
        self._check(self.lib.pg_profile_reset(self._h))

    def profile_get(self, which: int) -> Tuple[float, int]:
        ms, n = ctypes.c_double(0), ctypes.c_uint64(0)
        self._check(self.lib.pg_profile_get(self._h, which, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def kernel_name(self, which: int) -> str:
        return self.lib.pg_kernel_name(which).decode()


_default: List[Optional[Engine]] = [None]


def default_engine() -> Engine:
    """Process-wide engine on LOCAL_RANK's GPU (created on first use)."""
    import os
    if _default[0] is None:
        _default[0] = Engine(int(os.environ.get("LOCAL_RANK", "0")))
    return _default[0]
