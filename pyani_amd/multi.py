"""MultiEngine — one process, several GPUs: the in-product answer to pyani's `--workers` (SURVEY.md §8 e).

pyani spreads its nucmer / blastn jobs over `--workers` CPU cores with a multiprocessing pool
(`scripts/subcommands/subcmd_anim.py:392-396` -> `run_multiprocessing.run_dependency_graph(jobs, workers=...)`,
`run_multiprocessing.py:113-152`): independent jobs, handed out as workers fall idle.  Here a "worker" is a GPU: one Engine (one
libpyani_gpu context) per device, every genome resident on every device (1.9 GB for 1000 x 5 Mb of the 288 GB each has), and the
ordered-pair list cut into chunks that the devices PULL from a shared queue — one host thread per device, the ctypes calls release
the GIL — so a device that drew cheap chunks (unrelated pairs cost microseconds, related ones milliseconds: ~60 x apart) simply
comes back for more.  No data-path collective: results are merged on the host in the caller's order.  The multi-process RCCL
path (`pyani_amd/parallel.py`, `bench.py --gpus N`: one process per GPU, one all-gather per step) stays what the driver's scaling
run measures; this class is what `run_anim(..., devices=[...])` and the module functions use.

A chunk keeps a pair and its reverse together (they share their seeding inside one `pg_anim_pairs` call, DESIGN.md §5c "Roles") and
keeps the pairs of one reference genome together (its seed table is built once per call).  Results do not depend on the number of
devices or on which device ran what (tests/test_parallel_multi_gpu.py: two engines on GPU 0 == one).
"""
import threading
from collections import defaultdict
from typing import Iterable, List, Optional, Sequence

import numpy as np

from .engine import Engine


def _chunks_by_hub(a: np.ndarray, b: np.ndarray, target: int) -> List[np.ndarray]:
    """Indices of the pair list grouped so that (x, y) and (y, x) share a chunk and the pairs of one hub genome stay together;
    chunks of about `target` pairs, heaviest first is not knowable, so: in hub order."""
    n = len(a)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    # the hub of an unordered pair {lo, hi}: lo when lo + hi is even, hi otherwise — every genome hubs half of its partners
    hub = np.where(((lo + hi) & 1) == 0, lo, hi)
    order = np.argsort(hub, kind="stable")
    bounds = np.flatnonzero(np.diff(hub[order])) + 1
    groups = np.split(order, bounds)
    out, cur, size = [], [], 0
    for g in groups:
        cur.append(g)
        size += len(g)
        if size >= target:
            out.append(np.concatenate(cur)); cur, size = [], 0
    if cur:
        out.append(np.concatenate(cur))
    assert sum(len(c) for c in out) == n
    return out


def _static_parts_by_hub(a: np.ndarray, b: np.ndarray, parts: int) -> List[np.ndarray]:
    """The pair list dealt into `parts` shares, ONE call each: the hub groups of _chunks_by_hub in a fixed scrambled order
    (multiplicative hash of the hub id: related genomes sit next to each other in sorted input lists and at a fixed stride in the
    synthetic sets), dealt round-robin.  Why one big call per device and not a queue of small ones (measured on MI355X, C4,
    profiles/r05_deal_probe.json): a call's kernels each end with their longest item — one forced re-alignment can take 100 ms — so a
    call of 100 hub rows costs 20 ms per row, 25 rows 26 ms, 6 rows 51 ms, 2 rows 72 ms; the dynamic deal of round 4 (chunks down to
    2 rows) ran at 48 % of the plain rate, while the scrambled static shares of 8 ranks are within 2 - 5 % of each other."""
    n = len(a)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    hub = np.where(((lo + hi) & 1) == 0, lo, hi)
    order = np.argsort(hub, kind="stable")
    bounds = np.flatnonzero(np.diff(hub[order])) + 1
    groups = np.split(order, bounds)
    groups.sort(key=lambda g: ((int(hub[g[0]]) * 0x9E3779B1) & 0xFFFFFFFF, int(hub[g[0]])) if len(g) else (0, 0))
    out = [np.concatenate(groups[p::parts]) if groups[p::parts] else np.zeros(0, dtype=np.int64) for p in range(parts)]
    out = [c for c in out if len(c)]
    assert sum(len(c) for c in out) == n
    return out


class MultiEngine:
    """Engines on several devices behind the Engine calls the module functions use: genome store (replicated), anim_pairs / anib_pairs /
    anim_alignments_batch (pairs pulled in chunks by the devices), tetra_counts / tetra_matrix (genomes counted in shards)."""

    def __init__(self, devices: Sequence[int], chunk_pairs: int = 0):
        if not devices:
            raise ValueError("MultiEngine needs at least one device")
        self.devices = list(devices)
        self.engines = [Engine(d) for d in self.devices]
        self.chunk_pairs = int(chunk_pairs)
        self.last_chunks_per_engine: List[int] = []

    # -- plumbing -------------------------------------------------------------------------------------------------------
    def close(self):
        for e in self.engines:
            e.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _all(self, fn):
        """fn(engine) on every engine, one thread each; returns the results in engine order, re-raises the first error."""
        out, err = [None] * len(self.engines), []

        def run(k):
            try:
                out[k] = fn(self.engines[k])
            except BaseException as e:  # noqa: BLE001 — re-raised below
                err.append(e)
        ts = [threading.Thread(target=run, args=(k,)) for k in range(len(self.engines))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if err:
            raise err[0]
        return out

    # -- genome store: replicated ---------------------------------------------------------------------------------------
    def add_genome(self, seq, rec_off) -> int:
        ids = [e.add_genome(seq, rec_off) for e in self.engines]
        assert len(set(ids)) == 1
        return ids[0]

    def add_fasta(self, path):
        res = [e.add_fasta(path) for e in self.engines]
        assert len({r[0] for r in res}) == 1
        return res[0]

    def add_fasta_batch(self, paths, threads: int = 0):
        paths = list(paths)
        res = self._all(lambda e: e.add_fasta_batch(paths, threads))
        assert all(r == res[0] for r in res)
        return res[0]

    def genome_count(self) -> int:
        return self.engines[0].genome_count()

    def genome_length(self, gid: int):
        return self.engines[0].genome_length(gid)

    def clear_genomes(self):
        self._all(lambda e: e.clear_genomes())

    def upload(self):
        self._all(lambda e: e.upload())

    def anim_set_workers(self, workers: int = 2):
        """Engine.anim_set_workers on every device (host worker threads / streams per device)."""
        for e in self.engines:
            e.anim_set_workers(workers)

    def anim_counters(self, reset: bool = False) -> np.ndarray:
        """Engine.anim_counters summed over the devices."""
        return np.sum([e.anim_counters(reset) for e in self.engines], axis=0).astype(np.uint64)

    # -- the work queue ----------------------------------------------------------------------------------------------------
    def _pull(self, chunks: List[np.ndarray], run_chunk, out: np.ndarray):
        lock = threading.Lock()
        first = {id(e): i for i, e in enumerate(self.engines)}      # engine i starts on chunk i (the static shares: one each) ...
        nxt = [min(len(self.engines), len(chunks))]                 # ... and then pulls what is left
        taken = defaultdict(int)

        def worker(e):
            k = first[id(e)]
            while True:
                if k >= len(chunks):
                    return
                idx = chunks[k]
                out[idx] = run_chunk(e, idx)
                taken[id(e)] += 1
                with lock:
                    k = nxt[0]
                    nxt[0] += 1
        self._all(worker)
        self.last_chunks_per_engine = [taken[id(e)] for e in self.engines]

    def _target(self, n: int) -> int:
        if self.chunk_pairs > 0:
            return self.chunk_pairs
        # ~24 chunks per device so that the tail is short, but never launches so small that the GPU idles inside them
        return max(2048, n // (24 * len(self.engines)) + 1)

    def anim_pairs(self, ref_ids, qry_ids, filter_1to1: bool = True, maxmatch: bool = False) -> np.ndarray:
        r = np.ascontiguousarray(list(ref_ids), dtype=np.int32)
        q = np.ascontiguousarray(list(qry_ids), dtype=np.int32)
        if len(r) != len(q):
            raise ValueError("ref_ids and qry_ids must have the same length")
        out = np.zeros(len(r), dtype=Engine.ANIM_DTYPE)
        if len(r):
            # one scrambled share per device, ONE call each (_static_parts_by_hub says why); chunk_pairs > 0: a queue of chunks of
            # that size, pulled by the devices (jobs whose cost is concentrated in a few genomes; the tests)
            chunks = (_chunks_by_hub(r, q, self.chunk_pairs) if self.chunk_pairs > 0
                      else _static_parts_by_hub(r, q, max(1, min(len(self.engines), len(r) // 64))))
            self._pull(chunks, lambda e, idx: e.anim_pairs(r[idx], q[idx], filter_1to1=filter_1to1, maxmatch=maxmatch), out)
        return out

    def anib_pairs(self, qry_ids, sbj_ids, fragsize: int = 1020) -> np.ndarray:
        qa = np.ascontiguousarray(list(qry_ids), dtype=np.int32)
        sa = np.ascontiguousarray(list(sbj_ids), dtype=np.int32)
        if len(qa) != len(sa):
            raise ValueError("qry_ids and sbj_ids must have the same length")
        out = np.zeros(len(qa), dtype=Engine.ANIB_DTYPE)
        if len(qa):
            # fragment mode: the fragmented (query) genome is the expensive side to set up — keep its pairs together
            order = np.argsort(qa, kind="stable")
            bounds = np.flatnonzero(np.diff(qa[order])) + 1
            groups, chunks, cur, size = np.split(order, bounds), [], [], 0
            target = max(64, len(qa) // (24 * len(self.engines)) + 1) if self.chunk_pairs <= 0 else self.chunk_pairs
            for g in groups:
                cur.append(g); size += len(g)
                if size >= target:
                    chunks.append(np.concatenate(cur)); cur, size = [], 0
            if cur:
                chunks.append(np.concatenate(cur))
            self._pull(chunks, lambda e, idx: e.anib_pairs(qa[idx], sa[idx], fragsize), out)
        return out

    # -- alignment records (run_anim(write_output=True): the .delta / .filter files) -----------------------------------------------
    def anim_alignments_batch(self, ref_ids, qry_ids, maxmatch: bool = False, with_indels: bool = False):
        """Engine.anim_alignments_batch over all devices: the pair list is cut by hub genome as for anim_pairs, the devices pull
        chunks, and the records (and indel lists) are put back together in the caller's pair order."""
        r = np.ascontiguousarray(list(ref_ids), dtype=np.int32)
        q = np.ascontiguousarray(list(qry_ids), dtype=np.int32)
        if len(r) != len(q):
            raise ValueError("ref_ids and qry_ids must have the same length")
        n = len(r)
        if n == 0 or len(self.engines) == 1:
            return self.engines[0].anim_alignments_batch(r, q, maxmatch=maxmatch, with_indels=with_indels)
        # traceback launches keep their walks' pieces on the host: smaller chunks
        chunks = _chunks_by_hub(r, q, max(16 if with_indels else 256, n // (8 * len(self.engines)) + 1))
        parts = [None] * len(chunks)
        lock = threading.Lock()
        nxt = [0]

        def worker(e):
            while True:
                with lock:
                    k = nxt[0]
                    nxt[0] += 1
                if k >= len(chunks):
                    return
                idx = chunks[k]
                parts[k] = e.anim_alignments_batch(r[idx], q[idx], maxmatch=maxmatch, with_indels=with_indels)
        self._all(worker)
        return merge_alignment_parts(n, chunks, parts, with_indels)

    # -- TETRA: genomes are independent -> counted in shards, the small N x N matrix on one device ----------------------------
    def _id_shards(self, ids) -> List[np.ndarray]:
        ids = np.ascontiguousarray(list(ids), dtype=np.int32)
        if len(ids) == 0:
            return [ids]
        lens = np.array([self.engines[0].genome_length(int(g))[0] for g in ids], dtype=np.int64)
        # contiguous blocks of about equal bases (results are concatenated in order)
        cuts = np.searchsorted(np.cumsum(lens), np.linspace(0, lens.sum(), len(self.engines) + 1)[1:-1], side="left")
        return [b for b in np.split(ids, cuts) if len(b)]

    def tetra_counts(self, ids):
        shards = self._id_shards(ids)
        # one thread per engine through _all: an engine error (PG_E_NOMEM, a bad id) is re-raised here, not lost in a thread
        res = self._all(lambda e: e.tetra_counts(shards[self.engines.index(e)]) if self.engines.index(e) < len(shards) else None)
        res = [x for x in res if x is not None]
        return tuple(np.concatenate([x[j] for x in res]) for j in range(3))

    def tetra_matrix(self, ids, want_corr: bool = True):
        c2, c3, c4 = self.tetra_counts(ids)
        z, present = self.engines[0].tetra_zscores_from_counts(c2, c3, c4)
        corr = self.engines[0].tetra_corr(z, present) if want_corr else None
        return z, present, corr

    # -- single-pair / reduction calls go to the first engine ---------------------------------------------------------------
    def anim_pair_alignments(self, ref_id: int, qry_id: int):
        return self.engines[0].anim_pair_alignments(ref_id, qry_id)

    def anib_pair_rows(self, qry_id: int, sbj_id: int, fragsize: int = 1020):
        return self.engines[0].anib_pair_rows(qry_id, sbj_id, fragsize)

    def anim_reduce(self, pairs, apply_filter: bool = False):
        return self.engines[0].anim_reduce(pairs, apply_filter)

    def anib_reduce(self, pairs):
        return self.engines[0].anib_reduce(pairs)


def merge_alignment_parts(n: int, chunks: List[np.ndarray], parts, with_indels: bool):
    """Put the per-chunk results of Engine.anim_alignments_batch — (offsets, records, indel offsets, indels) over the chunk's own
    pairs — back into ONE result over the caller's n pairs, in the caller's order."""
    counts = np.zeros(n, dtype=np.int64)
    for idx, (off, _, _, _) in zip(chunks, parts):
        counts[idx] = np.diff(np.asarray(off, dtype=np.int64))
    offsets = np.zeros(n + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum(counts)
    total = int(offsets[-1])
    rec_dtype = next((p[1].dtype for p in parts if len(p[1])), Engine.ALN_DTYPE)
    recs = np.zeros(total, dtype=rec_dtype)
    icounts = np.zeros(total, dtype=np.int64)
    for idx, (off, rc, ioff, _) in zip(chunks, parts):
        off = np.asarray(off, dtype=np.int64)
        for j, p in enumerate(idx):
            a, b = int(off[j]), int(off[j + 1])
            if b > a:
                dst = int(offsets[p])
                recs[dst:dst + b - a] = rc[a:b]
                if with_indels:
                    icounts[dst:dst + b - a] = np.diff(np.asarray(ioff[a:b + 1], dtype=np.int64))
    if not with_indels:
        return offsets, recs, None, None
    ioffsets = np.zeros(total + 1, dtype=np.uint64)
    ioffsets[1:] = np.cumsum(icounts)
    ind_dtype = next((p[3].dtype for p in parts if p[3] is not None and len(p[3])), np.int64)
    indels = np.zeros(int(ioffsets[-1]), dtype=ind_dtype)
    for idx, (off, _, ioff, ind) in zip(chunks, parts):
        off = np.asarray(off, dtype=np.int64)
        for j, p in enumerate(idx):
            a, b = int(off[j]), int(off[j + 1])
            if b > a:
                dst = int(offsets[p])
                lo, hi = int(ioff[a]), int(ioff[b])
                indels[int(ioffsets[dst]):int(ioffsets[dst]) + hi - lo] = ind[lo:hi]
    return offsets, recs, ioffsets, indels


def engine_for(devices: Optional[Iterable[int]] = None, workers: Optional[int] = None):
    """The engine a caller should use: `devices` as given; else `workers` GPUs starting at 0 (pyani's --workers, capped at the
    GPUs present); else the process-wide single engine."""
    from .engine import default_engine
    if devices is None and workers:
        import ctypes
        n = ctypes.c_int(0)
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            if hip.hipGetDeviceCount(ctypes.byref(n)) != 0:
                n.value = 0
        except OSError:
            n.value = 0
        devices = list(range(max(1, min(int(workers), n.value or 1))))
    if devices is None:
        return default_engine()
    devices = list(devices)
    return MultiEngine(devices) if len(devices) > 1 or devices != [0] else default_engine()
