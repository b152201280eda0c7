"""Multi-GPU orchestration of the TETRA matrix: one process per GPU, genomes sharded by rank,
two small all-gathers (Z vectors, then matrix row blocks) over RCCL/xGMI — no other collective.

The compute steps are injected as callables so that the sharding / padding / gather logic below is exactly what
runs on the GPUs (bench.py, backend "nccl" = RCCL) AND what the CPU tests exercise with world_size 2 over gloo.
"""
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced block of [0, n) owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def max_shard(n: int, world: int) -> int:
    return (n + world - 1) // world


class TetraAllGather:
    """Pre-allocated buffers for repeated passes over the same job size (collectives need equal-size pieces, so
    every rank's piece is padded to the largest shard)."""

    def __init__(self, n_total: int, device: torch.device, group=None):
        self.n = n_total
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.lo, self.hi = shard_range(n_total, self.rank, self.world)
        self.m = max_shard(n_total, self.world)
        kw = dict(device=device)
        self.z_loc = torch.zeros((self.m, 256), dtype=torch.float64, **kw)
        self.p_loc = torch.zeros((self.m, 256), dtype=torch.uint8, **kw)
        self.z_pad = torch.zeros((self.world * self.m, 256), dtype=torch.float64, **kw)
        self.p_pad = torch.zeros((self.world * self.m, 256), dtype=torch.uint8, **kw)
        self.z_all = torch.zeros((n_total, 256), dtype=torch.float64, **kw)
        self.p_all = torch.zeros((n_total, 256), dtype=torch.uint8, **kw)
        self.rows = torch.zeros((self.m, n_total), dtype=torch.float64, **kw)
        self.rows_pad = torch.zeros((self.world * self.m, n_total), dtype=torch.float64, **kw)
        self.corr = torch.zeros((n_total, n_total), dtype=torch.float64, **kw)

    def _unpad(self, padded: torch.Tensor, out: torch.Tensor):
        for r in range(self.world):
            lo, hi = shard_range(self.n, r, self.world)
            out[lo:hi] = padded[r * self.m: r * self.m + (hi - lo)]

    def run(self, compute_z: Callable[[torch.Tensor, torch.Tensor], None],
            compute_rows: Callable[[torch.Tensor, torch.Tensor, int, int, torch.Tensor], None]) -> torch.Tensor:
        """compute_z(z_loc, p_loc): fill the first (hi-lo) rows with this rank's genomes' Z / presence.
        compute_rows(z_all, p_all, lo, nrows, rows): fill rows[0:nrows] = matrix rows lo..lo+nrows of the full job.
        Returns the full n x n matrix (identical on every rank)."""
        compute_z(self.z_loc, self.p_loc)
        dist.all_gather_into_tensor(self.z_pad, self.z_loc, group=self.group)     # collective #1 (Z, 2 KiB/genome)
        dist.all_gather_into_tensor(self.p_pad, self.p_loc, group=self.group)
        self._unpad(self.z_pad, self.z_all)
        self._unpad(self.p_pad, self.p_all)
        if self.z_all.is_cuda:
            torch.cuda.current_stream().synchronize()  # the engine computes on its own stream
        compute_rows(self.z_all, self.p_all, self.lo, self.hi - self.lo, self.rows)
        dist.all_gather_into_tensor(self.rows_pad, self.rows, group=self.group)   # collective #2 (matrix rows)
        self._unpad(self.rows_pad, self.corr)
        return self.corr


# ---- ANIm: the ordered-pair grid is sharded, results assembled with ONE all-gather ---------------
ANIM_FIELDS = 6  # ref_aln_len, qry_aln_len, sim_errors, n_alignments, identity (bit pattern), status


def anim_row_shard(rows: Sequence[int], rank: int, world: int) -> List[int]:
    """The reference genomes (rows of the ordered-pair grid) of `rows` owned by `rank`.  Whole rows stay on one rank so that
    every reference k-mer table is built once per pass.  Pair cost varies ~60x with relatedness, and related genomes tend
    to sit next to each other in a sorted input list (or, in the synthetic sets, at a fixed stride), so rows are dealt
    round-robin in a fixed scrambled order (multiplicative hash of the row number) rather than in blocks or by stride."""
    order = sorted(rows, key=lambda q: ((q * 0x9E3779B1) & 0xFFFFFFFF, q))
    return sorted(order[rank::world])


def anim_pair_shard(n_genomes: int, rank: int, world: int, rows: Optional[Sequence[int]] = None) -> List[Tuple[int, int]]:
    """Ordered pairs (q, s), q != s, owned by `rank` (grouped by reference q)."""
    rows = range(n_genomes) if rows is None else rows
    return [(q, s) for q in anim_row_shard(rows, rank, world) for s in range(n_genomes) if s != q]


def anim_pair_array(n_genomes: int, rows: Sequence[int], symmetric: bool = False):
    """The ordered pairs of the rows `rows` as an int64 [m, 2] numpy array of (reference, query).
    symmetric = False: row q = reference q against every other genome (grouped by q).
    symmetric = True: row g = the UNORDERED pairs {g, h} that g owns, each in both directions ((g, h) then, in a second
    block, (h, g)).  {g, h} is owned by the smaller id when g + h is even, by the larger one otherwise, so every genome owns
    about half of its pairs and the rows of all genomes together are exactly the N x N grid.  The engine seeds a pair and its
    reverse once when both are in one call (they have the same maximal exact matches), which is what this layout is for."""
    import numpy as np
    rows = np.asarray(list(rows), dtype=np.int64)
    q = np.repeat(rows, n_genomes)
    s = np.tile(np.arange(n_genomes, dtype=np.int64), len(rows))
    keep = q != s
    if symmetric:
        keep &= (((q + s) % 2) == 0) == (q < s)
        fwd = np.stack([q[keep], s[keep]], axis=1)
        return np.concatenate([fwd, fwd[:, ::-1]])
    return np.stack([q[keep], s[keep]], axis=1)


def anim_allgather(compute_pairs: Callable, n_genomes: int, device: torch.device, group=None,
                   rows: Optional[Sequence[int]] = None, symmetric: bool = False, stats: Optional[dict] = None) -> torch.Tensor:
    """compute_pairs(pairs: int64 ndarray [m, 2] of (q, s)) -> int64 tensor [m, ANIM_FIELDS] on `device` (identity as its
    IEEE-754 bit pattern).
    rows = None: the whole grid -> [n, n, ANIM_FIELDS] on every rank (diagonal zero).  rows = a list of genomes (one step
    of a tiled run): only those rows are computed -> [len(rows), n, ANIM_FIELDS], row i = reference rows[i]; with
    symmetric = True a row is what anim_pair_array(symmetric=True) says and the result is [n, n, ANIM_FIELDS] with the
    computed cells filled.  Rows are dealt over the ranks by anim_row_shard either way.
    One collective of 64 B per pair (results + the pair's own (q, s), so the gathered block is self-describing)."""
    import time as _time
    _t = [_time.perf_counter()]

    def _mark():      # (only when the caller asked for stats: the device is drained so that the sections do not bleed into each other)
        if stats is not None:
            if device.type == "cuda":
                torch.cuda.synchronize(device)
            _t.append(_time.perf_counter())
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    all_rows = list(range(n_genomes)) if rows is None else list(rows)
    shards = [anim_row_shard(all_rows, r, world) for r in range(world)]
    mine = anim_pair_array(n_genomes, shards[rank], symmetric)
    if symmetric:   # rows own different numbers of pairs: the padded length is the largest shard's
        import numpy as np
        own = np.zeros(n_genomes, dtype=np.int64)
        g = np.arange(n_genomes, dtype=np.int64)
        for parity in (0, 1):     # same parity as g: the larger ids; other parity: the smaller ids
            ids = g[g % 2 == parity]
            own[ids] += len(ids) - 1 - np.arange(len(ids))
            other = g[g % 2 != parity]
            own[ids] += np.searchsorted(other, ids)
        cap = int(max(2 * own[np.asarray(sh, dtype=np.int64)].sum() if len(sh) else 0 for sh in shards))
    else:
        rows_max = (len(all_rows) + world - 1) // world
        cap = rows_max * max(n_genomes - 1, 0)
    assert len(mine) <= cap
    loc = torch.zeros((cap, ANIM_FIELDS + 2), dtype=torch.int64, device=device)
    _mark()
    busy = 0.0
    if len(mine):
        import time
        t0 = time.perf_counter()
        vals = compute_pairs(mine)
        if vals.is_cuda:
            torch.cuda.synchronize(vals.device)
        busy = time.perf_counter() - t0
        loc[: len(mine), :ANIM_FIELDS] = vals
        loc[: len(mine), ANIM_FIELDS:] = torch.from_numpy(mine).to(device)
    loc[len(mine):, ANIM_FIELDS] = -1
    _mark()
    if stats is not None:      # (a second, 8-byte collective: the ranks' busy seconds of this step)
        allb = torch.zeros(world, dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(allb, torch.tensor([busy], dtype=torch.float64, device=device), group=group)
        b = allb.cpu().tolist()
        mean = sum(b) / len(b)
        stats.update({"busy_s": b, "chunks": [1] * world, "imbalance": (max(b) / mean) if mean > 0 else 1.0})
    allv = torch.zeros((world * cap, ANIM_FIELDS + 2), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(allv, loc, group=group)
    _mark()
    valid = allv[:, ANIM_FIELDS] >= 0
    q, s = allv[valid, ANIM_FIELDS], allv[valid, ANIM_FIELDS + 1]
    if symmetric:
        grid = torch.zeros((n_genomes, n_genomes, ANIM_FIELDS), dtype=torch.int64, device=device)
        grid[q, s] = allv[valid, :ANIM_FIELDS]
        _mark()
        if stats is not None and len(_t) == 5:
            stats["host_ms"] = {k: round((b - a) * 1e3, 2) for k, a, b in zip(("deal", "compute", "gathers", "grid"), _t[:-1], _t[1:])}
        return grid
    row_slot = torch.full((n_genomes,), -1, dtype=torch.int64, device=device)
    row_slot[torch.tensor(all_rows, dtype=torch.int64, device=device)] = torch.arange(len(all_rows), dtype=torch.int64, device=device)
    grid = torch.zeros((len(all_rows), n_genomes, ANIM_FIELDS), dtype=torch.int64, device=device)
    grid[row_slot[q], s] = allv[valid, :ANIM_FIELDS]
    return grid


# ---- dynamic dealing: the ranks PULL row chunks from a cross-rank counter ------------------------------------------------------
# Pair cost varies ~60 x with relatedness and row cost several-fold with the divergence of a genome's family; a static deal (the
# hash above) leaves the step waiting for its unluckiest rank, and every launch inside a rank ends on its slowest unit.  pyani's
# own runner is a pool that hands the next job to whichever worker falls idle (run_multiprocessing.py:113-152); the counterpart
# across processes: the step's rows, in the scrambled order, are cut into GUIDED chunks (each about half of an even share of
# what is left: large first, small at the end — the classic guided self-scheduling), every rank computes the same chunk list, and
# an atomic counter in a TCPStore hands out chunk numbers.  One engine call per chunk; results gathered once per step.
def guided_chunks(n_rows: int, world: int, min_rows: int = 2) -> List[Tuple[int, int]]:
    """[lo, hi) spans of the guided chunk sequence over n_rows rows for `world` ranks."""
    spans, lo = [], 0
    while lo < n_rows:
        c = max(min_rows, -(-(n_rows - lo) // (2 * world)))
        hi = min(n_rows, lo + c)
        spans.append((lo, hi))
        lo = hi
    return spans


class RowQueue:
    """The cross-rank counter: `next_chunk(token)` returns the next unclaimed chunk number of a step (atomic over all ranks).
    The counter lives in a key-value store with an atomic add: by default THE JOB'S OWN rendezvous store (the TCPStore that
    `init_process_group` already opened on MASTER_PORT: no second port to collide with another job on the node — the driver runs
    N = 1, 2, 4, 8 back to back); `port` opens a store of its own instead (tests; a process group built without a store): rank 0
    binds it (the requested port, or any free one if that is taken) and tells the others the port it got through the process group.
    Which of the two it is, is AGREED between the ranks (a store that works on some ranks only would leave the others waiting in a
    broadcast).  A step's counter is named by `step_token(step_key)`: the key plus how often this queue has begun that key, so a
    caller that reuses a key (a rerun, two loops that both say "k0") gets a fresh counter each time — every rank calls step_token
    once per collective step, in the same order."""

    def __init__(self, rank: int, world: int, host: str = "127.0.0.1", port: int = 0, timeout_s: float = 1800.0, prefix: str = "pyani_rows"):
        import datetime
        self.world = world
        self.store = None
        self._local = {}
        self._uses = {}
        self._prefix = prefix
        self.kind = "local"          # "job-store": the process group's rendezvous store; "own-store": a TCPStore of its own
        if world <= 1:
            return
        to = datetime.timedelta(seconds=timeout_s)
        if not dist.is_initialized():      # no process group to talk through: everybody is given the same port (tests)
            self.store = dist.TCPStore(host, port, world, is_master=(rank == 0), timeout=to, wait_for_workers=True)
            self.kind = "own-store"
            return
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        if not port:
            store, ok = None, 1
            try:
                from torch.distributed.distributed_c10d import _get_default_store
                store = dist.PrefixStore(prefix, _get_default_store())
                store.add("probe", 0)      # (a store without `add` — a FileStore on some file systems — fails here, not mid-step)
            except Exception:
                store, ok = None, 0
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)      # every rank takes the same branch
            if int(flag.item()) == 1:
                self.store, self.kind = store, "job-store"
                return
        # a store of its own: rank 0 binds (no probe-then-bind window: the store itself takes the port), the others learn the port
        chosen = [0]
        if rank == 0:
            for cand in ([port] if port else []) + [0]:
                try:
                    self.store = dist.TCPStore(host, cand, world, is_master=True, timeout=to, wait_for_workers=False)
                    chosen[0] = int(self.store.port)
                    break
                except Exception:
                    self.store = None
        dist.broadcast_object_list(chosen, src=0)
        if not chosen[0]:
            raise RuntimeError("RowQueue: rank 0 could not open a TCPStore")
        if rank != 0:
            self.store = dist.TCPStore(host, int(chosen[0]), world, is_master=False, timeout=to)
        self.kind = "own-store"

    def step_token(self, step_key: str) -> str:
        """The counter name of the next collective step that uses `step_key` (call once per step on every rank)."""
        n = self._uses.get(step_key, 0)
        self._uses[step_key] = n + 1
        return f"{step_key}#{n}"

    def next_chunk(self, token: str) -> int:
        if self.store is None:
            k = self._local.get(token, 0)
            self._local[token] = k + 1
            return k
        return int(self.store.add(f"rows/{token}", 1)) - 1


def anim_allgather_dynamic(compute_pairs: Callable, n_genomes: int, device: torch.device, queue: RowQueue, step_key: str,
                           rows: Sequence[int], symmetric: bool = True, group=None, min_rows: int = 2):
    """anim_allgather with the rows PULLED in guided chunks from `queue` instead of dealt statically.  Returns (grid, stats):
    grid as anim_allgather returns it; stats = {"busy_s": [per rank], "chunks": [per rank], "imbalance": max / mean of busy_s}.
    Two small collectives (per-rank counts and busy times, then the padded results) per step."""
    import time
    import numpy as np
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_initialized() else (0, 1)
    order = sorted(rows, key=lambda q: ((q * 0x9E3779B1) & 0xFFFFFFFF, q))      # related genomes sit side by side in sorted lists
    spans = guided_chunks(len(order), world, min_rows)
    mine_pairs, mine_vals, busy, taken = [], [], 0.0, 0
    token = queue.step_token(step_key)      # (a key that is used again gets a counter of its own)
    while True:
        k = queue.next_chunk(token)
        if k >= len(spans):
            break
        lo, hi = spans[k]
        pairs = anim_pair_array(n_genomes, order[lo:hi], symmetric)
        if len(pairs):
            t0 = time.perf_counter()
            vals = compute_pairs(pairs)
            if vals.is_cuda:
                torch.cuda.synchronize(vals.device)
            busy += time.perf_counter() - t0
            mine_pairs.append(pairs); mine_vals.append(vals)
        taken += 1
    n_mine = sum(len(p) for p in mine_pairs)
    meta = torch.tensor([float(n_mine), busy, float(taken)], dtype=torch.float64, device=device)
    if world > 1:
        metas = torch.zeros(world * 3, dtype=torch.float64, device=device)
        dist.all_gather_into_tensor(metas, meta, group=group)
    else:
        metas = meta
    metas = metas.cpu().numpy().reshape(world, 3)
    cap = int(metas[:, 0].max())
    loc = torch.zeros((max(cap, 1), ANIM_FIELDS + 2), dtype=torch.int64, device=device)
    loc[:, ANIM_FIELDS] = -1
    if n_mine:
        loc[:n_mine, :ANIM_FIELDS] = torch.cat(mine_vals)
        loc[:n_mine, ANIM_FIELDS:] = torch.from_numpy(np.concatenate(mine_pairs)).to(device)
    if world > 1:
        allv = torch.zeros((world * max(cap, 1), ANIM_FIELDS + 2), dtype=torch.int64, device=device)
        dist.all_gather_into_tensor(allv, loc, group=group)
    else:
        allv = loc
    valid = allv[:, ANIM_FIELDS] >= 0
    q, s_ = allv[valid, ANIM_FIELDS], allv[valid, ANIM_FIELDS + 1]
    grid = torch.zeros((n_genomes, n_genomes, ANIM_FIELDS), dtype=torch.int64, device=device)
    grid[q, s_] = allv[valid, :ANIM_FIELDS]
    busy_s = metas[:, 1].tolist()
    mean = sum(busy_s) / len(busy_s)
    return grid, {"busy_s": busy_s, "chunks": [int(x) for x in metas[:, 2]], "imbalance": (max(busy_s) / mean) if mean > 0 else 1.0}


def anib_records_to_tensor(recs, device: torch.device) -> torch.Tensor:
    """Engine.anib_pairs structured array -> int64 [n, ANIM_FIELDS] tensor: aln_length, sim_errors, n_frags, n_kept, mean
    pident (bit-cast, lossless), status — fragment mode shards and gathers exactly like ANIm (anim_allgather)."""
    import numpy as np
    a = np.zeros((len(recs), ANIM_FIELDS), dtype=np.int64)
    a[:, 0], a[:, 1], a[:, 2], a[:, 3] = recs["aln_length"], recs["sim_errors"], recs["n_frags"], recs["n_kept"]
    a[:, 4] = recs["pid"].view(np.int64)
    a[:, 5] = recs["status"]
    return torch.from_numpy(a).to(device)


def anim_records_to_tensor(recs, device: torch.device) -> torch.Tensor:
    """Engine.anim_pairs structured array -> int64 [n, ANIM_FIELDS] tensor (identity bit-cast, lossless)."""
    import numpy as np
    a = np.zeros((len(recs), ANIM_FIELDS), dtype=np.int64)
    a[:, 0], a[:, 1], a[:, 2], a[:, 3] = recs["ref_aln_len"], recs["qry_aln_len"], recs["sim_errors"], recs["n_alignments"]
    a[:, 4] = recs["identity"].view(np.int64)
    a[:, 5] = recs["status"]
    return torch.from_numpy(a).to(device)


# ---- the product's cross-process path: one engine per rank behind the Engine calls run_anim uses ------------------------------------
class DistributedEngine:
    """One process per GPU (torch.distributed: backend "nccl" = RCCL over xGMI on a node of MI355X, "gloo" in the CPU tests), every
    rank holding a LOCAL engine with all genomes resident (they are small: 1.9 GB for 1000 x 5 Mb).  `anim_pairs` is a COLLECTIVE
    call — every rank makes it with the same arguments, as every rank runs the same `run_anim` — that

      * deals the pair list by hub genome (a pair and its reverse stay together: they share their seeding) into one scrambled share
        per rank, ONE engine call each (pyani_amd.multi._static_parts_by_hub: measured on C4 the shares of 8 ranks are within 2 - 5 %
        of each other, and every extra call pays its kernels' tails again); `dynamic=True`: chunks of 1 / chunks_per_rank of a
        share handed out through the cross-rank counter (RowQueue: the rank that drew cheap pairs comes back for more — pyani's own
        runner is such a pool, run_multiprocessing.py:113-152 — for jobs whose cost sits in a few genomes),
      * computes the rank's share on its engine, and
      * assembles the result with ONE all-gather per call (the 40-byte records + the pair's index, padded to the largest share),

    so that every rank returns the complete array in the caller's order: the counterpart of pyani's `--workers` (subcmd_anim.py:
    392-396) across the GPUs of a node, with the collective in the PRODUCT path (pyani_amd.subcmd_anim.run_anim wraps its engine in
    one of these whenever a process group with more than one rank is initialised), not only in bench.py.  Everything else — the
    genome store, reductions of existing files, alignment records for output files — is the local engine's call, made by every
    rank alike.  Results do not depend on the number of ranks or on who computed what (tests/test_parallel_gloo.py, world size 2 on
    CPU with a recording engine; tests/test_parallel_multi_gpu.py: two gloo ranks on GPU 0 and one RCCL rank == a plain engine)."""

    def __init__(self, local_engine, group=None, queue: Optional["RowQueue"] = None, chunks_per_rank: int = 24, dynamic: bool = False):
        self.local = local_engine
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.dynamic = bool(dynamic) or queue is not None
        self.queue = (queue or RowQueue(self.rank, self.world)) if self.dynamic else None
        self.chunks_per_rank = chunks_per_rank
        # collective buffers live on the LOCAL ENGINE's device (torch.cuda.current_device() is GPU 0 for every rank of a
        # caller that never called torch.cuda.set_device)
        dev_index = getattr(local_engine, "device", None)
        if dist.get_backend(group) == "nccl":
            self.device = torch.device("cuda", int(dev_index) if dev_index is not None else torch.cuda.current_device())
        else:
            self.device = torch.device("cpu")
        self.last_stats = None

    def __getattr__(self, name):      # genome store, reductions, single-pair calls: the local engine's
        return getattr(self.local, name)

    def anim_pairs(self, ref_ids, qry_ids, filter_1to1: bool = True, maxmatch: bool = False):
        import time
        import numpy as np
        from .multi import _chunks_by_hub, _static_parts_by_hub
        r = np.ascontiguousarray(list(ref_ids), dtype=np.int32)
        q = np.ascontiguousarray(list(qry_ids), dtype=np.int32)
        if len(r) != len(q):
            raise ValueError("ref_ids and qry_ids must have the same length")
        n = len(r)
        proto = self.local.anim_pairs(r[:0], q[:0], filter_1to1=filter_1to1, maxmatch=maxmatch)      # (dtype of the records)
        if n == 0 or self.world == 1:
            return self.local.anim_pairs(r, q, filter_1to1=filter_1to1, maxmatch=maxmatch)
        if self.dynamic:
            chunks = _chunks_by_hub(r, q, max(64, n // (self.chunks_per_rank * self.world) + 1))
            token = self.queue.step_token("anim_pairs")
            draw = lambda: self.queue.next_chunk(token)      # noqa: E731
        else:
            chunks = _static_parts_by_hub(r, q, self.world)
            mine = iter([self.rank] if self.rank < len(chunks) else [])
            draw = lambda: next(mine, len(chunks))           # noqa: E731
        mine_idx, mine_rec, busy = [], [], 0.0
        # error agreement: a rank whose engine call fails (PG_E_NOMEM, PG_E_CAPACITY, a bad id) must not leave the others
        # waiting in the all-gathers until the backend's timeout — it keeps drawing (so a shared queue still drains), reports the failure
        # in the `meta` all-gather, and EVERY rank raises
        failure = None
        while True:
            k = draw()
            if k >= len(chunks):
                break
            if failure is not None:
                continue
            idx = chunks[k]
            t0 = time.perf_counter()
            try:
                mine_rec.append(self.local.anim_pairs(r[idx], q[idx], filter_1to1=filter_1to1, maxmatch=maxmatch))
            except Exception as exc:  # noqa: BLE001
                failure = exc
                continue
            busy += time.perf_counter() - t0
            mine_idx.append(idx)
        n_mine = sum(len(i) for i in mine_idx)
        meta = torch.tensor([float(n_mine), busy, 0.0 if failure is None else 1.0], dtype=torch.float64, device=self.device)
        metas = torch.zeros(self.world * 3, dtype=torch.float64, device=self.device)
        dist.all_gather_into_tensor(metas, meta, group=self.group)
        metas = metas.cpu().numpy().reshape(self.world, 3)
        failed_ranks = [int(k) for k in np.nonzero(metas[:, 2])[0]]
        if failed_ranks:
            if failure is not None:
                raise failure
            raise RuntimeError(f"DistributedEngine.anim_pairs: the engine call failed on rank(s) {failed_ranks} (see their tracebacks); no rank returns a result")
        cap = max(1, int(metas[:, 0].max()))
        words = proto.dtype.itemsize // 8      # (the record is 40 bytes = five 8-byte words; one more for the pair's index)
        loc = np.zeros((cap, words + 1), dtype=np.int64)
        loc[:, words] = -1
        if n_mine:
            loc[:n_mine, :words] = np.concatenate(mine_rec).view(np.int64).reshape(n_mine, words)
            loc[:n_mine, words] = np.concatenate(mine_idx)
        loc_t = torch.from_numpy(loc).to(self.device)
        all_t = torch.zeros((self.world * cap, words + 1), dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(all_t, loc_t, group=self.group)      # THE collective of the call
        allv = all_t.cpu().numpy()
        valid = allv[:, words] >= 0
        if int(valid.sum()) != n:
            raise RuntimeError(f"DistributedEngine.anim_pairs: {int(valid.sum())} of {n} pairs came back")
        out = np.zeros(n, dtype=proto.dtype)
        out.view(np.int64).reshape(n, words)[allv[valid, words]] = allv[valid, :words]
        mean = float(metas[:, 1].mean())
        self.last_stats = {"busy_s": metas[:, 1].tolist(), "pairs": metas[:, 0].astype(int).tolist(), "chunks": len(chunks),
                           "imbalance": float(metas[:, 1].max()) / mean if mean > 0 else 1.0}
        return out


def engine_for_process_group(local_engine, group=None):
    """`local_engine` wrapped in a DistributedEngine when a process group with more than one rank is initialised (one process per
    GPU, launched with torch.distributed.run), else `local_engine` itself.  run_anim calls this only when asked to
    (`run_anim(..., distributed=True)` / `group=`): a collective must never start behind the caller's back."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        return DistributedEngine(local_engine, group)
    return local_engine
