"""Multi-GPU orchestration of the TETRA matrix: one process per GPU, genomes sharded by rank,
two small all-gathers (Z vectors, then matrix row blocks) over RCCL/xGMI — no other collective.

The compute steps are injected as callables so that the sharding / padding / gather logic below is exactly what
runs on the GPUs (bench.py, backend "nccl" = RCCL) AND what the CPU tests exercise with world_size 2 over gloo.
"""
from typing import Callable, Tuple

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
