"""Multi-GPU placement for the hot path: one process per GPU (torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" on CPU for tests).

The reference has no distributed layer.  The path shards two ways (section 8e):
  1. independent units -- the column commits / column FFTs of a prover phase
     (plonk/prover.rs:93-96,305-309) are independent: round-robin them over ranks, no collective;
  2. one range-split MSM -- each rank sums a contiguous range of (scalar, base) pairs and the 96-byte
     Jacobian partials are exchanged with ONE all_gather, then added locally (RCCL cannot reduce curve
     points itself).  Payload is 96 B per rank: latency-bound, bucket/link sizing is irrelevant here.
"""
from __future__ import annotations

import numpy as np


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous range of [0, n) owned by `rank` (sizes differ by at most one)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def columns_for_rank(num_columns: int, rank: int, world: int) -> list[int]:
    """Round-robin placement of independent column commits / FFTs."""
    return list(range(rank, num_columns, world))


def allgather_points(local_xyz: np.ndarray, device=None) -> np.ndarray:
    """All-gather one Jacobian point (12 uint64 limbs) per rank -> (world, 12)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    t = torch.from_numpy(np.ascontiguousarray(local_xyz, dtype=np.uint64).view(np.int64).reshape(12).copy())
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return np.stack([o.cpu().numpy().view(np.uint64) for o in out])


def split_msm(scalars: np.ndarray, bases: np.ndarray, curve: int, rank: int, world: int, device=None,
              msm=None, points_sum=None) -> np.ndarray:
    """One MSM split by point range across ranks.  `msm` / `points_sum` default to the HIP path; the CPU
    (gloo) tests inject the oracle so the sharding logic is covered without a GPU."""
    if msm is None or points_sum is None:
        from . import arithmetic
        msm = msm or (lambda s, b: arithmetic.best_multiexp(s, b, curve))
        points_sum = points_sum or (lambda pts: arithmetic.points_sum(pts, curve))
    lo, hi = shard_range(scalars.shape[0], rank, world)
    partial = msm(scalars[lo:hi], bases[lo:hi])
    gathered = allgather_points(np.asarray(partial), device=device)
    return points_sum(gathered)
