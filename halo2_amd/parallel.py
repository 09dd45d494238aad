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


class PeerFailure(RuntimeError):
    """A rank of a split multiexp / commit failed: EVERY rank raises (the failing rank its own error, the others this one)
    instead of summing the partials that did arrive into a plausible-looking wrong point."""


def allgather_points(local_xyz: np.ndarray, device=None, status: int = 0):
    """All-gather one Jacobian point (12 uint64 limbs) per rank -> (world, 12).  The payload carries a status word behind the
    point (13 words per rank): a rank whose local work failed still enters the collective -- its peers would wait for it forever
    otherwise -- with status != 0, and the caller on every rank sees which ranks failed.  Returns (points (world, 12), statuses
    (world,))."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    payload = np.zeros(13, dtype=np.uint64)
    if local_xyz is not None:
        payload[:12] = np.ascontiguousarray(local_xyz, dtype=np.uint64).reshape(12)
    payload[12] = np.uint64(status & 0xFFFFFFFF)
    t = torch.from_numpy(payload.view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    got = np.stack([o.cpu().numpy().view(np.uint64) for o in out])
    return np.ascontiguousarray(got[:, :12]), got[:, 12].astype(np.int64)


def _raise_if_any_failed(statuses, what: str, local_exc=None):
    bad = [int(r) for r in np.nonzero(np.asarray(statuses))[0]]
    if local_exc is not None:
        raise local_exc
    if bad:
        raise PeerFailure("%s: rank(s) %s failed; no result (a partial sum would be a wrong point)" % (what, bad))


def split_msm(scalars: np.ndarray, bases: np.ndarray, curve: int, rank: int, world: int, device=None,
              msm=None, points_sum=None) -> np.ndarray:
    """One MSM split by point range across ranks.  `msm` / `points_sum` default to the HIP path; the CPU
    (gloo) tests inject the oracle so the sharding logic is covered without a GPU.  A rank whose multiexp raises still joins the
    all-gather (status word set) and every rank raises: nobody hangs, nobody returns a sum with a range missing."""
    if msm is None or points_sum is None:
        from . import arithmetic
        msm = msm or (lambda s, b: arithmetic.best_multiexp(s, b, curve))
        points_sum = points_sum or (lambda pts: arithmetic.points_sum(pts, curve))
    lo, hi = shard_range(scalars.shape[0], rank, world)
    partial, exc = None, None
    try:
        partial = np.asarray(msm(scalars[lo:hi], bases[lo:hi]))
    except Exception as e:                          # noqa: BLE001 -- reported after the exchange, on every rank
        exc = e
    gathered, statuses = allgather_points(partial, device=device, status=0 if exc is None else 1)
    _raise_if_any_failed(statuses, "split_msm", exc)
    return points_sum(gathered)


# ---- one process, several GPUs: the C ABI's own fan-out (csrc/multi.hip) ------------------------------------------------
def commit_batch_multi(handles, devices, columns, n: int, w=None, blinds=None, affine: bool = False) -> np.ndarray:
    """h2_commit_batch_multi: `columns` (host limb arrays, (n, 4) each) committed round-robin over `devices`; handles[d] is the
    h2_bases_t (int, or a Params' registered handle) of the same bases on devices[d].  Returns (len(columns), 12 | 8) limbs."""
    import ctypes as C
    from ._lib import FORM_MONTGOMERY, OUT_AFFINE, OUT_JACOBIAN, check, lib
    count = len(columns)
    out = np.zeros((count, 8 if affine else 12), dtype=np.uint64)
    cols = [np.ascontiguousarray(c, dtype=np.uint64) for c in columns]
    for c in cols:
        if c.shape != (n, 4):
            raise ValueError("commit_batch_multi: every column must hold n scalars")
    if (w is None) != (blinds is None):
        raise ValueError("commit_batch_multi: w and blinds go together")
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    h_arr = (C.c_uint64 * len(handles))(*[int(getattr(h_, "value", h_)) for h_ in handles])
    d_arr = (C.c_int * len(devices))(*devices)
    s_arr = (C.c_void_p * count)(*[ptr(c) for c in cols])
    o_arr = (C.c_void_p * count)(*[out[i].ctypes.data_as(C.c_void_p) for i in range(count)])
    if w is not None:
        w = np.ascontiguousarray(w, dtype=np.uint64).reshape(8)
        bl = [np.ascontiguousarray(b, dtype=np.uint64).reshape(4) for b in blinds]
        b_arr = (C.c_void_p * count)(*[ptr(b) for b in bl])
        check(lib().h2_commit_batch_multi(h_arr, d_arr, len(devices), s_arr, count, n, ptr(w), b_arr, FORM_MONTGOMERY,
                                          OUT_AFFINE if affine else OUT_JACOBIAN, o_arr), "h2_commit_batch_multi")
    else:
        check(lib().h2_commit_batch_multi(h_arr, d_arr, len(devices), s_arr, count, n, None, None, FORM_MONTGOMERY,
                                          OUT_AFFINE if affine else OUT_JACOBIAN, o_arr), "h2_commit_batch_multi")
    return out


def split_msm_multi(scalars, bases, curve: int, devices, affine: bool = False) -> np.ndarray:
    """h2_msm_split_multi: one best_multiexp cut into len(devices) point ranges, partials added on devices[0]."""
    import ctypes as C
    from ._lib import FORM_MONTGOMERY, OUT_AFFINE, OUT_JACOBIAN, check, lib
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    if scalars.shape[0] != bases.shape[0]:
        raise ValueError("split_msm_multi: coeffs and bases differ in length")          # arithmetic.rs:144
    out = np.zeros(8 if affine else 12, dtype=np.uint64)
    d_arr = (C.c_int * len(devices))(*devices)
    check(lib().h2_msm_split_multi(curve, scalars.ctypes.data_as(C.c_void_p), bases.ctypes.data_as(C.c_void_p), scalars.shape[0], d_arr,
                                   len(devices), FORM_MONTGOMERY, OUT_AFFINE if affine else OUT_JACOBIAN, out.ctypes.data_as(C.c_void_p)),
          "h2_msm_split_multi")
    return out


# ---- one process per GPU, exchange step by RCCL inside the library ------------------------------------------------------------
def rccl_init(rank: int, world: int) -> None:
    """Creates the library's own RCCL communicator: rank 0's unique id travels over the caller's torch.distributed group
    (any backend -- it is 128 bytes), then every rank joins with its GPU current."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from ._lib import check, lib
    buf = np.zeros(128, dtype=np.uint8)
    if rank == 0:
        check(lib().h2_rccl_unique_id(buf.ctypes.data_as(C.c_void_p)), "h2_rccl_unique_id")
    if world > 1:
        t = torch.from_numpy(buf.copy())
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.broadcast(t, 0)
        buf = t.cpu().numpy()
    check(lib().h2_rccl_init(np.ascontiguousarray(buf).ctypes.data_as(C.c_void_p), rank, world), "h2_rccl_init")


def rccl_finalize() -> None:
    from ._lib import lib
    lib().h2_rccl_finalize()


def split_msm_rccl(d_scalars, d_bases, curve: int, affine: bool = False):
    """h2_msm_split_rccl_device on CUDA tensors holding the whole problem on every rank: this rank's range, one 96-byte
    all-gather over xGMI, local sum.  Returns a CUDA tensor (12 | 8 limbs); asynchronous on the current stream."""
    import torch
    from ._lib import FORM_MONTGOMERY, OUT_AFFINE, OUT_JACOBIAN, check, lib
    from .arithmetic import _stream_ptr
    if d_scalars.shape[0] != d_bases.shape[0]:
        raise ValueError("split_msm_rccl: coeffs and bases differ in length")
    out = torch.empty(8 if affine else 12, dtype=torch.int64, device=d_scalars.device)
    check(lib().h2_msm_split_rccl_device(curve, d_scalars.data_ptr(), d_bases.data_ptr(), d_scalars.shape[0], FORM_MONTGOMERY,
                                         OUT_AFFINE if affine else OUT_JACOBIAN, out.data_ptr(), _stream_ptr()), "h2_msm_split_rccl_device")
    return out


def split_commit_rccl(handle, d_scalars, d_blind=None, affine: bool = False):
    """h2_commit_split_rccl_device: one commit over REGISTERED bases (every rank holds the table and the column) split by table
    column range, the last rank carrying the blind term; one 96-byte all-gather over xGMI, local sum.  `handle`: an h2_bases_t."""
    import torch
    from ._lib import FORM_MONTGOMERY, OUT_AFFINE, OUT_JACOBIAN, check, lib
    from .arithmetic import _stream_ptr
    out = torch.empty(8 if affine else 12, dtype=torch.int64, device=d_scalars.device)
    check(lib().h2_commit_split_rccl_device(int(getattr(handle, "value", handle)), d_scalars.data_ptr(), d_scalars.shape[0],
                                            d_blind.data_ptr() if d_blind is not None else None, FORM_MONTGOMERY,
                                            OUT_AFFINE if affine else OUT_JACOBIAN, out.data_ptr(), _stream_ptr()), "h2_commit_split_rccl_device")
    return out


def split_commit(handle, d_scalars, rank: int, world: int, d_blind=None):
    """The same split with the exchange step carried by torch.distributed (backend nccl = RCCL; gloo goes through the host):
    rank r commits table columns shard_range(n, r, world) with h2_commit_range_device, the 96-byte Jacobian partials are
    all-gathered, every rank adds them with h2_points_sum_device.  Returns a (12,) CUDA tensor."""
    import ctypes as C
    import torch
    import torch.distributed as dist
    from ._lib import FORM_MONTGOMERY, OUT_JACOBIAN, check, lib
    from .arithmetic import _stream_ptr
    n = d_scalars.shape[0]
    lo, hi = shard_range(n, rank, world)
    dev = d_scalars.device
    mine = torch.zeros(12, dtype=torch.int64, device=dev)          # all-zero limbs = the identity (Z = 0)
    hv = int(getattr(handle, "value", handle))
    # rank-independent preconditions on EVERY rank, before anyone enters the gather: only the last rank passes the blind down, and
    # its failure alone would leave the others waiting in the collective
    if d_blind is not None and lib().h2_bases_blind_base_set(hv) != 1:
        raise ValueError("split_commit: a blind scalar but the handle has no blind base (h2_bases_set_blind_base on every rank)")
    # a rank-LOCAL failure still takes part in the exchange -- with its status word set (word 12 of the payload), so that EVERY rank
    # raises instead of adding up the partials that did arrive
    local_rc = lib().h2_commit_range_device(hv, d_scalars[lo:].data_ptr() if hi > lo else None, lo, hi - lo,
                                            d_blind.data_ptr() if (d_blind is not None and rank == world - 1) else None, FORM_MONTGOMERY,
                                            OUT_JACOBIAN, mine.data_ptr(), _stream_ptr())
    payload = torch.zeros(13, dtype=torch.int64, device=dev)
    if local_rc == 0:
        payload[:12].copy_(mine)
    else:
        payload[12] = 1
    gathered13 = torch.empty((world, 13), dtype=torch.int64, device=dev)
    if world > 1:
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(gathered13, payload)
        else:
            parts = [torch.empty(13, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(parts, payload.cpu())
            gathered13.copy_(torch.stack(parts))
    else:
        gathered13[0].copy_(payload)
    check(local_rc, "h2_commit_range_device")
    statuses = gathered13[:, 12].cpu().numpy()                       # one small read-back: a peer's failure must not become a wrong point
    _raise_if_any_failed(statuses, "split_commit")
    gathered = gathered13[:, :12].contiguous()
    out = torch.empty(12, dtype=torch.int64, device=dev)
    curve = C.c_int(0)
    check(lib().h2_bases_info(hv, None, None, C.byref(curve)), "h2_bases_info")
    check(lib().h2_points_sum_device(curve.value, gathered.data_ptr(), world, FORM_MONTGOMERY, OUT_JACOBIAN, out.data_ptr(), _stream_ptr()),
          "h2_points_sum_device")
    return out
