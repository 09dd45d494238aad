"""N > 1 path on CPU: world_size-2 gloo process group; the sharding and the 96-byte all-gather of a
range-split MSM are exercised with the oracle injected as the per-rank MSM (no GPU here)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, n=300):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from halo2_amd import parallel
    from oracle import c_oracle as co
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    curve = 0
    sf = co.field_of_curve(curve, "scalar")
    scal = co.random_field(sf, 11, max(n, 1))[:n]
    bases = co.generate_bases(curve, 12, max(n, 1))[:n]

    def msm(s, b):
        return co.best_multiexp(curve, np.ascontiguousarray(s), np.ascontiguousarray(b))

    def psum(pts):
        acc = np.zeros(12, np.uint64)
        for p in pts:
            out = np.zeros(12, np.uint64)
            co.lib().orc_point_add(curve, co._p(out), co._p(acc), co._p(np.ascontiguousarray(p)))
            acc = out
        return acc

    total = parallel.split_msm(scal, bases, curve, rank, world, msm=msm, points_sum=psum)
    whole = msm(scal, bases)
    ok = co.jac_to_affine_ints(curve, total) == co.jac_to_affine_ints(curve, whole)
    cols = parallel.columns_for_rank(7, rank, world)
    q.put((rank, ok, cols, parallel.shard_range(n, rank, world)))
    dist.barrier()
    dist.destroy_process_group()


def test_split_msm_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
    assert res[0][2] == [0, 2, 4, 6] and res[1][2] == [1, 3, 5]
    assert res[0][3] == (0, 150) and res[1][3] == (150, 300)


@pytest.mark.parametrize("n", [301, 5])
def test_split_msm_world8_gloo(n):
    """the node's size: eight ranks, ragged ranges (301 = 5 x 38 + 3 x 37) and fewer points than ranks (three ranks contribute the
    identity): the 96-byte all-gather and the local sum return the whole multiexp on every rank"""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, n)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res)
    spans = [r[3] for r in res]
    assert spans[0][0] == 0 and spans[-1][1] == n and all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
    assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
    assert sorted(c for r in res for c in r[2]) == list(range(7))          # seven columns dealt round-robin over eight ranks


def _failing_worker(rank, world, port, q, bad_rank):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from halo2_amd import parallel
    from oracle import c_oracle as co
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    curve, n = 0, 64
    sf = co.field_of_curve(curve, "scalar")
    scal, bases = co.random_field(sf, 21, n), co.generate_bases(curve, 22, n)

    def msm(s, b):
        if rank == bad_rank:
            raise RuntimeError("injected: this rank's multiexp failed")
        return co.best_multiexp(curve, np.ascontiguousarray(s), np.ascontiguousarray(b))

    summed = []
    try:
        parallel.split_msm(scal, bases, curve, rank, world, msm=msm, points_sum=lambda pts: summed.append(1) or pts[0])
        outcome = "returned"
    except parallel.PeerFailure as e:
        outcome = "peer:" + str(e)
    except RuntimeError as e:
        outcome = "own:" + str(e)
    q.put((rank, outcome, len(summed)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,bad_rank", [(2, 1), (3, 0)])
def test_split_msm_one_rank_fails_every_rank_raises(world, bad_rank):
    """A rank whose range multiexp fails still joins the all-gather (status word behind its point): IT raises its own error, every
    OTHER rank raises PeerFailure naming it, nobody hangs in the collective and nobody sums the partials that did arrive into a
    plausible-looking wrong point (the round-4 advisor's finding on the identity-as-partial scheme)."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, q, bad_rank)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, outcome, nsum in res:
        assert nsum == 0, (rank, outcome)                          # no rank added anything up
        if rank == bad_rank:
            assert outcome.startswith("own:injected"), outcome
        else:
            assert outcome.startswith("peer:") and ("[%d]" % bad_rank) in outcome, outcome


def test_shard_range_covers_everything():
    from halo2_amd import parallel
    for n in (0, 1, 7, 8, 1 << 20, (1 << 20) + 1):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    with pytest.raises(ValueError):
        parallel.shard_range(10, 3, 3)
