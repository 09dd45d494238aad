"""Multi-GPU entry points of the C ABI (csrc/multi.hip) on the one device a test box has: the fan-out code runs with the
device list [0, 0] (two host threads, round-robin columns) and [0], the range-split multiexp with one to three ranges, the
RCCL path with a communicator of size 1; and the one-process-per-GPU split (parallel.split_msm) with two gloo ranks sharing
the GPU, the HIP path as each rank's multiexp.  Expected values come from the oracle."""
import os
import socket
import sys

import numpy as np
import pytest

import halo2_amd as h
from halo2_amd import parallel
from oracle import c_oracle as co

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def aff(curve, out):
    out = np.ascontiguousarray(out, dtype=np.uint64)
    return co.jac_to_affine_ints(curve, out) if out.shape[0] == 12 else co.affine_to_ints(curve, out)


@pytest.mark.parametrize("curve,n", [(h.VESTA, 1 << 11), (h.PALLAS, 777)])
def test_commit_batch_multi_two_lanes_on_one_device(curve, n):
    sf = co.field_of_curve(curve, "scalar")
    k = max(1, (n - 1).bit_length())
    g = co.generate_bases(curve, 5150 + n, 1 << k)
    w = co.generate_bases(curve, 9, 1)[0]
    params = h.Params(curve, k, g, g, w, w)
    cols = [co.random_field(sf, 300 + i, n) for i in range(7)]
    blinds = co.random_field(sf, 77, 7)
    want = [aff(curve, co.commit(curve, np.ascontiguousarray(g[:n]), w, c, blinds[i])) for i, c in enumerate(cols)]
    for devices in ([0, 0], [0]):
        got = parallel.commit_batch_multi([params._h_g] * len(devices), devices, cols, n, w=w, blinds=list(blinds))
        assert [aff(curve, got[i]) for i in range(7)] == want
    got = parallel.commit_batch_multi([params._h_g, params._h_g], [0, 0], cols[:3], n, affine=True)        # no blind, affine out
    assert [aff(curve, got[i]) for i in range(3)] == [aff(curve, co.best_multiexp(curve, c, np.ascontiguousarray(g[:n]))) for c in cols[:3]]
    assert parallel.commit_batch_multi([params._h_g], [0], [], n).shape == (0, 12)
    with pytest.raises(ValueError):
        parallel.commit_batch_multi([params._h_g], [5], cols[:1], n)            # no such device
    params.close()


@pytest.mark.parametrize("ndev", [1, 2, 3])
def test_msm_split_multi(ndev):
    curve, n = h.PALLAS, 5000
    sf = co.field_of_curve(curve, "scalar")
    sc, bs = co.random_field(sf, 41, n), co.generate_bases(curve, 42, n)
    want = aff(curve, co.best_multiexp(curve, sc, bs))
    assert aff(curve, parallel.split_msm_multi(sc, bs, curve, [0] * ndev)) == want
    assert aff(curve, parallel.split_msm_multi(sc, bs, curve, [0] * ndev, affine=True)) == want
    assert aff(curve, parallel.split_msm_multi(sc[:2], bs[:2], curve, [0] * ndev)) == aff(curve, co.best_multiexp(curve, sc[:2], bs[:2]))


def test_points_sum_device_forms():
    import torch
    curve = h.VESTA
    sf = co.field_of_curve(curve, "scalar")
    parts = [h.best_multiexp(co.random_field(sf, 60 + i, 50), co.generate_bases(curve, 70 + i, 50), curve) for i in range(4)]
    want = aff(curve, h.points_sum(np.stack(parts), curve))
    d = torch.from_numpy(np.stack(parts).view(np.int64)).cuda()
    out = torch.zeros(8, dtype=torch.int64, device="cuda")
    from halo2_amd._lib import check, lib
    check(lib().h2_points_sum_device(curve, d.data_ptr(), 4, h.FORM_MONTGOMERY, 1, out.data_ptr(), None), "h2_points_sum_device")
    torch.cuda.synchronize()
    assert aff(curve, out.cpu().numpy().view(np.uint64)) == want


def test_split_msm_rccl_world_of_one():
    """The library's own RCCL communicator (dlopen librccl.so): unique id, init, one all-gather, local sum, finalize."""
    import torch
    curve, n = h.PALLAS, 3000
    sf = co.field_of_curve(curve, "scalar")
    sc, bs = co.random_field(sf, 91, n), co.generate_bases(curve, 92, n)
    parallel.rccl_init(0, 1)
    try:
        out = parallel.split_msm_rccl(torch.from_numpy(sc.view(np.int64)).cuda(), torch.from_numpy(bs.view(np.int64)).cuda(), curve)
        torch.cuda.synchronize()
        assert aff(curve, out.cpu().numpy().view(np.uint64)) == aff(curve, co.best_multiexp(curve, sc, bs))
    finally:
        parallel.rccl_finalize()


def test_split_commit_over_registered_bases_world_of_one():
    """The split of a commit over REGISTERED bases (h2_commit_range_device per rank, blind on the last one, 96-byte exchange,
    local sum) through the library's RCCL communicator and through parallel.split_commit, each with a world of one; and
    h2_bases_info."""
    import ctypes as C
    import torch
    from halo2_amd._lib import lib
    from halo2_amd.arithmetic import _p
    curve, n = h.VESTA, 20000
    sf = co.field_of_curve(curve, "scalar")
    g, col = co.generate_bases(curve, 93, n), co.random_field(sf, 94, n)
    w, blind = co.generate_bases(curve, 95, 1)[0], co.random_field(sf, 96, 1)
    hd = C.c_uint64(0)
    assert lib().h2_bases_register(curve, _p(g), n, h.FORM_MONTGOMERY, C.byref(hd)) == 0
    assert lib().h2_bases_set_blind_base(hd, _p(np.ascontiguousarray(w)), h.FORM_MONTGOMERY) == 0
    nn, cc, cv = C.c_size_t(0), C.c_int(0), C.c_int(-1)
    assert lib().h2_bases_info(hd, C.byref(nn), C.byref(cc), C.byref(cv)) == 0
    assert (nn.value, cc.value, cv.value) == (n, 16, curve)
    assert lib().h2_bases_info(C.c_uint64(987654), None, None, None) != 0
    d_col = torch.from_numpy(col.view(np.int64)).cuda()
    d_bl = torch.from_numpy(blind.view(np.int64)).cuda()[0].contiguous()
    want = aff(curve, co.commit(curve, g, w, col, blind[0]))
    out = parallel.split_commit(hd, d_col, 0, 1, d_bl)
    torch.cuda.synchronize()
    assert aff(curve, out.cpu().numpy().view(np.uint64)) == want
    parallel.rccl_init(0, 1)
    try:
        out = parallel.split_commit_rccl(hd, d_col, d_bl, affine=True)
        torch.cuda.synchronize()
        assert aff(curve, out.cpu().numpy().view(np.uint64)) == want
        out = parallel.split_commit_rccl(hd, d_col[:12345])                                   # a prefix, no blind
        torch.cuda.synchronize()
        assert aff(curve, out.cpu().numpy().view(np.uint64)) == aff(curve, co.best_multiexp(curve, col[:12345], g[:12345]))
    finally:
        parallel.rccl_finalize()
    assert lib().h2_bases_free(hd) == 0


def test_split_commit_rccl_injected_local_failure_is_an_error_not_a_point():
    """h2_commit_split_rccl_device with a rank-LOCAL failure injected after the range commit (H2_TEST_FAIL_RANK, read once per
    process: a fresh interpreter): the failing rank still runs the exchange (status word behind its partial) and returns ITS error;
    the output buffer is never written.  (With a world of one there is no peer to see H2_ERR_PEER; the gloo test
    test_split_msm_one_rank_fails_every_rank_raises covers the all-ranks-raise side on the torch.distributed path.)"""
    import subprocess
    code = r"""
import ctypes as C, numpy as np, torch, sys
sys.path.insert(0, %r)
import halo2_amd as h
from halo2_amd import parallel
from halo2_amd._lib import lib, H2Error
from halo2_amd.arithmetic import _p
from oracle import c_oracle as co
curve, n = h.PALLAS, 5000
sf = co.field_of_curve(curve, "scalar")
g, col = co.generate_bases(curve, 3, n), co.random_field(sf, 4, n)
hd = C.c_uint64(0)
assert lib().h2_bases_register(curve, _p(g), n, h.FORM_MONTGOMERY, C.byref(hd)) == 0
d_col = torch.from_numpy(col.view(np.int64)).cuda()
parallel.rccl_init(0, 1)
try:
    parallel.split_commit_rccl(hd, d_col)
    print("RETURNED")
except H2Error as e:
    print("H2ERROR", e)
finally:
    parallel.rccl_finalize()
""" % ROOT
    from conftest import ab_env
    env = ab_env(H2_TEST_FAIL_RANK="0")           # the hook is compiled out of the shipped library (csrc/common.h ab_env): the child loads the laboratory build
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "H2ERROR" in out.stdout and "injected local failure" in out.stdout and "RETURNED" not in out.stdout, out.stdout + out.stderr


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import halo2_amd as hh
    from halo2_amd import parallel as par
    from oracle import c_oracle as oc
    curve, n = 0, 4097
    sf = oc.field_of_curve(curve, "scalar")
    scal, bases = oc.random_field(sf, 11, n), oc.generate_bases(curve, 12, n)
    total = par.split_msm(scal, bases, curve, rank, world)                     # HIP multiexp per rank, gloo all_gather, HIP sum
    whole = oc.best_multiexp(curve, scal, bases)
    q.put((rank, oc.jac_to_affine_ints(curve, total) == oc.jac_to_affine_ints(curve, whole)))
    dist.barrier()
    dist.destroy_process_group()


def test_split_msm_two_ranks_hip_path():
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def test_split_commit_blind_precondition_is_checked_on_every_rank():
    """only the LAST rank of a split commit hands the blind to its range commit; if the handle has no blind base
    that rank alone would fail and leave the others waiting in the all-gather.  h2_bases_blind_base_set is the rank-independent
    precondition: parallel.split_commit (any rank) and h2_commit_split_rccl_device refuse BEFORE entering the exchange."""
    import ctypes as C
    import torch
    from halo2_amd.arithmetic import _p
    lib = h.lib()
    curve, n = h.PALLAS, 1 << 12
    sf = co.field_of_curve(curve, "scalar")
    g = co.generate_bases(curve, 0x5151, n)
    hd = C.c_uint64(0)
    assert lib.h2_bases_register(curve, _p(g), n, h.FORM_MONTGOMERY, C.byref(hd)) == 0
    assert lib.h2_bases_blind_base_set(hd) == 0
    assert lib.h2_bases_blind_base_set(C.c_uint64(0xDEAD0001)) < 0                       # no such handle
    d_col = torch.from_numpy(co.random_field(sf, 1, n).view(np.int64)).to("cuda:0")
    d_bl = torch.from_numpy(co.random_field(sf, 2, 1).view(np.int64)).to("cuda:0")[0].contiguous()
    for rank in (0, 1):                    # rank 0 of a world of two never touches the blind itself -- and still refuses
        with pytest.raises(ValueError):
            parallel.split_commit(hd, d_col, rank, 2, d_bl)
    w = co.generate_bases(curve, 0x77, 1)[0]
    assert lib.h2_bases_set_blind_base(hd, _p(w), h.FORM_MONTGOMERY) == 0
    assert lib.h2_bases_blind_base_set(hd) == 1
    out = parallel.split_commit(hd, d_col, 0, 1, d_bl)
    want = co.commit(curve, g, w, co.random_field(sf, 1, n), co.random_field(sf, 2, 1)[0])
    assert co.jac_to_affine_ints(curve, out.cpu().numpy().view(np.uint64)) == co.jac_to_affine_ints(curve, want)
    assert lib.h2_bases_free(hd) == 0
