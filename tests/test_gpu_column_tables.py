"""The configuration the headline is measured on, under test: column tables with 17-bit windows (`h2_bases_register_ex`,
what `Params` registers `g` / `g_lagrange` with from 2^18 points on: 2^16 buckets, row / column fold) at 2^18, 2^19 and 2^20 points
on both curves, against the C restatement of `Params::commit` / `best_multiexp` (poly/commitment.rs:119-150,
arithmetic.rs:143-180) -- dense and skewed columns, blinds, prefix lengths, the batch entry point -- and the blind base as a
property of the handle (`Params::w`, commitment.rs:26-33): content-checked, never keyed by an address.  Also the pipelined
host-pointer commit (column ranges committed as they cross PCIe)."""
import os
import ctypes as C

import numpy as np
import pytest

import halo2_amd as h
from halo2_amd import fields
from halo2_amd.arithmetic import _p
from oracle import c_oracle as co
from oracle import pasta as o

pytestmark = pytest.mark.gpu


def affine_of(curve, out):
    out = np.ascontiguousarray(out, dtype=np.uint64)
    return co.jac_to_affine_ints(curve, out) if out.shape[0] == 12 else co.affine_to_ints(curve, out)


def _points(curve, ks):
    bm, _ = o.CURVES[curve]
    return co.points_to_mont(curve, [o.ec_mul(k, (bm - 1, 2), bm) for k in ks])


def _skewed_columns(sf, sm, n):
    dense = co.random_field(sf, 4100 + n % 97, n)
    zeros90 = dense.copy()
    zeros90[np.arange(n) % 10 != 0] = 0
    small = fields.to_limbs([((i * 2654435761) & 0xFFFF) for i in range(1 << 12)], sf)
    top = co.to_mont(sf, co.ints_to_limbs([sm - 1 - i for i in range(1 << 12)]))
    return {
        "dense": dense,
        "zeros90": zeros90,
        "all_equal": np.ascontiguousarray(np.tile(fields.scalar_limbs(0xDEADBEEFCAFE0123456789, sf), (n, 1))),
        "below_2^16": np.ascontiguousarray(np.tile(small, (n >> 12, 1)) if n >= (1 << 12) else small[:n]),
        "q-1-i": np.ascontiguousarray(np.tile(top, (n >> 12, 1)) if n >= (1 << 12) else top[:n]),          # maximal digits, every window negative after the recode
    }


@pytest.mark.parametrize("curve,k", [(h.VESTA, 18), (h.PALLAS, 19), (h.PALLAS, 20), (h.VESTA, 19), (h.VESTA, 20)])
def test_17_bit_column_tables_match_oracle(curve, k):
    import torch
    lib = h.lib()
    n = 1 << k
    _, sm = o.CURVES[curve]
    sf = co.field_of_curve(curve, "scalar")
    assert lib.h2_commit_column_window_bits(n) == 17                      # what Params and bench.py register with
    g = co.generate_bases(curve, 0x1700 + k + curve, n)
    hd = C.c_uint64(0)
    assert lib.h2_bases_register_ex(curve, _p(g), n, h.FORM_MONTGOMERY, 17, C.byref(hd)) == 0
    w1, w2 = _points(curve, [0x77, 0x1234567])
    cols = _skewed_columns(sf, sm, n)
    blinds = co.random_field(sf, 0xB11D, 4)
    out = np.zeros(12, np.uint64)
    dev = torch.device("cuda", 0)
    d_out = torch.zeros(12, dtype=torch.int64, device=dev)

    def dev_commit(d_col, n_used, d_w, d_bl):
        rc = lib.h2_commit_device(hd, d_col.data_ptr(), n_used, d_w.data_ptr() if d_w is not None else None,
                                  d_bl.data_ptr() if d_bl is not None else None, h.FORM_MONTGOMERY, 0, d_out.data_ptr(), None)
        assert rc == 0, lib.h2_last_error()
        torch.cuda.synchronize()
        return affine_of(curve, d_out.cpu().numpy().view(np.uint64))

    # a blind scalar before any blind base was installed: refused, not a commitment to garbage
    d_dense = torch.from_numpy(cols["dense"].view(np.int64)).to(dev)
    d_bl = torch.from_numpy(blinds.view(np.int64)).to(dev)
    assert lib.h2_commit_device(hd, d_dense.data_ptr(), n, None, d_bl[0].data_ptr(), h.FORM_MONTGOMERY, 0, d_out.data_ptr(), None) == 1
    assert lib.h2_bases_set_blind_base(hd, _p(w1), h.FORM_MONTGOMERY) == 0

    # every column shape, with the handle's blind base, device path; the dense one also through the host path
    want_w1 = {}
    for name, col in cols.items():
        want_w1[name] = co.jac_to_affine_ints(curve, co.commit(curve, g, w1, col, blinds[0]))
        d_col = d_dense if name == "dense" else torch.from_numpy(col.view(np.int64)).to(dev)
        assert dev_commit(d_col, n, None, d_bl[0]) == want_w1[name], name
    assert lib.h2_commit(hd, _p(cols["dense"]), n, None, _p(blinds[0]), h.FORM_MONTGOMERY, 0, _p(out)) == 0     # pipelined ranges
    assert affine_of(curve, out) == want_w1["dense"]

    # two different w in sequence, by content: host pointers ...
    want_w2 = co.jac_to_affine_ints(curve, co.commit(curve, g, w2, cols["dense"], blinds[1]))
    for w, bl, want in ((w2, blinds[1], want_w2), (w1, blinds[0], want_w1["dense"]), (w2, blinds[1], want_w2)):
        assert lib.h2_commit(hd, _p(cols["dense"]), n, _p(w), _p(bl), h.FORM_MONTGOMERY, 0, _p(out)) == 0
        assert affine_of(curve, out) == want
    # ... and ONE device pointer whose 64 bytes are rewritten between commits (the address-keyed cache of round 2 returned the
    # commitment to the old w here)
    d_w = torch.from_numpy(w1.view(np.int64)).to(dev)
    assert dev_commit(d_dense, n, d_w, d_bl[0]) == want_w1["dense"]
    d_w.copy_(torch.from_numpy(w2.view(np.int64)))
    assert dev_commit(d_dense, n, d_w, d_bl[1]) == want_w2
    assert dev_commit(d_dense, n, None, d_bl[1]) == want_w2                   # the presented w became the handle's
    d_w.copy_(torch.from_numpy(w1.view(np.int64)))
    assert dev_commit(d_dense, n, d_w, d_bl[0]) == want_w1["dense"]
    # the canonical form of the same point is the same point
    w1_canon = np.ascontiguousarray(co.from_mont(co.field_of_curve(curve, "base"), w1.reshape(2, 4)).reshape(8))
    assert lib.h2_bases_set_blind_base(hd, _p(w1_canon), h.FORM_CANONICAL) == 0
    assert dev_commit(d_dense, n, None, d_bl[0]) == want_w1["dense"]

    # prefix lengths (IPA rounds commit over the first n' bases, poly/commitment/prover.rs:107-108), with and without blind
    for n_used in (0, 1, 17, n - 1):
        want = co.jac_to_affine_ints(curve, co.best_multiexp(curve, cols["dense"][:n_used], g[:n_used]))
        assert dev_commit(d_dense, n_used, None, None) == want, n_used
        assert lib.h2_commit(hd, _p(cols["dense"]), n_used, None, None, h.FORM_MONTGOMERY, 0, _p(out)) == 0
        assert affine_of(curve, out) == want, n_used
        want_b = co.jac_to_affine_ints(curve, co.commit(curve, np.ascontiguousarray(g[:n_used]), w1, np.ascontiguousarray(cols["dense"][:n_used]), blinds[2]))
        assert dev_commit(d_dense, n_used, None, d_bl[2]) == want_b, n_used

    # the batch entry point (plonk/prover.rs:305-313) over four different columns with their blinds
    names = ["dense", "zeros90", "q-1-i", "below_2^16"]
    d_cols = [torch.from_numpy(cols[nm].view(np.int64)).to(dev) for nm in names]
    d_outs = torch.zeros((4, 8), dtype=torch.int64, device=dev)
    arr = C.c_void_p * 4
    rc = lib.h2_commit_batch_device(hd, arr(*[c_.data_ptr() for c_ in d_cols]), 4, n, None, arr(*[d_bl[i].data_ptr() for i in range(4)]),
                                    h.FORM_MONTGOMERY, 1, arr(*[d_outs[i].data_ptr() for i in range(4)]), None)
    assert rc == 0, lib.h2_last_error()
    torch.cuda.synchronize()
    got = d_outs.cpu().numpy().view(np.uint64)
    for i, nm in enumerate(names):
        want = want_w1[nm] if i == 0 else co.jac_to_affine_ints(curve, co.commit(curve, g, w1, cols[nm], blinds[i]))
        assert affine_of(curve, got[i]) == want, nm
    assert lib.h2_bases_free(hd) == 0


@pytest.mark.parametrize("n,chunk", [(3000, 1000), (3000, 4096), (20000, 8192), (70001, 16384)])
def test_pipelined_host_commit_ranges(n, chunk):
    """h2_commit cuts a host column into table-column ranges that are committed as they land (MsmArgs::col0) and summed:
    ragged last range, one-pass and two-pass sorts, blind on the last range, prefix lengths that end inside a range."""
    curve = h.VESTA
    lib = h.lib()
    sf = co.field_of_curve(curve, "scalar")
    g = co.generate_bases(curve, 0x9000 + n, n)
    col = co.random_field(sf, n, n)
    w, = _points(curve, [0xABCDEF])
    blind = fields.scalar_limbs(0x1234567890ABCDEF, sf)
    hd = C.c_uint64(0)
    assert lib.h2_bases_register(curve, _p(g), n, h.FORM_MONTGOMERY, C.byref(hd)) == 0
    assert lib.h2_set_option(b"host_commit_chunk", float(chunk)) == 0
    out = np.zeros(12, np.uint64)
    try:
        for n_used in (n, n - 1, 2 * chunk, 2 * chunk + 1, 1, 0):
            if n_used > n:
                continue
            assert lib.h2_commit(hd, _p(col), n_used, _p(w), _p(blind), h.FORM_MONTGOMERY, 0, _p(out)) == 0, lib.h2_last_error()
            want = co.jac_to_affine_ints(curve, co.commit(curve, np.ascontiguousarray(g[:n_used]), w, np.ascontiguousarray(col[:n_used]), blind))
            assert affine_of(curve, out) == want, n_used
            assert lib.h2_commit(hd, _p(col), n_used, None, None, h.FORM_MONTGOMERY, 1, _p(out)) == 0
            assert affine_of(curve, out[:8]) == co.jac_to_affine_ints(curve, co.best_multiexp(curve, col[:n_used], g[:n_used])), n_used
        # canonical scalars through the same ranges
        canon = np.ascontiguousarray(co.from_mont(sf, col))
        assert lib.h2_commit(hd, _p(canon), n, None, None, h.FORM_CANONICAL, 0, _p(out)) == 0
        got = co.jac_to_affine_ints(curve, co.to_mont(co.field_of_curve(curve, "base"), out.reshape(3, 4)).reshape(12))
        assert got == co.jac_to_affine_ints(curve, co.best_multiexp(curve, col, g))
    finally:
        assert lib.h2_set_option(b"host_commit_chunk", 0.0) == 0
        assert lib.h2_bases_free(hd) == 0


def test_commit_range_device_partials_sum_to_the_commit():
    """h2_commit_range_device: the commit restricted to table columns [first, first + n) -- what one rank of a commit split over
    GPUs computes (h2_commit_split_rccl_device, parallel.split_commit); ragged ranges, blind on the last one, both sort paths."""
    import torch
    curve = h.PALLAS
    lib = h.lib()
    sf = co.field_of_curve(curve, "scalar")
    dev = torch.device("cuda", 0)
    for n, cuts in ((3000, [0, 1, 1000, 2999, 3000]), (40000, [0, 13000, 13001, 40000])):
        g = co.generate_bases(curve, 0x5150 + n, n)
        col = co.random_field(sf, 0x5151 + n, n)
        w, = _points(curve, [0x99])
        blind = co.random_field(sf, 0x5152, 1)
        hd = C.c_uint64(0)
        assert lib.h2_bases_register(curve, _p(g), n, h.FORM_MONTGOMERY, C.byref(hd)) == 0
        assert lib.h2_bases_set_blind_base(hd, _p(w), h.FORM_MONTGOMERY) == 0
        d_col = torch.from_numpy(col.view(np.int64)).to(dev)
        d_bl = torch.from_numpy(blind.view(np.int64)).to(dev)
        parts = torch.zeros((len(cuts) - 1, 12), dtype=torch.int64, device=dev)
        for r in range(len(cuts) - 1):
            lo, hi = cuts[r], cuts[r + 1]
            last = r == len(cuts) - 2
            rc = lib.h2_commit_range_device(hd, d_col[lo:].data_ptr() if hi > lo else None, lo, hi - lo, d_bl.data_ptr() if last else None,
                                            h.FORM_MONTGOMERY, 0, parts[r].data_ptr(), None)
            assert rc == 0, lib.h2_last_error()
            torch.cuda.synchronize()
            want = (co.commit(curve, np.ascontiguousarray(g[lo:hi]), w, np.ascontiguousarray(col[lo:hi]), blind[0]) if last
                    else co.best_multiexp(curve, col[lo:hi], g[lo:hi]))
            assert affine_of(curve, parts[r].cpu().numpy().view(np.uint64)) == co.jac_to_affine_ints(curve, want), (n, r)
        total = h.points_sum(parts.cpu().numpy().view(np.uint64), curve)
        assert affine_of(curve, total) == co.jac_to_affine_ints(curve, co.commit(curve, g, w, col, blind[0]))
        assert lib.h2_commit_range_device(hd, d_col.data_ptr(), 1, n, None, h.FORM_MONTGOMERY, 0, parts[0].data_ptr(), None) == 1   # past the table
        assert lib.h2_bases_free(hd) == 0


def test_commit_batch_multi_two_blind_bases_one_handle():
    """h2_commit_batch_multi with the same handles and two different w in sequence (the staging slot's fixed address
    made the second call reuse the first w's multiples)."""
    curve = h.PALLAS
    lib = h.lib()
    sf = co.field_of_curve(curve, "scalar")
    n = 5000
    g = co.generate_bases(curve, 0x4444, n)
    hd = C.c_uint64(0)
    assert lib.h2_bases_register(curve, _p(g), n, h.FORM_MONTGOMERY, C.byref(hd)) == 0
    cols = [co.random_field(sf, 70 + i, n) for i in range(3)]
    blinds = co.random_field(sf, 71, 3)
    outs = [np.zeros(12, np.uint64) for _ in range(3)]
    vp3 = C.c_void_p * 3
    for wk in (0x55, 0x66, 0x55):
        w, = _points(curve, [wk])
        rc = lib.h2_commit_batch_multi((C.c_uint64 * 1)(hd.value), (C.c_int * 1)(0), 1, vp3(*[c_.ctypes.data for c_ in cols]), 3, n,
                                       C.c_void_p(w.ctypes.data), vp3(*[blinds[i].ctypes.data for i in range(3)]), h.FORM_MONTGOMERY, 0,
                                       vp3(*[o_.ctypes.data for o_ in outs]))
        assert rc == 0, lib.h2_last_error()
        for i in range(3):
            assert affine_of(curve, outs[i]) == co.jac_to_affine_ints(curve, co.commit(curve, g, w, cols[i], blinds[i])), (wk, i)
    assert lib.h2_bases_free(hd) == 0


@pytest.mark.parametrize("curve", [h.PALLAS, h.VESTA])
def test_small_registered_tables_quad_lane_chain(curve):
    """Tables of up to 2^16 points are filled by the quad-lane doubling chain on the carry-free layer (msm_table_chain_wide,
    curve9_wide.cuh); registered from host and from device memory, 16-bit windows (240 doublings per point), commits against
    `best_multiexp` (arithmetic.rs:143-180).  The Vesta table of 2^14 points from seed 0x219 holds the point whose chain meets a
    product with low limb 2^29 (field_check's recorded state)."""
    import torch
    lib = h.lib()
    sf = co.field_of_curve(curve, "scalar")
    for n in (16384, 32772, 65536):
        g = co.generate_bases(curve, 0x99 + n % 1000, n)
        col = co.random_field(sf, 5 + n, n)
        d_col = torch.from_numpy(col.view(np.int64)).cuda()
        want = co.jac_to_affine_ints(curve, co.best_multiexp(curve, col, g))
        for on_device in (False, True):
            hd = C.c_uint64(0)
            if on_device:
                dg = torch.from_numpy(g.view(np.int64)).cuda()
                assert lib.h2_bases_register_device(curve, dg.data_ptr(), n, h.FORM_MONTGOMERY, C.byref(hd)) == 0
            else:
                assert lib.h2_bases_register(curve, _p(g), n, h.FORM_MONTGOMERY, C.byref(hd)) == 0
            out = torch.empty(12, dtype=torch.int64, device="cuda:0")
            assert lib.h2_commit_device(hd, d_col.data_ptr(), n, None, None, h.FORM_MONTGOMERY, 0, out.data_ptr(), None) == 0
            torch.cuda.synchronize()
            assert affine_of(curve, out.cpu().numpy().view(np.uint64)) == want, (n, on_device)
            lib.h2_bases_free(hd)


@pytest.mark.parametrize("curve,k,bits", [(h.PALLAS, 18, 17), (h.VESTA, 16, 16), (h.PALLAS, 20, 17), (h.VESTA, 12, 13), (h.PALLAS, 11, 13),
                                          (h.VESTA, 13, 16)])
def test_column_batched_commit_matches_oracle(curve, k, bits):
    """The column-batched form of h2_commit_batch_device (one sort / accumulate / fold launch set for K columns, blockIdx.z =
    column; csrc/msm.hip ColIn / ColStride): every column shape side by side -- dense, 90 %-zero, all-equal (oversized pass-2 bins
    and heavy buckets IN ONE column of the batch), < 2^16, q - 1 - i --, with and without blinds, counts that fill one group
    (2, 5, 8), spill into two (9: groups of 5 + 4 on two internal streams) and a prefix length, each output against the C
    restatement of Params::commit (poly/commitment.rs:119-130).  2^20 at 17 bits is the configuration bench.py times."""
    import torch
    lib = h.lib()
    n = 1 << k
    _, sm = o.CURVES[curve]
    sf = co.field_of_curve(curve, "scalar")
    g = co.generate_bases(curve, 0x2200 + k + curve, n)
    hd = C.c_uint64(0)
    assert lib.h2_bases_register_ex(curve, _p(g), n, h.FORM_MONTGOMERY, bits, C.byref(hd)) == 0
    (w1,) = _points(curve, [0x4242])
    assert lib.h2_bases_set_blind_base(hd, _p(w1), h.FORM_MONTGOMERY) == 0
    cols = _skewed_columns(sf, sm, n)
    names = list(cols) + ["dense2", "dense3", "dense4", "dense5", "all_zero"]
    for i in range(2, 6):
        cols[f"dense{i}"] = co.random_field(sf, 5200 + i, n)
    cols["all_zero"] = np.zeros_like(cols["dense"])         # no entries at all: an empty stretch in the middle of the joined sorted list
    blinds = co.random_field(sf, 0xB22D, len(names))
    dev = torch.device("cuda", 0)
    d_cols = {nm: torch.from_numpy(cols[nm].view(np.int64)).to(dev) for nm in names}
    d_bl = torch.from_numpy(blinds.view(np.int64)).to(dev)
    want_b = {}
    want_nb = {}

    def want(nm, with_blind, n_used=n):
        key = (nm, n_used)
        tab = want_b if with_blind else want_nb
        if key not in tab:
            i = names.index(nm)
            tab[key] = (co.jac_to_affine_ints(curve, co.commit(curve, np.ascontiguousarray(g[:n_used]), w1, np.ascontiguousarray(cols[nm][:n_used]), blinds[i]))
                        if with_blind else co.jac_to_affine_ints(curve, co.best_multiexp(curve, cols[nm][:n_used], g[:n_used])))
        return tab[key]

    def batch(sel, with_blind, out_kind, n_used=n):
        cnt = len(sel)
        arr = C.c_void_p * cnt
        d_outs = torch.zeros((cnt, 8 if out_kind else 12), dtype=torch.int64, device=dev)
        bl = arr(*[d_bl[names.index(nm)].data_ptr() for nm in sel]) if with_blind else None
        rc = lib.h2_commit_batch_device(hd, arr(*[d_cols[nm].data_ptr() for nm in sel]), cnt, n_used, None, bl, h.FORM_MONTGOMERY, out_kind,
                                        arr(*[d_outs[i].data_ptr() for i in range(cnt)]), None)
        assert rc == 0, lib.h2_last_error()
        torch.cuda.synchronize()
        got = d_outs.cpu().numpy().view(np.uint64)
        for i, nm in enumerate(sel):
            assert affine_of(curve, got[i]) == want(nm, with_blind, n_used), (nm, cnt, with_blind, n_used)

    full = k <= 18
    batch(names[:5], True, 0)                               # the five shapes in one launch set, Jacobian out
    batch(["dense", "dense2"], False, 1)                    # two columns, no blind, affine out
    batch(names[:8], True, 1)                               # a full group
    batch(["dense", "all_zero", "dense2"], False, 0)        # a column without a single entry between two dense ones
    batch(["all_zero", "all_zero"], True, 0)                # only the blinds
    if full:
        batch(names[:9], True, 0)                           # nine: two groups on two internal streams
        batch(["all_equal", "all_equal", "zeros90"], True, 0)   # the same column twice (shared input, separate work areas)
        batch(["dense", "q-1-i", "dense3"], True, 0, n_used=n - 5)      # prefix of the table (IPA rounds commit over the first n' bases)
        batch(["dense", "zeros90"], False, 0, n_used=17)    # tiny prefix: the shape falls back to one commit per column
    # a repeat on the warm workspaces gives the same points, and a lone commit right after a batch is unaffected
    batch(names[:5], True, 0)
    d_one = torch.zeros(12, dtype=torch.int64, device=dev)
    assert lib.h2_commit_device(hd, d_cols["dense"].data_ptr(), n, None, d_bl[0].data_ptr(), h.FORM_MONTGOMERY, 0, d_one.data_ptr(), None) == 0
    torch.cuda.synchronize()
    assert affine_of(curve, d_one.cpu().numpy().view(np.uint64)) == want("dense", True)
    assert lib.h2_bases_free(hd) == 0


def test_column_batched_commit_one_launch_per_column_form():
    """H2_BATCH_JOIN=0 (laboratory build only) keeps the form that launches msm_accumulate once per column (blockIdx.z); the switch is read
    once per process, so the same parity test runs in a child process that loads that build with it set."""
    from conftest import run_test_in_ab_child
    run_test_in_ab_child(__file__, "test_column_batched_commit_matches_oracle and 16-16", H2_BATCH_JOIN="0")
