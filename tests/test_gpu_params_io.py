"""`Params::write` / `Params::read` (halo2_proofs/src/poly/commitment.rs:169-205) on the device: compression and the
square-root decompression against the oracle's restatement of pasta_curves' to_bytes / from_bytes, bit-exact, plus the
reference's own write -> read round trip (poly/commitment.rs:323-326).  Runs only on a real MI355X (`-m gpu`)."""
import io

import numpy as np
import pytest

import halo2_amd as h
from halo2_amd import fields
from oracle import c_oracle as co
from oracle import pasta as o

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("curve", [h.PALLAS, h.VESTA])
def test_compress_decompress_match_oracle(curve):
    bm = o.CURVES[curve][0]
    n = 300
    pts = co.generate_bases(curve, 77, n)
    pts[5] = 0                                                     # the identity
    ints = [co.affine_to_ints(curve, p) for p in pts]
    want = b"".join(o.point_to_bytes(pt, bm) for pt in ints)
    got = h.points_to_bytes(pts, curve)
    assert got == want
    # canonical-form input gives the same bytes
    bf = fields.CURVE_FIELDS[curve][0]
    assert h.points_to_bytes(co.from_mont(bf, pts.reshape(-1, 4)).reshape(-1, 8), curve, h.FORM_CANONICAL) == want
    back = h.points_from_bytes(got, curve)
    assert np.array_equal(back, pts)
    assert [o.point_from_bytes(want[32 * i: 32 * i + 32], bm) for i in range(n)] == ints
    # the other root: flip the sign bit
    flipped = bytearray(want[32:64])
    flipped[31] ^= 0x80
    neg = h.points_from_bytes(bytes(flipped), curve)[0]
    assert co.affine_to_ints(curve, neg) == (ints[1][0], bm - ints[1][1])


def test_invalid_encodings_rejected():
    curve, bm = h.VESTA, o.CURVES[h.VESTA][0]
    good = h.points_to_bytes(co.generate_bases(curve, 78, 4), curve)
    for bad in (bytes(31) + b"\x80",                               # (0, odd)
                b"\xff" * 31 + b"\x7f",                            # x >= p
                (bm).to_bytes(32, "little")):                      # x == p
        with pytest.raises(ValueError):
            h.points_from_bytes(good[:64] + bad + good[64:], curve)
    # an x whose x^3 + 5 is a non-residue
    x = 1
    while o.sqrt_mod((x * x * x + 5) % bm, bm) is not None:
        x += 1
    with pytest.raises(ValueError):
        h.points_from_bytes(x.to_bytes(32, "little"), curve)
    with pytest.raises(ValueError):
        h.points_from_bytes(good[:33], curve)


def test_params_write_read_roundtrip_and_oracle_bytes():
    curve, k = h.PALLAS, 8
    n = 1 << k
    bm = o.CURVES[curve][0]
    g = co.generate_bases(curve, 80, n)
    w, u = co.generate_bases(curve, 81, 1)[0], co.generate_bases(curve, 82, 1)[0]
    params = h.Params.from_generators(curve, k, g, None, w, u)
    buf = io.BytesIO()
    params.write(buf)
    raw = buf.getvalue()
    ai = lambda pts: [co.affine_to_ints(curve, p) for p in pts]
    assert raw == o.params_write(k, ai(params.g), ai(params.g_lagrange), ai([w])[0], ai([u])[0], bm)
    again = h.Params.read(io.BytesIO(raw), curve)                  # commitment.rs:323-326
    assert again.k == k and np.array_equal(again.g, params.g) and np.array_equal(again.g_lagrange, params.g_lagrange)
    assert np.array_equal(again.w, params.w) and np.array_equal(again.u, params.u)
    sf = fields.CURVE_FIELDS[curve][1]
    poly, blind = co.random_field(sf, 83, n), h.Blind(co.random_field(sf, 84, 1)[0])
    assert np.array_equal(again.commit(poly, blind, affine=True), params.commit(poly, blind, affine=True))
    with pytest.raises(ValueError):
        h.Params.read(io.BytesIO(raw[:-1]), curve)
    params.close(), again.close()


def test_roundtrip_2_20_device():
    """Full-size property: decompress(compress(P)) == P for 2^20 points, device-resident."""
    import ctypes as C
    import torch
    curve, n = h.VESTA, 1 << 20
    pts = co.generate_bases(curve, 85, n)
    dev = torch.device("cuda:0")
    d_p = torch.from_numpy(pts.view(np.int64)).to(dev)
    d_b = torch.empty((n, 4), dtype=torch.int64, device=dev)
    d_q = torch.empty_like(d_p)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    lib = h.lib()
    assert lib.h2_points_compress_device(curve, d_p.data_ptr(), n, h.FORM_MONTGOMERY, d_b.data_ptr(), st) == 0
    assert lib.h2_points_decompress_device(curve, d_b.data_ptr(), n, h.FORM_MONTGOMERY, d_q.data_ptr(), st) == 0
    assert torch.equal(d_p, d_q)
