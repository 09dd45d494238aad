"""The grouped form of the generic multiexp (csrc/msm_generic.hip; best_multiexp without a registered table, arithmetic.rs:143-180, from
2^18 + 1 device-resident points): the endomorphism split once into digit rows, window slices sorted / accumulated / folded in groups.  Parity
against the C oracle (bit-exact canonical affine coordinates) over the column shapes the reference's callers produce and the degenerate ones
that stress a group's sort (every entry of a slice in ONE bucket, slices with no entry at all), over odd sizes, both curves, both forms --
a lone call (three groups on the library's own streams) and a call that finds another stream's multiexp in flight (one group on the caller's
stream) -- and at a size whose pass-1 workgroups take 8192 columns.  `h2_msm_device` is the entry point: device tensors in, device point out."""
import numpy as np
import pytest

import halo2_amd as h
from halo2_amd import fields
from oracle import c_oracle as co
from oracle import pasta as o

pytestmark = pytest.mark.gpu


def affine_of(curve, jac):
    return co.jac_to_affine_ints(curve, np.ascontiguousarray(jac, dtype=np.uint64))


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64)).cuda()


def _shapes(curve, n, seed):
    sf = fields.CURVE_FIELDS[curve][1]
    sm = o.CURVES[curve][1]
    dense = co.random_field(sf, seed, n)
    zeros90 = dense.copy()
    zeros90[np.arange(n) % 10 != 0] = 0
    equal = np.tile(dense[:1], (n, 1))                                   # every scalar equal: each slice's entries in one bucket (the big-bin kernels, heavy buckets)
    small = fields.to_limbs([((i * 2654435761) & 0xFFFF) for i in range(4096)], sf, True)
    small = np.ascontiguousarray(np.tile(small, ((n + 4095) // 4096, 1))[:n])      # all below 2^16: seven of nine slices hold nothing
    top = fields.to_limbs([(sm - 1 - i) % sm for i in range(4096)], sf, True)
    top = np.ascontiguousarray(np.tile(top, ((n + 4095) // 4096, 1))[:n])          # q - 1 - i: both halves of the split near their bounds
    return {"dense": dense, "zeros90": zeros90, "all_equal": equal, "below_2^16": small, "q-1-i": top, "all_zero": np.zeros_like(dense)}


@pytest.mark.parametrize("curve", [h.PALLAS, h.VESTA])
def test_grouped_generic_multiexp_column_shapes(curve):
    import torch
    n = (1 << 18) + 4099
    bases = co.generate_bases(curve, 4100 + curve, n)
    bases[5] = 0                                                         # an identity base, and a repeated one
    bases[7] = bases[6]
    d_b = _dev(bases)
    for name, sc in _shapes(curve, n, 4200 + curve).items():
        got = h.best_multiexp(_dev(sc), d_b, curve)
        torch.cuda.synchronize()
        want = co.best_multiexp(curve, sc, bases)
        assert affine_of(curve, got.cpu().numpy().view(np.uint64)) == co.jac_to_affine_ints(curve, want), name


@pytest.mark.parametrize("curve,n", [(h.PALLAS, (1 << 18) + 1), (h.VESTA, (1 << 19) + 12345), (h.PALLAS, (1 << 20) + 1)])
def test_grouped_generic_multiexp_both_forms_side_by_side(curve, n):
    """Calls enqueued back to back on three streams: the first finds nothing in flight (latency form), the ones behind it find a multiexp of
    another stream in flight (throughput form below 2^22 points); affine and Jacobian outputs, canonical and Montgomery inputs."""
    import torch
    sf, bf = fields.CURVE_FIELDS[curve][1], fields.CURVE_FIELDS[curve][0]
    bases = co.generate_bases(curve, 4300 + curve, n)
    sc = [co.random_field(sf, 4310 + i, n) for i in range(3)]
    want = [co.jac_to_affine_ints(curve, co.best_multiexp(curve, s, bases)) for s in sc]
    d_b, d_s = _dev(bases), [_dev(s) for s in sc]
    d_b_can = _dev(co.from_mont(bf, bases.reshape(2 * n, 4)).reshape(n, 8))
    d_s_can = _dev(co.from_mont(sf, sc[2]))
    streams = [torch.cuda.Stream() for _ in range(3)]
    for rep in range(3):
        outs = []
        for i in range(6):
            with torch.cuda.stream(streams[i % 3]):
                if i == 5:
                    outs.append((2, True, h.best_multiexp(d_s_can, d_b_can, curve, form=h.FORM_CANONICAL, affine=True)))
                else:
                    outs.append((i % 3, False, h.best_multiexp(d_s[i % 3], d_b, curve)))
        torch.cuda.synchronize()
        for which, is_affine, out in outs:
            got = out.cpu().numpy().view(np.uint64)
            if is_affine:                                                # canonical affine coordinates out: compare as integers
                x, y = fields.from_limbs(got[:4].reshape(1, 4), bf, False)[0], fields.from_limbs(got[4:].reshape(1, 4), bf, False)[0]
                assert (x, y) == want[which], (rep, which)
            else:
                assert affine_of(curve, got) == want[which], (rep, which)


def test_grouped_generic_multiexp_large_pass1_workgroups():
    """2^21 + 3 points: the groups' pass 1 takes 8192 digit columns per workgroup and 12-bit column indices no longer leave room for more than
    a few low key bits in the tagged entry (more pass-1 bins); the carry slice's single bucket holds ~220 K entries."""
    import torch
    curve, n = h.PALLAS, (1 << 21) + 3
    sf = fields.CURVE_FIELDS[curve][1]
    bases = co.generate_bases(curve, 4400, n)
    sc = co.random_field(sf, 4401, n)
    got = h.best_multiexp(_dev(sc), _dev(bases), curve)
    torch.cuda.synchronize()
    assert affine_of(curve, got.cpu().numpy().view(np.uint64)) == co.jac_to_affine_ints(curve, co.best_multiexp(curve, sc, bases))
