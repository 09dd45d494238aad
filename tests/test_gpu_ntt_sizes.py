"""Elementwise parity of the NTT and the EvaluationDomain transforms AT THE MEASURED SIZES (BASELINE.json configs 3 and 4):
`best_fft` (halo2_proofs/src/arithmetic.rs:192-295) at 2^20 and 2^22 on both Pasta fields (the 2- and 3-pass plans of
csrc/ntt.hip), an arbitrary non-root omega at 2^20 (benches/fft.rs:17), `ifft` (poly/domain.rs:375-383),
`coeff_to_extended` 2^20 -> 2^21 and -> 2^22 (:241-255), `extended_to_coeff` at 2^21 and 2^22 (:303-325) and the batched
column entry point at 2^20 -- every index compared with the C oracle.  A round trip alone cannot see an error shared by the
forward and the inverse plan; these do.  `-m gpu` only; the oracle needs 0.1-0.6 s per transform."""
import numpy as np
import pytest

import halo2_amd as h
from halo2_amd import fields
from oracle import c_oracle as co
from oracle import pasta as o

pytestmark = pytest.mark.gpu


def mont(field, v):
    return fields.scalar_limbs(v, field, True)


@pytest.mark.parametrize("field", [h.FP, h.FQ])
@pytest.mark.parametrize("log_n", [19, 20, 21, 22])
def test_best_fft_elementwise_at_size(field, log_n):
    """host-pointer h2_ntt, root-of-unity omega: 2^19 / 2^20 run as two passes, 2^21 / 2^22 as three."""
    m = fields.MODULUS[field]
    a = co.random_field(field, 7100 + log_n, 1 << log_n)
    omega = mont(field, o.omega_for(m, log_n))
    want = co.best_fft(field, a, omega, log_n)
    got = h.best_fft(a.copy(), omega, log_n, field)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("field", [h.FP, h.FQ])
def test_best_fft_nonroot_omega_2_20(field):
    """benches/fft.rs:17 hands best_fft a RANDOM field element as omega: the butterfly network itself must match."""
    m = fields.MODULUS[field]
    log_n = 20
    a = co.random_field(field, 7200, 1 << log_n)
    w = co.random_field(field, 7201, 1)[0]              # already a Montgomery-form element of the field
    want = co.best_fft(field, a, w, log_n)
    got = h.best_fft(a.copy(), w, log_n, field)
    assert np.array_equal(got, want)
    # canonical-form buffers (what 100 % safe Rust would pass: to_repr() limbs)
    a_can, w_can = co.from_mont(field, a), co.from_mont(field, w.reshape(1, 4))[0]
    got_can = h.best_fft(a_can.copy(), w_can, log_n, field, form=h.FORM_CANONICAL)
    assert np.array_equal(got_can, co.from_mont(field, want))
    assert m  # (silence linters: m documents which modulus the field id selects)


@pytest.mark.parametrize("field,j,k", [(h.FP, 3, 20), (h.FQ, 3, 20), (h.FP, 5, 20), (h.FQ, 5, 19)])
def test_domain_transforms_elementwise_at_size(field, j, k):
    """config 4's transforms at its sizes: lagrange_to_coeff 2^k, coeff_to_extended 2^k -> 2^(k+1) (simple-example, j = 3)
    and -> 2^(k+2) (degree-5 circuits), extended_to_coeff and divide_by_vanishing_poly at the extended size."""
    dom = h.EvaluationDomain(j, k, field)
    ref = o.EvaluationDomain(j, k, fields.MODULUS[field])
    assert dom.extended_k == ref.extended_k == k + (1 if j == 3 else 2)
    a = co.random_field(field, 7300 + 5 * k + j, dom.n)
    coeff_want = co.ifft(field, a, mont(field, ref.omega_inv), k, mont(field, ref.ifft_divisor))
    coeff = dom.lagrange_to_coeff(a.copy())
    assert np.array_equal(coeff, coeff_want)
    ext_want = co.coeff_to_extended(field, coeff_want, k, ref.extended_k, mont(field, ref.g_coset), mont(field, ref.g_coset_inv),
                                    mont(field, ref.extended_omega))
    ext = dom.coeff_to_extended(coeff)
    assert np.array_equal(ext, ext_want)
    t = co.to_mont(field, co.ints_to_limbs(ref.t_evaluations))
    assert np.array_equal(dom.divide_by_vanishing_poly(ext.copy()), co.divide_by_vanishing_poly(field, ext_want, ref.extended_k, t))
    # a generic extended vector (not the image of a low-degree polynomial): every output index carries information
    e = co.random_field(field, 7400 + k + j, 1 << ref.extended_k)
    back_want = co.extended_to_coeff(field, e, ref.extended_k, mont(field, ref.g_coset), mont(field, ref.g_coset_inv),
                                     mont(field, ref.extended_omega_inv), mont(field, ref.extended_ifft_divisor))
    back = dom.extended_to_coeff(e.copy())
    assert np.array_equal(back, back_want[: dom.n * dom.quotient_poly_degree])
    # and the round trip through the coset
    assert np.array_equal(dom.extended_to_coeff(ext.copy())[: dom.n], coeff_want)


@pytest.mark.parametrize("field", [h.FP, h.FQ])
def test_fft_batch_device_2_20(field):
    """h2_ntt_batch_device / h2_ifft_batch_device at 2^20: the small-tile plan over internal streams (what bench.py's
    independent_columns leg and create_proof's column phases run) against the oracle, column by column."""
    import torch
    k = 20
    m = fields.MODULUS[field]
    dev = torch.device("cuda:0")
    cols = [co.random_field(field, 7500 + i, 1 << k) for i in range(4)]
    omega = mont(field, o.omega_for(m, k))
    d = [torch.from_numpy(c.view(np.int64)).to(dev) for c in cols]
    h.best_fft_batch(d, omega, k, field)
    torch.cuda.synchronize()
    for c, t in zip(cols, d):
        assert np.array_equal(t.cpu().numpy().view(np.uint64), co.best_fft(field, c, omega, k))
    dom = h.EvaluationDomain(3, k, field)
    d = [torch.from_numpy(c.view(np.int64)).to(dev) for c in cols]
    dom.lagrange_to_coeff_batch(d)
    torch.cuda.synchronize()
    for c, t in zip(cols, d):
        want = co.ifft(field, c, mont(field, dom.omega_inv), k, mont(field, dom.ifft_divisor))
        assert np.array_equal(t.cpu().numpy().view(np.uint64), want)


def test_device_resident_fft_2_22_in_place_and_inverse():
    """config 3 on device-resident data: forward 2^22 elementwise, then the fused inverse returns the input."""
    import torch
    field, log_n = h.FP, 22
    m = fields.MODULUS[field]
    a = co.random_field(field, 7600, 1 << log_n)
    omega = o.omega_for(m, log_n)
    d = torch.from_numpy(a.view(np.int64)).to("cuda:0")
    h.best_fft(d, mont(field, omega), log_n, field)
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy().view(np.uint64), co.best_fft(field, a, mont(field, omega), log_n))
    from halo2_amd.arithmetic import _p, _stream_ptr
    from halo2_amd._lib import check, lib
    check(lib().h2_ifft_device(field, d.data_ptr(), log_n, _p(mont(field, pow(omega, -1, m))), _p(mont(field, pow(1 << log_n, -1, m))),
                               h.FORM_MONTGOMERY, _stream_ptr()), "h2_ifft_device")
    torch.cuda.synchronize()
    assert np.array_equal(d.cpu().numpy().view(np.uint64), a)
