"""Parity of the HIP path (through the C ABI) against the oracle on seeded inputs.  Bit-exact bar:
identical canonical affine (x, y) for MSM results, identical field elements at every index for NTTs.
Runs only on a real MI355X (`-m gpu`)."""
import os

import numpy as np
import pytest

import halo2_amd as h
from halo2_amd import fields
from oracle import c_oracle as co
from oracle import pasta as o

pytestmark = pytest.mark.gpu


def mont(field, v):
    return fields.scalar_limbs(v, field, True)


def affine_of(curve, out):
    """Jacobian (12,) or affine (8,) Montgomery limbs -> canonical (x, y) / None, via the oracle."""
    out = np.ascontiguousarray(out, dtype=np.uint64)
    return co.jac_to_affine_ints(curve, out) if out.shape[0] == 12 else co.affine_to_ints(curve, out)


# ------------------------------------------------------------------ NTT
@pytest.mark.parametrize("field", [h.FP, h.FQ])
@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 8, 9, 10, 13, 16, 17])
def test_best_fft_matches_oracle(field, log_n):
    m = fields.MODULUS[field]
    a = co.random_field(field, 900 + log_n, 1 << log_n)
    omega = mont(field, o.omega_for(m, log_n))
    want = co.best_fft(field, a, omega, log_n)
    got = h.best_fft(a.copy(), omega, log_n, field)
    assert np.array_equal(got, want)


def test_best_fft_canonical_form_and_nonroot_omega():
    field, m, log_n = h.FP, o.P, 11
    a = co.random_field(field, 77, 1 << log_n)
    w = 0x1234567 % m                           # benches/fft.rs:17 style arbitrary omega
    want = co.best_fft(field, a, mont(field, w), log_n)
    got = h.best_fft(a.copy(), mont(field, w), log_n, field)
    assert np.array_equal(got, want)
    # canonical-form buffers in, canonical out
    a_can = co.from_mont(field, a)
    got_can = h.best_fft(a_can.copy(), fields.scalar_limbs(w, field, False), log_n, field, form=h.FORM_CANONICAL)
    assert np.array_equal(got_can, co.from_mont(field, want))


def test_best_fft_rejects_bad_length():
    a = co.random_field(h.FP, 1, 12)
    with pytest.raises(ValueError):
        h.best_fft(a, mont(h.FP, 1), 4, h.FP)


@pytest.mark.parametrize("field,j,k", [(h.FP, 3, 6), (h.FP, 5, 9), (h.FQ, 3, 10), (h.FP, 4, 12)])
def test_domain_transforms_match_oracle(field, j, k):
    dom = h.EvaluationDomain(j, k, field)
    ref = o.EvaluationDomain(j, k, fields.MODULUS[field])
    assert (dom.extended_k, dom.omega, dom.extended_omega, dom.g_coset) == (ref.extended_k, ref.omega, ref.extended_omega, ref.g_coset)
    assert dom.t_evaluations == ref.t_evaluations
    a = co.random_field(field, 5 * k + j, dom.n)
    coeff_want = co.ifft(field, a, mont(field, ref.omega_inv), k, mont(field, ref.ifft_divisor))
    coeff = dom.lagrange_to_coeff(a.copy())
    assert np.array_equal(coeff, coeff_want)
    ext_want = co.coeff_to_extended(field, coeff_want, k, ref.extended_k, mont(field, ref.g_coset), mont(field, ref.g_coset_inv),
                                    mont(field, ref.extended_omega))
    ext = dom.coeff_to_extended(coeff)
    assert np.array_equal(ext, ext_want)
    back_want = co.extended_to_coeff(field, ext_want, ref.extended_k, mont(field, ref.g_coset), mont(field, ref.g_coset_inv),
                                     mont(field, ref.extended_omega_inv), mont(field, ref.extended_ifft_divisor))
    back = dom.extended_to_coeff(ext.copy())
    assert np.array_equal(back, back_want[: dom.n * dom.quotient_poly_degree])
    assert np.array_equal(back[: dom.n], coeff_want)          # round trip
    t = co.to_mont(field, co.ints_to_limbs(ref.t_evaluations))
    assert np.array_equal(dom.divide_by_vanishing_poly(ext.copy()), co.divide_by_vanishing_poly(field, ext_want, ref.extended_k, t))


def test_fft_roundtrip_2_22():
    """BASELINE config 3: 2^22 Fp forward + inverse returns the input (size-independent property);
    forward checked elementwise against the oracle at 2^18."""
    field, m = h.FP, o.P
    for log_n, check_oracle in ((18, True), (22, False)):
        a = co.random_field(field, 4000 + log_n, 1 << log_n)
        omega = o.omega_for(m, log_n)
        fwd = h.best_fft(a.copy(), mont(field, omega), log_n, field)
        if check_oracle:
            assert np.array_equal(fwd, co.best_fft(field, a, mont(field, omega), log_n))
        dom_div = pow(1 << log_n, -1, m)
        back = h.best_fft(fwd.copy(), mont(field, pow(omega, -1, m)), log_n, field)
        lib = h.lib()
        # scale by 1/n through the fused entry point as well
        from halo2_amd.arithmetic import _p
        fused = fwd.copy()
        assert lib.h2_ifft(field, _p(fused), log_n, _p(mont(field, pow(omega, -1, m))), _p(mont(field, dom_div)), h.FORM_MONTGOMERY) == 0
        assert np.array_equal(fused, a)
        # unscaled inverse equals n * a: check a sample through the oracle's field mul
        idx = [0, 1, 12345 % (1 << log_n), (1 << log_n) - 1]
        got = fields.from_limbs(back[idx], field)
        want = [(v << log_n) % m for v in fields.from_limbs(a[idx], field)]
        assert got == want


# ------------------------------------------------------------------ MSM
SIZES = [0, 1, 2, 3, 4, 31, 32, 33, 255, 256, 257, 1000, 4096, 65537]


@pytest.mark.parametrize("curve", [h.PALLAS, h.VESTA])
def test_best_multiexp_matches_oracle(curve):
    """arithmetic.rs:440-458 test_multiexp, with the oracle's best_multiexp as the expected value."""
    sf = co.field_of_curve(curve, "scalar")
    for n in SIZES:
        scal = co.random_field(sf, 1000 + n, n)
        bases = co.generate_bases(curve, 5000 + n, n)
        want = co.jac_to_affine_ints(curve, co.best_multiexp(curve, scal, bases))
        got = h.best_multiexp(scal, bases, curve)
        assert affine_of(curve, got) == want, n
        got_aff = h.best_multiexp(scal, bases, curve, affine=True)
        assert affine_of(curve, got_aff) == want, n
        if n:
            assert co.lib().orc_point_on_curve(curve, co._p(np.ascontiguousarray(got_aff))) == 1


def test_best_multiexp_rejects_length_mismatch():
    with pytest.raises(ValueError):
        h.best_multiexp(np.zeros((4, 4), np.uint64), np.zeros((5, 8), np.uint64), h.PALLAS)


def test_msm_edge_cases():
    """zero scalars, q-1, identity bases, duplicate bases, a base with its negation, sparse columns."""
    curve = h.PALLAS
    bm, sm = o.CURVES[curve]
    sf = co.field_of_curve(curve, "scalar")
    n = 600
    bases = co.generate_bases(curve, 31337, n)
    b_int = [co.affine_to_ints(curve, bases[i]) for i in range(n)]
    b_int[3] = None
    b_int[5] = b_int[4]
    b_int[7] = o.ec_neg(b_int[6], bm)
    for i in range(100, 140):
        b_int[i] = b_int[100]                    # 40 copies of one base
    bases = co.points_to_mont(curve, b_int)
    s_int = co.limbs_to_ints(co.from_mont(sf, co.random_field(sf, 4242, n)))
    s_int[0] = 0
    s_int[1] = sm - 1
    s_int[4] = s_int[5] = 12345
    s_int[6] = s_int[7] = 777
    for i in range(100, 140):
        s_int[i] = 99                            # same digit everywhere -> one heavy bucket, P + P path
    for i in range(200, 600):
        s_int[i] = 0 if i % 10 else s_int[i]     # 90 % zeros
    scal = co.to_mont(sf, co.ints_to_limbs(s_int))
    want = co.jac_to_affine_ints(curve, co.best_multiexp(curve, scal, bases))
    assert affine_of(curve, h.best_multiexp(scal, bases, curve)) == want
    # all scalars equal + all bases equal -> n * s * B
    s_same = co.to_mont(sf, co.ints_to_limbs([9] * n))
    b_same = co.points_to_mont(curve, [b_int[0]] * n)
    assert affine_of(curve, h.best_multiexp(s_same, b_same, curve)) == o.ec_mul(9 * n, b_int[0], bm)
    # everything cancels
    half = n // 2
    b_pm = co.points_to_mont(curve, b_int[8:8 + half] + [o.ec_neg(p, bm) for p in b_int[8:8 + half]])
    s_pm = co.to_mont(sf, co.ints_to_limbs(s_int[8:8 + half] * 2))
    out = h.best_multiexp(s_pm, b_pm, curve)
    assert affine_of(curve, out) is None and not out[8:].any()
    # canonical form in / out
    got = h.best_multiexp(co.from_mont(sf, scal), co.from_mont(co.field_of_curve(curve, "base"), bases.reshape(-1, 4)).reshape(-1, 8),
                          curve, form=h.FORM_CANONICAL, affine=True)
    assert (fields.from_limbs(got.reshape(2, 4), montgomery=False)[0], fields.from_limbs(got.reshape(2, 4), montgomery=False)[1]) == want


def test_msm_linearity_and_split_2_20():
    """BASELINE config 2 size: properties that do not need the oracle at full size -- linearity in the
    scalars and split-and-sum over 8 ranges (the multi-GPU partition) -- plus the oracle on a 2^16 prefix."""
    curve = h.PALLAS
    sf = co.field_of_curve(curve, "scalar")
    n = 1 << 20
    bases = co.generate_bases(curve, 20, n)
    a = co.random_field(sf, 21, n)
    b = co.random_field(sf, 22, n)
    # linearity MSM(a) + MSM(b) == MSM(a + b) on the first 2^14 entries (a + b formed with Python ints)
    m_ = 1 << 14
    ai = fields.from_limbs(a[:m_], sf)
    bi = fields.from_limbs(b[:m_], sf)
    s = fields.to_limbs([(x + y) % fields.MODULUS[sf] for x, y in zip(ai, bi)], sf)
    ra = h.best_multiexp(a[:m_], bases[:m_], curve)
    rb = h.best_multiexp(b[:m_], bases[:m_], curve)
    rs = h.best_multiexp(s, bases[:m_], curve)
    assert affine_of(curve, h.points_sum(np.stack([ra, rb]), curve)) == affine_of(curve, rs)
    # full size: whole == sum of 8 contiguous range partials
    whole = h.best_multiexp(a, bases, curve)
    parts = [h.best_multiexp(a[i * n // 8:(i + 1) * n // 8], bases[i * n // 8:(i + 1) * n // 8], curve) for i in range(8)]
    assert affine_of(curve, h.points_sum(np.stack(parts), curve)) == affine_of(curve, whole)
    # oracle on a 2^16 prefix
    k = 1 << 16
    assert affine_of(curve, h.best_multiexp(a[:k], bases[:k], curve)) == co.jac_to_affine_ints(curve, co.best_multiexp(curve, a[:k], bases[:k]))


@pytest.mark.parametrize("curve", [h.PALLAS, h.VESTA])
def test_params_commit_matches_oracle(curve):
    """Params::commit / commit_lagrange with the blind term (poly/commitment.rs:119-150) and
    test_commit_lagrange (:258-302): commit(iFFT(a)) == commit_lagrange(a)."""
    k = 5
    n = 1 << k
    bm, sm = o.CURVES[curve]
    sf = co.field_of_curve(curve, "scalar")
    g = co.generate_bases(curve, 2024 + curve, n)
    g_int = [co.affine_to_ints(curve, g[i]) for i in range(n)]
    w_int = o.ec_mul(424242, (bm - 1, 2), bm)
    u_int = o.ec_mul(171717, (bm - 1, 2), bm)
    dom = h.EvaluationDomain(1, k, sf)
    g_lag = []
    for i in range(n):
        acc = None
        for j in range(n):
            acc = o.ec_add(acc, o.ec_mul(pow(dom.omega_inv, i * j, sm) * dom.ifft_divisor % sm, g_int[j], bm), bm)
        g_lag.append(acc)
    params = h.Params.from_generators(curve, k, g, co.points_to_mont(curve, g_lag), co.points_to_mont(curve, [w_int])[0],
                                      co.points_to_mont(curve, [u_int])[0])
    a = co.random_field(sf, 31, n)
    blind = h.Blind(fields.scalar_limbs(987654321, sf))
    coeff = dom.lagrange_to_coeff(a.copy())
    c1 = params.commit(coeff, blind)
    c2 = params.commit_lagrange(a, blind)
    want = co.jac_to_affine_ints(curve, co.commit(curve, g, co.points_to_mont(curve, [w_int])[0], coeff, blind.value))
    assert affine_of(curve, c1) == want
    assert affine_of(curve, c2) == want
    assert affine_of(curve, params.commit(coeff, h.Blind(field=sf), affine=True)) == co.jac_to_affine_ints(
        curve, co.commit(curve, g, co.points_to_mont(curve, [w_int])[0], coeff, fields.scalar_limbs(1, sf)))
    params.close()


def test_registered_bases_prefix_blind_and_skew():
    """h2_commit on a registered (precomputed-table) basis: full commits with two different blind bases,
    prefix MSMs over the first n' bases (IPA rounds, poly/commitment/prover.rs:107-108), and heavily
    skewed columns (all-equal scalars, tiny scalars, 99 % zeros) -- the work partition must not care."""
    import ctypes as C
    from halo2_amd.arithmetic import _p
    curve = h.VESTA
    bm, sm = o.CURVES[curve]
    sf = co.field_of_curve(curve, "scalar")
    n = 3000
    g = co.generate_bases(curve, 777, n)
    lib = h.lib()
    hd = C.c_uint64(0)
    assert lib.h2_bases_register(curve, _p(g), n, h.FORM_MONTGOMERY, C.byref(hd)) == 0
    out = np.zeros(12, np.uint64)
    cols = {
        "uniform": co.random_field(sf, 1, n),
        "all_equal": co.to_mont(sf, co.ints_to_limbs([0xDEADBEEFCAFE] * n)),
        "tiny": co.to_mont(sf, co.ints_to_limbs([i % 7 for i in range(n)])),
        "sparse": co.to_mont(sf, co.ints_to_limbs([0 if i % 100 else (i * 7919 + 1) for i in range(n)])),
        "max": co.to_mont(sf, co.ints_to_limbs([sm - 1 - i for i in range(n)])),
    }
    w1 = co.points_to_mont(curve, [o.ec_mul(5, (bm - 1, 2), bm)])[0]
    w2 = co.points_to_mont(curve, [o.ec_mul(7, (bm - 1, 2), bm)])[0]
    blind = fields.scalar_limbs(123456789, sf)
    for name, col in cols.items():
        for w in (w1, w2, w1):
            assert lib.h2_commit(hd, _p(col), n, _p(w), _p(blind), h.FORM_MONTGOMERY, 0, _p(out)) == 0
            want = co.jac_to_affine_ints(curve, co.commit(curve, g, w, col, blind))
            assert affine_of(curve, out) == want, name
        for n_used in (0, 1, 17, 1500, n):
            assert lib.h2_commit(hd, _p(col), n_used, None, None, h.FORM_MONTGOMERY, 0, _p(out)) == 0
            want = co.jac_to_affine_ints(curve, co.best_multiexp(curve, col[:n_used], g[:n_used]))
            assert affine_of(curve, out) == want, (name, n_used)
    assert lib.h2_commit(hd, _p(cols["uniform"]), n + 1, None, None, h.FORM_MONTGOMERY, 0, _p(out)) == 1   # more than registered
    assert lib.h2_bases_free(hd) == 0
    assert lib.h2_bases_free(hd) == 4


def test_generic_msm_skewed_large():
    """2^17 points, every scalar identical (one bucket per window owns everything) and all < 2^16."""
    curve = h.PALLAS
    bm, sm = o.CURVES[curve]
    sf = co.field_of_curve(curve, "scalar")
    n = 1 << 17
    bases = co.generate_bases(curve, 99, n)
    same = co.to_mont(sf, co.ints_to_limbs([0x1234567] * n))
    assert affine_of(curve, h.best_multiexp(same, bases, curve)) == co.jac_to_affine_ints(curve, co.best_multiexp(curve, same, bases))
    small = co.to_mont(sf, co.ints_to_limbs([(i * 2654435761) & 0xFFFF for i in range(n)]))
    assert affine_of(curve, h.best_multiexp(small, bases, curve)) == co.jac_to_affine_ints(curve, co.best_multiexp(curve, small, bases))


def test_device_resident_path():
    """torch CUDA tensors in, device tensors out, on torch's current stream."""
    torch = pytest.importorskip("torch")
    curve, n = h.VESTA, 1 << 12
    sf = co.field_of_curve(curve, "scalar")
    scal = co.random_field(sf, 1, n)
    bases = co.generate_bases(curve, 2, n)
    d_s = torch.from_numpy(scal.view(np.int64)).cuda()
    d_b = torch.from_numpy(bases.view(np.int64)).cuda()
    out = h.best_multiexp(d_s, d_b, curve)
    torch.cuda.synchronize()
    assert affine_of(curve, out.cpu().numpy().view(np.uint64)) == co.jac_to_affine_ints(curve, co.best_multiexp(curve, scal, bases))
    field, log_n = h.FP, 12
    a = co.random_field(field, 3, 1 << log_n)
    omega = mont(field, o.omega_for(o.P, log_n))
    d_a = torch.from_numpy(a.view(np.int64)).cuda()
    h.best_fft(d_a, omega, log_n, field)
    torch.cuda.synchronize()
    assert np.array_equal(d_a.cpu().numpy().view(np.uint64), co.best_fft(field, a, omega, log_n))


@pytest.mark.parametrize("binary,marker", [("field_check", "FIELD CHECK OK"), ("host_mirror_check", "HOST MIRROR CHECK OK")])
def test_native_drivers(binary, marker):
    """tests/native/*: the gfx950 field arithmetic (every multiplier variant, add, sub, inverse; both moduli; edge
    values) against the C oracle, and the C++ host mirror (halo2_amd/host/halo2_host.hpp) through the C ABI from a
    plain g++ program -- no Python, no torch in the process.  Built by __graft_entry__.build()."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", binary)
    if not os.path.exists(exe):
        pytest.skip(f"{exe} not built (run __graft_entry__.build())")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and marker in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_native_h2bench_parity():
    """bench/native/h2bench.cpp `parity`: the native timing / A-B driver's own sweep through the C ABI -- generic multiexps of small and
    odd sizes from device and from host pointers on both curves, device-resident transforms 2^1 .. 2^21 on both fields with a random
    (non-root) omega, registered commits with blinds at 2^14 and 2^18 over three streams -- every result against the C oracle."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "h2bench")
    if not os.path.exists(exe):
        pytest.skip(f"{exe} not built (run __graft_entry__.build())")
    out = subprocess.run([exe, "parity"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "H2BENCH OK" in out.stdout and "FAIL" not in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


@pytest.mark.parametrize("config,k", [("plonk-bench", 8), ("simple-example", 10)])
def test_create_proof_trace_replay(config, k):
    """BASELINE configs[0] / configs[3] at test size: the whole MSM / FFT call trace of one create_proof
    (bench/replay_create_proof.py), every output compared with the oracle."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "replay", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench", "replay_create_proof.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.run_trace(config, k, check=True)
    assert res["mismatches"] == [], res
    lag = mod.CONFIGS[config]["lagrange_columns"]
    assert res["ops"] == 3 * lag + 1 + 1 + mod.CONFIGS[config]["h_pieces"] + 2 + 2 * k
    if config == "plonk-bench":
        assert (res["msm_full"], res["ipa_msm"], res["extended_k"]) == (11, 16, 10)


def test_concurrent_streams_and_lane_fraction():
    """bench.py spreads independent column commits over several HIP streams with a reduced accumulate lane
    fraction: every commit must still be bit-exact (per-(device, stream) workspaces, one shared table)."""
    import ctypes as C
    torch = pytest.importorskip("torch")
    from halo2_amd.arithmetic import _p
    curve, n, ncols = h.PALLAS, 1 << 14, 9
    sf = co.field_of_curve(curve, "scalar")
    lib = h.lib()
    g = co.generate_bases(curve, 55, n)
    hd = C.c_uint64(0)
    assert lib.h2_bases_register(curve, _p(g), n, h.FORM_MONTGOMERY, C.byref(hd)) == 0
    cols = [co.random_field(sf, 600 + i, n) for i in range(ncols)]
    d_cols = [torch.from_numpy(c.view(np.int64)).cuda() for c in cols]
    d_out = torch.zeros((ncols, 12), dtype=torch.int64, device="cuda")
    assert lib.h2_set_option(b"msm_lane_fraction", 0.5) == 0
    assert lib.h2_set_option(b"msm_lane_fraction", 7.0) == 1 and lib.h2_set_option(b"no_such_option", 1.0) == 1
    streams = [torch.cuda.Stream() for _ in range(3)]
    torch.cuda.synchronize()
    for rep in range(2):
        for i in range(ncols):
            rc = lib.h2_commit_device(hd, d_cols[i].data_ptr(), n, None, None, h.FORM_MONTGOMERY, 0, d_out[i].data_ptr(),
                                      C.c_void_p(streams[i % 3].cuda_stream))
            assert rc == 0
    torch.cuda.synchronize()
    assert lib.h2_set_option(b"msm_lane_fraction", 1.0) == 0
    res = d_out.cpu().numpy().view(np.uint64)
    for i in range(ncols):
        assert affine_of(curve, res[i]) == co.jac_to_affine_ints(curve, co.best_multiexp(curve, cols[i], g)), i
    assert lib.h2_bases_free(hd) == 0


def test_commit_batch_matches_oracle():
    """h2_commit_batch_device: the independent column commits of a prover phase in one call (internal streams)."""
    torch = pytest.importorskip("torch")
    curve, k = h.VESTA, 12
    n = 1 << k
    sf = co.field_of_curve(curve, "scalar")
    g = co.generate_bases(curve, 901, n)
    gl = co.generate_bases(curve, 902, n)
    w = co.generate_bases(curve, 903, 1)[0]
    u = co.generate_bases(curve, 904, 1)[0]
    params = h.Params.from_generators(curve, k, g, gl, w, u)
    cols = [co.random_field(sf, 910 + i, n) for i in range(7)]
    blinds = [h.Blind(co.random_field(sf, 920 + i, 1)[0]) for i in range(7)]
    d_cols = [torch.from_numpy(c.view(np.int64)).cuda() for c in cols]
    for lagrange, basis in ((True, gl), (False, g)):
        out = params.commit_batch(d_cols, blinds, lagrange=lagrange)
        torch.cuda.synchronize()
        res = out.cpu().numpy().view(np.uint64)
        for i in range(7):
            assert affine_of(curve, res[i]) == co.jac_to_affine_ints(curve, co.commit(curve, basis, w, cols[i], blinds[i].value)), (lagrange, i)
    params.close()


@pytest.mark.parametrize("curve", [h.PALLAS, h.VESTA])
def test_ipa_round_kernels_match_oracle(curve):
    """parallel_generator_collapse (poly/commitment/prover.rs:154-166) and the p'/b fold (:128-131), run for every
    round of a k = 9 argument with the vectors kept on the device, against the oracle's restatement."""
    torch = pytest.importorskip("torch")
    bm, sm = o.CURVES[curve]
    sf = co.field_of_curve(curve, "scalar")
    k = 9
    n = 1 << k
    g = co.generate_bases(curve, 77 + curve, n)
    g[5] = 0                                            # an identity generator must survive the collapse
    p_vec = co.random_field(sf, 78, n)
    challenges = [fields.scalar_limbs(v, sf) for v in (0, 1, sm - 1, 2)] + [co.random_field(sf, 80 + j, 1)[0] for j in range(k - 4)]
    d_g = torch.from_numpy(g.view(np.int64)).cuda()
    d_p = torch.from_numpy(p_vec.view(np.int64)).cuda()
    ref_g, ref_p = g, p_vec
    for j, u in enumerate(challenges):
        ref_g = co.generator_collapse(curve, ref_g, u)
        ref_p = co.fold_scalars(sf, ref_p, u)
        d_g = h.parallel_generator_collapse(d_g, u, curve).contiguous()
        d_p = h.fold_scalars(d_p, u, sf).contiguous()
        torch.cuda.synchronize()
        assert np.array_equal(d_g.cpu().numpy().view(np.uint64), ref_g), j
        assert np.array_equal(d_p.cpu().numpy().view(np.uint64), ref_p), j
    assert ref_g.shape[0] == 1
    # host-pointer + canonical-form entry points
    bf = co.field_of_curve(curve, "base")
    g_can = co.from_mont(bf, g.reshape(-1, 4)).reshape(-1, 8)
    u = co.random_field(sf, 99, 1)[0]
    got = h.parallel_generator_collapse(g_can, co.from_mont(sf, u.reshape(1, 4))[0], curve, form=h.FORM_CANONICAL)
    assert np.array_equal(co.to_mont(bf, got.reshape(-1, 4)).reshape(-1, 8), co.generator_collapse(curve, g, u))
    with pytest.raises(ValueError):
        h.parallel_generator_collapse(g[:3], u, curve)


@pytest.mark.parametrize("curve", [h.PALLAS, h.VESTA])
def test_lagrange_basis_and_commit_lagrange_identity(curve):
    """Params::new's point FFT (poly/commitment.rs:77-100) on the device vs the oracle's restatement, then the
    reference's own test_commit_lagrange_{epaffine,eqaffine} (:258-302) at k = 6:
    commit(lagrange_to_coeff(a)) == commit_lagrange(a), which ties the MSM, the field iFFT and the point FFT together."""
    sf = co.field_of_curve(curve, "scalar")
    for k in (0, 1, 3, 6, 8, 9, 10):          # from k = 9 on the early stages run one twiddle per wave with the endomorphism split
        g = co.generate_bases(curve, 300 + k, 1 << k)
        assert np.array_equal(h.lagrange_basis(g, curve, k), co.lagrange_basis(curve, g, k)), k
    k = 6
    n = 1 << k
    g = co.generate_bases(curve, 306, n)
    w = co.generate_bases(curve, 307, 1)[0]
    u = co.generate_bases(curve, 308, 1)[0]
    params = h.Params.from_generators(curve, k, g, None, w, u)
    dom = h.EvaluationDomain(1, k, sf)
    a = co.random_field(sf, 309, n)
    alpha = h.Blind(co.random_field(sf, 310, 1)[0])
    b = dom.lagrange_to_coeff(a.copy())
    assert affine_of(curve, params.commit(b, alpha)) == affine_of(curve, params.commit_lagrange(a, alpha))
    params.close()


def test_lagrange_basis_k12_properties():
    """Larger point FFT without the oracle: sum_j g_lagrange[j] = g[0] (all Lagrange polynomials sum to 1) and
    commit_lagrange of the evaluations of X^3 equals g[3]."""
    curve, k = h.VESTA, 12
    n = 1 << k
    sf = co.field_of_curve(curve, "scalar")
    g = co.generate_bases(curve, 400, n)
    gl = h.lagrange_basis(g, curve, k)
    ones = np.tile(fields.scalar_limbs(1, sf), (n, 1))
    assert affine_of(curve, h.best_multiexp(ones, gl, curve)) == co.affine_to_ints(curve, g[0])
    dom = h.EvaluationDomain(1, k, sf)
    evals = fields.to_limbs([pow(dom.omega, 3 * i, dom.m) for i in range(n)], sf)
    assert affine_of(curve, h.best_multiexp(evals, gl, curve)) == co.affine_to_ints(curve, g[3])


def test_registered_heavy_buckets_2_16():
    """Registered-bases path with 2^16 identical scalars (a handful of buckets own everything: the workgroup-per-bucket
    finisher) and with more heavy buckets than the finisher's list holds (falls back to in-place sums)."""
    import ctypes as C
    from halo2_amd.arithmetic import _p
    curve = h.PALLAS
    bm, sm = o.CURVES[curve]
    sf = co.field_of_curve(curve, "scalar")
    n = 1 << 16
    g = co.generate_bases(curve, 1234, n)
    lib = h.lib()
    hd = C.c_uint64(0)
    assert lib.h2_bases_register(curve, _p(g), n, h.FORM_MONTGOMERY, C.byref(hd)) == 0
    out = np.zeros(12, np.uint64)
    same = co.to_mont(sf, co.ints_to_limbs([0x0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF] * n))
    assert lib.h2_commit(hd, _p(same), n, None, None, h.FORM_MONTGOMERY, 0, _p(out)) == 0
    assert affine_of(curve, out) == co.jac_to_affine_ints(curve, co.best_multiexp(curve, same, g))
    few = co.to_mont(sf, co.ints_to_limbs([(i % 3 + 1) * 0x0001000100010001000100010001000100010001000100010001000100010001 % sm for i in range(n)]))
    assert lib.h2_commit(hd, _p(few), n, None, None, h.FORM_MONTGOMERY, 0, _p(out)) == 0
    assert affine_of(curve, out) == co.jac_to_affine_ints(curve, co.best_multiexp(curve, few, g))
    rep = 0x0001000100010001000100010001000100010001000100010001000100010001
    many = co.to_mont(sf, co.ints_to_limbs([(((i * 2654435761) >> 7) % 600 + 1) * rep % sm for i in range(n)]))   # ~600 heavy buckets > the finisher's list
    assert lib.h2_commit(hd, _p(many), n, None, None, h.FORM_MONTGOMERY, 0, _p(out)) == 0
    assert affine_of(curve, out) == co.jac_to_affine_ints(curve, co.best_multiexp(curve, many, g))
    assert lib.h2_bases_free(hd) == 0


def test_msm_batch_matches_oracle():
    """h2_msm_batch_device: ragged independent multiexps in one call (the L_j / R_j pairs of the opening argument)."""
    import torch
    from halo2_amd.arithmetic import best_multiexp_batch
    curve = h.VESTA
    sf = fields.CURVE_FIELDS[curve][1]
    dev = torch.device("cuda:0")
    sizes = [1, 2, 33, 1000, 4097, 1 << 15, 0, 700]
    pairs, want = [], []
    for i, n in enumerate(sizes):
        sc, bs = co.random_field(sf, 300 + i, max(n, 1))[:n], co.generate_bases(curve, 400 + i, max(n, 1))[:n]
        pairs.append((torch.from_numpy(sc.view(np.int64)).to(dev), torch.from_numpy(bs.view(np.int64)).to(dev)))
        want.append(affine_of(curve, co.best_multiexp(curve, sc, bs)))
    got = best_multiexp_batch(pairs, curve).cpu().numpy().view(np.uint64)
    assert [affine_of(curve, g) for g in got] == want
    got_aff = best_multiexp_batch(pairs, curve, affine=True).cpu().numpy().view(np.uint64)
    assert [affine_of(curve, g) for g in got_aff] == want
    with pytest.raises(ValueError):
        best_multiexp_batch([(pairs[3][0], pairs[4][1])], curve)


def test_concurrent_host_threads():
    """SURVEY.md section 8b 'Threading': entry points are re-entrant -- rayon workers of the reference call best_multiexp /
    best_fft concurrently.  Four host threads hammer the host-pointer entry points; every result must equal the
    single-threaded one."""
    from concurrent.futures import ThreadPoolExecutor
    curve, field = h.VESTA, h.FP
    sf = fields.CURVE_FIELDS[curve][1]
    jobs = []
    for i in range(12):
        n = [257, 1 << 12, 3000, 1 << 14][i % 4]
        sc, bs = co.random_field(sf, 500 + i, n), co.generate_bases(curve, 600 + i, n)
        log_n = 9 + i % 4
        a = co.random_field(field, 700 + i, 1 << log_n)
        omega = mont(field, o.omega_for(fields.MODULUS[field], log_n))
        jobs.append((sc, bs, a, omega, log_n))

    def run(job):
        sc, bs, a, omega, log_n = job
        return (affine_of(curve, h.best_multiexp(sc, bs, curve)), h.best_fft(a.copy(), omega, log_n, field),
                h.eval_polynomial(a, omega, field), h.kate_division(a, omega, field))
    want = [run(j) for j in jobs]
    with ThreadPoolExecutor(max_workers=4) as ex:
        for rep in range(3):
            got = list(ex.map(run, jobs))
            for g, w in zip(got, want):
                assert g[0] == w[0] and all(np.array_equal(x, y) for x, y in zip(g[1:], w[1:]))


def test_registered_wide_windows_c20(monkeypatch):
    """Windows beyond 16 bits on the registered path (H2_MSM_C sweep knob: 13 windows of 20 bits, 2^19 buckets): 32-bit digit
    codes, the two-pass sort with 4096 bins, HBM-counted giant windows for sparse columns, row / column-sum fold."""
    import ctypes as C
    from halo2_amd.arithmetic import _p
    if not os.environ.get("H2_AB_CHILD"):       # the knob lives in the laboratory build only: the body runs in a child that loads it
        from conftest import run_test_in_ab_child
        run_test_in_ab_child(__file__, "test_registered_wide_windows_c20")
        return
    monkeypatch.setenv("H2_MSM_C", "20")
    curve, n = h.VESTA, 1 << 15
    sf = fields.CURVE_FIELDS[curve][1]
    g = co.generate_bases(curve, 910, n)
    w, blind = co.generate_bases(curve, 911, 1)[0], co.random_field(sf, 912, 1)[0]
    lib = h.lib()
    hd = C.c_uint64(0)
    assert lib.h2_bases_register(curve, _p(g), n, h.FORM_MONTGOMERY, C.byref(hd)) == 0
    monkeypatch.delenv("H2_MSM_C")
    dense = co.random_field(sf, 913, n)
    sparse = dense.copy()
    sparse[np.arange(n) % 50 != 0] = 0
    out = np.zeros(12, dtype=np.uint64)
    for col, nn in ((dense, n), (sparse, n), (dense, 1000), (dense, 1)):
        assert lib.h2_commit(hd, _p(col), nn, _p(w), _p(blind), h.FORM_MONTGOMERY, 0, _p(out)) == 0
        assert affine_of(curve, out) == co.jac_to_affine_ints(curve, co.commit(curve, g[:nn], w, col[:nn], blind))
    assert lib.h2_bases_free(hd) == 0


def test_fft_batch_matches_single():
    """h2_ntt_batch_device / h2_ifft_batch_device: independent columns on internal streams (small-tile plan) give the same
    vectors as one-at-a-time transforms (fewest-passes plan) and as the oracle."""
    import torch
    field, k = h.FP, 13
    m = fields.MODULUS[field]
    dev = torch.device("cuda:0")
    dom = h.EvaluationDomain(3, k, field)
    cols = [co.random_field(field, 1200 + i, 1 << k) for i in range(5)]
    omega = mont(field, o.omega_for(m, k))
    d = [torch.from_numpy(c.view(np.int64)).to(dev) for c in cols]
    h.best_fft_batch(d, omega, k, field)
    for c, t in zip(cols, d):
        assert np.array_equal(t.cpu().numpy().view(np.uint64), co.best_fft(field, c, omega, k))
    d = [torch.from_numpy(c.view(np.int64)).to(dev) for c in cols]
    dom.lagrange_to_coeff_batch(d)
    for c, t in zip(cols, d):
        assert np.array_equal(t.cpu().numpy().view(np.uint64), dom.lagrange_to_coeff(c.copy()))
    assert h.best_fft_batch([], omega, k, field) == []
    with pytest.raises(ValueError):
        h.best_fft_batch([d[0][:-1]], omega, k, field)


def test_trim_releases_and_recreates_workspaces():
    """h2_trim hands cached scratch back to the allocator; the next calls rebuild it and give the same results."""
    import torch
    curve, field = h.PALLAS, h.FP
    sf = fields.CURVE_FIELDS[curve][1]
    n = 1 << 12
    sc, bs = co.random_field(sf, 31, n), co.generate_bases(curve, 32, n)
    a = co.random_field(field, 33, n)
    omega = mont(field, o.omega_for(fields.MODULUS[field], 12))
    before = (affine_of(curve, h.best_multiexp(sc, bs, curve)), h.best_fft(a.copy(), omega, 12, field))
    free0 = torch.cuda.mem_get_info()[0]
    assert h.lib().h2_trim() == 0
    assert torch.cuda.mem_get_info()[0] >= free0
    after = (affine_of(curve, h.best_multiexp(sc, bs, curve)), h.best_fft(a.copy(), omega, 12, field))
    assert before[0] == after[0] and np.array_equal(before[1], after[1])


def test_sizes_beyond_the_bench_config_2_21():
    """Maximum-size edge of the parity suite: a 2^21-point registered commit (two-pass sort geometry changes: 27-bit table
    index, 5 low bucket bits, 1024 bins) and the generic multiexp (2^22 split columns), against the C restatement."""
    import ctypes as C
    from halo2_amd.arithmetic import _p
    curve, n = h.VESTA, 1 << 21
    sf = fields.CURVE_FIELDS[curve][1]
    g, col = co.generate_bases(curve, 977, n), co.random_field(sf, 978, n)
    want = co.jac_to_affine_ints(curve, co.best_multiexp(curve, col, g))
    lib = h.lib()
    hd = C.c_uint64(0)
    assert lib.h2_bases_register(curve, _p(g), n, h.FORM_MONTGOMERY, C.byref(hd)) == 0
    out = np.zeros(12, dtype=np.uint64)
    assert lib.h2_commit(hd, _p(col), n, None, None, h.FORM_MONTGOMERY, 0, _p(out)) == 0
    assert affine_of(curve, out) == want
    assert lib.h2_bases_free(hd) == 0
    assert affine_of(curve, h.best_multiexp(col, g, curve)) == want


@pytest.mark.parametrize("curve", [h.PALLAS, h.VESTA])
def test_generic_multiexp_endomorphism_edge_scalars(curve):
    """Scalars that stress the device-side endomorphism split of the generic path (glv.cuh): 0, +-1, lambda and its
    neighbours (k2 = +-1, k1 = 0), powers of two around the 128-bit half length, the largest scalars, and every digit
    position set to +-2^(c-1) (the extreme signed digits)."""
    sm = o.CURVES[curve][1]
    sf = fields.CURVE_FIELDS[curve][1]
    lam = {0: 0x6819a58283e528e511db4d81cf70f5a0fed467d47c033af2aa9d2e050aa0e4f,
           1: 0x2d33357cb532458ed3552a23a8554e5005270d29d19fc7d27b7fd22f0201b547}[curve]
    assert (lam * lam + lam + 1) % sm == 0
    vals = [0, 1, 2, sm - 1, sm - 2, lam, lam + 1, lam - 1, sm - lam, (sm - lam) - 1, lam * lam % sm, 1 << 127, 1 << 128,
            (1 << 128) - 1, (1 << 128) + 1, 1 << 129, 1 << 254, (sm - 1) // 2, (sm + 1) // 2, 0x8000, 0x8001, 0x7FFF,
            sum(0x8000 << (16 * i) for i in range(15)), sum(0x8000 << (13 * i) for i in range(19)) % sm,
            sum(0x200 << (10 * i) for i in range(25)) % sm, (lam << 1) % sm, (lam * 0x8000) % sm]
    vals = [v % sm for v in vals]
    for n in (len(vals), 5000):                       # c = 10 and c = 13 shapes
        reps = -(-n // len(vals))
        sc = fields.to_limbs((vals * reps)[:n], sf, True)
        bs = co.generate_bases(curve, 990 + n, n)
        assert affine_of(curve, h.best_multiexp(sc, bs, curve)) == co.jac_to_affine_ints(curve, co.best_multiexp(curve, sc, bs))
        # one scalar at a time: isolates a wrong split from cancellation between terms
        if n == len(vals):
            for i in range(n):
                got = affine_of(curve, h.best_multiexp(sc[i:i + 1], bs[i:i + 1], curve))
                assert got == co.jac_to_affine_ints(curve, co.best_multiexp(curve, sc[i:i + 1], bs[i:i + 1])), hex(vals[i])


@pytest.mark.parametrize("curve", [h.PALLAS, h.VESTA])
def test_generic_multiexp_top_window_boundaries_at_size(curve):
    """The large generic multiexp (16-bit windows, from 2^18 + 1 points: nine window slices over the two 128-bit halves of every
    scalar, the ninth holding nothing but the carry out of the eighth).  2^19 scalars k = +-k1 +- lambda k2 whose halves carry every
    boundary value of the top window -- 0, 1, 2^15 - 1, 2^15, 2^15 + 1 (the first digit that recodes negative and carries), 2^16 - 1
    with and without a carry from below -- among random ones, against the C oracle; then the same boundary scalars alone (everything
    else zero), so that no cancellation between terms can hide a wrong digit.  (Real halves rarely get there: the top window of a
    split Pallas scalar exceeds 2^15 for 5 % of the halves, 23 % on Vesta -- which is why cutting that window unsigned to save the
    carry slice, tried in round 5, bought nothing: DESIGN.md section 4.4.)"""
    sm = o.CURVES[curve][1]
    sf = fields.CURVE_FIELDS[curve][1]
    lam = {0: 0x6819a58283e528e511db4d81cf70f5a0fed467d47c033af2aa9d2e050aa0e4f,
           1: 0x2d33357cb532458ed3552a23a8554e5005270d29d19fc7d27b7fd22f0201b547}[curve]
    n = 1 << 19
    import random
    rng = random.Random(1900 + curve)
    tops = [0, 1, 0x7FFF, 0x8000, 0x8001, 0xFFFE, 0xFFFF]
    below = [0, 0x7FFF, 0x8000, 0x8001, 0xFFFF]
    halves = [(t << 112) | (b << 96) | rng.getrandbits(96) for t in tops for b in below] + [(1 << 128) - 1, 1 << 127, (1 << 127) - 1]
    crafted = []
    for k1 in halves:
        for k2 in (0, halves[rng.randrange(len(halves))]):
            for s1, s2 in ((1, 1), (-1, 1), (1, -1)):
                crafted.append((s1 * k1 + s2 * k2 * lam) % sm)
    bases = co.generate_bases(curve, 1919, n)
    sc = co.random_field(sf, 1920 + curve, n)
    idx = rng.sample(range(n), len(crafted))
    sc[idx] = fields.to_limbs(crafted, sf, True)
    assert affine_of(curve, h.best_multiexp(sc, bases, curve)) == co.jac_to_affine_ints(curve, co.best_multiexp(curve, sc, bases))
    alone = np.zeros_like(sc)
    alone[idx] = sc[idx]
    assert affine_of(curve, h.best_multiexp(alone, bases, curve)) == co.jac_to_affine_ints(curve, co.best_multiexp(curve, alone, bases))


def test_generator_collapse_edge_challenges():
    """Challenges that stress the host-side endomorphism split of h2_generator_collapse: 0, 1, -1, lambda, -lambda,
    lambda +- 1, 2^127, 2^128 +- 1, the largest scalar; narrow (2^16 points) and quad-wide (64 points) kernels."""
    curve = h.VESTA
    sm = o.CURVES[curve][1]
    sf = fields.CURVE_FIELDS[curve][1]
    lam = 0x2d33357cb532458ed3552a23a8554e5005270d29d19fc7d27b7fd22f0201b547
    vals = [0, 1, sm - 1, lam, sm - lam, lam + 1, lam - 1, 1 << 127, (1 << 128) - 1, (1 << 128) + 1, sm - 2, (sm - 1) // 2]
    g_small = co.generate_bases(curve, 61, 128)
    g_small[7] = 0                                                   # identity in the low half
    g_small[64 + 9] = 0                                              # identity in the high half
    for v in vals:
        u = fields.scalar_limbs(v, sf, True)
        assert np.array_equal(h.parallel_generator_collapse(g_small, u, curve), co.generator_collapse(curve, g_small, u)), hex(v)
    g_big = co.generate_bases(curve, 62, 1 << 17)
    for v in (lam, sm - 1, (1 << 128) + 1):
        u = fields.scalar_limbs(v, sf, True)
        assert np.array_equal(h.parallel_generator_collapse(g_big, u, curve), co.generator_collapse(curve, g_big, u)), hex(v)


# ------------------------------------------------------------------ h2_msm from host slices: the range pipeline (round 5)
@pytest.mark.parametrize("curve,canonical,affine", [(h.PALLAS, False, False), (h.VESTA, True, True)])
def test_best_multiexp_host_slices_range_pipeline(curve, canonical, affine):
    """From 2^19 points on h2_msm cuts the multiexp into point ranges that start behind their bases' upload (csrc/msm.hip,
    msm_host_chunked): ragged ranges (n = 2^19 + 4097 over three), both data forms, both output kinds -- and FOUR calls with fresh
    scalars each, so that the first plain-launch call, the call that captures the launch sequences as hipGraphs and the calls that
    replay them (same staging buffers, new bytes) are all compared with the oracle."""
    n = (1 << 19) + 4097
    sf, bf = co.field_of_curve(curve, "scalar"), co.field_of_curve(curve, "base")
    bases = co.generate_bases(curve, 4242, n)
    bases_in = np.ascontiguousarray(co.from_mont(bf, bases.reshape(2 * n, 4)).reshape(n, 8)) if canonical else bases
    form = h.FORM_CANONICAL if canonical else h.FORM_MONTGOMERY
    for rep in range(4):
        scal = co.random_field(sf, 8800 + rep, n)
        if rep == 2:
            scal[: n // 2] = 0                     # half the scalars zero: empty digits everywhere in the first ranges
        scal_in = co.from_mont(sf, scal) if canonical else scal
        got = np.ascontiguousarray(h.best_multiexp(scal_in, bases_in, curve, form=form, affine=affine), dtype=np.uint64)
        if canonical:                              # canonical coordinates out: back to Montgomery limbs for the oracle's reader
            got = co.to_mont(bf, got.reshape(-1, 4)).reshape(-1)
        assert affine_of(curve, got) == affine_of(curve, co.best_multiexp(curve, scal, bases)), rep


@pytest.mark.parametrize("env,args", [
    ({"H2_NTT_MAXR": "11"}, ["ntt", "5,11,13,21,22", "0", "1"]),          # 11-stage passes: 2^21 = 11 + 10, 2^22 = 11 + 11 (odd stage counts open with a radix-2 round)
    ({"H2_NTT_MAXR": "12"}, ["ntt", "12,23,24", "1", "1"]),               # 12-stage passes on Fq: 2^24 = 12 + 12, one 4096-row column per tile
    ({"H2_GENERIC_SPLIT": "0", "H2_MSM_HOST_CHUNKS": "1"}, ["msm", "19", "0"]),      # round 4's forms: one accumulate, one piece from host slices
    ({"H2_GENERIC_SPLIT": "5"}, ["msm", "19", "1"]),                      # another cut of the window slices
    ({"H2_MSM_HOST_CHUNKS": "5", "H2_MSM_HOST_GRAPHS": "0"}, ["msm", "19", "0"]),    # five ragged ranges, plain launches
    ({"H2_MSM_HOST_CHUNKS": "2", "H2_MSM_HOST_THREAD": "1"}, ["msm", "20", "1"]),    # two ranges, captured sequences replayed from a helper thread
])
def test_switched_forms_keep_parity(env, args):
    """The A/B arms left behind switches (live only in the laboratory build, build/ab/libhalo2_mi355x_ab.so, which the native driver loads here;
    read once per process): every one of them is the same
    mathematics and must stay bit-exact against the C oracle -- the 11- and 12-stage NTT passes that the default plan does not use, the
    generic multiexp without its slice split or with another cut, the host-slice multiexp in one piece, in ragged ranges with plain
    launches, and with the runtime calls made by a helper thread."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build", "h2bench")
    if not os.path.exists(exe):
        pytest.skip(f"{exe} not built (run __graft_entry__.build())")
    from conftest import ab_env
    out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=600, env=ab_env(**env))
    assert out.returncode == 0 and "H2BENCH OK" in out.stdout and "FAIL" not in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
