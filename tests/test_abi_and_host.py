"""CPU-only checks of the drop-in boundary and the host logic: the C-ABI library loads, exports every
symbol include/halo2_mi355x.h declares, validates arguments without a GPU, fails loudly (no CPU fallback)
when no device is present; EvaluationDomain constants match the reference's pinned values."""
import ctypes as C
import json
import os
import re
import subprocess

import numpy as np
import pytest

import halo2_amd as h
from halo2_amd import fields
from halo2_amd import _lib
from oracle import pasta as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "halo2_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(h2_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    syms = declared_symbols()
    assert len(syms) >= 20
    lib = h.lib()
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in the header but not exported"
        assert s in _lib.SIGNATURES, f"{s} has no ctypes signature"
    # and nothing bound that the header does not declare
    assert sorted(_lib.SIGNATURES) == syms
    out = subprocess.check_output(["nm", "-D", "--defined-only", h.LIB_PATH], text=True)
    exported = set(re.findall(r"\bT (h2_[a-z0-9_]+)", out))
    assert exported == set(syms)


def test_no_torch_types_in_abi():
    text = open(os.path.join(ROOT, "include", "halo2_mi355x.h")).read()
    assert "torch" not in text.lower() and "at::" not in text


def test_argument_validation_without_gpu():
    lib = h.lib()
    z4 = np.zeros((4, 4), np.uint64)
    z8 = np.zeros((4, 8), np.uint64)
    out = np.zeros(12, np.uint64)
    p = lambda a: a.ctypes.data_as(_lib.u64p)
    assert lib.h2_msm(7, p(z4), p(z8), 4, 1, 0, p(out)) == _lib.H2_ERR_ARGS          # bad curve
    assert lib.h2_msm(0, p(z4), p(z8), 4, 9, 0, p(out)) == _lib.H2_ERR_ARGS          # bad form
    assert lib.h2_msm(0, None, p(z8), 4, 1, 0, p(out)) == _lib.H2_ERR_ARGS           # null scalars
    assert lib.h2_ntt(0, p(z4), 33, p(z4[0]), 1) == _lib.H2_ERR_ARGS                 # log_n > 32
    assert lib.h2_ntt(3, p(z4), 2, p(z4[0]), 1) == _lib.H2_ERR_ARGS                  # bad field
    assert lib.h2_bases_free(123456) == _lib.H2_ERR_HANDLE
    # round-2 entries: the opening argument's policies are host logic; bad handles / curves are refused before any device work
    assert lib.h2_ipa_default_switch_rounds(20, 1) == 6 and lib.h2_ipa_default_switch_rounds(16, 1) == 2 and lib.h2_ipa_default_switch_rounds(24, 1) == 5
    assert lib.h2_ipa_default_switch_rounds(19, 1) == 5 and lib.h2_ipa_default_switch_rounds(21, 1) == 6 and lib.h2_ipa_default_switch_rounds(22, 1) == 5
    assert lib.h2_ipa_default_switch_rounds(15, 1) == 0 and lib.h2_ipa_default_switch_rounds(20, 0) == 0
    assert lib.h2_commit_pair_supported(8192 + 4) == 1 and lib.h2_commit_pair_supported(4096 + 4) == 0
    assert [lib.h2_commit_window_bits(1 << k) for k in (4, 10, 12, 13, 14, 20)] == [8, 13, 13, 16, 16, 16]
    hh = C.c_uint64(0)
    assert lib.h2_bases_register_device(9, None, 4, 1, C.byref(hh)) == _lib.H2_ERR_ARGS
    assert lib.h2_ipa_collapsed_generators_device(424242, 16, 2, p(z4), 1, None, None) == _lib.H2_ERR_HANDLE
    assert lib.h2_transcript_free(987654) == _lib.H2_ERR_HANDLE
    with pytest.raises(ValueError):                                                   # arithmetic.rs:144
        h.best_multiexp(z4, np.zeros((5, 8), np.uint64), h.PALLAS)
    with pytest.raises(ValueError):                                                   # arithmetic.rs:205
        h.best_fft(z4, z4[0], 3, h.FP)


def test_fails_loudly_without_device():
    """The product path has no CPU fallback: on a box without a GPU every compute call errors out."""
    if h.lib().h2_device_count() > 0:
        pytest.skip("a GPU is present")
    z4 = np.zeros((4, 4), np.uint64)
    with pytest.raises(h.H2Error, match="no MI355X device|no HIP device"):
        h.best_multiexp(z4, np.zeros((4, 8), np.uint64), h.PALLAS)
    with pytest.raises(h.H2Error):
        h.best_fft(z4, z4[0], 2, h.FP)


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "halo2_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".inc")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("the oracle", "").replace("Python oracle", "").replace("inject the oracle", ""), f


def test_shipped_library_reads_no_algorithm_switch():
    """The laboratory is not in the product (round-5 review, item 4): every A/B arm and sweep knob is read through ab_env()
    (csrc/common.h), a constant null in the shipped build -- so the sources call getenv("H2_...") for the diagnostic H2_TIMELINE only, no
    switch name survives as a string in halo2_amd/libhalo2_mi355x.so, every one of them is live in the laboratory build
    (build/ab/libhalo2_mi355x_ab.so, which only tests and bench/tools load), and no source file is a catch-all again (<= 1500 lines)."""
    import glob
    import re
    csrc = os.path.join(ROOT, "halo2_amd", "csrc")
    names, direct = set(), []
    for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.cuh")) + glob.glob(os.path.join(csrc, "*.h")):
        src = open(f).read()
        assert src.count("\n") <= 1500 or not f.endswith(".hip"), (f, src.count("\n"))
        names |= set(re.findall(r'ab_env\("(H2_\w+)"\)', src))
        direct += re.findall(r'[^_]getenv\("(H2_\w+)"\)', src)
    assert direct == ["H2_TIMELINE"], direct
    assert len(names) >= 30 and "H2_TIMELINE" not in names, sorted(names)
    blob = open(os.path.join(ROOT, "halo2_amd", "libhalo2_mi355x.so"), "rb").read()
    leaked = [n for n in sorted(names) if (n + "\0").encode() in blob]
    assert not leaked, leaked
    ab = os.path.join(ROOT, "build", "ab", "libhalo2_mi355x_ab.so")
    if os.path.exists(ab):
        ab_blob = open(ab, "rb").read()
        missing = [n for n in sorted(names) if (n + "\0").encode() not in ab_blob]
        assert not missing, missing


def test_domain_constants_match_reference_pins(golden_dir):
    vks = json.load(open(os.path.join(golden_dir, "pinned_vk.json")))
    for v in vks:
        # tests/plonk_api.rs:593-597 and circuit_data/vk_*.rdata:4-8: k, extended_k, omega
        j = 5 if v["k"] == 5 else None
        for cs_degree in range(2, 12):
            d = h.EvaluationDomain(cs_degree, v["k"], h.FP)
            if d.extended_k == v["extended_k"]:
                assert d.omega == int(v["omega"], 16)
                break
        else:
            raise AssertionError("no degree reproduces extended_k")
    d = h.EvaluationDomain(3, 20, h.FP)        # simple-example at k = 20
    assert d.extended_k == 21 and len(d.t_evaluations) == 2
    d5 = h.EvaluationDomain(5, 8, h.FP)        # benches/plonk.rs at k = 8
    assert d5.extended_k == 10
    ref = o.EvaluationDomain(5, 8, o.P)
    assert (d5.omega, d5.extended_omega, d5.g_coset, d5.g_coset_inv, d5.ifft_divisor, d5.extended_ifft_divisor,
            d5.t_evaluations) == (ref.omega, ref.extended_omega, ref.g_coset, ref.g_coset_inv, ref.ifft_divisor,
                                  ref.extended_ifft_divisor, ref.t_evaluations)
    assert pow(d5.g_coset, 3, o.P) == 1 and d5.g_coset != 1


def test_limb_encoding_roundtrip():
    vals = [0, 1, o.P - 1, 0x123456789ABCDEF0123456789ABCDEF]
    a = fields.to_limbs(vals, h.FP)
    assert fields.from_limbs(a, h.FP) == vals
    assert fields.from_limbs(fields.to_limbs(vals, None, montgomery=False), None, montgomery=False) == vals


def test_cpp_host_mirror_header_compiles():
    """The C++ mirror of the reference interface is self-contained: it compiles against the C header alone."""
    src = '#include "halo2_amd/host/halo2_host.hpp"\nint main() { halo2::EvaluationDomain<H2_FP> d(3, 4); return d.extended_k == 5 ? 0 : 1; }\n'
    out = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", ROOT, "-x", "c++", "-"], input=src, text=True,
                         capture_output=True, cwd=ROOT)
    assert out.returncode == 0, out.stderr


def test_native_bench_driver_binds_the_abi_and_refuses_without_device():
    """bench/native/h2bench.cpp (built by __graft_entry__.build()) binds the entry points it drives from the in-tree library by name and,
    like the library itself, has no CPU path: without a GPU it says so and leaves with status 2."""
    exe = os.path.join(ROOT, "build", "h2bench")
    if not os.path.exists(exe):
        pytest.skip("build/h2bench not built (run __graft_entry__.build())")
    if h.lib().h2_device_count() > 0:
        pytest.skip("a GPU is present")
    out = subprocess.run([exe, "commit", "10"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 2 and "library:" in out.stdout and "no GPU" in out.stdout, out.stdout + out.stderr


def test_transcript_mirror_matches_oracle_restatement():
    """halo2_amd.transcript (host logic) against the oracle's independent copy of transcript.rs:150-300, and the oracle's
    opening-argument prover against its verifier (CPU only, k = 3)."""
    import numpy as np
    from halo2_amd import fields
    from halo2_amd.transcript import Blake2bWrite
    from oracle import c_oracle as co
    from oracle import ipa
    curve, k = 1, 3
    n = 1 << k
    sf = fields.CURVE_FIELDS[curve][1]
    g = co.generate_bases(curve, 5, n)
    w, u = co.generate_bases(curve, 6, 1)[0], co.generate_bases(curve, 7, 1)[0]
    pt = co.affine_to_ints(curve, g[3])
    sc = co.random_field(sf, 8, 1)[0]
    a, b = Blake2bWrite(curve), ipa.Transcript(curve)
    a.write_point(g[3]); b.write_point(pt)
    assert a.squeeze_challenge() == b.squeeze_challenge()
    a.write_scalar(sc); b.write_scalar(fields.from_limbs(sc, sf, True)[0])
    a.common_point(g[4]); b.common_point(co.affine_to_ints(curve, g[4]))
    assert a.squeeze_challenge() == b.squeeze_challenge()
    assert a.finalize() == bytes(b.out) and len(a.finalize()) == 64
    with pytest.raises(ValueError):
        a.write_point(np.zeros(8, dtype=np.uint64))

    ctr = [40]

    def rng(count):
        ctr[0] += 1
        return co.random_field(sf, ctr[0], count)
    px = fields.to_limbs(range(n), sf, True)
    blind = co.random_field(sf, 9, 1)[0]
    p = co.jac_to_affine_ints(curve, co.commit(curve, g, w, px, blind))
    t = ipa.Transcript(curve)
    t.write_point(p)
    x = t.squeeze_challenge()
    xl = fields.scalar_limbs(x, sf, True)
    v = fields.from_limbs(co.eval_polynomial(sf, px, xl), sf, True)[0]
    t.write_scalar(v)
    ipa.create_proof(curve, k, g, w, u, rng, t, px, blind, xl)
    vt = ipa.Transcript(curve, bytes(t.out))
    assert vt.read_point() == p and vt.squeeze_challenge() == x and vt.read_scalar() == v
    assert ipa.verify_proof(curve, k, g, w, u, vt, p, x, v)
    vt = ipa.Transcript(curve, bytes(t.out))
    vt.read_point(), vt.squeeze_challenge(), vt.read_scalar()
    assert not ipa.verify_proof(curve, k, g, w, u, vt, p, x, (v + 1))


def test_domain_rotations_and_lagrange_basis_evaluations():
    """EvaluationDomain host logic (domain.rs:258-274, 408-472), shaped after the reference's test_rotate and test_l_i
    (domain.rs:500-569): rotate_extended is a row rotation; l_i_range matches the direct product formula."""
    import numpy as np
    k = 4
    d = h.EvaluationDomain(3, k, h.FP)
    m, n = d.m, d.n
    assert d.get_omega() == d.omega and pow(d.omega, n, m) == 1 and d.get_quotient_poly_degree() == 2
    assert d.rotate_omega(7, 3) == 7 * pow(d.omega, 3, m) % m and d.rotate_omega(7, -2) == 7 * pow(d.omega_inv, 2, m) % m
    ext = np.arange(d.extended_len() * 4, dtype=np.uint64).reshape(-1, 4)
    step = 1 << (d.extended_k - d.k)
    assert np.array_equal(d.rotate_extended(ext, 1), np.concatenate([ext[step:], ext[:step]]))         # rotate_left
    assert np.array_equal(d.rotate_extended(ext, -2), np.concatenate([ext[-2 * step:], ext[:-2 * step]]))
    assert d.constant_lagrange(5).shape == (n, 4) and d.empty_extended().shape == (d.extended_len(), 4)
    with pytest.raises(ValueError):
        d.rotate_extended(ext[:-1], 1)
    # l_i(x) = prod_{j != i} (x - w^j) / (w^i - w^j)   (domain.rs:541-569)
    x = 0x1234567890ABCDEF % m
    xn = pow(x, n, m)
    pts = [pow(d.omega, i, m) for i in range(n)]

    def l_direct(i):
        num = den = 1
        for j in range(n):
            if j != i:
                num = num * (x - pts[j]) % m
                den = den * (pts[i] - pts[j]) % m
        return num * pow(den, -1, m) % m
    rots = [0, 1, 5, -1, -3]
    assert d.l_i_range(x, xn, rots) == [l_direct(r % n) for r in rots]


def test_polynomial_basis_tags():
    """`Polynomial<F, B>` (poly.rs:30-57): the basis marker travels with the values; the domain's constructors hand out tagged
    polynomials and a transform applied to the wrong basis is refused before anything reaches the device."""
    import halo2_amd as h
    d = h.EvaluationDomain(3, 4, h.FP)
    lag = d.empty_lagrange()
    assert isinstance(lag, h.Polynomial) and lag.basis is h.LagrangeCoeff and len(lag) == 16 and lag.shape == (16, 4)
    assert d.empty_coeff().basis is h.Coeff and d.empty_extended().basis is h.ExtendedLagrangeCoeff
    assert d.constant_extended(7).basis is h.ExtendedLagrangeCoeff and len(d.constant_extended(7)) == d.extended_len()
    import numpy as np
    vals = np.zeros((16, 4), dtype=np.uint64)
    assert d.coeff_from_vec(vals).basis is h.Coeff and d.lagrange_from_vec(vals).basis is h.LagrangeCoeff
    lag[3] = np.array([1, 2, 3, 4], dtype=np.uint64)
    assert list(lag[3]) == [1, 2, 3, 4]
    import pytest
    with pytest.raises(TypeError):
        d.lagrange_to_coeff(d.empty_coeff())                   # already coefficients
    with pytest.raises(TypeError):
        d.coeff_to_extended(lag)
    with pytest.raises(TypeError):
        d.extended_to_coeff(d.empty_coeff())
    with pytest.raises(TypeError):
        d.divide_by_vanishing_poly(lag)
    with pytest.raises(TypeError):
        h.Polynomial(vals, "Coeff")
    with pytest.raises(ValueError):
        h.Polynomial(np.zeros(16, dtype=np.uint64), h.Coeff)
    with pytest.raises(ValueError):
        d.coeff_from_vec(np.zeros((15, 4), dtype=np.uint64))


@pytest.mark.parametrize("curve", [h.PALLAS, h.VESTA])
def test_library_transcript_matches_hashlib_restatement(curve):
    """h2_transcript_* (the library's Blake2bWrite / Challenge255, host arithmetic only) against the oracle's hashlib restatement
    of transcript.rs: a few hundred mixed operations -- affine and Jacobian points, scalars, challenges at block boundaries of the
    128-byte BLAKE2b buffer -- give the same challenges and the same written bytes; the identity is refused."""
    from halo2_amd import fields
    from halo2_amd.transcript import Blake2bWrite
    from oracle import c_oracle as co
    from oracle import ipa
    bf, sf = fields.CURVE_FIELDS[curve]
    bm, sm = fields.MODULUS[bf], fields.MODULUS[sf]
    pts = co.generate_bases(curve, 5, 24)
    ints = [co.affine_to_ints(curve, p) for p in pts]
    rs = np.random.RandomState(11)
    tr, ot = Blake2bWrite(curve), ipa.Transcript(curve)
    for step in range(300):
        op = int(rs.randint(0, 5))
        i = int(rs.randint(0, len(pts)))
        if op == 0:
            tr.write_point(pts[i])
            ot.write_point(ints[i])
        elif op == 1:                                   # Jacobian (x z^2, y z^3, z): the prover's .to_affine() happens inside
            z = int(rs.randint(2, 1 << 30))
            jac = fields.to_limbs([ints[i][0] * z * z % bm, ints[i][1] * z * z * z % bm, z], bf, True).reshape(12)
            tr.write_point(jac)
            ot.write_point(ints[i])
        elif op == 2:
            s = int(rs.randint(0, 1 << 62)) * int(rs.randint(1, 1 << 62)) % sm
            tr.write_scalar(fields.scalar_limbs(s, sf, True))
            ot.write_scalar(s)
        elif op == 3:
            tr.common_point(pts[i])
            ot.common_point(ints[i])
        else:
            assert tr.squeeze_challenge() == ot.squeeze_challenge(), step
    assert tr.squeeze_challenge() == ot.squeeze_challenge()
    assert tr.finalize() == bytes(ot.out) and len(tr.finalize()) > 1000
    with pytest.raises(ValueError):
        tr.write_point(np.zeros(8, dtype=np.uint64))
    with pytest.raises(ValueError):
        tr.write_point(np.zeros(12, dtype=np.uint64))


def test_divstep_inversion_on_the_host():
    """csrc/field_inv.cuh (fe_inv on the device: Bernstein-Yang divsteps on signed 30-bit limbs) is plain __host__ __device__ integer code:
    tests/native/modinv_check.cpp compiles the same functions for the CPU and checks x * modinv30(x) = 1 mod p for 2 x 600 values of both
    Pasta moduli -- 0, 1, 2, p - 1, powers of two and random ones -- with the result in [0, p).  The library's HOST inversion (host_inv of
    csrc/host_field.h: the same divsteps, Montgomery in and out -- two per round of the opening argument) is checked there too, against the
    binary Euclid it replaced and against a * a^-1 = 1, for 2 x 4000 values including raw limbs at or above p."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "build", "modinv_check")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(root, "tests", "native", "modinv_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "1200 cases, 0 failures" in out.stdout and "host_inv: 8000 cases, 0 failures" in out.stdout, out.stdout + out.stderr
