"""The HIP path against values the REFERENCE TREE pins (not against the builder's restatement): the verifying key of
halo2_proofs/tests/plonk_api.rs:958-981 recomputed through the C ABI, and group-law anchors on the 288 reference-pinned
Vesta points.  The oracle supplies only inputs here (hash-to-curve generators, the keygen columns); the expected values
are the reference's own numbers from tests/golden/pinned_vk.json."""
import numpy as np
import pytest

import halo2_amd as h
from oracle import c_oracle as co
from oracle import pasta as o
from oracle import plonk_api as pa
from test_reference_goldens import PINNED, PINNED_FIXED, _all_pinned_points

pytestmark = pytest.mark.gpu
VESTA = h.VESTA


def affine_of(curve, out):
    out = np.ascontiguousarray(out, dtype=np.uint64)
    return co.jac_to_affine_ints(curve, out) if out.shape[0] == 12 else co.affine_to_ints(curve, out)


@pytest.fixture(scope="module")
def keygen():
    g, _, w, u = pa.params_new("vesta", 5, with_lagrange=False)
    fixed, mapping = pa.keygen_columns(o.P)
    om = [pow(o.omega_for(o.P, 5), j, o.P) for j in range(32)]
    sigmas = [[pow(pa.DELTA[o.P], mapping[i][j][0], o.P) * om[mapping[i][j][1]] % o.P for j in range(32)] for i in range(12)]
    return g, w, u, fixed + sigmas


def test_pinned_vk_through_registered_commit(keygen):
    """Params (g_lagrange by the device point FFT, h2_lagrange_basis) -> commit_lagrange (h2_commit over the registered
    table, blind = Blind::default()) for the 7 fixed and 12 permutation columns == plonk_api.rs:958-981."""
    g, w, u, columns = keygen
    sf = co.field_of_curve(VESTA, "scalar")
    params = h.Params.from_generators(VESTA, 5, co.points_to_mont(VESTA, g), None, co.points_to_mont(VESTA, [w])[0],
                                      co.points_to_mont(VESTA, [u])[0])
    got = []
    for col in columns:
        poly = co.to_mont(sf, co.ints_to_limbs(col))
        got.append(affine_of(VESTA, params.commit_lagrange(poly, h.Blind(field=sf))))
    assert got == PINNED
    # affine output straight from the device (to_affine on the GPU), and device-resident columns in one batch call
    assert affine_of(VESTA, params.commit_lagrange(co.to_mont(sf, co.ints_to_limbs(columns[6])), h.Blind(field=sf), affine=True)) \
        == PINNED_FIXED[6]
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    polys = [torch.from_numpy(co.to_mont(sf, co.ints_to_limbs(c)).view(np.int64)).to(dev) for c in columns]
    out = params.commit_batch(polys, [h.Blind(field=sf)] * len(polys), lagrange=True, affine=True)
    torch.cuda.synchronize()
    assert [affine_of(VESTA, out[i].cpu().numpy().view(np.uint64)) for i in range(len(polys))] == PINNED
    params.close()


def test_pinned_vk_through_generic_multiexp(keygen):
    """The free function best_multiexp (arithmetic.rs:143) over g_lagrange || w, as commit_lagrange calls it
    (commitment.rs:143-149)."""
    g, w, _, columns = keygen
    sf = co.field_of_curve(VESTA, "scalar")
    gl = h.lagrange_basis(co.points_to_mont(VESTA, g), VESTA, 5)
    bases = np.concatenate([gl, co.points_to_mont(VESTA, [w])])
    for col, want in zip(columns, PINNED):
        sc = co.to_mont(sf, co.ints_to_limbs(col + [1]))
        assert affine_of(VESTA, h.best_multiexp(sc, bases, VESTA)) == want


def test_group_law_on_reference_pinned_points():
    """[q-1]P = -P and [q-1]P + P = O for every reference-pinned Vesta point, and [2^k]P by doubling == best_multiexp,
    through h2_msm and through a registered commit."""
    pts = _all_pinned_points()
    n = len(pts)
    sf = co.field_of_curve(VESTA, "scalar")
    bases = co.points_to_mont(VESTA, pts)
    neg1 = co.to_mont(sf, co.ints_to_limbs([o.P - 1]))
    one = co.to_mont(sf, co.ints_to_limbs([1]))
    for i in range(n):
        assert affine_of(VESTA, h.best_multiexp(neg1, bases[i:i + 1], VESTA)) == o.ec_neg(pts[i], o.Q)
        both = h.best_multiexp(np.concatenate([neg1, one]), np.concatenate([bases[i:i + 1], bases[i:i + 1]]), VESTA)
        assert affine_of(VESTA, both) is None
    for k in (1, 7, 64, 200, 253):
        sc = co.to_mont(sf, co.ints_to_limbs([pow(2, k, o.P)] * n))
        want = None
        for P_ in pts:
            d = P_
            for _ in range(k):
                d = o.ec_add(d, d, o.Q)
            want = o.ec_add(want, d, o.Q)
        assert affine_of(VESTA, h.best_multiexp(sc, bases, VESTA)) == want
    # all of them at once with scalar q - 1: sum of negations, on the registered (table) path; 288 -> pad to 512
    pad = np.tile(bases[0], (512, 1))
    pad[:n] = bases
    sc = np.zeros((512, 4), dtype=np.uint64)
    sc[:n] = neg1
    params = h.Params(VESTA, 9, pad, pad, bases[0], bases[1])
    got = affine_of(VESTA, params.commit(sc, h.Blind(np.zeros(4, dtype=np.uint64))))
    want = None
    for P_ in pts:
        want = o.ec_add(want, o.ec_neg(P_, o.Q), o.Q)
    assert got == want
    params.close()
    G = (o.P - 1, 2)                                                                  # Pallas (-1, 2), msm.rs:181
    s0 = co.field_of_curve(h.PALLAS, "scalar")
    got = h.best_multiexp(co.to_mont(s0, co.ints_to_limbs([o.Q - 1])), co.points_to_mont(h.PALLAS, [G]), h.PALLAS)
    assert affine_of(h.PALLAS, got) == o.ec_neg(G, o.P)


def test_reference_stored_proof_through_the_device_verifier_and_prover():
    """The product's keygen (device commits) reproduces the pinned key; its verifier (MSM::eval on the device) accepts the
    proof the reference stores (tests/plonk_api_proof.bin, two circuit instances) and rejects perturbations; and a fresh proof
    of the same circuit from the device prover is accepted by both verifiers."""
    import os
    from halo2_amd import verifier as hv
    from halo2_amd.plonk import ConstraintSystem, create_proof_many, keygen_pk
    from halo2_amd.transcript import Blake2bWrite
    from oracle import plonk as op
    from test_reference_goldens import GOLDEN
    g, _, w, u = pa.params_new("vesta", 5, with_lagrange=False)
    gm, wm, um = co.points_to_mont(VESTA, g), co.points_to_mont(VESTA, [w])[0], co.points_to_mont(VESTA, [u])[0]
    params = h.Params.from_generators(VESTA, 5, gm, None, wm, um)
    cs = pa.constraint_system(ConstraintSystem)
    fixed, mapping = pa.keygen_columns(o.P)
    vk_repr = pa.transcript_repr(open(os.path.join(GOLDEN, "plonk_api_pinned_vk.txt")).read())
    pk = keygen_pk(params, cs, fixed, mapping, vk_repr)
    dvk = hv.keygen_vk(params, pk)
    assert dvk.fixed_commitments + dvk.permutation_commitments == PINNED
    proof = open(os.path.join(GOLDEN, "plonk_api_proof.bin"), "rb").read()
    instances = [[[2]], [[2]]]
    assert hv.verify_proof_many(params, dvk, instances, proof)
    assert not hv.verify_proof_many(params, dvk, [[[2]], [[3]]], proof)
    for pos in (0, 1000, 2500, len(proof) - 1):
        bad = bytearray(proof)
        bad[pos] ^= 1
        assert not hv.verify_proof_many(params, dvk, instances, bytes(bad))
    # a fresh two-instance proof of the reference's circuit and witness from the device prover
    sf = co.field_of_curve(VESTA, "scalar")
    ctr = [31337]

    def rng(count):
        ctr[0] += 1
        return co.random_field(sf, ctr[0], count)
    adv, inst = pa.witness(o.P)
    tr = Blake2bWrite(VESTA)
    create_proof_many(params, pk, [(adv, inst), ([list(c) for c in adv], inst)], rng, tr)
    fresh = tr.finalize()
    assert len(fresh) == len(proof)                                   # CircuitCost::proof_size(2), plonk_api.rs:492-497
    assert hv.verify_proof_many(params, dvk, instances, fresh)
    ovk = {"cs": cs, "vk_repr": vk_repr, "domain": o.EvaluationDomain(cs.degree, 5, o.P),
           "fixed_commitments": PINNED[:7], "permutation_commitments": PINNED[7:]}
    assert op.verify_proof_many(VESTA, 5, gm, wm, um, ovk, instances, fresh)
    params.close()


def test_small_multiexp_on_the_reference_bench_shape(keygen):
    """benches/arithmetic.rs:15-33 through the product: Params::new(5) generators split in halves, 16 two-point calls of
    `small_multiexp`; each equals the oracle's double-and-add restatement (and the definition)."""
    g = keygen[0]
    sf = co.field_of_curve(VESTA, "scalar")
    rng = o.SplitMix64(0x736D616C6C)
    c = [rng.next() * rng.next() * rng.next() * rng.next() % o.P for _ in range(2)]
    coeffs = co.to_mont(sf, co.ints_to_limbs(c))
    gm = co.points_to_mont(VESTA, g)
    for i in range(16):
        got = affine_of(VESTA, h.small_multiexp(coeffs, np.stack([gm[i], gm[16 + i]]), VESTA))
        assert got == o.small_multiexp(c, [g[i], g[16 + i]], o.Q)
    assert affine_of(VESTA, h.small_multiexp(coeffs[:0], gm[:0], VESTA)) is None


def test_polynomial_tags_through_the_device_transforms(keygen):
    """Tagged polynomials through lagrange_to_coeff / coeff_to_extended / commit* return the right basis and the same numbers
    as raw arrays; commit(iFFT(a)) == commit_lagrange(a) (poly/commitment.rs:258-302) with the typed API."""
    g, w, u, columns = keygen
    sf = co.field_of_curve(VESTA, "scalar")
    dom = h.EvaluationDomain(4, 5, sf)
    params = h.Params.from_generators(VESTA, 5, co.points_to_mont(VESTA, g), None, co.points_to_mont(VESTA, [w])[0],
                                      co.points_to_mont(VESTA, [u])[0])
    lag = dom.lagrange_from_vec(co.to_mont(sf, co.ints_to_limbs(columns[6])))
    blind = h.Blind(field=sf)
    c_lag = affine_of(VESTA, params.commit_lagrange(lag, blind))
    assert c_lag == PINNED_FIXED[6]
    coeff = dom.lagrange_to_coeff(h.Polynomial(lag.values.copy(), h.LagrangeCoeff))
    assert coeff.basis is h.Coeff
    assert affine_of(VESTA, params.commit(coeff, blind)) == c_lag
    ext = dom.coeff_to_extended(coeff)
    assert ext.basis is h.ExtendedLagrangeCoeff and len(ext) == dom.extended_len()
    back = dom.extended_to_coeff(dom.divide_by_vanishing_poly(ext))          # plain vector, as in the reference
    assert back.shape[0] == dom.n * dom.quotient_poly_degree
    with pytest.raises(TypeError):
        params.commit(lag, blind)
    with pytest.raises(TypeError):
        params.commit_lagrange(coeff, blind)
    params.close()


def test_hash_to_curve_on_the_device():
    """h2_hash_to_curve == the restated pasta_curves map, both curves, several prefixes and message lengths (0 .. 64 bytes);
    w of Params::new is the value the reference pins."""
    from oracle import hash_to_curve as oh
    for curve, cid in ((h.VESTA, "vesta"), (h.PALLAS, "pallas")):
        for prefix, msgs in (("Halo2-Parameters", [b"\x00" + i.to_bytes(4, "little") for i in (0, 1, 2, 31, 0xDEADBEEF)]),
                             ("z.cash:test", [b"Trans rights now!"]), ("", [b""]), ("x" * 64, [bytes(range(64))])):
            got = h.hash_to_curve(curve, prefix, msgs)
            want = [oh.hash_to_curve(cid, prefix)(m_) for m_ in msgs]
            assert [co.affine_to_ints(curve, got[i]) for i in range(len(msgs))] == want
    assert co.affine_to_ints(VESTA, h.hash_to_curve(VESTA, "Halo2-Parameters", [b"\x01"])[0]) == PINNED_FIXED[0]
    can = h.hash_to_curve(VESTA, "Halo2-Parameters", [b"\x01"], form=h.FORM_CANONICAL)[0]
    assert (int(can[0]) | int(can[1]) << 64 | int(can[2]) << 128 | int(can[3]) << 192) == PINNED_FIXED[0][0]
    with pytest.raises(ValueError):
        h.hash_to_curve(VESTA, "y" * 65, [b"\x01"])            # prefix too long for the DST buffer


def test_params_new_reproduces_the_pinned_verifying_key():
    """`Params.new(VESTA, 5)` -- generators hashed on the device, Lagrange basis by the device point FFT -- then commit_lagrange of
    the reference circuit's keygen columns: all 19 commitments of tests/plonk_api.rs:958-981, with nothing but the column
    values coming from the test side."""
    params = h.Params.new(VESTA, 5)
    g_ref, _, w_ref, u_ref = pa.params_new("vesta", 5, with_lagrange=False)
    assert [co.affine_to_ints(VESTA, params.g[i]) for i in range(32)] == g_ref
    assert co.affine_to_ints(VESTA, params.w) == w_ref == PINNED_FIXED[0] and co.affine_to_ints(VESTA, params.u) == u_ref
    sf = co.field_of_curve(VESTA, "scalar")
    fixed, mapping = pa.keygen_columns(o.P)
    om = [pow(o.omega_for(o.P, 5), j, o.P) for j in range(32)]
    sigmas = [[pow(pa.DELTA[o.P], mapping[i][j][0], o.P) * om[mapping[i][j][1]] % o.P for j in range(32)] for i in range(12)]
    got = [affine_of(VESTA, params.commit_lagrange(co.to_mont(sf, co.ints_to_limbs(col)), h.Blind(field=sf))) for col in fixed + sigmas]
    assert got == PINNED
    params.close()
    big = h.Params.new(h.PALLAS, 12)                           # a larger one: every generator on the curve and distinct
    pts = [co.affine_to_ints(h.PALLAS, big.g[i]) for i in range(0, 4096, 97)]
    assert all(o.on_curve(p_, o.P) for p_ in pts) and len(set(pts)) == len(pts)
    big.close()
