"""The device-resident multi-point opening argument (halo2_amd/multiopen.py) against the oracle's sequential restatement of
`poly::multiopen::create_proof` (oracle/multiopen.py): identical proof BYTES for the same randomness, and the oracle's
restatement of the reference verifier accepts them.  Query shapes: the reference's `test_roundtrip`
(halo2_proofs/src/poly/multiopen.rs:278-393) and the shape plonk::create_proof produces (columns at x, some also at
omega x and omega^-1 x; plonk/prover.rs:664-722).  Runs only on a real MI355X (`-m gpu`)."""
import numpy as np
import pytest

import halo2_amd as h
from halo2_amd import fields
from halo2_amd.multiopen import ProverQuery, create_proof
from halo2_amd.transcript import Blake2bWrite
from halo2_amd import verifier as hv
from oracle import c_oracle as co
from oracle import ipa, multiopen as om

pytestmark = pytest.mark.gpu


def _rng(sf, seed):
    ctr = [seed]

    def rng(count):
        ctr[0] += 1
        return co.random_field(sf, ctr[0], count)
    return rng


def _run(curve, k, polys, blinds, pattern, points, schedule=None):
    """pattern: (poly index, point index) pairs in query order."""
    import torch
    n = 1 << k
    sf = fields.CURVE_FIELDS[curve][1]
    I = lambda limbs: fields.from_limbs(np.ascontiguousarray(limbs).reshape(1, 4), sf, True)[0]
    g = co.generate_bases(curve, 50 + k, n)
    w, u = co.generate_bases(curve, 60, 1)[0], co.generate_bases(curve, 61, 1)[0]
    params = h.Params(curve, k, g, g, w, u)
    dev = torch.device("cuda:0")
    d_polys = [torch.from_numpy(p.view(np.int64)).to(dev) for p in polys]
    tr = Blake2bWrite(curve)
    create_proof(params, _rng(sf, 3000), tr, [ProverQuery(points[j], d_polys[i], h.Blind(blinds[i])) for i, j in pattern],
                 schedule=schedule)
    proof = tr.finalize()
    for d_p, p in zip(d_polys, polys):                              # the prover leaves its inputs untouched
        assert np.array_equal(d_p.cpu().numpy().view(np.uint64), p)
    ot = ipa.Transcript(curve)
    om.create_proof(curve, k, g, w, u, _rng(sf, 3000), ot, [(I(points[j]), polys[i], blinds[i]) for i, j in pattern])
    assert bytes(ot.out) == proof
    comms = [co.jac_to_affine_ints(curve, co.commit(curve, g, w, p, b)) for p, b in zip(polys, blinds)]
    evals = {(i, j): I(co.eval_polynomial(sf, polys[i], points[j])) for i, j in pattern}
    vq = [(I(points[j]), comms[i], evals[(i, j)]) for i, j in pattern]
    assert om.verify_proof(curve, k, g, w, u, ipa.Transcript(curve, proof), vq)
    i0, j0 = pattern[0]
    vq_bad = [(I(points[j0]), comms[i0], (evals[(i0, j0)] + 1) % fields.MODULUS[sf])] + vq[1:]
    assert not om.verify_proof(curve, k, g, w, u, ipa.Transcript(curve, proof), vq_bad)
    # the product's verifier (multiopen::verify_proof -> commitment::verify_proof -> Guard::use_challenges -> MSM::eval) agrees
    for queries, want in ((vq, True), (vq_bad, False)):
        guard = hv.multiopen_verify_proof(params, hv.Blake2bRead(curve, proof), [hv.VerifierQuery(*q) for q in queries], hv.MSM(params))
        assert guard.use_challenges().eval() is want
    params.close()
    return proof


@pytest.mark.parametrize("curve", [h.VESTA, h.PALLAS])
def test_multiopen_reference_roundtrip_shape(curve):
    k = 4
    n = 1 << k
    sf = fields.CURVE_FIELDS[curve][1]
    ax = fields.to_limbs([10 + i for i in range(n)], sf, True)            # multiopen.rs:293-306
    bx = fields.to_limbs([100 + i for i in range(n)], sf, True)
    cx = fields.to_limbs([100 + i for i in range(n)], sf, True)
    blind = co.random_field(sf, 93, 1)[0]
    pts = co.random_field(sf, 94, 2)
    proof = _run(curve, k, [ax, bx, cx], [blind] * 3, [(0, 0), (1, 0), (2, 1)], pts)
    assert len(proof) == 32 + 2 * 32 + 32 + 64 * k + 64


@pytest.mark.parametrize("k,schedule", [(6, None), (6, "collapse"), (10, None)])
def test_multiopen_plonk_shaped_queries(k, schedule):
    """8 columns at x; two of them also at omega x, one of those at omega^-1 x as well, one column only at omega x."""
    curve = h.VESTA
    n = 1 << k
    sf = fields.CURVE_FIELDS[curve][1]
    polys = [co.random_field(sf, 200 + i, n) for i in range(8)]
    blinds = list(co.random_field(sf, 220, 8))
    pts = co.random_field(sf, 221, 3)                                 # stand-ins for x, omega x, omega^-1 x
    pattern = [(i, 0) for i in range(7)] + [(1, 1), (2, 1), (2, 2), (7, 1)]
    _run(curve, k, polys, blinds, pattern, pts, schedule)


def test_multiopen_repeated_query_is_refused():
    import torch
    curve, k = h.VESTA, 4
    n = 1 << k
    sf = fields.CURVE_FIELDS[curve][1]
    g = co.generate_bases(curve, 54, n)
    params = h.Params(curve, k, g, g, g[1], g[2])
    d_p = torch.from_numpy(co.random_field(sf, 5, n).view(np.int64)).cuda()
    pt = co.random_field(sf, 6, 1)[0]
    q = ProverQuery(pt, d_p, h.Blind(field=sf))
    with pytest.raises(ValueError):
        create_proof(params, _rng(sf, 1), Blake2bWrite(curve), [q, q])            # prover.rs:41-46
    with pytest.raises(ValueError):
        create_proof(params, _rng(sf, 1), Blake2bWrite(curve), [ProverQuery(pt, d_p[:n // 2], h.Blind(field=sf))])
    params.close()
