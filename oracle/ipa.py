"""TEST INFRASTRUCTURE ONLY: CPU restatement of the polynomial-commitment opening argument --
`create_proof` (halo2_proofs/src/poly/commitment/prover.rs:26-151) and `verify_proof` + `Guard::use_challenges`
(poly/commitment/verifier.rs:65-171, :35-41) -- with its own copy of the Blake2b transcript
(transcript.rs:150-300).  Sequential, as the reference writes it, on the C oracle's field / curve functions and Python
integers.  Nothing here is imported by the product; tests use it to check the device-resident prover byte for byte and
to verify the proofs it emits."""
from __future__ import annotations

import hashlib

import numpy as np

from . import c_oracle as co
from . import pasta as o


def _le32(v: int) -> bytes:
    return int(v).to_bytes(32, "little")


class Transcript:
    """Blake2bWrite / Blake2bRead with Challenge255 (transcript.rs:68-300)."""

    def __init__(self, curve: int, proof: bytes | None = None):
        self.curve = curve
        self.bm, self.sm = o.CURVES[curve]
        self.state = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
        self.out = bytearray()
        self.inp, self.pos = proof, 0

    def squeeze_challenge(self) -> int:                                # :200-205, :286-296
        self.state.update(b"\x00")
        return int.from_bytes(self.state.copy().digest(), "little") % self.sm

    def common_point(self, pt):                                        # :207-219
        assert pt is not None, "cannot write points at infinity to the transcript"
        self.state.update(b"\x01")
        self.state.update(_le32(pt[0]))
        self.state.update(_le32(pt[1]))

    def common_scalar(self, s: int):                                   # :221-226
        self.state.update(b"\x02")
        self.state.update(_le32(s))

    def write_point(self, pt):                                         # :183-187
        self.common_point(pt)
        b = bytearray(_le32(pt[0]))
        b[31] |= (pt[1] & 1) << 7
        self.out += b

    def write_scalar(self, s: int):                                    # :188-192
        self.common_scalar(s)
        self.out += _le32(s)

    def read_point(self):                                              # :89-101 (Blake2bRead)
        raw = bytearray(self.inp[self.pos:self.pos + 32])
        self.pos += 32
        sign = raw[31] >> 7
        raw[31] &= 0x7F
        x = int.from_bytes(raw, "little")
        assert x < self.bm
        y = o.sqrt_mod((x * x * x + o.CURVE_B) % self.bm, self.bm)
        assert y is not None, "not a curve point"
        if y & 1 != sign:
            y = self.bm - y
        self.common_point((x, y))
        return (x, y)

    def read_scalar(self) -> int:                                      # :103-116
        s = int.from_bytes(self.inp[self.pos:self.pos + 32], "little")
        self.pos += 32
        assert s < self.sm
        self.common_scalar(s)
        return s


def _aff(curve, jac):
    return co.jac_to_affine_ints(curve, jac)


def _pt_limbs(curve, pt) -> np.ndarray:
    bf = co.field_of_curve(curve, "base")
    return co.to_mont(bf, co.ints_to_limbs([pt[0], pt[1]])).reshape(8)


def create_proof(curve, k, g, w, u, rng, transcript: Transcript, p_poly, p_blind, x_3):
    """prover.rs:26-151.  g (n, 8), w, u (8,), p_poly (n, 4), p_blind, x_3 (4,): Montgomery limbs; rng(count) -> (count, 4)."""
    sf = co.field_of_curve(curve, "scalar")
    m = o.CURVES[curve][1]
    n = 1 << k
    assert p_poly.shape[0] == n
    I = lambda limbs: co.limbs_to_ints(co.from_mont(sf, np.ascontiguousarray(limbs).reshape(-1, 4)))[0]
    L = lambda v: co.to_mont(sf, co.ints_to_limbs([v % m]))[0]
    s_poly = rng(n).copy()
    s_at_x3 = I(co.eval_polynomial(sf, s_poly, x_3))
    s_poly[0] = L(I(s_poly[0]) - s_at_x3)
    s_blind = rng(1)[0].copy()
    transcript.write_point(_aff(curve, co.commit(curve, g, w, s_poly, s_blind)))
    xi = transcript.squeeze_challenge()
    z = transcript.squeeze_challenge()
    p_prime = co.scale_add(sf, s_poly, L(xi), p_poly)
    v = I(co.eval_polynomial(sf, p_prime, x_3))
    p_prime[0] = L(I(p_prime[0]) - v)
    f = (I(s_blind) * xi + I(p_blind)) % m
    b = co.powers(sf, x_3, n)
    g_prime = np.ascontiguousarray(g, dtype=np.uint64).copy()
    uw = np.stack([u, w])
    for j in range(k):
        half = 1 << (k - j - 1)
        l_j = co.best_multiexp(curve, p_prime[half:], g_prime[:half])
        r_j = co.best_multiexp(curve, p_prime[:half], g_prime[half:])
        value_l = I(co.inner_product(sf, p_prime[half:], b[:half]))
        value_r = I(co.inner_product(sf, p_prime[:half], b[half:]))
        l_rand, r_rand = rng(2)
        lib = co.lib()
        for acc, val, rnd in ((l_j, value_l, l_rand), (r_j, value_r, r_rand)):
            extra = co.best_multiexp(curve, np.stack([L(val * z), rnd]), uw)
            lib.orc_point_add(curve, co._p(acc), co._p(acc), co._p(extra))
        transcript.write_point(_aff(curve, l_j))
        transcript.write_point(_aff(curve, r_j))
        u_j = transcript.squeeze_challenge()
        u_inv = pow(u_j, -1, m)
        p_prime = co.fold_scalars(sf, p_prime, L(u_inv))
        b = co.fold_scalars(sf, b, L(u_j))
        g_prime = co.generator_collapse(curve, g_prime, L(u_j))
        f = (f + I(l_rand) * u_inv + I(r_rand) * u_j) % m
    transcript.write_scalar(I(p_prime[0]))
    transcript.write_scalar(f)


def verify_proof(curve, k, g, w, u, transcript: Transcript, commitment, x: int, v: int, terms=None) -> bool:
    """verifier.rs:65-141 followed by Guard::use_challenges (:35-41) and MSM::eval: True iff the opening is accepted.
    `commitment`: canonical affine (x, y) of P; x, v canonical integers.  `terms`: the caller's `msm` as a list of
    (scalar, point) pairs when it is more than [1] P (multiopen/verifier.rs:127-139); then `commitment` is ignored."""
    sf = co.field_of_curve(curve, "scalar")
    m = o.CURVES[curve][1]
    n = 1 << k
    L = lambda vals: co.to_mont(sf, co.ints_to_limbs([t % m for t in vals]))
    terms = [(1, commitment)] if terms is None else list(terms)   # the caller's msm: [1] P
    g_scalars = [0] * n
    g_scalars[0] = -v                                           # add_constant_term(-v): [-v] G_0
    s_commitment = transcript.read_point()
    xi = transcript.squeeze_challenge()
    terms.append((xi, s_commitment))
    z = transcript.squeeze_challenge()
    rounds = []
    for _ in range(k):
        l = transcript.read_point()
        r = transcript.read_point()
        rounds.append((l, r, transcript.squeeze_challenge()))
    us = []
    for l, r, u_j in rounds:
        terms.append((pow(u_j, -1, m), l))
        terms.append((u_j, r))
        us.append(u_j)
    c = transcript.read_scalar()
    f = transcript.read_scalar()
    b, cur = 1, x                                               # compute_b (:144-152)
    for u_j in reversed(us):
        b = b * (1 + u_j * cur) % m
        cur = cur * cur % m
    u_scalar = -c * b * z
    w_scalar = -f
    s = [0] * n                                                 # compute_s(u, -c) (:155-171)
    s[0] = -c % m
    for i, u_j in enumerate(reversed(us)):
        ln = 1 << i
        for t in range(ln):
            s[ln + t] = s[t] * u_j % m
    g_scalars = [(a + t) % m for a, t in zip(g_scalars, s)]
    scalars = L([t[0] for t in terms] + g_scalars + [u_scalar, w_scalar])
    bases = np.concatenate([np.stack([_pt_limbs(curve, t[1]) for t in terms]), np.ascontiguousarray(g, dtype=np.uint64),
                            np.stack([u, w])])
    return _aff(curve, co.best_multiexp(curve, scalars, bases)) is None
