"""TEST INFRASTRUCTURE ONLY: CPU restatement of the multi-point opening argument -- `construct_intermediate_sets`
(halo2_proofs/src/poly/multiopen.rs:152-276), `create_proof` (poly/multiopen/prover.rs:21-125) and `verify_proof`
(poly/multiopen/verifier.rs:15-141), with `lagrange_interpolate` (arithmetic.rs:379-432) -- sequential, on the C oracle's
field functions and Python integers.  Nothing here is imported by the product; tests use it to check the device-resident
multiopen prover byte for byte and to verify the proofs it emits."""
from __future__ import annotations

import numpy as np

from . import c_oracle as co
from . import ipa
from . import pasta as o


def construct_intermediate_sets(queries):
    """multiopen.rs:152-276.  queries: (point, commitment key, eval) triples; eval None for the prover (its Eval is `()`).
    Returns ([{key, set_index, point_indices, evals}], point_sets) or None for a repeated (commitment, point) pair."""
    commitment_map = {}                       # IndexMap: insertion-ordered
    point_index_map = {}
    for point, key, _ in queries:             # :169-180
        idx = point_index_map.setdefault(point, len(point_index_map))
        commitment_map.setdefault(key, {"key": key, "set_index": 0, "point_indices": [], "evals": []})["point_indices"].append(idx)
    inverse = {i: pt for pt, i in point_index_map.items()}              # :183-186
    point_idx_sets = {}                       # ordered set of point indices -> set index, in order of first appearance
    commitment_set = {}
    for key, data in commitment_map.items():                            # :193-209
        s = tuple(sorted(set(data["point_indices"])))
        commitment_set[key] = s
        point_idx_sets.setdefault(s, len(point_idx_sets))
        data["evals"] = [None] * len(s)
    for point, key, ev in queries:                                      # :212-250
        data = commitment_map[key]
        s = commitment_set[key]
        data["set_index"] = point_idx_sets[s]
        pos = s.index(point_index_map[point])
        if data["evals"][pos] is not None:
            return None
        data["evals"][pos] = () if ev is None else ev
    point_sets = [None] * len(point_idx_sets)                           # :267-273
    for s, set_idx in point_idx_sets.items():
        point_sets[set_idx] = [inverse[i] for i in s]
    return list(commitment_map.values()), point_sets


def lagrange_interpolate(points, evals, m):
    """arithmetic.rs:379-432: coefficients (low degree first) of the polynomial through (points[i], evals[i])."""
    assert len(points) == len(evals)
    if len(points) == 1:
        return [evals[0] % m]
    final = [0] * len(points)
    for j, (x_j, ev) in enumerate(zip(points, evals)):
        tmp = [1]
        for kk, x_k in enumerate(points):
            if kk == j:
                continue
            denom = pow((x_j - x_k) % m, -1, m)
            a = tmp + [0]
            b = [0] + tmp
            tmp = [(ai * (-denom * x_k) + bi * denom) % m for ai, bi in zip(a, b)]
        for i, c in enumerate(tmp):
            final[i] = (final[i] + c * ev) % m
    return final


def create_proof(curve, k, g, w, u, rng, transcript: ipa.Transcript, queries):
    """multiopen/prover.rs:21-125.  queries: (point int, poly (n, 4) Montgomery limbs, blind (4,) limbs); polynomials are
    identified by object identity, as the reference's PolynomialPointer does (:128-140)."""
    sf = co.field_of_curve(curve, "scalar")
    m = o.CURVES[curve][1]
    n = 1 << k
    I = lambda limbs: co.limbs_to_ints(co.from_mont(sf, np.ascontiguousarray(limbs).reshape(-1, 4)))[0]
    L = lambda v: co.to_mont(sf, co.ints_to_limbs([v % m]))[0]
    x_1 = transcript.squeeze_challenge()
    x_2 = transcript.squeeze_challenge()
    polys = {id(q[1]): (q[1], q[2]) for q in queries}
    sets = construct_intermediate_sets([(pt, id(poly), None) for pt, poly, _ in queries])
    if sets is None:
        raise ValueError("queries iterator contains mismatching evaluations")       # :41-46
    poly_map, point_sets = sets
    q_polys = [None] * len(point_sets)
    q_blinds = [0] * len(point_sets)
    for data in poly_map:                                                           # :53-72
        poly, blind = polys[data["key"]]
        si = data["set_index"]
        q_polys[si] = np.ascontiguousarray(poly).copy() if q_polys[si] is None else co.scale_add(sf, q_polys[si], L(x_1), poly)
        q_blinds[si] = (q_blinds[si] * x_1 + I(blind)) % m
    q_prime = None
    for points, poly in zip(point_sets, q_polys):                                   # :75-97
        cur = poly
        for pt in points:
            cur = co.kate_division(sf, cur, L(pt))
        padded = np.zeros((n, 4), dtype=np.uint64)
        padded[:cur.shape[0]] = cur
        q_prime = padded if q_prime is None else co.scale_add(sf, q_prime, L(x_2), padded)
    q_prime_blind = rng(1)[0].copy()                                                # :99
    transcript.write_point(co.jac_to_affine_ints(curve, co.commit(curve, g, w, q_prime, q_prime_blind)))
    x_3 = transcript.squeeze_challenge()
    for q in q_polys:                                                               # :108-110
        transcript.write_scalar(I(co.eval_polynomial(sf, q, L(x_3))))
    x_4 = transcript.squeeze_challenge()
    p_poly, p_blind = q_prime, I(q_prime_blind)
    for q, b in zip(q_polys, q_blinds):                                             # :114-122
        p_poly = co.scale_add(sf, p_poly, L(x_4), q)
        p_blind = (p_blind * x_4 + b) % m
    ipa.create_proof(curve, k, g, w, u, rng, transcript, p_poly, L(p_blind), L(x_3))


def verify_proof(curve, k, g, w, u, transcript: ipa.Transcript, queries) -> bool:
    """multiopen/verifier.rs:15-141 with an empty caller msm, then commitment::verify_proof + use_challenges + eval.
    queries: (point int, commitment (x, y) -- identified by object identity --, eval int).  Returns False where the
    reference returns Err(OpeningError) for contradictory queries."""
    m = o.CURVES[curve][1]
    x_1 = transcript.squeeze_challenge()
    x_2 = transcript.squeeze_challenge()
    comms = {id(q[1]): q[1] for q in queries}
    sets = construct_intermediate_sets([(pt, id(c), ev) for pt, c, ev in queries])
    if sets is None:
        return False
    commitment_map, point_sets = sets
    q_commitments = [[[], 1] for _ in point_sets]                                   # (terms, next x_1 power)
    q_eval_sets = [[0] * len(ps) for ps in point_sets]
    for data in reversed(commitment_map):                                           # :75-81
        terms, power = q_commitments[data["set_index"]]
        terms.append((power, comms[data["key"]]))
        for i, ev in enumerate(data["evals"]):
            q_eval_sets[data["set_index"]][i] = (q_eval_sets[data["set_index"]][i] + ev * power) % m
        q_commitments[data["set_index"]][1] = power * x_1 % m
    q_prime_commitment = transcript.read_point()
    x_3 = transcript.squeeze_challenge()
    us = [transcript.read_scalar() for _ in q_eval_sets]
    msm_eval = 0
    for points, evals, proof_eval in zip(point_sets, q_eval_sets, us):              # :101-116
        r_poly = lagrange_interpolate(points, evals, m)
        r_eval = sum(c * pow(x_3, i, m) for i, c in enumerate(r_poly)) % m
        ev = (proof_eval - r_eval) % m
        for pt in points:
            ev = ev * pow((x_3 - pt) % m, -1, m) % m
        msm_eval = (msm_eval * x_2 + ev) % m
    x_4 = transcript.squeeze_challenge()
    terms = [(1, q_prime_commitment)]                                               # :123-136
    v = msm_eval
    for (q_terms, _), q_eval in zip(q_commitments, us):
        terms = [(s * x_4 % m, pt) for s, pt in terms] + q_terms
        v = (v * x_4 + q_eval) % m
    return ipa.verify_proof(curve, k, g, w, u, transcript, None, x_3, v, terms=terms)
