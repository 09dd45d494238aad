"""TEST INFRASTRUCTURE ONLY: restatement of the permutation argument on Python integers -- the sigma columns of keygen
(halo2_proofs/src/plonk/permutation/keygen.rs:110-190), the prover's grand products, constraint expressions and evaluations
(plonk/permutation/prover.rs:46-381) and the verifier's expressions (plonk/permutation/verifier.rs:102-190).  Nothing here is
imported by the product; tests compare the device-resident prover with it."""
from __future__ import annotations

import numpy as np

from . import c_oracle as co
from . import pasta as o

DELTA = {m: pow(5, 1 << 32, m) for m in (o.P, o.Q)}          # ff::PrimeField::DELTA = GENERATOR^(2^S)


def build_sigma(mapping, dom: o.EvaluationDomain):
    """keygen.rs:163-190: sigma_i(omega^j) = delta^i' omega^j' for mapping[i][j] = (i', j')."""
    m, n = dom.m, dom.n
    om = [pow(dom.omega, j, m) for j in range(n)]
    return [[pow(DELTA[m], mapping[i][j][0], m) * om[mapping[i][j][1]] % m for j in range(n)] for i in range(len(mapping))]


def commit(curve, dom: o.EvaluationDomain, g_lagrange, w, cs_degree: int, blinding_factors: int, columns, sigmas, beta: int,
           gamma: int, rng, transcript):
    """prover.rs:46-197 -> [(z Lagrange ints, blind int)] per set; writes each set's commitment."""
    sf = co.field_of_curve(curve, "scalar")
    m, n = dom.m, dom.n
    I = lambda limbs: co.limbs_to_ints(co.from_mont(sf, np.ascontiguousarray(limbs).reshape(-1, 4)))
    L = lambda vals: co.to_mont(sf, co.ints_to_limbs([v % m for v in vals]))
    chunk_len = cs_degree - 2
    deltaomega, last_z = 1, 1
    sets = []
    for c0 in range(0, len(columns), chunk_len):
        modified = [1] * n
        for v, s in zip(columns[c0:c0 + chunk_len], sigmas[c0:c0 + chunk_len]):
            modified = [a * (beta * si + gamma + vi) % m for a, vi, si in zip(modified, v, s)]
        modified = [pow(a, -1, m) if a else 0 for a in modified]
        for v in columns[c0:c0 + chunk_len]:
            cur = deltaomega
            for i in range(n):
                modified[i] = modified[i] * (cur * beta + gamma + v[i]) % m
                cur = cur * dom.omega % m
            deltaomega = deltaomega * DELTA[m] % m
        z = [last_z]
        for row in range(1, n):
            z.append(z[row - 1] * modified[row - 1] % m)
        if blinding_factors:
            z[n - blinding_factors:] = I(rng(blinding_factors))
        last_z = z[n - (blinding_factors + 1)]
        blind_l = rng(1)[0].copy()
        transcript.write_point(co.jac_to_affine_ints(curve, co.commit(curve, g_lagrange, w, L(z), blind_l)))
        sets.append((z, I(blind_l)[0]))
    return sets


def constraint_trees(n_sets: int, n_columns: int, cs_degree: int, blinding_factors: int, beta: int, gamma: int, m: int,
                     col0: int, sig0: int, z0: int, l0: int, l_blind: int, l_last: int):
    """prover.rs:229-303 as oracle/evaluator.py trees; col0 / sig0 / z0 / l0 / l_blind / l_last index the polynomial list."""
    P = lambda i, r=0: ("poly", i, r)
    neg = lambda t: ("scale", t, m - 1)
    one = ("constant", 1)
    chunk_len = cs_degree - 2
    last_rot = -(blinding_factors + 1)
    out = [("mul", ("add", one, neg(P(z0))), P(l0))]
    zl = z0 + n_sets - 1
    out.append(("mul", ("add", ("mul", P(zl), P(zl)), neg(P(zl))), P(l_last)))
    for i in range(1, n_sets):
        out.append(("mul", ("add", P(z0 + i), neg(P(z0 + i - 1, last_rot))), P(l0)))
    for ci in range(n_sets):
        cols = range(ci * chunk_len, min((ci + 1) * chunk_len, n_columns))
        left = P(z0 + ci, 1)
        for j in cols:
            left = ("mul", left, ("add", ("add", P(col0 + j), ("mul", ("constant", beta), P(sig0 + j))), ("constant", gamma)))
        right = P(z0 + ci)
        cur = beta * pow(DELTA[m], ci * chunk_len, m) % m
        for j in cols:
            right = ("mul", right, ("add", ("add", P(col0 + j), ("linear", cur)), ("constant", gamma)))
            cur = cur * DELTA[m] % m
        out.append(("mul", ("add", left, neg(right)), ("add", one, neg(("add", P(l_last), P(l_blind))))))
    return out


def verifier_expressions(cs_degree: int, column_evals, sigma_evals, z_evals, l_0: int, l_last: int, l_blind: int, beta: int,
                         gamma: int, x: int, m: int):
    """verifier.rs:102-190.  z_evals: per set (eval at x, eval at omega x, eval at omega^last x or None)."""
    chunk_len = cs_degree - 2
    out = [l_0 * (1 - z_evals[0][0]) % m]
    zl = z_evals[-1][0]
    out.append((zl * zl - zl) * l_last % m)
    for i in range(1, len(z_evals)):
        out.append((z_evals[i][0] - z_evals[i - 1][2]) * l_0 % m)
    for ci, (z_x, z_next, _) in enumerate(z_evals):
        cols = range(ci * chunk_len, min((ci + 1) * chunk_len, len(column_evals)))
        left = z_next
        for j in cols:
            left = left * (column_evals[j] + beta * sigma_evals[j] + gamma) % m
        right = z_x
        cur = beta * x % m * pow(DELTA[m], ci * chunk_len, m) % m
        for j in cols:
            right = right * (column_evals[j] + cur + gamma) % m
            cur = cur * DELTA[m] % m
        out.append((left - right) * (1 - (l_last + l_blind)) % m)
    return out
