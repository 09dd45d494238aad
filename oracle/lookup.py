"""TEST INFRASTRUCTURE ONLY: restatement of the lookup argument's prover on Python integers --
`permute_expression_pair` (halo2_proofs/src/plonk/lookup/prover.rs:557-647), `commit_product` (:246-386), the constraint
expressions of `construct` (:395-470) and the verifier's expressions (plonk/lookup/verifier.rs:96-170).  Nothing here is
imported by the product; tests compare the device-resident prover with it."""
from __future__ import annotations


def permute_expression_pair(inputs, table, usable_rows: int):
    """prover.rs:567-623 over the usable rows (the blinding tail of :624-627 is the caller's).  Field elements order by their
    canonical integer value (pasta_curves' `Ord`).  Returns (A', S') or None for Error::ConstraintSystemFailure."""
    a = sorted(inputs[:usable_rows])                                    # :571-574
    leftover = {}                                                       # BTreeMap value -> count (:577-583)
    for v in table[:usable_rows]:
        leftover[v] = leftover.get(v, 0) + 1
    s = [0] * usable_rows
    repeated = []
    for row, v in enumerate(a):                                         # :586-613
        if row == 0 or v != a[row - 1]:
            s[row] = v
            if v not in leftover:
                return None
            assert leftover[v] > 0
            leftover[v] -= 1
        else:
            repeated.append(row)
    for v in sorted(leftover):                                          # :616-621: BTreeMap iterates ascending, rows pop off the end
        for _ in range(leftover[v]):
            s[repeated.pop()] = v
    assert not repeated
    return a, s


def product(compressed_input, compressed_table, permuted_input, permuted_table, beta: int, gamma: int, blinding_rows, m: int):
    """commit_product's z (:263-330): n - blinding_factors running-product rows, then the given random rows."""
    n = len(compressed_input)
    frac = [(beta + a_) * (gamma + s_) % m for a_, s_ in zip(permuted_input, permuted_table)]
    frac = [pow(v, -1, m) if v else 0 for v in frac]
    frac = [f * (ci + beta) % m * (ct + gamma) % m for f, ci, ct in zip(frac, compressed_input, compressed_table)]
    z, state = [], 1
    for cur in [1] + frac:
        state = state * cur % m
        z.append(state)
    return z[: n - len(blinding_rows)] + list(blinding_rows)


def constraint_trees(beta: int, gamma: int, m: int, z: int, a_perm: int, s_perm: int, a_comp, s_comp, l0: int, l_blind: int, l_last: int):
    """construct's five expressions (:417-466) as oracle/evaluator.py trees; z / a_perm / s_perm / l* index the polynomial list,
    a_comp / s_comp are the trees of the compressed input / table cosets."""
    P = lambda i, r=0: ("poly", i, r)
    neg = lambda t: ("scale", t, m - 1)
    sub = lambda x, y: ("add", x, neg(y))
    one = ("constant", 1)
    active = sub(one, ("add", P(l_last), P(l_blind)))
    left = ("mul", ("mul", P(z, 1), ("add", P(a_perm), ("constant", beta))), ("add", P(s_perm), ("constant", gamma)))
    right = ("mul", ("mul", P(z), ("add", a_comp, ("constant", beta))), ("add", s_comp, ("constant", gamma)))
    return [
        ("mul", sub(one, P(z)), P(l0)),
        ("mul", sub(("mul", P(z), P(z)), P(z)), P(l_last)),
        ("mul", sub(left, right), active),
        ("mul", sub(P(a_perm), P(s_perm)), P(l0)),
        ("mul", ("mul", sub(P(a_perm), P(s_perm)), sub(P(a_perm), P(a_perm, -1))), active),
    ]


def verifier_expressions(product_eval, product_next_eval, permuted_input_eval, permuted_input_inv_eval, permuted_table_eval,
                         compressed_input_eval, compressed_table_eval, l_0, l_last, l_blind, beta, gamma, m):
    """verifier.rs:96-170, with the theta-compression of the argument's expressions already applied by the caller."""
    active = (1 - (l_last + l_blind)) % m
    left = product_next_eval * (permuted_input_eval + beta) % m * (permuted_table_eval + gamma) % m
    right = product_eval * (compressed_input_eval + beta) % m * (compressed_table_eval + gamma) % m
    return [
        l_0 * (1 - product_eval) % m,
        l_last * (product_eval * product_eval - product_eval) % m,
        (left - right) * active % m,
        l_0 * (permuted_input_eval - permuted_table_eval) % m,
        (permuted_input_eval - permuted_table_eval) * (permuted_input_eval - permuted_input_inv_eval) % m * active % m,
    ]


def commit_permuted(curve, g_lagrange, w, blinding_factors: int, compressed_input, compressed_table, rng, transcript):
    """prover.rs:192-222 on integers: permute, append the blinding rows, commit A' then S'.  Returns
    (A', S', A' blind, S' blind) or raises on ConstraintSystemFailure."""
    import numpy as np
    from . import c_oracle as co
    sf = co.field_of_curve(curve, "scalar")
    I = lambda limbs: co.limbs_to_ints(co.from_mont(sf, np.ascontiguousarray(limbs).reshape(-1, 4)))
    L = lambda vals: co.to_mont(sf, co.ints_to_limbs(list(vals)))
    n = len(compressed_input)
    usable = n - (blinding_factors + 1)
    pair = permute_expression_pair(compressed_input, compressed_table, usable)
    if pair is None:
        raise ValueError("ConstraintSystemFailure")
    a = pair[0] + I(rng(blinding_factors + 1))
    s = pair[1] + I(rng(blinding_factors + 1))
    a_blind_l = rng(1)[0].copy()
    a_comm = co.jac_to_affine_ints(curve, co.commit(curve, g_lagrange, w, L(a), a_blind_l))
    s_blind_l = rng(1)[0].copy()
    s_comm = co.jac_to_affine_ints(curve, co.commit(curve, g_lagrange, w, L(s), s_blind_l))
    transcript.write_point(a_comm)
    transcript.write_point(s_comm)
    return a, s, I(a_blind_l)[0], I(s_blind_l)[0]


def commit_product(curve, g_lagrange, w, blinding_factors: int, compressed_input, compressed_table, permuted_input, permuted_table,
                   beta: int, gamma: int, m: int, rng, transcript):
    """prover.rs:263-370 on integers -> (z Lagrange, blind)."""
    import numpy as np
    from . import c_oracle as co
    sf = co.field_of_curve(curve, "scalar")
    I = lambda limbs: co.limbs_to_ints(co.from_mont(sf, np.ascontiguousarray(limbs).reshape(-1, 4)))
    L = lambda vals: co.to_mont(sf, co.ints_to_limbs(list(vals)))
    z = product(compressed_input, compressed_table, permuted_input, permuted_table, beta, gamma,
                I(rng(blinding_factors)) if blinding_factors else [], m)
    blind_l = rng(1)[0].copy()
    transcript.write_point(co.jac_to_affine_ints(curve, co.commit(curve, g_lagrange, w, L(z), blind_l)))
    return z, I(blind_l)[0]
