"""TEST INFRASTRUCTURE ONLY: restatement of `plonk::verify_proof` (halo2_proofs/src/plonk/verifier.rs:65-347, with
vanishing/verifier.rs, permutation/verifier.rs, lookup/verifier.rs) and of the verifying-key part of keygen
(plonk/keygen.rs:200-290) on Python integers and the C oracle's commits, for one or several circuit instances and a constraint system in
the lowered form halo2_amd/plonk.py documents (gate / lookup expressions are callables over `cells`).  Nothing here is imported
by the product; tests use it to verify the proofs the device prover writes."""
from __future__ import annotations


from . import c_oracle as co
from . import ipa
from . import lookup as olk
from . import multiopen as om
from . import pasta as o
from . import permutation as operm


class _EvalCells:
    """`Expression::evaluate` with the verifier's closures (verifier.rs:251-265): a query is its evaluation."""

    def __init__(self, cs, fixed_evals, advice_evals, instance_evals, m):
        self.cs, self.f, self.a, self.i, self.m = cs, fixed_evals, advice_evals, instance_evals, m

    def fixed(self, col, rot=0):
        return self.f[self.cs.fixed_queries.index((col, rot))]

    def advice(self, col, rot=0):
        return self.a[self.cs.advice_queries.index((col, rot))]

    def instance(self, col, rot=0):
        return self.i[self.cs.instance_queries.index((col, rot))]


def keygen_vk(curve, k, g, w, cs, fixed_columns, mapping, vk_repr: int):
    """Fixed and sigma commitments (blind 1: `Blind::default()`, keygen.rs:252, permutation/keygen.rs:150)."""
    sf = co.field_of_curve(curve, "scalar")
    m = o.CURVES[curve][1]
    dom = o.EvaluationDomain(cs.degree, k, m)
    L = lambda vals: co.to_mont(sf, co.ints_to_limbs([v % m for v in vals]))
    one = L([1])[0]
    comm = lambda lagrange: co.jac_to_affine_ints(curve, co.commit(curve, g, w, L(dom.lagrange_to_coeff(lagrange)), one))
    sigmas = operm.build_sigma(mapping, dom) if cs.permutation_columns else []
    return {"cs": cs, "vk_repr": vk_repr, "domain": dom, "fixed_commitments": [comm(c) for c in fixed_columns],
            "permutation_commitments": [comm(s) for s in sigmas]}


def verify_proof(curve, k, g, w, u, vk, instance_columns, proof: bytes) -> bool:
    return verify_proof_many(curve, k, g, w, u, vk, [instance_columns], proof)


def verify_proof_many(curve, k, g, w, u, vk, instances, proof: bytes) -> bool:
    """instances[i] = the instance columns of circuit i (`instances: &[&[&[C::Scalar]]]`, verifier.rs:76)."""
    cs, dom = vk["cs"], vk["domain"]
    sf = co.field_of_curve(curve, "scalar")
    m, n = dom.m, dom.n
    bf = cs.blinding_factors
    L = lambda vals: co.to_mont(sf, co.ints_to_limbs([v % m for v in vals]))
    one = L([1])[0]
    per_proof = range(len(instances))
    instance_commitments = []
    for instance_columns in instances:                                                  # :77-101
        if len(instance_columns) != cs.num_instance_columns:
            return False
        cms = []
        for values in instance_columns:
            if len(values) > n - (bf + 1):
                return False
            lag = list(values) + [0] * (n - len(values))
            cms.append(co.jac_to_affine_ints(curve, co.commit(curve, g, w, L(dom.lagrange_to_coeff(lag)), one)))
        instance_commitments.append(cms)
    t = ipa.Transcript(curve, proof)
    t.common_scalar(vk["vk_repr"] % m)                                                  # :106
    for cms in instance_commitments:                                                    # :108-112
        for c in cms:
            t.common_point(c)
    advice_commitments = [[t.read_point() for _ in range(cs.num_advice_columns)] for _ in per_proof]          # :115-120
    theta = t.squeeze_challenge()
    lookups_permuted = [[(t.read_point(), t.read_point()) for _ in cs.lookups] for _ in per_proof]            # :125-135
    beta = t.squeeze_challenge()
    gamma = t.squeeze_challenge()
    chunk_len = cs.degree - 2
    n_perm = len(cs.permutation_columns)
    n_sets = -(-n_perm // chunk_len) if n_perm else 0
    perm_product_commitments = [[t.read_point() for _ in range(n_sets)] for _ in per_proof]                   # :144-149
    lookup_product_commitments = [[t.read_point() for _ in cs.lookups] for _ in per_proof]                    # :151-159
    random_poly_commitment = t.read_point()                                             # :161
    y = t.squeeze_challenge()
    h_commitments = [t.read_point() for _ in range(dom.quotient_poly_degree)]           # :166
    x = t.squeeze_challenge()
    instance_evals = [[t.read_scalar() for _ in cs.instance_queries] for _ in per_proof]                      # :171-179
    advice_evals = [[t.read_scalar() for _ in cs.advice_queries] for _ in per_proof]
    fixed_evals = [t.read_scalar() for _ in cs.fixed_queries]
    random_eval = t.read_scalar()                                                       # :181
    sigma_evals = [t.read_scalar() for _ in range(n_perm)]                              # :183
    z_evals = []
    for _ in per_proof:                                                                 # permutation/verifier.rs:70-96
        zs = []
        for i in range(n_sets):
            e, e_next = t.read_scalar(), t.read_scalar()
            zs.append((e, e_next, t.read_scalar() if i + 1 < n_sets else None))
        z_evals.append(zs)
    lookup_evals = [[[t.read_scalar() for _ in range(5)] for _ in cs.lookups] for _ in per_proof]             # lookup/verifier.rs:72-93

    xn = pow(x, n, m)
    l_i = lambda row: pow(dom.omega, row, m) * pow(n, -1, m) % m * (xn - 1) % m * pow((x - pow(dom.omega, row, m)) % m, -1, m) % m
    usable = n - (bf + 1)
    l_0, l_last = l_i(0), l_i(usable)                                                   # :205-215
    l_blind = sum(l_i(r) for r in range(usable + 1, n)) % m
    exprs = []
    for p_ in per_proof:                                                                # :217-271
        cells = _EvalCells(cs, fixed_evals, advice_evals[p_], instance_evals[p_], m)
        exprs += [int(gate(cells)) % m for gate in cs.gates]
        if n_perm:
            pick = {"advice": cells.advice, "fixed": cells.fixed, "instance": cells.instance}
            exprs += operm.verifier_expressions(cs.degree, [pick[kind](idx, 0) for kind, idx in cs.permutation_columns], sigma_evals,
                                                z_evals[p_], l_0, l_last, l_blind, beta, gamma, x, m)
        for (ins, tabs), ev in zip(cs.lookups, lookup_evals[p_]):
            comp = lambda es: _fold(theta, [int(e(cells)) % m for e in es], m)
            exprs += olk.verifier_expressions(ev[0], ev[1], ev[2], ev[3], ev[4], comp(ins), comp(tabs), l_0, l_last, l_blind, beta, gamma, m)
    expected_h_eval = _fold(y, exprs, m) * pow((xn - 1) % m, -1, m) % m                  # vanishing/verifier.rs:103-105
    # h commitment = sum_i xn^i H_i (:107-116), as one point
    bm = o.CURVES[curve][0]
    h_commitment = None
    for c in reversed(h_commitments):
        h_commitment = o.ec_add(o.ec_mul(xn, h_commitment, bm) if h_commitment is not None else None, c, bm)

    rot = lambda r: x * pow(dom.omega if r >= 0 else dom.omega_inv, abs(r), m) % m
    x_next, x_last, x_inv = rot(1), rot(-(bf + 1)), rot(-1)
    queries = []                                                                        # :277-345
    for p_ in per_proof:
        queries += [(rot(r), instance_commitments[p_][c], e) for (c, r), e in zip(cs.instance_queries, instance_evals[p_])]
        queries += [(rot(r), advice_commitments[p_][c], e) for (c, r), e in zip(cs.advice_queries, advice_evals[p_])]
        for c, (e, e_next, _) in zip(perm_product_commitments[p_], z_evals[p_]):        # permutation/verifier.rs:192-241
            queries += [(x, c, e), (x_next, c, e_next)]
        for c, (_, _, e_last) in reversed(list(zip(perm_product_commitments[p_], z_evals[p_]))[:-1]):
            queries.append((x_last, c, e_last))
        for (pa, ps), pz, ev in zip(lookups_permuted[p_], lookup_product_commitments[p_], lookup_evals[p_]):  # lookup/verifier.rs:172-208
            queries += [(x, pz, ev[0]), (x, pa, ev[2]), (x, ps, ev[4]), (x_inv, pa, ev[3]), (x_next, pz, ev[1])]
    queries += [(rot(r), vk["fixed_commitments"][c], e) for (c, r), e in zip(cs.fixed_queries, fixed_evals)]
    queries += [(x, c, e) for c, e in zip(vk["permutation_commitments"], sigma_evals)]
    queries += [(x, h_commitment, expected_h_eval), (x, random_poly_commitment, random_eval)]
    return om.verify_proof(curve, k, g, w, u, t, queries)


def _fold(base: int, values, m: int) -> int:
    acc = 0
    for v in values:
        acc = (acc * base + v) % m
    return acc
