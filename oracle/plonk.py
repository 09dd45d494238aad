"""TEST INFRASTRUCTURE ONLY: restatement of `plonk::verify_proof` (halo2_proofs/src/plonk/verifier.rs:65-347, with
vanishing/verifier.rs, permutation/verifier.rs, lookup/verifier.rs) and of the verifying-key part of keygen
(plonk/keygen.rs:200-290) on Python integers and the C oracle's commits, for one or several circuit instances and a constraint system in
the lowered form halo2_amd/plonk.py documents (gate / lookup expressions are callables over `cells`).  Nothing here is imported
by the product; tests use it to verify the proofs the device prover writes."""
from __future__ import annotations

import numpy as np


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


def create_proof(curve, k, g, w, u, cs, fixed_columns, mapping, vk_repr: int, advice_columns, instance_columns, rng, transcript) -> None:
    create_proof_many(curve, k, g, w, u, cs, fixed_columns, mapping, vk_repr, [(advice_columns, instance_columns)], rng, transcript)


def verify_proof(curve, k, g, w, u, vk, instance_columns, proof: bytes) -> bool:
    return verify_proof_many(curve, k, g, w, u, vk, [instance_columns], proof)


def verify_proof_many(curve, k, g, w, u, vk, instances, proof: bytes) -> bool:
    """instances[i] = the instance columns of circuit i (`instances: &[&[&[C::Scalar]]]`, verifier.rs:76).  A proof that does not
    parse (the transcript's assertions: short read, non-canonical scalar, not a curve point) is a rejected proof."""
    try:
        return _verify_many(curve, k, g, w, u, vk, instances, proof)
    except (AssertionError, ValueError, IndexError):
        return False


def _verify_many(curve, k, g, w, u, vk, instances, proof: bytes) -> bool:
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


class _T:
    """Expression algebra producing oracle/evaluator.py trees: what a lowered gate callable sees on the oracle prover's side."""

    def __init__(self, tree):
        self.tree = tree

    @staticmethod
    def of(v, m):
        return v if isinstance(v, _T) else _T(("constant", int(v) % m))

    def _bin(self, kind, other):
        return _T((kind, self.tree, _T.of(other, _T.m).tree))

    def __add__(self, other):
        return self._bin("add", other)

    __radd__ = __add__

    def __neg__(self):
        return _T(("scale", self.tree, _T.m - 1))

    def __sub__(self, other):
        return self + (-_T.of(other, _T.m))

    def __rsub__(self, other):
        return _T.of(other, _T.m) - self

    def __mul__(self, other):
        return self._bin("mul", other) if isinstance(other, _T) else _T(("scale", self.tree, int(other) % _T.m))

    __rmul__ = __mul__


class _TreeCells:
    def __init__(self, nf, na, circuit_base):
        self.nf, self.na, self.base = nf, na, circuit_base

    def fixed(self, col, rot=0):
        return _T(("poly", col, rot))

    def advice(self, col, rot=0):
        return _T(("poly", self.base + col, rot))

    def instance(self, col, rot=0):
        return _T(("poly", self.base + self.na + col, rot))


def create_proof_many(curve, k, g, w, u, cs, fixed_columns, mapping, vk_repr: int, circuits, rng, transcript: ipa.Transcript):
    """plonk/prover.rs:35-724 on integers, sequentially: circuits = [(advice columns, instance columns)] as integer lists.
    Draws from rng in the reference's order; every commitment is the C oracle's `best_multiexp`."""
    from . import evaluator as oev
    from . import vanishing as ov
    sf = co.field_of_curve(curve, "scalar")
    m = o.CURVES[curve][1]
    _T.m = m
    dom = o.EvaluationDomain(cs.degree, k, m)
    n, bf = dom.n, cs.blinding_factors
    usable = n - (bf + 1)
    I = lambda limbs: co.limbs_to_ints(co.from_mont(sf, np.ascontiguousarray(limbs).reshape(-1, 4)))
    L = lambda vals: co.to_mont(sf, co.ints_to_limbs([v % m for v in vals]))
    one = L([1])[0]
    evalp = lambda poly, pt: sum(c * pow(pt, i, m) for i, c in enumerate(poly)) % m
    g_lagrange = co.lagrange_basis(curve, g, k)                      # Params::new's point FFT (poly/commitment.rs:77-100)
    commit_lagrange = lambda values, blind_l: co.jac_to_affine_ints(curve, co.commit(curve, g_lagrange, w, L(values), blind_l))
    nf, na, ni = cs.num_fixed_columns, cs.num_advice_columns, cs.num_instance_columns

    transcript.common_scalar(vk_repr % m)
    inst_all = []
    for _, instance_columns in circuits:
        cols = [list(v) + [0] * (n - len(v)) for v in instance_columns]
        for col in cols:
            transcript.common_point(commit_lagrange(col, one))
        inst_all.append(cols)
    adv_all, blinds_all = [], []
    for advice_columns, _ in circuits:
        cols = [list(c) for c in advice_columns]
        for col in cols:
            col[usable:] = I(rng(bf + 1))
        blinds = [rng(1)[0].copy() for _ in cols]
        for col, b in zip(cols, blinds):
            transcript.write_point(commit_lagrange(col, b))
        adv_all.append(cols)
        blinds_all.append(I(np.stack(blinds)) if blinds else [])
    # polynomial lists (same indexing in both bases): fixed, then per circuit advice + instance
    lag = [list(c) for c in fixed_columns]
    bases_of = []
    for adv, inst in zip(adv_all, inst_all):
        bases_of.append(len(lag))
        lag += adv + inst
    coeff = [dom.lagrange_to_coeff(c) for c in lag]
    ext = [dom.coeff_to_extended(c) for c in coeff]
    cells = [_TreeCells(nf, na, b) for b in bases_of]

    def add_poly(lagrange_values):
        c = dom.lagrange_to_coeff(lagrange_values)
        coeff.append(c)
        ext.append(dom.coeff_to_extended(c))
        lag.append(list(lagrange_values))
        return len(ext) - 1
    sigmas = operm.build_sigma(mapping, dom) if cs.permutation_columns else []
    sig_idx = [add_poly(s_) for s_ in sigmas]
    l0_i = add_poly([1] + [0] * (n - 1))
    lblind_i = add_poly([1 if r > usable else 0 for r in range(n)])
    llast_i = add_poly([1 if r == usable else 0 for r in range(n)])

    theta = transcript.squeeze_challenge()
    lookups = []                                          # per circuit, per lookup: dict of indices / values
    for cc in cells:
        per = []
        for ins, tabs in cs.lookups:
            def compress(es):
                tree = ("constant", 0)
                for e in es:
                    tree = ("add", ("scale", tree, theta), _T.of(e(cc), m).tree)
                return tree
            in_tree, tb_tree = compress(ins), compress(tabs)
            lag_eval = lambda t_: oev.evaluate(t_, lag, oev.LAGRANGE, m, dom.k, dom.extended_k, dom.omega, dom.extended_omega, dom.g_coset)
            comp_in, comp_tb = lag_eval(in_tree), lag_eval(tb_tree)
            pa, ps, pa_blind, ps_blind = olk.commit_permuted(curve, g_lagrange, w, bf, comp_in, comp_tb, rng, transcript)
            per.append(dict(in_tree=in_tree, tb_tree=tb_tree, comp_in=comp_in, comp_tb=comp_tb, pa=pa, ps=ps, pa_blind=pa_blind,
                            ps_blind=ps_blind, pa_i=add_poly(pa), ps_i=add_poly(ps)))
        lookups.append(per)
    beta = transcript.squeeze_challenge()
    gamma = transcript.squeeze_challenge()
    perms = []
    for ci, (adv, inst) in enumerate(zip(adv_all, inst_all)):
        pick = {"advice": adv, "fixed": fixed_columns, "instance": inst}
        cols = [pick[kind][idx] for kind, idx in cs.permutation_columns]
        sets = operm.commit(curve, dom, g_lagrange, w, cs.degree, bf, cols, sigmas, beta, gamma, rng, transcript) if cs.permutation_columns else []
        col_idx = [add_poly(c) for c in cols]             # contiguous copies, the layout operm.constraint_trees expects
        z_idx = [add_poly(z) for z, _ in sets]
        perms.append(dict(sets=sets, col_idx=col_idx, z_idx=z_idx))
    for per in lookups:
        for lk in per:
            z, z_blind = olk.commit_product(curve, g_lagrange, w, bf, lk["comp_in"], lk["comp_tb"], lk["pa"], lk["ps"], beta, gamma, m, rng, transcript)
            lk.update(z=z, z_blind=z_blind, z_i=add_poly(z))
    random_poly, random_blind = ov.commit(curve, dom, g, w, rng, transcript)
    y = transcript.squeeze_challenge()
    trees = []
    for cc, pm, per in zip(cells, perms, lookups):
        trees += [_T.of(gate(cc), m).tree for gate in cs.gates]
        if cs.permutation_columns:
            # constraint_trees wants the columns, the sigmas and the z sets as contiguous runs: col_idx, sig_idx and z_idx are
            trees += operm.constraint_trees(len(pm["sets"]), len(cs.permutation_columns), cs.degree, bf, beta, gamma, m, pm["col_idx"][0],
                                            sig_idx[0], pm["z_idx"][0], l0_i, lblind_i, llast_i)
        for lk in per:
            trees += olk.constraint_trees(beta, gamma, m, lk["z_i"], lk["pa_i"], lk["ps_i"], lk["in_tree"], lk["tb_tree"], l0_i, lblind_i, llast_i)
    pieces, piece_blinds = ov.construct(curve, dom, g, w, rng, transcript, ext, trees, y)
    x = transcript.squeeze_challenge()
    rot = lambda r: x * pow(dom.omega if r >= 0 else dom.omega_inv, abs(r), m) % m
    for ci, b in enumerate(bases_of):
        for col, r in cs.instance_queries:
            transcript.write_scalar(evalp(coeff[b + na + col], rot(r)))
    for ci, b in enumerate(bases_of):
        for col, r in cs.advice_queries:
            transcript.write_scalar(evalp(coeff[b + col], rot(r)))
    for col, r in cs.fixed_queries:
        transcript.write_scalar(evalp(coeff[col], rot(r)))
    h_poly, h_blind = ov.evaluate(dom, pieces, piece_blinds, random_poly, x, transcript)
    for i in sig_idx:
        transcript.write_scalar(evalp(coeff[i], x))
    x_next, x_last, x_inv = rot(1), rot(-(bf + 1)), rot(-1)
    for pm in perms:
        for i, zi in enumerate(pm["z_idx"]):
            transcript.write_scalar(evalp(coeff[zi], x))
            transcript.write_scalar(evalp(coeff[zi], x_next))
            if i + 1 < len(pm["z_idx"]):
                transcript.write_scalar(evalp(coeff[zi], x_last))
    for per in lookups:
        for lk in per:
            for poly_i, pt in ((lk["z_i"], x), (lk["z_i"], x_next), (lk["pa_i"], x), (lk["pa_i"], x_inv), (lk["ps_i"], x)):
                transcript.write_scalar(evalp(coeff[poly_i], pt))
    # queries: (point, polynomial limbs, blind limbs); one limb array per polynomial so identities are shared
    limbs = {}

    def P(i):
        if i not in limbs:
            limbs[i] = L(coeff[i])
        return limbs[i]
    queries = []
    for b, blinds, pm, per in zip(bases_of, blinds_all, perms, lookups):
        queries += [(rot(r), P(b + na + col), one) for col, r in cs.instance_queries]
        queries += [(rot(r), P(b + col), L([blinds[col]])[0]) for col, r in cs.advice_queries]
        for zi, (_, zb) in zip(pm["z_idx"], pm["sets"]):
            queries += [(x, P(zi), L([zb])[0]), (x_next, P(zi), L([zb])[0])]
        for zi, (_, zb) in reversed(list(zip(pm["z_idx"], pm["sets"]))[:-1]):
            queries.append((x_last, P(zi), L([zb])[0]))
        for lk in per:
            zb, ab, sb = L([lk["z_blind"]])[0], L([lk["pa_blind"]])[0], L([lk["ps_blind"]])[0]
            queries += [(x, P(lk["z_i"]), zb), (x, P(lk["pa_i"]), ab), (x, P(lk["ps_i"]), sb), (x_inv, P(lk["pa_i"]), ab), (x_next, P(lk["z_i"]), zb)]
    queries += [(rot(r), P(col), one) for col, r in cs.fixed_queries]
    queries += [(x, P(i), one) for i in sig_idx]
    h_l, r_l = L(h_poly), L(random_poly)
    queries += [(x, h_l, L([h_blind])[0]), (x, r_l, L([random_blind])[0])]
    om.create_proof(curve, k, g, w, u, rng, transcript, queries)


def _fold(base: int, values, m: int) -> int:
    acc = 0
    for v in values:
        acc = (acc * base + v) % m
    return acc
