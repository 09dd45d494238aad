"""The permutation argument's prover, `plonk::permutation::Argument` (halo2_proofs/src/plonk/permutation/prover.rs:46-434):
the grand products z_i over column chunks, their commitments, the constraint expressions handed to the vanishing argument,
the evaluations and the opening queries.  Everything O(n) runs on the device:

* the row fractions prod_j (v_j + delta^j omega^i beta + gamma) / (v_j + beta s_j + gamma) (prover.rs:101-141) are two
  `h2_evaluate_device` programs over the Lagrange columns (products in the Lagrange basis are row-wise, evaluator.rs:370-392; the
  delta^j omega^i beta term is an `Ast.linear`) around one `h2_batch_invert_device`;
* the running product z (:153-160) is `h2_grand_product_device`;
* commit_lagrange, lagrange_to_coeff, coeff_to_extended (:172-178) are the registered commit and the NTT entry points.
Per set the host sees one 32-byte row (last_z) and one commitment.

The reference resolves a column to advice / fixed / instance storage (:107-111); here the caller passes the resolved
Lagrange columns in the argument's column order.  torch is plumbing; all arithmetic goes through the C ABI."""
from __future__ import annotations

import numpy as np

from . import fields
from .arithmetic import batch_invert, eval_polynomial, grand_product
from .commitment import Blind, Params
from .evaluator import LAGRANGE, Ast, Evaluator
from .multiopen import ProverQuery
from .transcript import write_evaluation


def _host(t) -> np.ndarray:
    return t.cpu().numpy().view(np.uint64)


class ProvingKey:
    """permutation::ProvingKey (plonk/permutation.rs:76-81): the sigma columns in the three bases.  permutations: Lagrange
    device tensors, polys: coefficient form, cosets: AstLeafs of the extended evaluator."""

    def __init__(self, permutations, polys, cosets):
        self.permutations, self.polys, self.cosets = list(permutations), list(polys), list(cosets)

    def open(self, x, field: int) -> list[ProverQuery]:                                   # prover.rs:310-319
        return [ProverQuery(x, poly, Blind(field=field)) for poly in self.polys]

    def evaluate(self, x, field: int, transcript) -> None:                                # prover.rs:321-333
        for poly in self.polys:
            write_evaluation(transcript, eval_polynomial(poly, x, field))


class Argument:
    def __init__(self, num_columns: int):
        self.num_columns = num_columns

    def commit(self, params: Params, domain, cs_degree: int, blinding_factors: int, pkey: ProvingKey, columns, beta: int,
               gamma: int, evaluator, rng, transcript) -> "Committed":
        """prover.rs:46-197.  columns: the argument's columns as Lagrange CUDA tensors (n, 4), in order; beta, gamma canonical
        integers; evaluator: the extended-basis Evaluator the product cosets are registered with; rng(count) -> (count, 4)
        Montgomery limbs (per set: `blinding_factors` rows, then the blind)."""
        import torch
        if cs_degree < 3:
            raise ValueError("the permutation argument needs a constraint system of degree >= 3")       # :69
        if len(columns) != self.num_columns or len(pkey.permutations) != self.num_columns:
            raise ValueError("permutation: column count mismatch")
        sf, m, n = domain.field, domain.m, params.n
        chunk_len = cs_degree - 2
        delta = fields.delta(sf)
        deltaomega = 1                                     # each column gets its own delta power (:74)
        last_z = fields.scalar_limbs(1, sf, True)          # the previous set's last value (:77)
        sets = []
        for c0 in range(0, self.num_columns, chunk_len):
            cols = columns[c0:c0 + chunk_len]
            perms = pkey.permutations[c0:c0 + chunk_len]
            rows = Evaluator(LAGRANGE)
            v_leaves = [rows.register_poly(v) for v in cols]
            s_leaves = [rows.register_poly(s) for s in perms]
            den = None
            for v, s in zip(v_leaves, s_leaves):           # prod_j (beta s_j + gamma + v_j), :101-120
                term = Ast.of(s) * beta + Ast.constant(gamma) + Ast.of(v)
                den = term if den is None else den * term
            inv = batch_invert(rows.evaluate(den, domain), sf)                                        # :123
            num = Ast.of(rows.register_poly(inv))
            for v in v_leaves:                             # prod_j (delta^j omega^i beta + gamma + v_j), :127-146
                num = num * (Ast.linear(deltaomega * beta % m) + Ast.constant(gamma) + Ast.of(v))
                deltaomega = deltaomega * delta % m
            z = grand_product(rows.evaluate(num, domain), n, last_z, sf)                              # :153-160
            if blinding_factors:
                z[n - blinding_factors:] = torch.from_numpy(
                    np.ascontiguousarray(rng(blinding_factors), dtype=np.uint64).view(np.int64)).to(z.device)     # :162-165
            last_z = _host(z[n - (blinding_factors + 1)]).copy()                                      # :167
            blind = Blind(np.ascontiguousarray(rng(1)[0]))                                            # :169
            transcript.write_point(_host(params.commit_lagrange(z, blind)))                           # :171, :182-186
            poly = domain.lagrange_to_coeff(z)                                                        # :173 (in place)
            coset = evaluator.register_poly(domain.coeff_to_extended(poly))                           # :176-177
            sets.append(CommittedSet(poly, coset, blind))
        return Committed(sets)


class CommittedSet:
    def __init__(self, poly, coset, blind: Blind):
        self.permutation_product_poly, self.permutation_product_coset, self.permutation_product_blind = poly, coset, blind


class Committed:
    def __init__(self, sets):
        self.sets = sets

    def construct(self, domain, cs_degree: int, blinding_factors: int, pkey: ProvingKey, column_cosets, l0, l_blind, l_last,
                  beta: int, gamma: int):
        """prover.rs:199-306 -> (Constructed, expressions).  column_cosets: the argument's columns as AstLeafs of the extended
        evaluator; l0, l_blind, l_last: AstLeafs of the same evaluator."""
        m = domain.m
        chunk_len = cs_degree - 2
        last_rotation = -(blinding_factors + 1)
        delta = fields.delta(domain.field)
        A = Ast.of
        exprs = []
        if self.sets:
            exprs.append((Ast.one() - A(self.sets[0].permutation_product_coset)) * A(l0))                       # :232-238
            zl = self.sets[-1].permutation_product_coset
            exprs.append((A(zl) * A(zl) - A(zl)) * A(l_last))                                                    # :239-246
        for prev, cur in zip(self.sets, self.sets[1:]):                                                         # :247-262
            exprs.append((A(cur.permutation_product_coset) - A(prev.permutation_product_coset.with_rotation(last_rotation))) * A(l0))
        for ci, s in enumerate(self.sets):                                                                      # :263-302
            cols = column_cosets[ci * chunk_len:(ci + 1) * chunk_len]
            sig = pkey.cosets[ci * chunk_len:(ci + 1) * chunk_len]
            left = A(s.permutation_product_coset.with_rotation(1))
            for v, p_ in zip(cols, sig):
                left = left * (A(v) + Ast.constant(beta) * A(p_) + Ast.constant(gamma))
            right = A(s.permutation_product_coset)
            cur_delta = beta * pow(delta, ci * chunk_len, m) % m
            for v in cols:
                right = right * (A(v) + Ast.linear(cur_delta) + Ast.constant(gamma))
                cur_delta = cur_delta * delta % m
            exprs.append((left - right) * (Ast.one() - (A(l_last) + A(l_blind))))
        return Constructed([ConstructedSet(s.permutation_product_poly, s.permutation_product_blind) for s in self.sets]), exprs


class ConstructedSet:
    def __init__(self, poly, blind: Blind):
        self.permutation_product_poly, self.permutation_product_blind = poly, blind


class Constructed:
    def __init__(self, sets):
        self.sets = sets

    def evaluate(self, domain, blinding_factors: int, x: int, transcript) -> "Evaluated":
        """prover.rs:336-381.  x: canonical integer."""
        sf = domain.field
        lim = lambda v: fields.scalar_limbs(v % domain.m, sf, True)
        for i, s in enumerate(self.sets):
            write_evaluation(transcript, eval_polynomial(s.permutation_product_poly, lim(x), sf))
            write_evaluation(transcript, eval_polynomial(s.permutation_product_poly, lim(domain.rotate_omega(x, 1)), sf))
            if i + 1 < len(self.sets):                                                     # chain to the next set (:366-373)
                write_evaluation(transcript, eval_polynomial(s.permutation_product_poly, lim(domain.rotate_omega(x, -(blinding_factors + 1))), sf))
        return Evaluated(self)


class Evaluated:
    def __init__(self, constructed: Constructed):
        self.constructed = constructed

    def open(self, domain, blinding_factors: int, x: int) -> list[ProverQuery]:
        """prover.rs:384-433."""
        sf = domain.field
        lim = lambda v: fields.scalar_limbs(v % domain.m, sf, True)
        x_l, x_next, x_last = lim(x), lim(domain.rotate_omega(x, 1)), lim(domain.rotate_omega(x, -(blinding_factors + 1)))
        out = []
        for s in self.constructed.sets:
            out.append(ProverQuery(x_l, s.permutation_product_poly, s.permutation_product_blind))
            out.append(ProverQuery(x_next, s.permutation_product_poly, s.permutation_product_blind))
        for s in reversed(self.constructed.sets[:-1]):                                     # `.rev().skip(1)` (:419-431)
            out.append(ProverQuery(x_last, s.permutation_product_poly, s.permutation_product_blind))
        return out
