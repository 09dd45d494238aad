"""The lookup argument's prover, `plonk::lookup::Argument` (halo2_proofs/src/plonk/lookup/prover.rs:66-555): compress the
input and table expressions with theta, permute them (`permute_expression_pair`), commit A' and S', build and commit the
grand product z, hand five constraint expressions to the vanishing argument, evaluate and open.  Everything O(n) and the
O(n log n) sort run on the device:

* the compressed expressions are `h2_evaluate_device` programs over the Lagrange columns (prover.rs:180-181);
* (A', S') is `h2_permute_expression_pair_device` (bitonic sort + flags / scans / gather, halo2_amd/csrc/lookup.hip);
* the fractions of `commit_product` (:263-306) are two Lagrange-basis programs around one `h2_batch_invert_device`, z is
  `h2_grand_product_device`;
* commit_lagrange, lagrange_to_coeff, coeff_to_extended are the registered commit and the NTT entry points.

The reference evaluates each `Expression<F>` twice, over the Lagrange columns and over their cosets (:115-166).  Here an
expression is a callable taking the list of column leaves (value leaves or coset leaves, same indexing) and returning an
`Ast`; the argument applies it to both.  torch is plumbing; all arithmetic goes through the C ABI."""
from __future__ import annotations

import numpy as np

from . import fields
from .arithmetic import batch_invert, eval_polynomial, grand_product, permute_expression_pair
from .commitment import Blind, Params
from .evaluator import LAGRANGE, Ast, Evaluator
from .multiopen import ProverQuery
from .transcript import write_evaluation


def _host(t) -> np.ndarray:
    return t.cpu().numpy().view(np.uint64)


def _to_dev(limbs, like):
    import torch
    return torch.from_numpy(np.ascontiguousarray(limbs, dtype=np.uint64).view(np.int64)).to(like.device)


class Argument:
    def __init__(self, input_expressions, table_expressions):
        """input_expressions / table_expressions: equally many callables `leaves -> Ast` (lookup.rs:10-14)."""
        if len(input_expressions) != len(table_expressions):
            raise ValueError("lookup: input and table expression counts differ")
        self.input_expressions, self.table_expressions = list(input_expressions), list(table_expressions)

    def commit_permuted(self, params: Params, domain, blinding_factors: int, value_evaluator, coset_evaluator, theta: int,
                        value_leaves, coset_leaves, rng, transcript) -> "Permuted":
        """prover.rs:80-245.  value_evaluator: Lagrange-basis Evaluator holding the columns (`value_leaves`); coset_evaluator:
        the extended-basis one (`coset_leaves`); theta a canonical integer; rng(count) -> (count, 4) Montgomery limbs
        (blinding rows of A', of S', then the two blinds)."""
        import torch
        n, sf = params.n, domain.field
        usable = n - (blinding_factors + 1)

        def compress(expressions):                                                            # :112-182
            lagrange = Ast.constant(0)
            for e in expressions:
                lagrange = lagrange * theta + e(value_leaves)                                 # &(acc * theta) + expression, :170-173
            coset = Ast.constant(0)
            for e in expressions:
                coset = coset * Ast.constant(theta) + e(coset_leaves)                         # :176-179
            return coset, value_evaluator.evaluate(lagrange, domain)
        compressed_input_coset, compressed_input = compress(self.input_expressions)
        compressed_table_coset, compressed_table = compress(self.table_expressions)

        a, s = permute_expression_pair(compressed_input, compressed_table, usable, sf)       # :192-199
        permuted_input = torch.cat([a, _to_dev(rng(blinding_factors + 1), a)])                # :624-627
        permuted_table = torch.cat([s, _to_dev(rng(blinding_factors + 1), s)])

        def commit_values(values):                                                            # :202-207
            blind = Blind(np.ascontiguousarray(rng(1)[0]))
            commitment = _host(params.commit_lagrange(values, blind))
            return domain.lagrange_to_coeff(values.clone()), blind, commitment
        input_poly, input_blind, input_commitment = commit_values(permuted_input)
        table_poly, table_blind, table_commitment = commit_values(permuted_table)
        transcript.write_point(input_commitment)                                              # :218-222
        transcript.write_point(table_commitment)
        input_coset = coset_evaluator.register_poly(domain.coeff_to_extended(input_poly))     # :224-227
        table_coset = coset_evaluator.register_poly(domain.coeff_to_extended(table_poly))
        return Permuted(compressed_input, compressed_input_coset, permuted_input, input_poly, input_coset, input_blind,
                        compressed_table, compressed_table_coset, permuted_table, table_poly, table_coset, table_blind)


class Permuted:
    def __init__(self, compressed_input_expression, compressed_input_coset, permuted_input_expression, permuted_input_poly,
                 permuted_input_coset, permuted_input_blind, compressed_table_expression, compressed_table_coset,
                 permuted_table_expression, permuted_table_poly, permuted_table_coset, permuted_table_blind):
        self.compressed_input_expression, self.compressed_input_coset = compressed_input_expression, compressed_input_coset
        self.permuted_input_expression, self.permuted_input_poly = permuted_input_expression, permuted_input_poly
        self.permuted_input_coset, self.permuted_input_blind = permuted_input_coset, permuted_input_blind
        self.compressed_table_expression, self.compressed_table_coset = compressed_table_expression, compressed_table_coset
        self.permuted_table_expression, self.permuted_table_poly = permuted_table_expression, permuted_table_poly
        self.permuted_table_coset, self.permuted_table_blind = permuted_table_coset, permuted_table_blind

    def commit_product(self, params: Params, domain, blinding_factors: int, beta: int, gamma: int, evaluator, rng,
                       transcript) -> "Committed":
        """prover.rs:246-386.  evaluator: the extended-basis Evaluator the product coset is registered with."""
        n, sf = params.n, domain.field
        rows = Evaluator(LAGRANGE)
        a_p, s_p = rows.register_poly(self.permuted_input_expression), rows.register_poly(self.permuted_table_expression)
        a_c, s_c = rows.register_poly(self.compressed_input_expression), rows.register_poly(self.compressed_table_expression)
        A = Ast.of
        den = (A(a_p) + Ast.constant(beta)) * (A(s_p) + Ast.constant(gamma))                  # :266-275
        inv = rows.register_poly(batch_invert(rows.evaluate(den, domain), sf))                # :279
        frac = A(inv) * (A(a_c) + Ast.constant(beta)) * (A(s_c) + Ast.constant(gamma))        # :283-293
        z = grand_product(rows.evaluate(frac, domain), n, fields.scalar_limbs(1, sf, True), sf)      # :309-321
        if blinding_factors:
            z[n - blinding_factors:] = _to_dev(rng(blinding_factors), z)
        blind = Blind(np.ascontiguousarray(rng(1)[0]))                                        # :364
        commitment = _host(params.commit_lagrange(z, blind))                                  # :365
        poly = domain.lagrange_to_coeff(z)                                                    # :366 (in place)
        coset = evaluator.register_poly(domain.coeff_to_extended(poly))                       # :367
        transcript.write_point(commitment)                                                    # :370
        return Committed(self, poly, coset, blind)


class Committed:
    def __init__(self, permuted: Permuted, product_poly, product_coset, product_blind: Blind):
        self.permuted, self.product_poly, self.product_coset, self.product_blind = permuted, product_poly, product_coset, product_blind

    def construct(self, beta: int, gamma: int, l0, l_blind, l_last):
        """prover.rs:395-482 -> (Constructed, five expressions)."""
        p = self.permuted
        A = Ast.of
        active = Ast.one() - (A(l_last) + A(l_blind))
        b, g = Ast.constant(beta), Ast.constant(gamma)
        z = self.product_coset
        left = A(z.with_rotation(1)) * (A(p.permuted_input_coset) + b) * (A(p.permuted_table_coset) + g)
        right = A(z) * (p.compressed_input_coset + b) * (p.compressed_table_coset + g)
        exprs = [
            (Ast.one() - A(z)) * A(l0),                                                                        # :419-420
            (A(z) * A(z) - A(z)) * A(l_last),                                                                  # :421-425
            (left - right) * active,                                                                           # :430-445
            (A(p.permuted_input_coset) - A(p.permuted_table_coset)) * A(l0),                                   # :449-452
            (A(p.permuted_input_coset) - A(p.permuted_table_coset))
            * (A(p.permuted_input_coset) - A(p.permuted_input_coset.with_rotation(-1))) * active,              # :457-465
        ]
        return Constructed(p.permuted_input_poly, p.permuted_input_blind, p.permuted_table_poly, p.permuted_table_blind,
                           self.product_poly, self.product_blind), exprs


class Constructed:
    def __init__(self, permuted_input_poly, permuted_input_blind, permuted_table_poly, permuted_table_blind, product_poly, product_blind):
        self.permuted_input_poly, self.permuted_input_blind = permuted_input_poly, permuted_input_blind
        self.permuted_table_poly, self.permuted_table_blind = permuted_table_poly, permuted_table_blind
        self.product_poly, self.product_blind = product_poly, product_blind

    def evaluate(self, domain, x: int, transcript) -> "Evaluated":
        """prover.rs:485-515.  x: canonical integer."""
        sf = domain.field
        lim = lambda v: fields.scalar_limbs(v % domain.m, sf, True)
        x_inv, x_next = domain.rotate_omega(x, -1), domain.rotate_omega(x, 1)
        for poly, pt in ((self.product_poly, x), (self.product_poly, x_next), (self.permuted_input_poly, x),
                         (self.permuted_input_poly, x_inv), (self.permuted_table_poly, x)):
            write_evaluation(transcript, eval_polynomial(poly, lim(pt), sf))
        return Evaluated(self)


class Evaluated:
    def __init__(self, constructed: Constructed):
        self.constructed = constructed

    def open(self, domain, x: int) -> list[ProverQuery]:
        """prover.rs:518-555."""
        sf = domain.field
        lim = lambda v: fields.scalar_limbs(v % domain.m, sf, True)
        x_l, x_inv, x_next = lim(x), lim(domain.rotate_omega(x, -1)), lim(domain.rotate_omega(x, 1))
        c = self.constructed
        return [ProverQuery(x_l, c.product_poly, c.product_blind),
                ProverQuery(x_l, c.permuted_input_poly, c.permuted_input_blind),
                ProverQuery(x_l, c.permuted_table_poly, c.permuted_table_blind),
                ProverQuery(x_inv, c.permuted_input_poly, c.permuted_input_blind),
                ProverQuery(x_next, c.product_poly, c.product_blind)]
