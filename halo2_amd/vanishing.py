"""The vanishing argument's prover, `plonk::vanishing::Argument` (halo2_proofs/src/plonk/vanishing/prover.rs:37-190):
commit a random polynomial, build h(X) = (sum_i y^i gate_i(X)) / (X^n - 1) over the extended domain, commit its n-sized
pieces, fold them at x^n, and hand two queries to the multi-point opening.  Every polynomial stays in HBM: the gates are
`h2_evaluate_device` (halo2_amd/evaluator.py), the division `h2_divide_by_vanishing_poly_device`, the way back to
coefficients `h2_extended_to_coeff_device`, the piece commits one `h2_commit_batch_device`, the fold `h2_scale_add_device`.

torch is plumbing (device buffers, slicing); all arithmetic goes through the C ABI."""
from __future__ import annotations

import numpy as np

from . import fields
from .arithmetic import eval_polynomial, scale_add
from .commitment import Blind, Params
from .evaluator import Ast
from .multiopen import ProverQuery
from .transcript import write_evaluation


def _host(t) -> np.ndarray:
    return t.cpu().numpy().view(np.uint64)


class Argument:
    @staticmethod
    def commit(params: Params, domain, rng, transcript, device=None) -> "Committed":
        """prover.rs:38-61.  rng(count) -> (count, 4) Montgomery limbs: n draws for the polynomial, then one blind."""
        import torch
        dev = torch.device(device) if device else fields.current_device()
        random_poly = fields.to_device_limbs(rng(params.n), dev)
        random_blind = Blind(np.ascontiguousarray(rng(1)[0]))
        transcript.write_point(_host(params.commit(random_poly, random_blind)))
        return Committed(random_poly, random_blind)


class Committed:
    def __init__(self, random_poly, random_blind: Blind):
        self.random_poly, self.random_blind = random_poly, random_blind

    def construct(self, params: Params, domain, evaluator, expressions, y: int, rng, transcript) -> "Constructed":
        """prover.rs:65-123.  `evaluator`: a halo2_amd.evaluator.Evaluator over the extended basis; `expressions`: its Asts,
        folded with powers of y (highest first, :84); y a canonical integer."""
        n = params.n
        if isinstance(expressions, tuple):          # (Compiled, LateConstant): the tree was flattened before y existed (plonk.create_proof)
            compiled, y_slot = expressions
            h_ext = evaluator.run(compiled, domain, {y_slot: y})                                      # :84-85
        else:
            h_ext = evaluator.evaluate(Ast.distribute_powers(list(expressions), y), domain)          # :84-85
        h_coeff = domain.extended_to_coeff(domain.divide_by_vanishing_poly(h_ext))                    # :88-91
        h_pieces = [h_coeff[i * n:(i + 1) * n] for i in range(h_coeff.shape[0] // n)]                 # chunks_exact, :94-97
        h_blinds = [Blind(np.ascontiguousarray(b)) for b in rng(len(h_pieces))]                       # :99-102
        commitments = params.commit_batch(h_pieces, h_blinds)                                        # :105-112
        for c in _host(commitments):                                                                 # :115-117
            transcript.write_point(c)
        return Constructed(h_pieces, h_blinds, self)


class Constructed:
    def __init__(self, h_pieces, h_blinds, committed: Committed):
        self.h_pieces, self.h_blinds, self.committed = h_pieces, h_blinds, committed

    def evaluate(self, x, xn: int, domain, transcript) -> "Evaluated":
        """prover.rs:127-156.  x: (4,) Montgomery limbs; xn = x^n as a canonical integer."""
        sf = domain.field
        m = fields.MODULUS[sf]
        xn_l = fields.scalar_limbs(xn % m, sf, True)
        h_poly, h_blind = None, 0
        for piece, blind in zip(reversed(self.h_pieces), reversed(self.h_blinds)):                    # :134-144
            h_poly = piece.clone() if h_poly is None else scale_add(h_poly, xn_l, piece, sf)
            h_blind = (h_blind * xn + fields.from_limbs(blind.value.reshape(1, 4), sf, True)[0]) % m
        write_evaluation(transcript, eval_polynomial(self.committed.random_poly, x, sf))           # :146-147
        return Evaluated(h_poly, Blind(fields.scalar_limbs(h_blind, sf, True)), self.committed)


class Evaluated:
    def __init__(self, h_poly, h_blind: Blind, committed: Committed):
        self.h_poly, self.h_blind, self.committed = h_poly, h_blind, committed

    def open(self, x) -> list[ProverQuery]:
        """prover.rs:160-177."""
        return [ProverQuery(x, self.h_poly, self.h_blind),
                ProverQuery(x, self.committed.random_poly, self.committed.random_blind)]
