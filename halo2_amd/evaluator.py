"""`poly::Evaluator` / `Ast` (halo2_proofs/src/poly/evaluator.rs) over the C ABI: expression trees over registered
polynomials, evaluated on the device in one kernel per tree (`h2_evaluate_device`).

    ev = new_evaluator(EXTENDED)                    # poly::new_evaluator(|| {})            evaluator.rs:105
    a = ev.register_poly(d_a); b = ev.register_poly(d_b)                                   # :118-127
    h = ev.evaluate((Ast.of(a) * Ast.of(b.with_rotation(1)) + Ast.constant(c)) * x, domain)   # :129-228

Polynomials are torch CUDA tensors (n, 4) of Montgomery limbs; scalars are canonical Python integers."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import fields
from ._lib import check, lib
from .arithmetic import _p, _stream_ptr

COEFF, LAGRANGE, EXTENDED = 0, 1, 2                      # poly::{Coeff, LagrangeCoeff, ExtendedLagrangeCoeff}
_POLY, _CONST, _LINEAR, _ADD, _MUL, _SCALE, _MULADD = 1, 2, 3, 4, 5, 6, 7


class AstLeaf:
    """A registered polynomial at a rotation (evaluator.rs:38-79)."""

    def __init__(self, evaluator, index: int, rotation: int = 0):
        self.evaluator, self.index, self.rotation = evaluator, index, rotation

    def with_rotation(self, rotation: int) -> "AstLeaf":
        """A NEW rotation, not relative to the current one (evaluator.rs:72-78)."""
        return AstLeaf(self.evaluator, self.index, rotation)


class LateConstant:
    """The base of a `distribute_powers` whose value is not known yet: the quotient's expressions are folded with powers of the challenge y
    (plonk/vanishing/prover.rs:84), which exists only after a commitment has crossed PCIe and been hashed -- while the tree, its flattening into the
    evaluation kernel's program and the table of its other constants ask for nothing of the kind.  Evaluator.compile takes the tree with a
    LateConstant in y's place, Evaluator.run takes the value."""

    def __init__(self):
        self.index = None            # its slot in the compiled program's constant table


class Compiled:
    """A flattened expression tree: the program words and constant table h2_evaluate_device takes (Evaluator.compile)."""

    def __init__(self, prog, n_words, consts, n_consts):
        self.prog, self.n_words, self.consts, self.n_consts = prog, n_words, consts, n_consts


class Ast:
    """evaluator.rs:236-258.  kind in {poly, add, mul, scale, distribute, linear, constant}."""

    def __init__(self, kind, *args):
        self.kind, self.args = kind, args

    @staticmethod
    def of(leaf: AstLeaf) -> "Ast":                       # From<AstLeaf> (:288-292)
        return Ast("poly", leaf)

    @staticmethod
    def constant(scalar: int) -> "Ast":                   # Ast::ConstantTerm
        return Ast("constant", int(scalar))

    @staticmethod
    def linear(scalar: int) -> "Ast":                     # Ast::LinearTerm
        return Ast("linear", int(scalar))

    @staticmethod
    def one() -> "Ast":                                   # :294-298
        return Ast.constant(1)

    @staticmethod
    def distribute_powers(terms, base) -> "Ast":          # :260-264, terms from the highest power down
        """base: a canonical integer, or a LateConstant whose value arrives after the tree has been compiled (Evaluator.compile / run)."""
        return Ast("distribute", list(terms), base if isinstance(base, LateConstant) else int(base))

    def __add__(self, other):                             # :318-340
        return Ast("add", self, _as_ast(other))

    def __sub__(self, other):                             # :342-360: a + (-b)
        return Ast("add", self, -_as_ast(other))

    def __neg__(self):                                    # :300-316: Scale(-1)
        return Ast("scale", self, -1)

    def __mul__(self, other):                             # Mul<Ast> (Lagrange and extended bases, :370-418) or Mul<F> = Scale (:420-434)
        if isinstance(other, Ast) or isinstance(other, AstLeaf):
            return Ast("mul", self, _as_ast(other))
        return Ast("scale", self, int(other))

    # integers on the left (`1 - a`, `3 * a`): lowered gate expressions are written once and applied to Asts or to evaluations
    def __radd__(self, other):
        return _as_ast(other) + self

    def __rsub__(self, other):
        return _as_ast(other) - self

    def __rmul__(self, other):
        return self * other


def _as_ast(x) -> Ast:
    if isinstance(x, AstLeaf):
        return Ast.of(x)
    return x if isinstance(x, Ast) else Ast.constant(int(x))         # a bare integer is Ast::ConstantTerm


class Evaluator:
    def __init__(self, basis: int):
        if basis not in (COEFF, LAGRANGE, EXTENDED):
            raise ValueError("unknown basis")
        self.basis, self.polys = basis, []

    def register_poly(self, poly) -> AstLeaf:             # evaluator.rs:118-127
        if self.polys and poly.shape != self.polys[0].shape:
            raise ValueError("register_poly: all polynomials of an evaluator have one length")
        assert poly.is_cuda and poly.is_contiguous() and poly.ndim == 2 and poly.shape[1] == 4
        self.polys.append(poly)
        return AstLeaf(self, len(self.polys) - 1)

    @staticmethod
    def _need(ast: Ast) -> int:
        """Stack slots a subtree needs (Sethi-Ullman numbering): lets commutative nodes run their deeper side first."""
        k = ast.kind
        if k in ("poly", "constant", "linear"):
            return 1
        if k == "scale":
            return Evaluator._need(ast.args[0])
        if k in ("add", "mul"):
            a, b = Evaluator._need(ast.args[0]), Evaluator._need(ast.args[1])
            return max(a, b) if a != b else a + 1
        terms = ast.args[0]                                  # distribute: the accumulator stays below every later term
        return max([1] + [Evaluator._need(t) + (1 if j else 0) for j, t in enumerate(terms)])

    def _compile(self, ast: Ast, domain, words: list, consts: list):
        m = domain.m

        def const_index(v) -> int:
            if isinstance(v, LateConstant):            # filled in by run()
                consts.append(0)
                v.index = len(consts) - 1
                return v.index
            consts.append(int(v) % m)
            return len(consts) - 1
        k = ast.kind
        if k == "poly":
            leaf = ast.args[0]
            if leaf.evaluator is not self:
                raise ValueError("AstLeaf belongs to another evaluator")      # the reference enforces this with a type parameter
            if self.basis == COEFF and leaf.rotation != 0:
                raise ValueError("Can't rotate polynomials in the standard basis")          # evaluator.rs:519
            step = 1 if self.basis != EXTENDED else 1 << (domain.extended_k - domain.k)
            words += [_POLY | leaf.index << 8, (leaf.rotation * step) & 0xFFFFFFFF]
        elif k == "constant":
            words.append(_CONST | const_index(ast.args[0]) << 8)
        elif k == "linear":
            zeta = domain.g_coset if self.basis == EXTENDED else 1                          # F::ZETA, evaluator.rs:595
            words.append(_LINEAR | const_index(ast.args[0] * zeta) << 8)
        elif k in ("add", "mul"):
            if k == "mul" and self.basis == COEFF:
                # `Mul` is implemented for Ast<_, _, LagrangeCoeff> and Ast<_, _, ExtendedLagrangeCoeff> (evaluator.rs:370-418)
                raise ValueError("Ast multiplication exists for the Lagrange bases only")
            first, second = ast.args
            if self._need(second) > self._need(first):       # + and * commute: deeper side first keeps the stack shallow
                first, second = second, first
            self._compile(first, domain, words, consts)
            self._compile(second, domain, words, consts)
            words.append(_ADD if k == "add" else _MUL)
        elif k == "scale":
            self._compile(ast.args[0], domain, words, consts)
            words.append(_SCALE | const_index(ast.args[1]) << 8)
        elif k == "distribute":
            terms, base = ast.args
            if not terms:
                words.append(_CONST | const_index(0) << 8)
            else:
                # fold(0, |acc, term| acc * base + term): the first step is 0 * base + term
                self._compile(terms[0], domain, words, consts)
                b = const_index(base)
                for term in terms[1:]:
                    self._compile(term, domain, words, consts)
                    words.append(_MULADD | b << 8)
        else:
            raise ValueError(k)

    def evaluate(self, ast: Ast, domain):
        """evaluator.rs:129-228: a new device tensor of the basis' length."""
        import torch
        if not self.polys:
            raise ValueError("evaluate: no polynomial registered")                          # `.first().unwrap()`, :140
        n = self.polys[0].shape[0]
        want = domain.extended_len() if self.basis == EXTENDED else domain.n
        if n != want:
            raise ValueError("evaluate: polynomial length does not match the domain")
        return self.run(self.compile(ast, domain), domain)

    def compile(self, ast: Ast, domain) -> Compiled:
        """The host half of `evaluate`: the tree flattened into program words and a constant table (Montgomery limbs).  A LateConstant in the tree
        leaves its slot open for `run`."""
        words, consts = [], []
        self._compile(_as_ast(ast), domain, words, consts)
        prog = (C.c_uint32 * len(words))(*words)
        cst = fields.to_limbs(consts, domain.field, True) if consts else np.zeros((1, 4), dtype=np.uint64)
        return Compiled(prog, len(words), np.ascontiguousarray(cst), len(consts))

    def run(self, compiled: Compiled, domain, late=None):
        """The device half: one launch of the evaluation kernel over the registered polynomials.  late: {LateConstant: canonical integer}."""
        import torch
        n = self.polys[0].shape[0]
        want = domain.extended_len() if self.basis == EXTENDED else domain.n
        if n != want:
            raise ValueError("evaluate: polynomial length does not match the domain")
        for slot, value in (late or {}).items():
            if slot.index is None:
                # the tree that was compiled never mentioned this slot (an empty expression list folds to the constant 0): nothing to fill.
                # Writing through consts[None] would broadcast the value over the WHOLE constant table and turn that 0 into `value`.
                continue
            compiled.consts[slot.index] = fields.scalar_limbs(int(value) % domain.m, domain.field, True)
        log_len = n.bit_length() - 1
        ptrs = (C.c_void_p * len(self.polys))(*[p.data_ptr() for p in self.polys])
        omega = fields.scalar_limbs(domain.extended_omega if self.basis == EXTENDED else domain.omega, domain.field, True)
        out = torch.empty_like(self.polys[0])
        check(lib().h2_evaluate_device(domain.field, self.basis, compiled.prog, compiled.n_words, _p(compiled.consts), compiled.n_consts, ptrs, len(self.polys),
                                       log_len, _p(omega), out.data_ptr(), _stream_ptr()), "h2_evaluate_device")
        return out


def new_evaluator(basis: int) -> Evaluator:
    return Evaluator(basis)
