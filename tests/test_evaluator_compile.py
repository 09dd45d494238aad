"""CPU-only: the host half of `poly::Evaluator` (halo2_amd/evaluator.py) -- the compiler from `Ast` trees to the post-order
bytecode `h2_evaluate_device` runs.  A reference interpreter of that bytecode on Python integers (the documented semantics of
the H2_EV_* opcodes, include/halo2_mi355x.h) is run on the compiled programs of random trees and compared with the oracle's
tree-walking restatement of `Evaluator::evaluate` (oracle/evaluator.py); the stack depth the compiler promises is checked on
the way.  The kernel itself is compared with the same oracle in the GPU suite."""
import random
from types import SimpleNamespace

import pytest

import halo2_amd as h
from halo2_amd import evaluator as hev
from halo2_amd.evaluator import COEFF, EXTENDED, LAGRANGE, Ast, AstLeaf, Evaluator
from oracle import evaluator as oev


def _interpret(words, consts, polys, basis, n, m, omega):
    """The bytecode's semantics on integers; returns (values, maximum stack depth)."""
    stack, depth, i = [], 0, 0
    while i < len(words):
        op, arg = words[i] & 0xFF, words[i] >> 8
        i += 1
        if op == hev._POLY:
            shift = words[i] if words[i] < (1 << 31) else words[i] - (1 << 32)
            i += 1
            stack.append([polys[arg][(r + shift) % n] for r in range(n)])
        elif op == hev._CONST:
            stack.append([consts[arg] if (basis != COEFF or r == 0) else 0 for r in range(n)])
        elif op == hev._LINEAR:
            if basis == COEFF:
                stack.append([consts[arg] if r == 1 else 0 for r in range(n)])
            else:
                stack.append([consts[arg] * pow(omega, r, m) % m for r in range(n)])
        elif op in (hev._ADD, hev._MUL):
            b, a = stack.pop(), stack.pop()
            stack.append([(x + y) % m if op == hev._ADD else x * y % m for x, y in zip(a, b)])
        elif op == hev._SCALE:
            stack.append([x * consts[arg] % m for x in stack.pop()])
        elif op == hev._MULADD:
            term, acc = stack.pop(), stack.pop()
            stack.append([(x * consts[arg] + y) % m for x, y in zip(acc, term)])
        else:
            raise AssertionError(f"unknown opcode {op}")
        depth = max(depth, len(stack))
    assert len(stack) == 1
    return stack[0], depth


def _random_tree(rnd, ev, n_polys, basis, m, depth):
    r = rnd.randrange(100)
    if depth == 0 or r < 20:
        pick = rnd.randrange(10)
        if pick < 7:
            i = rnd.randrange(n_polys)
            rot = 0 if basis == COEFF else rnd.randrange(-3, 4)
            return Ast.of(AstLeaf(ev, i).with_rotation(rot)), ("poly", i, rot)
        s = rnd.randrange(m)
        return (Ast.constant(s), ("constant", s)) if pick < 9 else (Ast.linear(s), ("linear", s))
    a, ta = _random_tree(rnd, ev, n_polys, basis, m, depth - 1)
    if r < 45:
        b, tb = _random_tree(rnd, ev, n_polys, basis, m, depth - 1)
        return a + b, ("add", ta, tb)
    if r < 60 and basis != COEFF:
        b, tb = _random_tree(rnd, ev, n_polys, basis, m, depth - 1)
        return a * b, ("mul", ta, tb)
    if r < 72:
        s = rnd.randrange(m)
        return a * s, ("scale", ta, s)
    if r < 80:
        b, tb = _random_tree(rnd, ev, n_polys, basis, m, depth - 1)
        return a - b, ("add", ta, ("scale", tb, m - 1))
    if r < 86:
        s = rnd.randrange(1000)
        return s - a, ("add", ("constant", s), ("scale", ta, m - 1))        # integers on the left
    terms = [(a, ta)] + [_random_tree(rnd, ev, n_polys, basis, m, depth - 1) for _ in range(rnd.randrange(0, 3))]
    base = rnd.randrange(m)
    return Ast.distribute_powers([t[0] for t in terms], base), ("distribute", [t[1] for t in terms], base)


@pytest.mark.parametrize("basis", [COEFF, LAGRANGE, EXTENDED])
@pytest.mark.parametrize("field", [h.FP, h.FQ])
def test_compiled_programs_mean_what_the_trees_mean(field, basis):
    k = 3
    dom = h.EvaluationDomain(4, k, field)
    m = dom.m
    n = dom.extended_len() if basis == EXTENDED else dom.n
    rnd = random.Random(100 * basis + field)
    polys = [[rnd.randrange(m) for _ in range(n)] for _ in range(3)]
    ev = Evaluator(basis)
    ev.polys = [SimpleNamespace(shape=(n, 4)) for _ in polys]          # the compiler only needs to know they exist
    omega = dom.extended_omega if basis == EXTENDED else dom.omega
    for _ in range(40):
        ast, tree = _random_tree(rnd, ev, len(polys), basis, m, 4)
        words, consts = [], []
        ev._compile(hev._as_ast(ast), dom, words, consts)
        got, depth = _interpret(words, consts, polys, basis, n, m, omega)
        want = oev.evaluate(tree, polys, basis, m, dom.k, dom.extended_k, dom.omega, dom.extended_omega, dom.g_coset)
        assert got == want, tree
        assert depth <= Evaluator._need(hev._as_ast(ast))          # the slots the Sethi-Ullman numbering promises suffice


def test_chains_compile_to_shallow_stacks_and_bad_trees_are_refused():
    dom = h.EvaluationDomain(3, 4, h.FP)
    ev = Evaluator(EXTENDED)
    ev.polys = [SimpleNamespace(shape=(dom.extended_len(), 4))]
    leaf = AstLeaf(ev, 0)
    for lean in ("left", "right"):
        chain = Ast.of(leaf)
        for _ in range(60):
            chain = chain + Ast.of(leaf) if lean == "left" else Ast.of(leaf) + chain
        words, consts = [], []
        ev._compile(chain, dom, words, consts)
        _, depth = _interpret(words, consts, [[1] * dom.extended_len()], EXTENDED, dom.extended_len(), dom.m, dom.extended_omega)
        assert depth == 2
    other = Evaluator(EXTENDED)
    other.polys = [SimpleNamespace(shape=(dom.extended_len(), 4))]
    with pytest.raises(ValueError):
        ev._compile(Ast.of(AstLeaf(other, 0)), dom, [], [])              # a leaf of another evaluator
    coeff = Evaluator(COEFF)
    coeff.polys = [SimpleNamespace(shape=(dom.n, 4))]
    with pytest.raises(ValueError):
        coeff._compile(Ast.of(AstLeaf(coeff, 0).with_rotation(1)), dom, [], [])      # evaluator.rs:519
    with pytest.raises(ValueError):
        coeff._compile(Ast.of(AstLeaf(coeff, 0)) * Ast.of(AstLeaf(coeff, 0)), dom, [], [])
