"""`plonk::create_proof` after synthesis (halo2_proofs/src/plonk/prover.rs:35-724) and the part of `keygen_pk` that
turns fixed columns and the copy-constraint mapping into the three bases (plonk/keygen.rs:296-380, permutation/keygen.rs:163-190),
for circuit instances whose columns are already assigned (one or several per proof).  This module is orchestration: it sequences the device
arguments (`halo2_amd.{permutation, lookup, vanishing, multiopen, opening}`), the column commits / iFFTs / coset FFTs and the
transcript exactly in the reference's order.  What it does not do is run a `Circuit` (floor planning, region assignment,
selector compression, `Expression<F>` construction): the caller hands over the constraint system in lowered form --

* a gate polynomial or a lookup expression is a callable `cells -> value` using `cells.fixed(col, rot)`, `cells.advice(col, rot)`,
  `cells.instance(col, rot)` and `+ - *` (integers are constants).  The prover applies it to `Ast` leaves, a verifier to
  evaluations -- the same role `Expression::evaluate` plays with its closures (prover.rs:553-580, verifier.rs:251-265);
* queries are (column, rotation) lists in the order the reference's `ConstraintSystem` would have recorded them.

`vk_repr` stands in for `VerifyingKey::transcript_repr` (plonk.rs:94-101; a hash of the Rust Debug print of the pinned
key, not reproducible without the reference's types).  torch is plumbing; all arithmetic goes through the C ABI."""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import fields
from . import lookup as lookup_arg
from . import multiopen, permutation
from . import vanishing as vanishing_arg
from .arithmetic import eval_polynomial
from .commitment import Blind, Params
from .evaluator import EXTENDED, LAGRANGE, Ast, Evaluator, LateConstant
from .multiopen import ProverQuery
from .transcript import DeferredScalars, write_evaluation


@dataclass
class ConstraintSystem:
    """The lowered plonk::ConstraintSystem (plonk/circuit.rs:955-1010)."""
    num_fixed_columns: int
    num_advice_columns: int
    num_instance_columns: int
    gates: list                       # callables cells -> value, one per gate polynomial (Gate::polynomials, flattened)
    advice_queries: list              # (column, rotation)
    instance_queries: list
    fixed_queries: list
    permutation_columns: list = field(default_factory=list)      # ("advice" | "fixed" | "instance", index), permutation::Argument
    lookups: list = field(default_factory=list)                  # (input expressions, table expressions): lists of callables
    degree: int = 3                   # ConstraintSystem::degree() (circuit.rs:1483-1531)
    blinding_factors: int = 5         # ConstraintSystem::blinding_factors() (circuit.rs:1535-1569)


class _Cells:
    """What a lowered expression sees: column queries as Ast leaves with rotations."""

    def __init__(self, fixed, advice, instance):
        self._f, self._a, self._i = fixed, advice, instance

    def fixed(self, col: int, rot: int = 0):
        return Ast.of(self._f[col].with_rotation(rot))

    def advice(self, col: int, rot: int = 0):
        return Ast.of(self._a[col].with_rotation(rot))

    def instance(self, col: int, rot: int = 0):
        return Ast.of(self._i[col].with_rotation(rot))


def _host(t) -> np.ndarray:
    return t.cpu().numpy().view(np.uint64)


def _up(col, sf, dev):
    """A column as an (n, 4) Montgomery CUDA tensor: integer lists are converted, CUDA tensors are copied."""
    import torch
    if type(col).__module__.startswith("torch"):
        return col.to(dev).clone()
    return torch.from_numpy(fields.to_limbs(col, sf, True).view(np.int64)).to(dev)


class ProvingKey:
    """plonk::ProvingKey (plonk.rs:118-130) with its VerifyingKey parts the prover needs."""

    def __init__(self, cs: ConstraintSystem, domain, vk_repr: int, fixed_values, fixed_polys, fixed_cosets, perm_values, perm_polys,
                 perm_cosets, l0, l_blind, l_last):
        self.cs, self.domain, self.vk_repr = cs, domain, vk_repr
        self.fixed_values, self.fixed_polys, self.fixed_cosets = fixed_values, fixed_polys, fixed_cosets
        self.perm_values, self.perm_polys, self.perm_cosets = perm_values, perm_polys, perm_cosets
        self.l0, self.l_blind, self.l_last = l0, l_blind, l_last


class _FingerprintCells:
    """Lowered expressions are callables, not trees, so they cannot be printed the way `PinnedConstraintSystem`'s Debug does
    (plonk.rs:81).  Their fingerprint is their value at points derived from the query's identity (Schwartz-Zippel: two
    different polynomials of degree <= 2^10 collide with probability ~2^-245)."""

    def __init__(self, m: int):
        self.m = m

    def _pt(self, kind: str, col: int, rot: int) -> int:
        import hashlib
        d = hashlib.blake2b(f"{kind}:{col}:{rot}".encode(), digest_size=64, person=b"Halo2-Vk-Queries").digest()
        return int.from_bytes(d, "little") % self.m

    def fixed(self, col: int, rot: int = 0):
        return self._pt("fixed", col, rot)

    def advice(self, col: int, rot: int = 0):
        return self._pt("advice", col, rot)

    def instance(self, col: int, rot: int = 0):
        return self._pt("instance", col, rot)

    def __getattr__(self, name):
        # an expression input this fingerprint does not know would silently stay out of the key's hash
        raise AttributeError(f"derive_vk_repr: lowered expressions may only query cells.fixed / advice / instance, not cells.{name}")


def derive_vk_repr(params: Params, cs: ConstraintSystem, domain, fixed_commitments, permutation_commitments) -> int:
    """Default `VerifyingKey::transcript_repr` (plonk.rs:75-98): Blake2b-512, personal "Halo2-Verify-Key", over a canonical
    serialisation of everything the reference's pinned key holds -- moduli, k, extended_k, omega, the constraint system's
    counts / query lists / permutation columns / degree / blinding factors, fingerprints of the gate and lookup expressions,
    and the fixed and permutation commitments -- so the Fiat-Shamir challenges are bound to the circuit.  For byte
    interoperability with a reference-produced proof pass the reference's own `transcript_repr` as `vk_repr` instead
    (tests/test_gpu_reference_goldens.py does, from the pinned key's Debug text)."""
    import hashlib
    bf_, sf = fields.CURVE_FIELDS[params.curve]
    m = fields.MODULUS[sf]
    cells = _FingerprintCells(m)
    fp = lambda e: int(e(cells)) % m
    doc = {
        "base_modulus": hex(fields.MODULUS[bf_]), "scalar_modulus": hex(m),
        "k": params.k, "extended_k": domain.extended_k, "omega": hex(domain.omega),
        "num_fixed_columns": cs.num_fixed_columns, "num_advice_columns": cs.num_advice_columns,
        "num_instance_columns": cs.num_instance_columns,
        "advice_queries": [list(q) for q in cs.advice_queries], "instance_queries": [list(q) for q in cs.instance_queries],
        "fixed_queries": [list(q) for q in cs.fixed_queries], "permutation": [list(c) for c in cs.permutation_columns],
        "degree": cs.degree, "blinding_factors": cs.blinding_factors,
        "gates": [hex(fp(g)) for g in cs.gates],
        "lookups": [[[hex(fp(e)) for e in ins], [hex(fp(e)) for e in tabs]] for ins, tabs in cs.lookups],
        "fixed_commitments": [None if c is None else [hex(c[0]), hex(c[1])] for c in fixed_commitments],
        "permutation_commitments": [None if c is None else [hex(c[0]), hex(c[1])] for c in permutation_commitments],
    }
    import json
    s = json.dumps(doc, sort_keys=True, separators=(",", ":")).encode()
    digest = hashlib.blake2b(len(s).to_bytes(8, "little") + s, digest_size=64, person=b"Halo2-Verify-Key").digest()
    return int.from_bytes(digest, "little") % m


def keygen_pk(params: Params, cs: ConstraintSystem, fixed_columns, mapping, vk_repr: int | None = None, device=None) -> ProvingKey:
    """keygen.rs:296-380 after `synthesize`.  fixed_columns: integer lists of n rows; mapping[c][r] = (c', r') the cell that
    follows (c, r) in its copy-constraint cycle (permutation/keygen.rs:24-107), identity where unconstrained.
    vk_repr=None (the default) derives the key's transcript representation from the key itself (`derive_vk_repr`): the same
    role as `VerifyingKey::from_parts` (plonk.rs:75-98), but NOT the same bytes -- the reference hashes the Debug text of its
    `PinnedVerificationKey`, this default hashes a JSON serialisation with expression fingerprints.  Proofs made with the default
    are therefore NOT byte-interoperable with a reference verifier: for interop pass the reference's own `transcript_repr`
    (tests/test_reference_goldens.py computes it from the pinned key's Debug text)."""
    import torch
    from .domain import EvaluationDomain
    dev = torch.device(device) if device else fields.current_device()
    sf = fields.CURVE_FIELDS[params.curve][1]
    n, k = params.n, params.k
    domain = EvaluationDomain(cs.degree, k, sf)
    m = domain.m

    def three(lagrange):
        coeff = domain.lagrange_to_coeff(lagrange.clone())
        return lagrange, coeff, domain.coeff_to_extended(coeff)
    fixed = [three(_up(col, sf, dev)) for col in fixed_columns]
    # sigma_c(omega^r) = delta^c' omega^r' (permutation/keygen.rs:163-190): the delta^c omega^r tables are Lagrange-basis linear terms
    n_perm = len(cs.permutation_columns)
    perms = []
    if n_perm:
        rows = Evaluator(LAGRANGE)
        rows.register_poly(torch.zeros((n, 4), dtype=torch.int64, device=dev))
        delta = fields.delta(sf)
        table = torch.cat([rows.evaluate(Ast.linear(pow(delta, c, m)), domain) for c in range(n_perm)])      # (n_perm * n, 4)
        for c in range(n_perm):
            if isinstance(mapping, np.ndarray):
                idx = torch.from_numpy(np.ascontiguousarray(mapping[c], dtype=np.int64)).to(dev)
            else:
                idx = torch.tensor([mapping[c][r][0] * n + mapping[c][r][1] for r in range(n)], dtype=torch.int64, device=dev)
            perms.append(three(table.index_select(0, idx).contiguous()))
    usable = n - (cs.blinding_factors + 1)
    one = _up([1], sf, dev)[0]

    def indicator(rows_):
        v = torch.zeros((n, 4), dtype=torch.int64, device=dev)
        for r in rows_:
            v[r] = one
        return three(v)[2]
    l0 = indicator([0])                                                     # keygen.rs:343-349
    l_blind = indicator(range(usable + 1, n))                               # :353-359
    l_last = indicator([usable])                                            # :363-368
    if vk_repr is None:
        from .verifier import _affine
        blind_one = Blind(field=sf)
        host = lambda t: t.cpu().numpy().view(np.uint64)
        commits = [_affine(params, host(params.commit_lagrange(t[0], blind_one))) for t in fixed + perms]
        vk_repr = derive_vk_repr(params, cs, domain, commits[:len(fixed)], commits[len(fixed):])
    return ProvingKey(cs, domain, vk_repr, [t[0] for t in fixed], [t[1] for t in fixed], [t[2] for t in fixed],
                      [t[0] for t in perms], [t[1] for t in perms], [t[2] for t in perms], l0, l_blind, l_last)


def create_proof(params: Params, pk: ProvingKey, advice_columns, instance_columns, rng, transcript, schedule: str | None = None) -> None:
    """prover.rs:35-724 for one circuit.  advice_columns: integer lists or (n, 4) CUDA tensors (n rows; the last
    blinding_factors + 1 are overwritten with randomness, :293-298); instance_columns: integer lists of at most
    n - (blinding_factors + 1) values (:84-86);
    rng(count) -> (count, 4) Montgomery limbs, drawn in the reference's order."""
    create_proof_many(params, pk, [(advice_columns, instance_columns)], rng, transcript, schedule=schedule)


def create_proof_many(params: Params, pk: ProvingKey, circuits, rng, transcript, schedule: str | None = None) -> None:
    """prover.rs:35-724: one proof for several instances of the same circuit (`circuits: &[ConcreteCircuit]`,
    `instances: &[&[&[C::Scalar]]]`).  circuits: [(advice_columns, instance_columns), ...] as `create_proof` takes them.  They
    share the verifying key, the challenges, one vanishing argument and one multi-point opening."""
    import torch
    cs, domain = pk.cs, pk.domain
    sf, m, n = domain.field, domain.m, params.n
    bf = cs.blinding_factors
    usable = n - (bf + 1)
    dev = pk.l0.device
    lim = lambda v: fields.scalar_limbs(v % m, sf, True)
    circuits = list(circuits)
    for advice_columns, instance_columns in circuits:
        if len(instance_columns) != cs.num_instance_columns or len(advice_columns) != cs.num_advice_columns:
            raise ValueError("InvalidInstances")                                          # :49-57

    transcript.common_scalar(lim(pk.vk_repr))                                             # vk.hash_into, :60

    def three(lagrange):
        coeff = domain.lagrange_to_coeff(lagrange.clone())
        return lagrange, coeff, domain.coeff_to_extended(coeff)

    # instance columns of every circuit: commit (blind 1), absorb, three bases (:77-130)
    inst_all = []
    for _, instance_columns in circuits:
        inst = []
        for values in instance_columns:
            if len(values) > usable:
                raise ValueError("InstanceTooLarge")                                      # :84-86
            lag = torch.zeros((n, 4), dtype=torch.int64, device=dev)                      # poly.resize(n, 0), :80-90
            if len(values):
                lag[:len(values)] = _up(values, sf, dev)
            transcript.common_point(_host(params.commit_lagrange(lag, Blind(field=sf))))  # :94-105
            inst.append(three(lag))
        inst_all.append(inst)

    # advice columns of every circuit: blinding rows, blinds, commitments (:293-341)
    adv_all, blinds_all = [], []
    for advice_columns, _ in circuits:
        adv_lag = [_up(col, sf, dev) for col in advice_columns]
        for lag in adv_lag:
            lag[usable:] = torch.from_numpy(np.ascontiguousarray(rng(bf + 1), dtype=np.uint64).view(np.int64)).to(dev)
        advice_blinds = [Blind(np.ascontiguousarray(rng(1)[0])) for _ in adv_lag]
        if adv_lag:
            for c in _host(params.commit_batch(adv_lag, advice_blinds, lagrange=True)):
                transcript.write_point(c)
        adv_all.append([three(lag) for lag in adv_lag])
        blinds_all.append(advice_blinds)

    # evaluators (:344-417)
    values, cosets = Evaluator(LAGRANGE), Evaluator(EXTENDED)
    fixed_v = [values.register_poly(t) for t in pk.fixed_values]
    fixed_c = [cosets.register_poly(t) for t in pk.fixed_cosets]
    cells_v, cells_c, advice_c_all, instance_c_all = [], [], [], []
    for adv, inst in zip(adv_all, inst_all):
        advice_v = [values.register_poly(t[0]) for t in adv]
        instance_v = [values.register_poly(t[0]) for t in inst]
        advice_c = [cosets.register_poly(t[2]) for t in adv]
        instance_c = [cosets.register_poly(t[2]) for t in inst]
        cells_v.append(_Cells(fixed_v, advice_v, instance_v))
        cells_c.append(_Cells(fixed_c, advice_c, instance_c))
        advice_c_all.append(advice_c)
        instance_c_all.append(instance_c)
    perm_c = [cosets.register_poly(t) for t in pk.perm_cosets]
    l0, l_blind, l_last = (cosets.register_poly(t) for t in (pk.l0, pk.l_blind, pk.l_last))

    theta = transcript.squeeze_challenge()                                                # :421
    lookups = [[lookup_arg.Argument(ins, tabs).commit_permuted(params, domain, bf, values, cosets, theta, cv, cc, rng, transcript)
                for ins, tabs in cs.lookups] for cv, cc in zip(cells_v, cells_c)]         # :423-454
    beta = transcript.squeeze_challenge()                                                 # :457
    gamma = transcript.squeeze_challenge()                                                # :460

    pkey = permutation.ProvingKey(pk.perm_values, pk.perm_polys, perm_c)
    perm_argument = permutation.Argument(len(cs.permutation_columns))
    perm_committed, perm_leaves_all = [], []
    for adv, inst, advice_c, instance_c in zip(adv_all, inst_all, advice_c_all, instance_c_all):      # :463-481
        by_kind = {"advice": (adv, advice_c), "fixed": (None, fixed_c), "instance": (inst, instance_c)}
        perm_lagrange, perm_leaves = [], []
        for kind, idx in cs.permutation_columns:
            perm_lagrange.append(pk.fixed_values[idx] if kind == "fixed" else by_kind[kind][0][idx][0])
            perm_leaves.append(by_kind[kind][1][idx])
        perm_leaves_all.append(perm_leaves)
        perm_committed.append(perm_argument.commit(params, domain, cs.degree, bf, pkey, perm_lagrange, beta, gamma, cosets, rng, transcript)
                              if cs.permutation_columns else permutation.Committed([]))
    lookups = [[p.commit_product(params, domain, bf, beta, gamma, cosets, rng, transcript) for p in ls] for ls in lookups]      # :483-502

    # the expression trees of the quotient (:511-586) ask for beta and gamma, not for y, and draw no randomness: built HERE, while the GPU still
    # works off the transforms the permutation / lookup products left queued, instead of behind the read-back of the next commitment
    perm_pairs = [pc.construct(domain, cs.degree, bf, pkey, leaves, l0, l_blind, l_last, beta, gamma)
                  for pc, leaves in zip(perm_committed, perm_leaves_all)]                 # :511-531
    lookup_pairs = [[p.construct(beta, gamma, l0, l_blind, l_last) for p in ls] for ls in lookups]             # :533-543
    expressions = []
    for cc, (_, perm_exprs), lps in zip(cells_c, perm_pairs, lookup_pairs):               # :545-586
        expressions += [g(cc) for g in cs.gates] + perm_exprs + [e for _, es in lps for e in es]
    expressions = [e if isinstance(e, Ast) else Ast.constant(int(e)) for e in expressions]
    # ... and flattened into the evaluation kernel's program with y's slot left open (evaluator.LateConstant): behind the commitment's read-back
    # only the challenge, one constant and the launch remain
    y_slot = LateConstant()
    expressions = (cosets.compile(Ast.distribute_powers(expressions, y_slot), domain), y_slot)

    vanishing = vanishing_arg.Argument.commit(params, domain, rng, transcript, device=dev)                      # :505
    y = transcript.squeeze_challenge()                                                    # :508
    vanishing = vanishing.construct(params, domain, cosets, expressions, y, rng, transcript)                    # :589-597

    x_l = transcript.squeeze_challenge_scalar()                                           # :598
    x = fields.from_limbs(x_l.reshape(1, 4), sf, True)[0]
    xn = pow(x, n, m)
    at = lambda rot: lim(domain.rotate_omega(x, rot))
    # every evaluation between x and the multi-point opening goes to the transcript with no challenge in between (:602-675): they are
    # enqueued one after the other and cross PCIe together (transcript.DeferredScalars) -- the same bytes, one synchronisation instead of ~27
    evals = DeferredScalars(transcript)
    for inst in inst_all:                                                                 # :602-619
        for col, rot in cs.instance_queries:
            write_evaluation(evals, eval_polynomial(inst[col][1], at(rot), sf))
    for adv in adv_all:                                                                   # :622-639
        for col, rot in cs.advice_queries:
            write_evaluation(evals, eval_polynomial(adv[col][1], at(rot), sf))
    for col, rot in cs.fixed_queries:                                                     # :642-653
        write_evaluation(evals, eval_polynomial(pk.fixed_polys[col], at(rot), sf))
    vanishing = vanishing.evaluate(x_l, xn, domain, evals)                                # :655
    pkey.evaluate(x_l, sf, evals)                                                         # :658
    perm_evaluated = [pc.evaluate(domain, bf, x, evals) for pc, _ in perm_pairs]          # :661-664
    lookups_evaluated = [[c.evaluate(domain, x, evals) for c, _ in lps] for lps in lookup_pairs]               # :667-675

    queries = []                                                                          # :677-722
    for inst, adv, advice_blinds, pe, les in zip(inst_all, adv_all, blinds_all, perm_evaluated, lookups_evaluated):
        queries += [ProverQuery(at(rot), inst[col][1], Blind(field=sf)) for col, rot in cs.instance_queries]
        queries += [ProverQuery(at(rot), adv[col][1], advice_blinds[col]) for col, rot in cs.advice_queries]
        queries += pe.open(domain, bf, x)
        for ev in les:
            queries += ev.open(domain, x)
    queries += [ProverQuery(at(rot), pk.fixed_polys[col], Blind(field=sf)) for col, rot in cs.fixed_queries]
    queries += pkey.open(x_l, sf)
    queries += vanishing.open(x_l)
    # (still through `evals`: the multi-point opening groups its queries on the host before it asks for its first challenge, which flushes)
    multiopen.create_proof(params, rng, evals, queries, schedule=schedule)                # :724
    evals.flush()
