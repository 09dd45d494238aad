"""The multi-point opening argument, `poly::multiopen::create_proof`
(halo2_proofs/src/poly/multiopen/prover.rs:21-125) with `construct_intermediate_sets` (poly/multiopen.rs:152-276): the
caller of `commitment::create_proof` inside `plonk::create_proof` (plonk/prover.rs:722).  Every polynomial stays in HBM:
the x_1 / x_2 / x_4 combinations are `h2_scale_add_device`, the divisions by (X - point) `h2_kate_division_device`, the
q evaluations `h2_eval_polynomial_device`, the q' commitment a registered commit, and the final opening is
`halo2_amd.opening.create_proof`.  The query bookkeeping (which polynomial is opened at which set of points) is host logic.

torch is plumbing (device buffers); all arithmetic goes through the C ABI."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import fields
from .arithmetic import eval_polynomial, kate_division, scale_add
from .commitment import Blind, Params
from .opening import create_proof as commitment_create_proof
from .transcript import DeferredScalars, write_evaluation


@dataclass
class ProverQuery:
    """multiopen.rs:43-51.  point: (4,) Montgomery limbs; poly: (n, 4) CUDA tensor of coefficients; blind: Blind."""
    point: np.ndarray
    poly: object
    blind: Blind


def construct_intermediate_sets(queries):
    """multiopen.rs:152-276 for (point key, commitment key, eval) triples; eval is None on the prover side.

    Returns (commitment_data, point_sets): commitment_data lists, per distinct commitment in order of first appearance,
    {"commitment", "set_index", "point_indices", "evals"}; point_sets[set_index] lists the point keys of that set in order of
    first appearance of the points.  None when a (commitment, point) pair occurs twice (:243-249)."""
    point_index = {}
    data = {}
    for point, commitment, _ in queries:
        idx = point_index.setdefault(point, len(point_index))
        data.setdefault(commitment, {"commitment": commitment, "set_index": 0, "point_indices": [], "evals": None})
        data[commitment]["point_indices"].append(idx)
    point_of = {idx: point for point, idx in point_index.items()}
    set_index = {}                                        # sorted tuple of point indices -> order of first appearance
    for d in data.values():
        d["set"] = tuple(sorted(set(d["point_indices"])))
        d["set_index"] = set_index.setdefault(d["set"], len(set_index))
        d["evals"] = [None] * len(d["set"])
    for point, commitment, ev in queries:
        d = data[commitment]
        pos = d["set"].index(point_index[point])
        if d["evals"][pos] is not None:
            return None
        d["evals"][pos] = () if ev is None else ev
    point_sets = [None] * len(set_index)
    for s, i in set_index.items():
        point_sets[i] = [point_of[j] for j in s]
    out = [{k: v for k, v in d.items() if k != "set"} for d in data.values()]
    return out, point_sets


def create_proof(params: Params, rng, transcript, queries, schedule: str | None = None) -> None:
    """Writes the multi-opening proof for `queries` (an iterable of ProverQuery) to `transcript`.

    rng(count) -> (count, 4) uniformly random scalars, Montgomery limbs: one draw for q' 's blind (prover.rs:99), then
    whatever `commitment::create_proof` draws.  Polynomials are told apart by object identity, as the reference's
    PolynomialPointer does (prover.rs:128-140).  Raises ValueError where the reference returns
    io::ErrorKind::InvalidInput (a polynomial queried twice at one point, :41-46)."""
    import torch
    queries = list(queries)
    sf = fields.CURVE_FIELDS[params.curve][1]
    m = fields.MODULUS[sf]
    n = params.n
    as_int = lambda limbs: fields.from_limbs(np.ascontiguousarray(limbs, dtype=np.uint64).reshape(1, 4), sf, True)[0]
    as_limbs = lambda v: fields.scalar_limbs(v % m, sf, True)
    host = lambda t: t.cpu().numpy().view(np.uint64)

    # (the grouping of the queries is host work that asks the transcript nothing: done BEFORE the challenges, so that a transcript that still has
    # evaluations queued in HBM -- transcript.DeferredScalars, plonk.create_proof -- is flushed after it, while the GPU finishes them)
    by_id = {id(q.poly): q for q in queries}
    for q in queries:
        if q.poly.shape[0] != n:
            raise ValueError("multiopen: polynomial length != params.n")
    sets = construct_intermediate_sets([(as_int(q.point), id(q.poly), None) for q in queries])
    if sets is None:
        raise ValueError("queries iterator contains mismatching evaluations")             # prover.rs:41-46
    poly_map, point_sets = sets
    x_1 = transcript.squeeze_challenge_scalar()                                           # prover.rs:38-39
    x_2 = transcript.squeeze_challenge_scalar()

    # openings at the same point set collapse into one polynomial with x_1 (prover.rs:50-72)
    x1_i = as_int(x_1)
    q_polys = [None] * len(point_sets)
    q_blinds = [0] * len(point_sets)
    for d in poly_map:
        q = by_id[d["commitment"]]
        si = d["set_index"]
        q_polys[si] = q.poly.clone() if q_polys[si] is None else scale_add(q_polys[si], x_1, q.poly, sf)
        q_blinds[si] = (q_blinds[si] * x1_i + as_int(q.blind.value)) % m

    # q'(X) = sum over sets of x_2^.. * q_set(X) / prod (X - point) (prover.rs:75-97)
    q_prime = None
    for points, poly in zip(point_sets, q_polys):
        cur = poly
        for pt in points:
            cur = kate_division(cur, as_limbs(pt), sf)
        padded = torch.zeros((n, 4), dtype=poly.dtype, device=poly.device)
        padded[:cur.shape[0]] = cur
        q_prime = padded if q_prime is None else scale_add(q_prime, x_2, padded, sf)

    q_prime_blind = Blind(np.ascontiguousarray(rng(1)[0]))                                # prover.rs:99-102
    transcript.write_point(host(params.commit(q_prime, q_prime_blind)))
    x_3 = transcript.squeeze_challenge_scalar()                                           # prover.rs:104
    evals = transcript if getattr(transcript, "defers", False) else DeferredScalars(transcript)      # the evaluations cross PCIe together (transcript.py)
    for q in q_polys:                                                                     # prover.rs:108-110
        write_evaluation(evals, eval_polynomial(q, x_3, sf))
    evals.flush()
    x_4 = transcript.squeeze_challenge_scalar()                                           # prover.rs:112

    p_poly, p_blind = q_prime, as_int(q_prime_blind.value)                                # prover.rs:114-122
    x4_i = as_int(x_4)
    for q, b in zip(q_polys, q_blinds):
        p_poly = scale_add(p_poly, x_4, q, sf)
        p_blind = (p_blind * x4_i + b) % m
    commitment_create_proof(params, rng, transcript, p_poly, Blind(as_limbs(p_blind)), x_3, schedule=schedule)   # :124
