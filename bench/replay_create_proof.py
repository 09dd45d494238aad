#!/usr/bin/env python3
"""Replay of the MSM / FFT call trace of one `create_proof` (halo2_proofs/src/plonk/prover.rs:35-725) on
the MI355X path, with seeded synthetic columns -- BASELINE.json configs[3] (simple-example at k = 20) and
configs[0] (benches/plonk.rs at k = 8).  The Rust prover itself cannot run here (no toolchain); SURVEY.md
section 3.2 derives the op sequence and sizes from the circuit shape:

    per Lagrange column (instance, advice, permutation z):  commit_lagrange, lagrange_to_coeff, coeff_to_extended
    vanishing: commit(random poly); extended_to_coeff on h(X); one commit per h piece
    multiopen: commit(q'); IPA: commit(s_poly), then k rounds of two half-size MSMs over the folded generators

What is NOT replayed (out of scope): witness synthesis, the gate evaluator between the FFTs,
the generator collapse of the IPA rounds (so the round MSMs run over stand-in bases of the right size), the
transcript.  Usage:
    python bench/replay_create_proof.py --config simple-example --k 20          # host-pointer entry points (PCIe included)
    python bench/replay_create_proof.py --config plonk-bench --k 8 --check     # every output vs the oracle
    python bench/replay_create_proof.py --config simple-example --k 20 --resident
        # columns resident in HBM: batched commits, device transforms, and the REAL opening argument
        # (halo2_amd/opening.py: collapse, folds and transcript included) instead of stand-in round MSMs
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# circuit shape -> trace parameters
CONFIGS = {
    # examples/simple-example.rs: 1 instance + 2 advice + 4 permutation z columns, cs_degree 3, 2 h pieces
    "simple-example": dict(cs_degree=3, lagrange_columns=7, h_pieces=2),
    # benches/plonk.rs: 3 advice + 1 permutation z, cs_degree 5, 4 h pieces
    "plonk-bench": dict(cs_degree=5, lagrange_columns=4, h_pieces=4),
}


def run_trace(config: str, k: int, check: bool, curve: int = 1):
    import halo2_amd as h
    from halo2_amd import fields
    from oracle import c_oracle as co
    from oracle import pasta as o

    cfg = CONFIGS[config]
    n = 1 << k
    sf = co.field_of_curve(curve, "scalar")     # every proof in the reference runs on Vesta: scalars in Fp
    bm, sm = o.CURVES[curve]
    dom = h.EvaluationDomain(cfg["cs_degree"], k, sf)
    ref = o.EvaluationDomain(cfg["cs_degree"], k, sm)
    mont = lambda v: fields.scalar_limbs(v, sf, True)

    # Params from seeded generators (Params::new's hash-to-curve is upstream of the hot path)
    g = co.generate_bases(curve, 101, n)
    g_lagrange = co.generate_bases(curve, 102, n)
    w = co.generate_bases(curve, 103, 1)[0]
    u = co.generate_bases(curve, 104, 1)[0]
    t0 = time.perf_counter()
    params = h.Params.from_generators(curve, k, g, g_lagrange, w, u)
    setup_s = time.perf_counter() - t0

    cols = [co.random_field(sf, 1000 + i, n) for i in range(cfg["lagrange_columns"])]
    blinds = [h.Blind(co.random_field(sf, 2000 + i, 1)[0]) for i in range(cfg["lagrange_columns"] + cfg["h_pieces"] + 3)]
    random_poly = co.random_field(sf, 3000, n)
    h_ext = co.random_field(sf, 3001, dom.extended_len())           # stand-in for the quotient evaluations
    q_poly, s_poly = co.random_field(sf, 3002, n), co.random_field(sf, 3003, n)
    ipa = []
    half = n >> 1
    rnd = 0
    while half >= 1:
        ipa.append((co.random_field(sf, 4000 + rnd, half), co.random_field(sf, 4100 + rnd, half), co.generate_bases(curve, 4200 + rnd, half)))
        half >>= 1
        rnd += 1

    outputs, mismatches = [], []

    def record(name, got, want_fn):
        outputs.append(name)
        if check:
            want = want_fn()
            ok = (co.jac_to_affine_ints(curve, got) == co.jac_to_affine_ints(curve, want)) if got.shape[0] == 12 else np.array_equal(got, want)
            if not ok:
                mismatches.append(name)

    import torch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bi = 0
    for i, col in enumerate(cols):                                   # prover.rs:95,308; permutation/prover.rs:172-178
        c = params.commit_lagrange(col, blinds[bi])
        record(f"commit_lagrange[{i}]", c, lambda col=col, b=blinds[bi]: co.commit(curve, g_lagrange, w, col, b.value))
        bi += 1
        coeff = dom.lagrange_to_coeff(col.copy())
        record(f"lagrange_to_coeff[{i}]", coeff, lambda col=col: co.ifft(sf, col, mont(ref.omega_inv), k, mont(ref.ifft_divisor)))
        ext = dom.coeff_to_extended(coeff)
        record(f"coeff_to_extended[{i}]", ext, lambda coeff=coeff: co.coeff_to_extended(
            sf, coeff, k, ref.extended_k, mont(ref.g_coset), mont(ref.g_coset_inv), mont(ref.extended_omega)))
    c = params.commit(random_poly, blinds[bi])                       # vanishing/prover.rs:53
    record("commit(random_poly)", c, lambda b=blinds[bi]: co.commit(curve, g, w, random_poly, b.value))
    bi += 1
    hq = dom.extended_to_coeff(h_ext.copy())                          # vanishing/prover.rs:88
    record("extended_to_coeff(h)", hq, lambda: co.extended_to_coeff(
        sf, h_ext, ref.extended_k, mont(ref.g_coset), mont(ref.g_coset_inv), mont(ref.extended_omega_inv),
        mont(ref.extended_ifft_divisor))[: n * dom.quotient_poly_degree])
    for piece in range(cfg["h_pieces"]):                             # vanishing/prover.rs:105
        poly = np.ascontiguousarray(hq[piece * n:(piece + 1) * n]) if (piece + 1) * n <= hq.shape[0] else random_poly
        c = params.commit(poly, blinds[bi])
        record(f"commit(h_piece[{piece}])", c, lambda poly=poly, b=blinds[bi]: co.commit(curve, g, w, poly, b.value))
        bi += 1
    for name, poly in (("q'", q_poly), ("s_poly", s_poly)):          # multiopen/prover.rs:97, commitment/prover.rs:57
        c = params.commit(poly, blinds[bi])
        record(f"commit({name})", c, lambda poly=poly, b=blinds[bi]: co.commit(curve, g, w, poly, b.value))
        bi += 1
    for j, (a_hi, a_lo, bases) in enumerate(ipa):                    # commitment/prover.rs:107-108
        for tag, sc in (("l", a_hi), ("r", a_lo)):
            c = h.best_multiexp(sc, bases, curve)
            record(f"ipa_round[{j}].{tag}", c, lambda sc=sc, bases=bases: co.best_multiexp(curve, sc, bases))
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    params.close()
    return dict(config=config, k=k, ops=len(outputs), mismatches=mismatches, gpu_trace_s=None if check else round(elapsed, 4),
                params_setup_s=round(setup_s, 3), msm_full=cfg["lagrange_columns"] + cfg["h_pieces"] + 3, ipa_msm=2 * len(ipa),
                ifft_n=cfg["lagrange_columns"], coset_fft=cfg["lagrange_columns"], ifft_ext=1, extended_k=dom.extended_k,
                note="host-pointer entry points: every call includes its PCIe transfers; check=True also runs the oracle inline")


def run_resident(config: str, k: int, curve: int = 1):
    """The same trace with every column resident in HBM (what a device-aware prover would do): phase commits through
    `Params.commit_batch`, transforms on device tensors, then the real multi-point opening (`halo2_amd.multiopen.create_proof`:
    q' construction, its commit, the q evaluations) ending in the real opening argument (`halo2_amd.opening.create_proof`)."""
    import torch
    import halo2_amd as h
    from halo2_amd import fields
    from halo2_amd.multiopen import ProverQuery, create_proof
    from halo2_amd.transcript import Blake2bWrite
    from oracle import c_oracle as co          # input generation only

    cfg = CONFIGS[config]
    n = 1 << k
    sf = co.field_of_curve(curve, "scalar")
    dev = torch.device("cuda:0")
    dom = h.EvaluationDomain(cfg["cs_degree"], k, sf)
    g = co.generate_bases(curve, 101, n)
    g_lagrange = co.generate_bases(curve, 102, n)
    w, u = co.generate_bases(curve, 103, 1)[0], co.generate_bases(curve, 104, 1)[0]
    params = h.Params.from_generators(curve, k, g, g_lagrange, w, u)
    up = lambda a: torch.from_numpy(a.view(np.int64)).to(dev)
    ncol = cfg["lagrange_columns"]
    d_cols = [up(co.random_field(sf, 1000 + i, n)) for i in range(ncol)]
    blinds = [h.Blind(co.random_field(sf, 2000 + i, 1)[0]) for i in range(ncol + cfg["h_pieces"] + 3)]
    d_random = up(co.random_field(sf, 3000, n))
    d_h_ext = up(co.random_field(sf, 3001, dom.extended_len()))
    pool = co.random_field(sf, 3003, n + 64)
    pos = [0]

    def rng(count):
        if count == n:
            return pool[:n]
        pos[0] = (pos[0] + count) % 32
        return pool[n + pos[0]: n + pos[0] + count]

    def once():
        t = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        params.commit_batch(d_cols, blinds[:ncol], lagrange=True)                        # prover.rs:95,308; permutation z
        coeffs = dom.lagrange_to_coeff_batch([c.clone() for c in d_cols])
        exts = [dom.coeff_to_extended(c) for c in coeffs]
        torch.cuda.synchronize()
        t["columns: commit_lagrange + iFFT + coset FFT"] = time.perf_counter() - t0
        t1 = time.perf_counter()
        params.commit(d_random, blinds[ncol])                                            # vanishing/prover.rs:53
        hq = dom.extended_to_coeff(dom.divide_by_vanishing_poly(d_h_ext.clone()))        # vanishing/prover.rs:85-88
        pieces = [hq[i * n:(i + 1) * n].contiguous() if (i + 1) * n <= hq.shape[0] else d_random for i in range(cfg["h_pieces"])]
        params.commit_batch(pieces, blinds[ncol + 1: ncol + 1 + cfg["h_pieces"]])         # vanishing/prover.rs:105
        torch.cuda.synchronize()
        t["vanishing: random_poly + h-piece commits, quotient iFFT"] = time.perf_counter() - t1
        t2 = time.perf_counter()
        tr = Blake2bWrite(curve)
        x, xw, xwi = (tr.squeeze_challenge_scalar() for _ in range(3))                   # stand-ins for x, omega x, omega^-1 x
        # the query shape of plonk/prover.rs:664-722: every column at x, two also at omega x, one of them at omega^-1 x as well;
        # the h pieces and random_poly at x
        queries = [ProverQuery(x, c, blinds[i]) for i, c in enumerate(coeffs)]
        queries += [ProverQuery(xw, coeffs[0], blinds[0]), ProverQuery(xw, coeffs[1], blinds[1]), ProverQuery(xwi, coeffs[1], blinds[1])]
        queries += [ProverQuery(x, pc, blinds[ncol + 1 + i]) for i, pc in enumerate(pieces) if pc is not d_random]
        queries += [ProverQuery(x, d_random, blinds[ncol])]
        create_proof(params, rng, tr, queries)                                           # multiopen/prover.rs:21-125
        torch.cuda.synchronize()
        t["multiopen (q', its commit, evaluations) + opening argument"] = time.perf_counter() - t2
        t["total"] = time.perf_counter() - t0
        del exts
        return t
    once()
    best = min((once() for _ in range(3)), key=lambda t: t["total"])
    params.close()
    return dict(config=config, k=k, mode="resident", seconds={k_: round(v, 4) for k_, v in best.items()},
                msm_full=ncol + cfg["h_pieces"] + 3, ifft_n=ncol, coset_fft=ncol, ifft_ext=1, extended_k=dom.extended_k,
                note="columns resident in HBM; ends with the real multi-point opening and opening argument (k rounds: two multiexps, "
                     "two inner products, folds, Blake2b transcript on the host; L_j / R_j over the original registered generators)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", choices=sorted(CONFIGS), default="simple-example")
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--check", action="store_true", help="compare every output with the oracle (small k)")
    ap.add_argument("--resident", action="store_true", help="columns resident in HBM, batched commits, real opening argument")
    a = ap.parse_args()
    print(json.dumps(run_resident(a.config, a.k) if a.resident else run_trace(a.config, a.k, a.check)))
