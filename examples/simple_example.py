#!/usr/bin/env python3
"""The reference's `examples/simple-example.rs` (prove knowledge of a, b with constant * a^2 * b^2 = c for a public c), proved and
verified on an MI355X with halo2_amd -- as a REAL proof (the reference's example stops at MockProver).

What the Rust example gets from `Circuit::configure` / `synthesize` and the floor planner is written out here in the lowered form
`halo2_amd.plonk` takes: the column layout `SimpleFloorPlanner` produces for FieldChip, the gate as a callable, the copy
constraints as cycles.

Parameters: `Params::new` derives its generators with pasta_curves' hash-to-curve, which this library does not have; pass
--params <file written by the reference's Params::write> for real ones.  Without it the example uses multiples of the curve
generator (-1, 2) -- fine for a demonstration, NOT binding (their discrete logarithms are known).

    python examples/simple_example.py [--k 4] [--params params.bin]
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def toy_params(h, curve, k):
    """g_i = [i + 1] G, w = [n + 1] G, u = [n + 2] G for G = (-1, 2) (poly/commitment/msm.rs:181 uses the same point)."""
    from halo2_amd import fields
    bf, sf = fields.CURVE_FIELDS[curve]
    gen = fields.to_limbs([fields.MODULUS[bf] - 1, 2], bf, True).reshape(1, 8)
    pts = [h.best_multiexp(fields.to_limbs([i + 1], sf, True), gen, curve, affine=True) for i in range((1 << k) + 2)]
    return h.Params.from_generators(curve, k, np.stack(pts[:1 << k]), None, pts[-2], pts[-1])


def build(m, n, a, b, constant):
    """The cells FieldChip assigns under SimpleFloorPlanner (simple-example.rs:269-301), as columns of n rows.
    advice: [a0, a1]; fixed: [constants, s_mul]; instance: [c]."""
    ab = a * b % m
    absq = ab * ab % m
    c = constant * absq % m
    a0, a1 = [0] * n, [0] * n
    constants, s_mul = [0] * n, [0] * n
    a0[0], a0[1], a0[2] = a, b, constant                 # load_private a, load_private b, load_constant
    constants[0] = constant
    a0[3], a1[3], s_mul[3], a0[4] = a, b, 1, ab          # mul: | lhs | rhs | s_mul |  /  | out |
    a0[5], a1[5], s_mul[5], a0[6] = ab, ab, 1, absq
    a0[7], a1[7], s_mul[7], a0[8] = constant, absq, 1, c
    # copy constraints, in permutation-column order [instance, constants, a0, a1] (enable_equality order, simple-example.rs:82-86)
    INST, CONST, A0, A1 = range(4)
    cycles = [[(A0, 0), (A0, 3)], [(A0, 1), (A1, 3)], [(A0, 2), (CONST, 0), (A0, 7)], [(A0, 4), (A0, 5), (A1, 5)],
              [(A0, 6), (A1, 7)], [(A0, 8), (INST, 0)]]                                   # the last one is expose_public
    mapping = [[(col, r) for r in range(n)] for col in range(4)]
    for cells in cycles:
        for i, (col, r) in enumerate(cells):
            mapping[col][r] = cells[(i + 1) % len(cells)]
    return [a0, a1], [constants, s_mul], mapping, c


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", type=int, default=4)
    ap.add_argument("--params", default=None, help="file written by the reference's Params::write (Vesta)")
    args = ap.parse_args(argv)
    import halo2_amd as h
    from halo2_amd import fields
    from halo2_amd.plonk import ConstraintSystem, create_proof, keygen_pk
    from halo2_amd.transcript import Blake2bWrite
    from halo2_amd.verifier import keygen_vk, verify_proof
    curve = h.VESTA                                        # EqAffine, as every proof in the reference
    sf = fields.CURVE_FIELDS[curve][1]
    m = fields.MODULUS[sf]
    if args.params:
        with open(args.params, "rb") as f:
            params = h.Params.read(f, curve)
    else:
        params = toy_params(h, curve, args.k)
    res = prove_and_verify(params)
    params.close()
    return res["ok"]


def prove_and_verify(params, quiet: bool = False) -> dict:
    """keygen, create_proof, verify_proof of the simple-example circuit on `params` (Vesta); returns the timings."""
    import halo2_amd as h
    from halo2_amd import fields
    from halo2_amd.plonk import ConstraintSystem, create_proof, keygen_pk
    from halo2_amd.transcript import Blake2bWrite
    from halo2_amd.verifier import keygen_vk, verify_proof
    curve = params.curve
    sf = fields.CURVE_FIELDS[curve][1]
    m = fields.MODULUS[sf]
    k, n = params.k, params.n
    cs = ConstraintSystem(
        num_fixed_columns=2, num_advice_columns=2, num_instance_columns=1,
        gates=[lambda q: q.fixed(1) * (q.advice(0) * q.advice(1) - q.advice(0, 1))],          # s_mul * (lhs * rhs - out), :116
        advice_queries=[(0, 0), (1, 0), (0, 1)], instance_queries=[(0, 0)], fixed_queries=[(0, 0), (1, 0)],
        permutation_columns=[("instance", 0), ("fixed", 0), ("advice", 0), ("advice", 1)], degree=3, blinding_factors=5)
    a, b, constant = 2, 3, 7                                # simple-example.rs:314-317
    advice, fixed, mapping, c = build(m, n, a, b, constant)
    assert c == 252

    gen = np.random.Generator(np.random.PCG64(0x9E3779B97F4A7C15))
    import torch
    tgen = torch.Generator(device=fields.current_device())
    tgen.manual_seed(0x9E3779B97F4A7C15)

    def rng(count):                                         # any source of uniform scalars; NOT cryptographic here
        if count >= 4096:                                   # the n coefficients of a random polynomial: drawn on the device
            out = torch.randint(-(1 << 63), (1 << 63) - 1, (count, 4), dtype=torch.int64, device=tgen.device, generator=tgen)
            out[:, 3] &= (1 << 62) - 1
            return out
        out = gen.integers(0, 1 << 64, size=(count, 4), dtype=np.uint64)
        out[:, 3] &= np.uint64((1 << 62) - 1)               # limbs of a value below 2^254 < p: a valid Montgomery representation
        return out

    # the assigned columns go to the device once (synthesis is the host's job; the prover starts from resident columns)
    dev = fields.current_device()
    up = lambda col: torch.from_numpy(fields.to_limbs(col, sf, True).view(np.int64)).to(dev)
    advice_host = [fields.to_limbs(col, sf, True) for col in advice]      # what synthesis leaves on the host: Vec<Fp>, Montgomery limbs
    advice, fixed = [torch.from_numpy(col.view(np.int64)).to(dev) for col in advice_host], [up(col) for col in fixed]
    flat = np.arange(4 * n, dtype=np.int64).reshape(4, n)     # mapping as flat cell indices c' * n + r' (identity except the copy cycles)
    for col in range(4):
        for r, (c2, r2) in enumerate(mapping[col][:16]):       # build() only ties cells in the first nine rows
            flat[col][r] = c2 * n + r2
    mapping = flat
    torch.cuda.synchronize()

    t0 = time.perf_counter()
    pk = keygen_pk(params, cs, fixed, mapping)          # transcript_repr derived from the key
    vk = keygen_vk(params, pk)
    t1 = time.perf_counter()
    transcript = Blake2bWrite(curve)
    create_proof(params, pk, advice, [[c]], rng, transcript)
    proof = transcript.finalize()
    t2 = time.perf_counter()
    ok = verify_proof(params, vk, [[c]], proof)
    t3 = time.perf_counter()
    wrong = verify_proof(params, vk, [[c + 1]], proof)
    # a second proof with everything warm (workspaces allocated, tables touched): the steady-state prover time
    transcript2 = Blake2bWrite(curve)
    t4 = time.perf_counter()
    create_proof(params, pk, advice, [[c]], rng, transcript2)
    transcript2.finalize()
    t5 = time.perf_counter()
    # a third one that starts from HOST-resident advice columns, as the reference's prover does after synthesis
    # (plonk/prover.rs:284-313): the PCIe upload of every advice column (2 x 32 MiB at k = 20) is inside the timed region
    transcript3 = Blake2bWrite(curve)
    t6 = time.perf_counter()
    advice_again = [torch.from_numpy(col.view(np.int64)).to(dev) for col in advice_host]
    create_proof(params, pk, advice_again, [[c]], rng, transcript3)
    transcript3.finalize()
    t7 = time.perf_counter()
    if not quiet:
        print(f"k = {k}: keygen {t1 - t0:.3f} s, create_proof {t2 - t1:.3f} s ({len(proof)} bytes; again, warm: {t5 - t4:.3f} s), "
              f"verify_proof {t3 - t2:.3f} s")
        print(f"public input c = {c}: {'accepted' if ok else 'REJECTED'};  c + 1: {'ACCEPTED' if wrong else 'rejected'}")
    return {"ok": bool(ok and not wrong), "k": k, "keygen_s": t1 - t0, "create_proof_first_s": t2 - t1, "create_proof_s": t5 - t4,
            "create_proof_from_host_columns_s": t7 - t6, "advice_columns": len(advice_host),
            "verify_proof_s": t3 - t2, "proof_bytes": len(proof)}


if __name__ == "__main__":
    sys.exit(0 if main() else 1)
