"""The polynomial-commitment opening argument, `poly::commitment::prover::create_proof`
(halo2_proofs/src/poly/commitment/prover.rs:26-151), with every vector resident in HBM.

Per round the reference runs two half-size MSMs, two inner products, the p' / b folds and the generator collapse
on the host and round-trips nothing; a naive offload would ship O(n) bytes each way 2k times.  Here p', b and G'
live on the device for the whole argument; per round the host sees L_j, R_j (192 bytes) and answers with one challenge.

Three schedules produce the same L_j, R_j (and so the same proof bytes):
* "collapse": the reference's -- multiexps over the collapsed G' (arbitrary bases each round) + the generator collapse,
  round by round from this file (k = 20: 0.062 s);
* "original": L_j, R_j as commits over the ORIGINAL, registered generators with scalars p' (x) s_j
  (`h2_ipa_round_scalars_device`): two registered multiexps per round over g || u || w, G' never exists;
* "paired" (the default where it applies, n >= 8192): the same scalars, but L_j and R_j have disjoint supports in g (the low /
  high half of every 2^(k-j) block), so they share ONE column and leave ONE sort, ONE bucket accumulation (two bucket slices)
  and one fold per round (`h2_commit_pair_device` over g || u || u || w || w).
For the last two the whole ARGUMENT is one C-ABI call (`h2_open_device` / `h2_open`, reached through `Params.open`: the commitment
to s_poly, xi and z, P', b, v and the round loop between the caller's rng and the caller's transcript); `native=False` keeps the
earlier form for A/B -- the steps before the loop from here, the loop alone native (`h2_ipa_rounds_device`, `Params.opening_rounds`).
From k = 16 on the loop moves to the collapsed generators, read off the registered table, after the rounds that leave a table of 2^14 points (k - 14, up to k = 20; 6 at k = 21, 5 beyond)
(`hybrid_rounds`; k = 20: 0.020 s against 0.037 s with every round on the original generators).

torch is plumbing (device buffers, slicing); all arithmetic goes through the C ABI."""
from __future__ import annotations

import numpy as np

from . import fields
from ._lib import FORM_MONTGOMERY
from .arithmetic import (best_multiexp_batch, compute_inner_product, eval_polynomial, fold_scalars, parallel_generator_collapse, powers,
                         scale_add)
from .commitment import Blind, Params


def _host(t) -> np.ndarray:
    return t.cpu().numpy().view(np.uint64)


def create_proof(params: Params, rng, transcript, p_poly, p_blind: Blind, x_3, device=None, schedule: str | None = None,
                 hybrid_rounds: int | None = None, native: bool = True) -> None:
    """Writes the opening proof of `p_poly` at `x_3` to `transcript`.

    rng(count) -> (count, 4) uniformly random scalars, Montgomery limbs (the reference draws `C::Scalar::random`
    n + 1 + 2k times in this order: s_poly coefficients, s_poly_blind, then l_j / r_j randomness per round).
    p_poly: (n, 4) numpy array or torch CUDA tensor; x_3: (4,) limbs."""
    import torch
    curve = params.curve
    sf = fields.CURVE_FIELDS[curve][1]
    m = fields.MODULUS[sf]
    n, k = params.n, params.k
    if p_poly.shape[0] != n:
        raise ValueError("create_proof: polynomial length != params.n")                  # prover.rs:41
    dev = p_poly.device if hasattr(p_poly, "device") and not isinstance(p_poly, np.ndarray) else (torch.device(device) if device else fields.current_device())
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64)).to(dev)
    as_int = lambda limbs: fields.from_limbs(limbs, sf, True)[0]
    as_limbs = lambda v: fields.scalar_limbs(v % m, sf, True)
    x3 = np.ascontiguousarray(x_3, dtype=np.uint64).reshape(4)
    if schedule is None:
        schedule = "paired" if n >= 8192 else "original"
    if schedule not in ("original", "collapse", "paired"):
        raise ValueError("create_proof: schedule must be 'paired', 'original' or 'collapse'")
    if schedule == "paired" and (n < 8192 or not params.pair_commit_supported()):
        schedule = "original"          # small tables use narrower windows, which the paired sort does not take
    if native and schedule != "collapse":
        # the whole argument as one C-ABI call; the randomness is drawn here, in the reference's order (:45-47, :53, :111-112)
        s_raw = rng(n)                         # host limbs, or a CUDA tensor from an rng that draws its large vectors on the device
        s_blind = Blind(np.ascontiguousarray(rng(1)[0]))
        rands = np.ascontiguousarray(np.concatenate([np.asarray(rng(2), dtype=np.uint64).reshape(2, 4) for _ in range(k)]))
        if isinstance(s_raw, np.ndarray):          # host randomness: it crosses PCIe inside the call, committed quarter by quarter as it lands
            c, f_final = params.open(p_poly if isinstance(p_poly, np.ndarray) else p_poly.contiguous(), p_blind, x3, s_raw, s_blind, rands, transcript,
                                     paired=schedule == "paired", hybrid_rounds=hybrid_rounds)
        else:
            d_p = p_poly.contiguous() if not isinstance(p_poly, np.ndarray) else to_dev(p_poly)
            c, f_final = params.open(d_p, p_blind, x3, fields.to_device_limbs(s_raw, dev).contiguous(), s_blind, rands, transcript,
                                     paired=schedule == "paired", hybrid_rounds=hybrid_rounds)
        transcript.write_scalar(c)                                                        # prover.rs:146-148
        transcript.write_scalar(f_final)
        return
    d_p = p_poly if not isinstance(p_poly, np.ndarray) else to_dev(p_poly)

    # random polynomial with a root at x_3 (prover.rs:43-53)
    d_s = fields.to_device_limbs(rng(n), dev)
    s_at_x3 = as_int(_host(eval_polynomial(d_s, x3, sf)))
    d_s[0] = to_dev(as_limbs(as_int(_host(d_s[0])) - s_at_x3))
    s_blind = Blind(np.ascontiguousarray(rng(1)[0]))
    transcript.write_point(_host(params.commit(d_s, s_blind)))               # prover.rs:56-57
    xi = transcript.squeeze_challenge_scalar()                                            # prover.rs:62
    z = transcript.squeeze_challenge_scalar()                                             # prover.rs:66

    # P' = P - [v] G_0 + [xi] S (prover.rs:70-73); the synthetic blind f starts as P''s blind (:74-78)
    d_pp = scale_add(d_s, xi, d_p, sf)                                                    # in place on d_s
    v = as_int(_host(eval_polynomial(d_pp, x3, sf)))
    d_pp[0] = to_dev(as_limbs(as_int(_host(d_pp[0])) - v))
    f = (as_int(s_blind.value) * as_int(xi) + as_int(p_blind.value)) % m
    z_i = as_int(z)

    d_b = powers(x3, n, sf, device=dev)                                                   # prover.rs:86-97
    if schedule != "collapse":
        # the round loop as one C-ABI call (h2_ipa_rounds_device): the host sees L_j, R_j once per round and answers with the challenge
        rands = np.ascontiguousarray(np.concatenate([np.asarray(rng(2), dtype=np.uint64).reshape(2, 4) for _ in range(k)]))
        c, f_delta = params.opening_rounds(d_pp, d_b, z, rands, transcript, paired=schedule == "paired", hybrid_rounds=hybrid_rounds)
        transcript.write_scalar(c)                                                        # prover.rs:146-148
        transcript.write_scalar(as_limbs(f + as_int(f_delta)))
        return
    d_g = to_dev(params.g)                                                                # G' (prover.rs:101)
    d_uw = to_dev(np.stack([params.u, params.w]))
    for j in range(k):                                                                    # prover.rs:104-142
        half = 1 << (k - j - 1)
        lo_p, hi_p = d_pp[:half], d_pp[half:2 * half]
        values = _host(torch.stack([compute_inner_product(hi_p, d_b[:half], sf),                 # prover.rs:109-110
                                    compute_inner_product(lo_p, d_b[half:2 * half], sf)]))
        value_l, value_r = as_int(values[0]), as_int(values[1])
        l_rand, r_rand = rng(2)
        # L_j = <p'_hi, G'_lo> + [value_l z] U + [l_rand] W as ONE multiexp (the reference's TODO, :108-110)
        tail_l = to_dev(np.stack([as_limbs(value_l * z_i), l_rand]))
        tail_r = to_dev(np.stack([as_limbs(value_r * z_i), r_rand]))
        lr = _host(best_multiexp_batch([(torch.cat([hi_p, tail_l]), torch.cat([d_g[:half], d_uw])),
                                        (torch.cat([lo_p, tail_r]), torch.cat([d_g[half:2 * half], d_uw]))],
                                       curve, FORM_MONTGOMERY, affine=False))
        transcript.write_point(lr[0])                                                     # prover.rs:121-122
        transcript.write_point(lr[1])
        u_j = transcript.squeeze_challenge_scalar()                                       # prover.rs:124
        u_i = as_int(u_j)
        u_inv_i = pow(u_i, -1, m)                                                         # prover.rs:125
        d_pp = fold_scalars(d_pp[:2 * half], as_limbs(u_inv_i), sf)                       # prover.rs:128-133
        d_b = fold_scalars(d_b[:2 * half], u_j, sf)
        d_g = parallel_generator_collapse(d_g[:2 * half], u_j, curve)                     # prover.rs:136-137
        f = (f + as_int(l_rand) * u_inv_i + as_int(r_rand) * u_i) % m                     # prover.rs:140-141

    transcript.write_scalar(_host(d_pp[0]))                                               # c  (prover.rs:146-148)
    transcript.write_scalar(as_limbs(f))
