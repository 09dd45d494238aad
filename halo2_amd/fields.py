"""Pasta field constants and limb encodings (host-side; the analogue of the constants the reference
takes from pasta_curves: MODULUS, ROOT_OF_UNITY, S, ZETA, TWO_INV -- halo2_proofs/src/poly/domain.rs:58-92)."""
from __future__ import annotations

import numpy as np

from ._lib import FP, FQ, PALLAS, VESTA

P = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001
Q = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001
S = 32
MULTIPLICATIVE_GENERATOR = 5
R = 1 << 256
MODULUS = {FP: P, FQ: Q}
# curve -> (base field id, scalar field id)
CURVE_FIELDS = {PALLAS: (FP, FQ), VESTA: (FQ, FP)}


def root_of_unity(field: int) -> int:
    m = MODULUS[field]
    return pow(MULTIPLICATIVE_GENERATOR, (m - 1) >> S, m)


def zeta(field: int) -> int:
    """Primitive cube root of unity (pasta_curves ZETA)."""
    m = MODULUS[field]
    z = pow(MULTIPLICATIVE_GENERATOR, (m - 1) // 3, m)
    return z * z % m if field == FP else z


def delta(field: int) -> int:
    """ff::PrimeField::DELTA = MULTIPLICATIVE_GENERATOR^(2^S): generator of the odd-order subgroup; separates the columns of the
    permutation argument (plonk/permutation/prover.rs:141, keygen.rs)."""
    return pow(MULTIPLICATIVE_GENERATOR, 1 << S, MODULUS[field])


def sqrt(a: int, field: int):
    """A square root of a, or None (ff::Field::sqrt; Tonelli-Shanks over the 2^32-torsion: both primes are 1 mod 2^32).  Host-side,
    for the handful of points a verifier decodes from a proof; bulk decoding (`Params::read`) is `h2_points_decompress`."""
    m = MODULUS[field]
    a %= m
    if a == 0:
        return 0
    t_odd = (m - 1) >> S
    w = pow(a, (t_odd - 1) // 2, m)              # a^((T-1)/2)
    r = a * w % m                                # a^((T+1)/2): a root once the 2-power part of a^T is cancelled
    t = r * w % m                                # a^T, of order dividing 2^S
    z = root_of_unity(field)                     # generator of the 2^S-torsion
    order = S
    while t != 1:
        i, probe = 0, t
        while probe != 1:
            probe = probe * probe % m
            i += 1
            if i == order:
                return None                      # a^T has full order: a is not a square
        b = pow(z, 1 << (order - i - 1), m)
        z = b * b % m
        r, t, order = r * b % m, t * z % m, i
    return r


_MASK256 = (1 << 256) - 1


def to_limbs(vals, field: int | None = None, montgomery: bool = True) -> np.ndarray:
    """ints -> (n, 4) uint64 limbs; Montgomery form (x * 2^256 mod p) when `montgomery`."""
    if montgomery:
        m = MODULUS[field]
        raw = b"".join((int(v) * R % m).to_bytes(32, "little") for v in vals)
    else:
        raw = b"".join((int(v) & _MASK256).to_bytes(32, "little") for v in vals)
    return np.frombuffer(raw, dtype="<u8").reshape(-1, 4).astype(np.uint64)


def from_limbs(a: np.ndarray, field: int | None = None, montgomery: bool = True) -> list[int]:
    raw = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4).astype("<u8").tobytes()
    vals = [int.from_bytes(raw[i:i + 32], "little") for i in range(0, len(raw), 32)]
    if montgomery:
        m = MODULUS[field]
        rinv = pow(R, -1, m)
        vals = [v * rinv % m for v in vals]
    return vals


def scalar_limbs(v: int, field: int, montgomery: bool = True) -> np.ndarray:
    return to_limbs([v], field, montgomery)[0]


def current_device():
    """The HIP device the C ABI launches on (the caller's current device): one process per GPU sets it once with
    `torch.cuda.set_device(local_rank)`; every tensor this package allocates follows it."""
    import torch
    return torch.device("cuda", torch.cuda.current_device())


def to_device_limbs(a, dev):
    """(count, 4) scalars as an int64 CUDA tensor on `dev`: numpy limbs are uploaded, CUDA tensors pass through (an rng that draws
    its large vectors on the device -- the n random coefficients of the opening / vanishing arguments -- saves the PCIe copy)."""
    import torch
    if isinstance(a, torch.Tensor):
        return a.to(dev).view(torch.int64) if a.dtype != torch.int64 else a.to(dev)
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.uint64).view(np.int64)).to(dev)
