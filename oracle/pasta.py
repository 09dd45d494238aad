"""Python big-int oracle for the halo2 prover hot path (TEST INFRASTRUCTURE ONLY).

This file is a CPU restatement, in plain Python integers, of the algorithms on
the hot path of the reference (zcash/halo2, halo2_proofs 0.3.2):

  * `best_multiexp`  halo2_proofs/src/arithmetic.rs:143-180 (+ Buckets :29-112)
  * `best_fft`       halo2_proofs/src/arithmetic.rs:192-295
  * `EvaluationDomain::{new, ifft, lagrange_to_coeff, coeff_to_extended,
     extended_to_coeff, divide_by_vanishing_poly}` halo2_proofs/src/poly/domain.rs:40-383
  * `Params::{commit, commit_lagrange}` halo2_proofs/src/poly/commitment.rs:119-150

The field / curve arithmetic of the reference lives in the un-vendored crate
`pasta_curves 0.5.1` (Cargo.lock:1303); it is restated here from the published
definition: Fp, Fq prime fields, curves y^2 = x^3 + 5, Montgomery R = 2^256.

Pinning: the field layer is pinned by the reference's own known-answer tests
(Poseidon permutation vectors over Fp and Fq, halo2_poseidon/src/test_vectors.rs;
root-of-unity constants tests/plonk_api.rs:596, circuit_data/vk_*.rdata:7) and
288 pinned on-curve points.  MSM *numeric* outputs have no reachable golden in
the reference tree (they need pasta_curves' hash-to-curve), so MSM parity is
anchored on the definition  sum_i [s_i] P_i  computed here with affine
chord-and-tangent arithmetic: "MSM golden parity unpinned; field layer pinned".

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path (halo2_amd/) never does.
"""
from __future__ import annotations

import math

# --- moduli (halo2_proofs/tests/plonk_api.rs:591-592) ------------------------
P = 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001  # Fp: Pallas base, Vesta scalar
Q = 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001  # Fq: Pallas scalar, Vesta base
S = 32                      # 2-adicity of both fields (book/src/background/fields.md)
GENERATOR = 5               # multiplicative generator of both fields
R = 1 << 256                # Montgomery radix used by pasta_curves' 4x64 limbs
CURVE_B = 5                 # y^2 = x^3 + 5 on both curves

FIELD_ID = {"fp": 0, "fq": 1}
MODULUS = {"fp": P, "fq": Q, 0: P, 1: Q}
# curve id -> (base field modulus, scalar field modulus)
CURVES = {"pallas": (P, Q), "vesta": (Q, P), 0: (P, Q), 1: (Q, P)}


def root_of_unity(m: int) -> int:
    """2^S-th primitive root: GENERATOR^((m-1)/2^S) (pasta_curves ROOT_OF_UNITY)."""
    return pow(GENERATOR, (m - 1) >> S, m)


def omega_for(m: int, k: int) -> int:
    """2^k-th root: ROOT_OF_UNITY^(2^(S-k)) (poly/domain.rs:58-78)."""
    w = root_of_unity(m)
    for _ in range(k, S):
        w = w * w % m
    return w


def zeta(m: int) -> int:
    """Primitive cube root of unity used as coset generator (pasta_curves ZETA).

    Fp: (5^((p-1)/3))^2, Fq: 5^((q-1)/3) -- values recorded in SURVEY.md section 8c.
    The product library never needs it (callers pass it in)."""
    z = pow(GENERATOR, (m - 1) // 3, m)
    return z * z % m if m == P else z


# --- encodings ---------------------------------------------------------------
def to_limbs(x: int) -> list[int]:
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def from_limbs(l) -> int:
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


def to_mont(x: int, m: int) -> int:
    return x * R % m


def from_mont(x: int, m: int) -> int:
    return x * pow(R, -1, m) % m


# --- curve arithmetic: affine chord and tangent, None = identity --------------
def on_curve(pt, m: int) -> bool:
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - CURVE_B) % m == 0


def ec_neg(pt, m: int):
    if pt is None:
        return None
    return (pt[0], (-pt[1]) % m)


def ec_add(a, b, m: int):
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % m == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, m) % m
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, m) % m
    x3 = (lam * lam - x1 - x2) % m
    y3 = (lam * (x1 - x3) - y1) % m
    return (x3, y3)


def ec_mul(k: int, pt, m: int):
    acc = None
    add = pt
    while k:
        if k & 1:
            acc = ec_add(acc, add, m)
        add = ec_add(add, add, m)
        k >>= 1
    return acc


def msm_naive(scalars, bases, m: int):
    """The definition the reference's own test uses (arithmetic.rs:448-455)."""
    acc = None
    for s, b in zip(scalars, bases):
        acc = ec_add(acc, ec_mul(s, b, m), m)
    return acc


def small_multiexp(scalars, bases, m: int):
    """`small_multiexp` restated (arithmetic.rs:116-136): double-and-add over the 256 bits of `to_repr()`, most significant
    first, the doubling shared by all points."""
    reprs = [int(s).to_bytes(32, "little") for s in scalars]
    acc = None
    for byte_idx in range(31, -1, -1):
        for bit_idx in range(7, -1, -1):
            acc = ec_add(acc, acc, m)
            for r, b in zip(reprs, bases):
                if (r[byte_idx] >> bit_idx) & 1:
                    acc = ec_add(acc, b, m)
    return acc


# --- best_multiexp restated (arithmetic.rs:143-180) --------------------------
def window_bits(n: int) -> int:
    """arithmetic.rs:146-152."""
    if n < 4:
        return 1
    if n < 32:
        return 3
    return math.ceil(math.log(float(n & 0xFFFFFFFF)))


def get_at(segment: int, c: int, repr_bytes: bytes) -> int:
    """Buckets::get_at, arithmetic.rs:95-111."""
    skip_bits = segment * c
    skip_bytes = skip_bits // 8
    if skip_bytes >= 32:
        return 0
    v = repr_bytes[skip_bytes:skip_bytes + 8].ljust(8, b"\0")
    tmp = int.from_bytes(v, "little")
    tmp >>= skip_bits - skip_bytes * 8
    return tmp % (1 << c)


def best_multiexp(scalars, bases, m: int):
    """Window/bucket Pippenger exactly as the reference (serial-branch Horner fold,
    arithmetic.rs:169-178).  scalars canonical ints, bases affine tuples / None."""
    assert len(scalars) == len(bases)
    n = len(bases)
    c = window_bits(n)
    reprs = [int(s).to_bytes(32, "little") for s in scalars]
    result = None
    for seg in reversed(range(256 // c + 1)):
        buckets = [None] * ((1 << c) - 1)
        for rb, base in zip(reprs, bases):
            d = get_at(seg, c, rb)
            if d != 0:
                buckets[d - 1] = ec_add(buckets[d - 1], base, m)
        acc = None
        run = None
        for b in reversed(buckets):
            run = ec_add(b, run, m)
            acc = ec_add(acc, run, m)
        for _ in range(c):
            result = ec_add(result, result, m)
        result = ec_add(result, acc, m)
    return result


# --- best_fft restated (arithmetic.rs:192-295) -------------------------------
def bitreverse(n: int, l: int) -> int:
    r = 0
    for _ in range(l):
        r = (r << 1) | (n & 1)
        n >>= 1
    return r


def best_fft(a: list[int], omega: int, log_n: int, m: int) -> None:
    """In place, natural order in and out (iterative form, arithmetic.rs:223-251)."""
    n = len(a)
    assert n == 1 << log_n
    for k in range(n):
        rk = bitreverse(k, log_n)
        if k < rk:
            a[rk], a[k] = a[k], a[rk]
    tw = [1] * max(n // 2, 1)
    for j in range(1, n // 2):
        tw[j] = tw[j - 1] * omega % m
    chunk = 2
    twiddle_chunk = n // 2
    for _ in range(log_n):
        half = chunk // 2
        for start in range(0, n, chunk):
            for i in range(half):
                t = a[start + half + i] * tw[i * twiddle_chunk] % m
                u = a[start + i]
                a[start + i] = (u + t) % m
                a[start + half + i] = (u - t) % m
        chunk *= 2
        twiddle_chunk //= 2


def fft_definition(a: list[int], omega: int, m: int) -> list[int]:
    """O(n^2) definition A[j] = sum_i a[i] omega^(ij)."""
    n = len(a)
    return [sum(a[i] * pow(omega, i * j, m) for i in range(n)) % m for j in range(n)]


# --- EvaluationDomain restated (poly/domain.rs:40-383) -----------------------
class EvaluationDomain:
    def __init__(self, j: int, k: int, m: int):
        self.m = m
        self.k = k
        self.n = 1 << k
        self.quotient_poly_degree = j - 1
        ek = k
        while (1 << ek) < self.n * self.quotient_poly_degree:
            ek += 1
        assert ek <= S
        self.extended_k = ek
        self.extended_omega = omega_for(m, ek)
        self.omega = omega_for(m, k)
        self.omega_inv = pow(self.omega, -1, m)
        self.extended_omega_inv = pow(self.extended_omega, -1, m)
        self.g_coset = zeta(m)
        self.g_coset_inv = self.g_coset * self.g_coset % m
        self.ifft_divisor = pow(1 << k, -1, m)
        self.extended_ifft_divisor = pow(1 << ek, -1, m)
        # t(X) = X^n - 1 over the coset (domain.rs:85-110)
        orig = pow(self.g_coset, self.n, m)
        step = pow(self.extended_omega, self.n, m)
        cur = orig
        t = []
        while True:
            t.append(cur)
            cur = cur * step % m
            if cur == orig:
                break
        assert len(t) == 1 << (ek - k)
        self.t_evaluations = [pow((v - 1) % m, -1, m) for v in t]

    def extended_len(self) -> int:
        return 1 << self.extended_k

    def ifft(self, a, omega_inv, log_n, divisor):
        best_fft(a, omega_inv, log_n, self.m)
        for i in range(len(a)):
            a[i] = a[i] * divisor % self.m

    def lagrange_to_coeff(self, a):
        a = list(a)
        assert len(a) == self.n
        self.ifft(a, self.omega_inv, self.k, self.ifft_divisor)
        return a

    def distribute_powers_zeta(self, a, into_coset: bool):
        cp = [self.g_coset, self.g_coset_inv] if into_coset else [self.g_coset_inv, self.g_coset]
        for idx in range(len(a)):
            i = idx % 3
            if i:
                a[idx] = a[idx] * cp[i - 1] % self.m

    def coeff_to_extended(self, a):
        a = list(a)
        assert len(a) == self.n
        self.distribute_powers_zeta(a, True)
        a += [0] * (self.extended_len() - len(a))
        best_fft(a, self.extended_omega, self.extended_k, self.m)
        return a

    def extended_to_coeff(self, a):
        a = list(a)
        assert len(a) == self.extended_len()
        self.ifft(a, self.extended_omega_inv, self.extended_k, self.extended_ifft_divisor)
        self.distribute_powers_zeta(a, False)
        return a[: self.n * self.quotient_poly_degree]

    def divide_by_vanishing_poly(self, a):
        assert len(a) == self.extended_len()
        t = self.t_evaluations
        return [v * t[i % len(t)] % self.m for i, v in enumerate(a)]


# --- Params::commit* restated (poly/commitment.rs:119-150) -------------------
def commit(g, w, poly, r, m):
    return best_multiexp(list(poly) + [r], list(g) + [w], m)


# --- Poseidon P128Pow5T3 permutation (halo2_poseidon/src/lib.rs:106-151) ------
def poseidon_permute(state, mds, rcs, m, r_f=8, r_p=56):
    state = list(state)

    def apply_mds(s):
        return [sum(mds[i][j] * s[j] for j in range(3)) % m for i in range(3)]

    half = r_f // 2
    for r, rc in enumerate(rcs):
        full = r < half or r >= half + r_p
        if full:
            state = [pow((w + c) % m, 5, m) for w, c in zip(state, rc)]
        else:
            state = [(w + c) % m for w, c in zip(state, rc)]
            state[0] = pow(state[0], 5, m)
        state = apply_mds(state)
    return state


# --- deterministic synthetic inputs (SplitMix64) --------
class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def field(self, m: int) -> int:
        """512 bits reduced mod m (mirrors ff::FromUniformBytes<64>)."""
        v = 0
        for i in range(8):
            v |= self.next() << (64 * i)
        return v % m


def sqrt_mod(a: int, m: int):
    """Tonelli-Shanks; returns a root or None."""
    a %= m
    if a == 0:
        return 0
    if pow(a, (m - 1) // 2, m) != 1:
        return None
    q, s = m - 1, 0
    while q % 2 == 0:
        q //= 2
        s += 1
    z = GENERATOR  # quadratic non-residue in both fields
    mm, c, t, r = s, pow(z, q, m), pow(a, q, m), pow(a, (q + 1) // 2, m)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % m
            i += 1
        b = pow(c, 1 << (mm - i - 1), m)
        mm, c = i, b * b % m
        t, r = t * c % m, r * b % m
    return r


def synth_point(rng: SplitMix64, m: int):
    """Try-and-increment: x random, y = sqrt(x^3+5), sign from the PRNG."""
    while True:
        x = rng.field(m)
        y = sqrt_mod(x * x * x + CURVE_B, m)
        if y is not None and y != 0:
            if rng.next() & 1:
                y = m - y
            return (x, y)


# --- compressed encoding (pasta_curves to_bytes / from_bytes; used by Params::write/read, poly/commitment.rs:169-205,
# --- and transcript write_point, transcript.rs:183-187) -------------------------------------------------------------
def point_to_bytes(pt, m: int) -> bytes:
    """pt: canonical affine (x, y) or None (identity -> 32 zero bytes)."""
    if pt is None:
        return bytes(32)
    b = bytearray(int(pt[0]).to_bytes(32, "little"))
    b[31] |= (pt[1] & 1) << 7
    return bytes(b)


def point_from_bytes(raw: bytes, m: int):
    """Returns (x, y), None for the identity; raises ValueError where from_bytes yields CtOption::none."""
    raw = bytearray(raw)
    sign = raw[31] >> 7
    raw[31] &= 0x7F
    x = int.from_bytes(raw, "little")
    if x >= m:
        raise ValueError("x not canonical")
    if x == 0:
        if sign:
            raise ValueError("(0, odd) is not a point")
        return None
    y = sqrt_mod((x * x * x + CURVE_B) % m, m)
    if y is None:
        raise ValueError("x^3 + 5 is not a square")
    if y & 1 != sign:
        y = m - y
    return (x, y)


def params_write(k: int, g, g_lagrange, w, u, m: int) -> bytes:
    """Params::write (poly/commitment.rs:169-181): k as u32 LE, then g, g_lagrange, w, u compressed."""
    out = bytearray(int(k).to_bytes(4, "little"))
    for pt in list(g) + list(g_lagrange) + [w, u]:
        out += point_to_bytes(pt, m)
    return bytes(out)


def params_read(raw: bytes, m: int):
    """Params::read (poly/commitment.rs:184-205)."""
    k = int.from_bytes(raw[:4], "little")
    n = 1 << k
    pts = [point_from_bytes(raw[4 + 32 * i: 36 + 32 * i], m) for i in range(2 * n + 2)]
    return k, pts[:n], pts[n:2 * n], pts[2 * n], pts[2 * n + 1]
