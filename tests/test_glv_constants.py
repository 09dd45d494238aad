"""The endomorphism split of the generic multiexp (halo2_amd/csrc/glv.cuh: k = k1 + k2 lambda, |k1|, |k2| < 2^129) checked on the CPU
from the constants the header SHIPS: they are parsed out of the source, not restated.  zeta is a primitive cube root of unity of the base
field and (zeta x, y) = [lambda](x, y) for the lambda the lattice vectors annihilate; g_i = floor(2^256 (b2, -b1) / q) exactly; and the
split as the kernel computes it (c_i = (k g_i) >> 256, 160-bit two's-complement differences) returns short halves that recombine to k for
random scalars and for the edges of the field.  `best_multiexp` arithmetic.rs:143-180 is what the split serves; no GPU needed."""
import os
import random
import re

import pytest

from oracle import pasta

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "halo2_amd", "csrc")
SRC = open(os.path.join(CSRC, "glv.cuh")).read()
FP, FQ = pasta.P, pasta.Q


def _array(name, fs_is_fq):
    body = re.search(r"const u32 %s\[\d\] = \{(.*?)\};" % name, SRC, re.S).group(1)
    out = []
    for item in re.split(r",(?![^?]*:)", body.replace("\n", " ")):      # commas outside a ternary's arms
        item = item.strip()
        m = re.match(r"FS == FQ \? (0x[0-9a-f]+)u : (0x[0-9a-f]+)u", item)
        out.append(int(m.group(1 if fs_is_fq else 2), 16) if m else int(item.rstrip("u"), 16))
    return sum(v << (32 * i) for i, v in enumerate(out))


def _zeta(base_is_fp):
    lines = re.search(r"glv_zeta\(\).*?\{(.*?)\n\}", SRC, re.S).group(1)
    rows = re.findall(r"fe\{\{(.*?)\}\}", lines)
    limbs = [int(x.strip().rstrip("u"), 16) for x in rows[0 if base_is_fp else 1].split(",")]
    return sum(v << (32 * i) for i, v in enumerate(limbs))


@pytest.mark.parametrize("curve", ["pallas", "vesta"])
def test_glv_constants_and_split(curve):
    # Pallas: coordinates in Fp, scalars in Fq (the kernel's FS == FQ branch); Vesta the other way round
    base, q, fs_is_fq = (FP, FQ, True) if curve == "pallas" else (FQ, FP, False)
    a1, b1m, a2, b2 = (_array(n, fs_is_fq) for n in ("a1", "b1", "a2", "b2"))
    g1, g2 = _array("g1", fs_is_fq), _array("g2", fs_is_fq)
    zeta = pasta.from_mont(_zeta(curve == "pallas"), base)
    assert zeta != 1 and pow(zeta, 3, base) == 1
    # lambda: the scalar the first lattice vector annihilates, a1 + b1 lambda = 0 with b1 = -|b1|
    lam = a1 * pow(b1m, -1, q) % q
    assert lam != 1 and pow(lam, 3, q) == 1
    assert (a2 + b2 * lam) % q == 0
    g = ((base - 1) % base, 2)
    pt = pasta.ec_mul(0x1234567, g, base)
    assert pasta.ec_mul(lam, pt, base) == (zeta * pt[0] % base, pt[1])          # phi(P) = [lambda] P
    assert g1 == (b2 << 256) // q and g2 == (b1m << 256) // q
    rng = random.Random(0x61C7 + fs_is_fq)
    worst = 0
    for k in [0, 1, 2, q - 1, q - 2, (q - 1) // 2, 1 << 254, (1 << 254) - 1, lam, q - lam] + [rng.randrange(q) for _ in range(4000)]:
        c1, c2 = (k * g1) >> 256, (k * g2) >> 256
        k1 = k - c1 * a1 - c2 * a2
        k2 = c1 * b1m - c2 * b2
        assert (k1 + k2 * lam - k) % q == 0
        worst = max(worst, abs(k1).bit_length(), abs(k2).bit_length())
        # the kernel forms both in 160-bit two's complement (sub160 / abs160): exact as long as the true values fit 159 bits
        for v in (k1, k2):
            w = v % (1 << 160)
            assert (w - (1 << 160) if w >> 159 else w) == v
    assert worst <= 129                                             # the digit code reserves 130 bits per half
