"""The generated carry-free multipliers (halo2_amd/csrc/field9_*.inc, written by gen_field9_mul.py), checked on the CPU: the
asm text of each statement is interpreted instruction by instruction (the nine instruction forms the generator emits, with
their gfx950 semantics on 32- / 64-bit registers) and the result compared with big-integer arithmetic -- value congruent to
a b 2^-261 (resp. a^2, a b + c d, a^2 - s 2^261) mod p, limbs normalised as field9.cuh promises, the 64-bit column accumulator
never overflowing.  Signed and un-normalised operands at the bounds the point formulas use (curve9.cuh), both Pasta moduli,
and the limb value 2^29 a product may leave in limb 0.  The device runs the same statements against the C oracle in
tests/native/field_check.hip; this test needs no GPU."""
import os
import random
import re

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "halo2_amd", "csrc")
P = {0: 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001,
     1: 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001}
M29 = (1 << 29) - 1


def _s32(x):
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x >> 31 else x


def _s64(x):
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >> 63 else x


def _statement(name):
    text = open(os.path.join(CSRC, name)).read()
    asm = re.search(r'asm\("(.*?)"\n\s*:', text, re.S).group(1)
    return asm.split("\\n\\t")


def _run(ins, operands, field):
    """Interprets one generated statement.  operands: {'a': [9 ints], ...}; returns the nine result limbs (signed 32-bit)."""
    p = P[field]
    limbs = [(p >> (29 * i)) & M29 for i in range(9)]
    reg = {}                                   # 32-bit registers by name, values as unsigned 32-bit
    for k, v in operands.items():
        for i, x in enumerate(v):
            assert -(1 << 31) <= x < (1 << 31)
            reg[f"%[{k}{i}]"] = x & 0xFFFFFFFF
    reg["%[k1]"], reg["%[k2]"], reg["%[k3]"] = limbs[1], limbs[2], limbs[3]
    peak = 0

    def rd32(tok):
        if tok in reg:
            return reg[tok]
        return int(tok, 0) & 0xFFFFFFFF       # inline constant / literal

    def rd64(tok):
        if tok.startswith("v["):
            lo, hi = re.match(r"v\[(\d+):(\d+)\]", tok).groups()
            return reg[f"v{lo}"] | (reg[f"v{hi}"] << 32)
        return int(tok, 0) & ((1 << 64) - 1)  # -1 / 0 as a 64-bit inline constant

    def wr64(tok, val):
        lo, hi = re.match(r"v\[(\d+):(\d+)\]", tok).groups()
        reg[f"v{lo}"], reg[f"v{hi}"] = val & 0xFFFFFFFF, (val >> 32) & 0xFFFFFFFF

    for line in ins:
        op, rest = line.split(None, 1)
        a = [t.strip() for t in rest.split(",")]
        if op in ("s_movk_i32", "s_mov_b32", "v_mov_b32"):
            reg[a[0]] = rd32(a[1])
        elif op == "v_mad_i64_i32":            # D = S0 * S1 + S2, signed, 64-bit; a[1] is the (unused) carry-out
            val = _s32(rd32(a[2])) * _s32(rd32(a[3])) + _s64(rd64(a[4]))
            assert -(1 << 63) <= val < (1 << 63), "column accumulator overflow"
            peak = max(peak, abs(val))
            wr64(a[0], val & ((1 << 64) - 1))
        elif op == "v_lshl_add_u64":           # D = (S0 << S1) + S2
            val = (_s64(rd64(a[1])) << int(a[2])) + _s64(rd64(a[3]))
            assert -(1 << 63) <= val < (1 << 63), "column accumulator overflow"
            wr64(a[0], val & ((1 << 64) - 1))
        elif op == "v_ashrrev_i64":            # D = S1 >> S0 (arithmetic)
            wr64(a[0], (_s64(rd64(a[2])) >> int(a[1])) & ((1 << 64) - 1))
        elif op == "v_bfi_b32":                # D = (S0 & S1) | (~S0 & S2)
            s0, s1, s2 = rd32(a[1]), rd32(a[2]), rd32(a[3])
            reg[a[0]] = ((s0 & s1) | (~s0 & s2)) & 0xFFFFFFFF
        elif op == "v_alignbit_b32":           # D = ({S0, S1} >> S2[4:0]) & 0xffffffff: the last column's limb 8
            reg[a[0]] = (((rd32(a[1]) << 32) | rd32(a[2])) >> (int(a[3], 0) & 31)) & 0xFFFFFFFF
        elif op == "v_and_b32":
            reg[a[0]] = rd32(a[1]) & rd32(a[2])
        elif op == "v_add_u32":
            reg[a[0]] = (rd32(a[1]) + rd32(a[2])) & 0xFFFFFFFF
        elif op == "v_sub_u32":
            reg[a[0]] = (rd32(a[1]) - rd32(a[2])) & 0xFFFFFFFF
        else:
            raise AssertionError(f"instruction form the model does not know: {line}")
    return [_s32(reg[f"%[r{i}]"]) for i in range(9)], peak


def _value(limbs):
    return sum(x << (29 * i) for i, x in enumerate(limbs))


def _limbs_of(value, rng, spread):
    """A signed 9-limb representation of `value` (limbs 0..7 within +-2^spread around their canonical digits)."""
    out, carry = [], 0
    for i in range(8):
        digit = ((value >> (29 * i)) & M29) + carry
        wobble = rng.randrange(-(1 << spread) // (1 << 29), (1 << spread) // (1 << 29) + 1) if spread > 29 else 0
        out.append(digit - (wobble << 29))
        carry = wobble
    out.append((value >> 232) + carry)
    assert _value(out) == value
    return out


def _operand(rng, p, kind):
    if kind == "normalised":                   # what a product leaves: value in (-2^255, 2^255 + p), limbs 0..7 in [0, 2^29)
        return _limbs_of(rng.randrange(-(1 << 255), (1 << 255) + p), rng, 29)
    if kind == "difference":                   # normalised - normalised: limbs in (-2^29, 2^29), |value| < 2^257
        a, b = _operand(rng, p, "normalised"), _operand(rng, p, "normalised")
        return [x - y for x, y in zip(a, b)]
    if kind == "wide":                         # limbs up to 2^30 in magnitude, |value| < 2^258
        return _limbs_of(rng.randrange(-(1 << 258) + 1, 1 << 258), rng, 30)
    raise ValueError(kind)


def _check_result(r, want_mod_p, p, value_bound):
    assert all(0 <= x < (1 << 29) for x in r[1:8]) and 1 <= r[0] <= (1 << 29), r       # limb 0: the pending +1 lands here
    v = _value(r)
    assert v % p == want_mod_p % p
    assert abs(v) < value_bound


@pytest.mark.parametrize("field", [0, 1])
def test_generated_multipliers_against_big_integers(field):
    p = P[field]
    rinv = pow(1 << 261, -1, p)
    rng = random.Random(0x9E37 + field)
    mul, sqr = _statement("field9_mul.inc"), _statement("field9_sqr.inc")
    dot2, sqr_minus = _statement("field9_dot2.inc"), _statement("field9_sqr_minus.inc")
    assert sum(i.startswith("v_mad") for i in mul) == 126 and sum(i.startswith("v_mad") for i in dot2) == 207
    for trial in range(60):
        kinds = ("normalised", "difference", "wide") if trial % 3 else ("normalised", "normalised", "normalised")
        a, b = _operand(rng, p, kinds[trial % 2]), _operand(rng, p, "normalised")
        if trial == 7:
            a[0] = 1 << 29                      # the limb value only a product's limb 0 takes
        r, _ = _run(mul, {"a": a, "b": b}, field)
        _check_result(r, _value(a) * _value(b) * rinv, p, (1 << 255) + p + 1)
        d = _operand(rng, p, "difference")
        r, _ = _run(sqr, {"a": d}, field)
        _check_result(r, _value(d) ** 2 * rinv, p, (1 << 255) + p + 1)
        # Y3 = R (Q - X3) + (-Y1) PPP: difference x difference + normalised x normalised (curve9.cuh)
        c, e = _operand(rng, p, "difference"), _operand(rng, p, "normalised")
        f = [-x for x in _operand(rng, p, "normalised")]
        r, peak = _run(dot2, {"a": d, "b": c, "c": f, "d": e}, field)
        _check_result(r, (_value(d) * _value(c) + _value(f) * _value(e)) * rinv, p, 1 << 256)
        assert peak < 1 << 63
        # X3 = R^2 - (PPP + 2 Q): the subtrahend's limbs reach 3 * 2^29
        s = [x + 2 * y for x, y in zip(_operand(rng, p, "normalised"), _operand(rng, p, "normalised"))]
        r, _ = _run(sqr_minus, {"a": d, "s": s}, field)
        _check_result(r, _value(d) ** 2 * rinv - _value(s), p, 1 << 258)
    # extreme limbs: every limb of every operand at +-2^29 (the column bound of fe9_dot2: 18 + 5 terms below 2^58)
    top = [(1 << 29)] * 8 + [1 << 24]
    neg = [-x for x in top]
    r, peak = _run(dot2, {"a": top, "b": top, "c": neg, "d": neg}, field)
    assert peak < 1 << 63 and _value(r) % p == (2 * _value(top) ** 2 * rinv) % p


@pytest.mark.parametrize("field", [0, 1])
def test_ntt_operand_regime_of_the_multiplier(field):
    """The carry-free NTT passes (csrc/ntt.hip, header of ntt_pass9) multiply data whose limbs are RAW -- the outputs of a radix-4 round,
    o = (e0 +- e1 wA) +- (e2 +- e3 wA) wB with e0 normalised and every product's limbs in [0, 2^29]: limbs in (-2^30, 3 x 2^29), limb 8
    carrying the growth of up to |value| < 2^261 -- by a NORMALISED twiddle in [0, p), with no carry pass in between.  Checked on the
    generated statement itself: the 64-bit column accumulator stays below 2^63, the product is congruent and leaves normalised, and it
    is small again (below 2^255 + p: what the next rounds add to an element), also at the corners of the limb range."""
    p = P[field]
    rinv = pow(1 << 261, -1, p)
    rng = random.Random(0x4E5454 + field)
    mul = _statement("field9_mul.inc")
    lo, hi = -(1 << 30) + 1, 3 * (1 << 29) - 1

    def twiddle(v):
        return [(v >> (29 * i)) & M29 for i in range(8)] + [v >> 232]

    worst = 0
    for trial in range(80):
        a = [rng.randrange(lo, hi + 1) for _ in range(8)] + [rng.randrange(-(1 << 29), (1 << 29) + 1)]
        if trial == 0:
            a = [hi] * 8 + [1 << 29]                   # every limb at the top of the RAW range
        elif trial == 1:
            a = [lo] * 8 + [-(1 << 29)]                # ... at the bottom
        elif trial == 2:
            a = [hi if i % 2 else lo for i in range(8)] + [1 << 29]
        w = p - 1 - trial if trial < 4 else rng.randrange(0, p)
        b = twiddle(w)
        assert _value(b) == w and all(0 <= x < (1 << 29) for x in b[:8])
        r, peak = _run(mul, {"a": a, "b": b}, field)
        worst = max(worst, peak)
        assert peak < 1 << 63, (trial, peak.bit_length())
        assert all(0 <= x < (1 << 29) for x in r[1:8]) and 1 <= r[0] <= (1 << 29), r
        v = _value(r)
        assert v % p == (_value(a) * w * rinv) % p
        # |a| < 2^261 (limb 8 within +-2^29) plus what raw low limbs add: the product stays within a few p
        assert abs(v) < (1 << 255) + 4 * p, v.bit_length()
    assert worst > 1 << 61                             # the corners do come close: the bound is tight, not vacuous


# ---- a whole 10-stage NTT pass at limb level ----------------------------------------------------------------------------------------
# csrc/ntt.hip, ntt_pass9<F, 10, FIRST>: the arithmetic of a 2^10-point transform exactly as the kernel sequences it -- unpack, five radix-4
# rounds with the consumer-side carry passes (e0 and e2 normalised when read, e1 and e3 multiplied RAW), the transform's first round
# without its multiplications by omega^0, one fold by the q p table and one sign-selected carry pass to the canonical value -- with the
# multiplications done by the generated statement under the interpreter above.  What the kernel's comments argue (limb ranges, |value|
# < 2^260 after five rounds, |q| <= 64, one conditional addition of p) is asserted on the way, and the output is compared index by index
# with the reference's radix-2 network on big integers (arithmetic.rs:192-255) for a RANDOM omega (benches/fft.rs:17).
def _norm(a):
    r, c = [], 0
    for i in range(8):
        t = a[i] + c
        assert -(1 << 31) <= t < (1 << 31)
        r.append(t & M29)
        c = t >> 29
    t8 = a[8] + c
    assert -(1 << 31) <= t8 < (1 << 31)
    return r + [t8]


def _i32(v):
    assert all(-(1 << 31) <= x < (1 << 31) for x in v), v
    return v


def _reference_network(a, omega, log_n, p):
    """best_fft of the reference: bit reversal, then log n butterfly stages with twiddle omega^(j n / 2m) for the j-th pair of a block"""
    n = 1 << log_n
    a = list(a)
    for k in range(n):
        rk = int(format(k, f"0{log_n}b")[::-1], 2)
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    m = 1
    for _ in range(log_n):
        w_m = pow(omega, n // (2 * m), p)
        for k in range(0, n, 2 * m):
            w = 1
            for j in range(m):
                t = a[k + j + m] * w % p
                a[k + j + m] = (a[k + j] - t) % p
                a[k + j] = (a[k + j] + t) % p
                w = w * w_m % p
        m *= 2
    return a


@pytest.mark.parametrize("field,L", [(0, 10), (1, 10), (0, 11), (1, 11), (0, 12)])
def test_ntt_pass_limb_model_against_reference_network(field, L):
    """L = 10: the shipped 10-stage pass.  L = 11 / 12 (round 5): the 11- and 12-stage passes behind H2_NTT_MAXR -- an ODD stage count opens
    with a radix-2 round on four elements per lane (the transform's first stage multiplies by omega^0 only: four loads, four additions),
    the radix-4 rounds then start at stage 1 and have no omega^0 shortcuts; six rounds of products instead of five must still leave
    |value| < 2^260 and |q| <= 64 for the fold.  The last step is ntt_canonical_folded9 as round 5 writes it: the folded value packed as a
    256-bit two's-complement word, p added by an 8-word carry chain exactly when bit 255 is set."""
    p = P[field]
    n = 1 << L
    rng = random.Random(0x70A55 + field + 97 * L)
    mul = _statement("field9_mul.inc")
    omega = rng.randrange(2, p)                                    # any field element: the network is the contract, not the DFT
    data = [rng.randrange(0, p) for _ in range(n)]
    data[3], data[5], data[7] = 0, p - 1, 1                        # edge residues among the inputs
    want = _reference_network(data, omega, L, p)
    R9 = pow(2, 261, p)
    plimbs = [(p >> (29 * i)) & M29 for i in range(8)] + [p >> 232]
    pwords = [(p >> (32 * i)) & 0xFFFFFFFF for i in range(8)]

    def tw(e):                                                     # omega^e in M9 form, normalised limbs (ntt_twiddles9)
        v = pow(omega, e, p) * R9 % p
        return [(v >> (29 * i)) & M29 for i in range(8)] + [v >> 232]

    peak_all, max_abs_value = 0, 0

    def fmul(a, b):
        nonlocal peak_all
        r, peak = _run(mul, {"a": _i32(a), "b": b}, field)
        peak_all = max(peak_all, peak)
        assert all(0 <= x < (1 << 29) for x in r[1:8]) and 1 <= r[0] <= (1 << 29)
        return r

    add = lambda x, y: [s + t for s, t in zip(x, y)]
    sub = lambda x, y: [s - t for s, t in zip(x, y)]
    raw_ok = lambda v: all(-(1 << 30) < x < 3 * (1 << 29) for x in v[:8])
    # first pass: LDS row `row` holds input element bitrev(row) (the bit reversal is folded into the gather)
    rev = lambda k: int(format(k, f"0{L}b")[::-1], 2)
    x = [[(data[rev(k)] >> (29 * i)) & M29 for i in range(8)] + [data[rev(k)] >> 232] for k in range(n)]     # fe9_unpack
    u0 = 0
    if L & 1:                                                      # odd stage count: stage 0 as a radix-2 round, no products (omega^0)
        nxt = list(x)
        for base in range(0, n, 2):
            nxt[base], nxt[base + 1] = _i32(add(x[base], x[base + 1])), _i32(sub(x[base], x[base + 1]))
            assert raw_ok(nxt[base]) and raw_ok(nxt[base + 1])
        x = nxt
        u0 = 1
    for u in range(u0, L, 2):                                      # round u: stages t = u and u + 1
        t = u
        nxt = list(x)
        for base in range(n):
            if base & (3 << t):
                continue
            low = base & ((1 << t) - 1)
            i00, i01, i10, i11 = base, base + (1 << t), base + (2 << t), base + (3 << t)
            first = u == 0
            # inputs of the transform's first round come unpacked (normalised); later rounds read RAW limbs and normalise e0, e2
            e0, e1, e2, e3 = (x[i00], x[i01], x[i10], x[i11]) if first else (_norm(x[i00]), x[i01], _norm(x[i10]), x[i11])
            if not first:
                assert raw_ok(x[i01]) and raw_ok(x[i11])
                wA = tw(low << (L - t - 1))
                e1, e3 = fmul(e1, wA), fmul(e3, wA)
            a0, a1 = add(e0, e1), sub(e0, e1)
            wB0, wB1 = tw(low << (L - t - 2)), tw((low + (1 << t)) << (L - t - 2))
            a2 = _norm(add(e2, e3)) if first else fmul(add(e2, e3), wB0)      # stage 1's omega^0: a carry pass stands in
            a3 = fmul(sub(e2, e3), wB1)
            nxt[i00], nxt[i10] = _i32(add(a0, a2)), _i32(sub(a0, a2))
            nxt[i01], nxt[i11] = _i32(add(a1, a3)), _i32(sub(a1, a3))
            for o in (nxt[i00], nxt[i10], nxt[i01], nxt[i11]):
                assert raw_ok(o), o                                # RAW: limbs in (-2^30, 3 x 2^29)
                max_abs_value = max(max_abs_value, abs(_value(o)))
        x = nxt
    assert max_abs_value < 1 << 260                                # two products per radix-4 round (one for the radix-2 round), absorbed by limb 8
    assert peak_all < 1 << 63
    got = []
    for k in range(n):
        v = x[k]
        q = (v[8] + (1 << 21)) >> 22                               # ntt_fold9: q = round(value / 2^254)
        assert -64 <= q <= 64
        qp, c = [], 0
        for i in range(8):                                         # the q p table entry as the kernel builds it
            tq = q * plimbs[i] + c
            qp.append(tq & M29)
            c = tq >> 29
        qp.append(q * plimbs[8] + c)
        f = _norm(_i32(sub(v, qp)))
        fv = _value(f)
        assert abs(fv) < (1 << 253) + (1 << 133)
        # ntt_canonical_folded9: fe9_pack of the normalised limbs IS the 256-bit two's-complement word; bit 255 selects + p, one carry chain
        words = [((fv % (1 << 256)) >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
        packed = sum(((f[i] % (1 << 32)) << (29 * i)) for i in range(8)) + ((f[8] % (1 << 32)) << 232)        # shifts and ors of the limbs
        assert packed % (1 << 256) == fv % (1 << 256)
        neg = words[7] >> 31
        assert neg == (1 if fv < 0 else 0)
        out, carry = [], 0
        for i in range(8):
            t_ = words[i] + (pwords[i] if neg else 0) + carry
            out.append(t_ & 0xFFFFFFFF)
            carry = t_ >> 32
        val = sum(w << (32 * i) for i, w in enumerate(out))
        assert 0 <= val < p
        got.append(val)
    assert got == want


# ---- the accumulate loop's mixed addition at limb level -----------------------------------------------------------------------------
# csrc/curve9.cuh, xyzz9_madd (the inner loop of msm_accumulate: madd-2008-s, 8M + 2S with X3 and Y3 in the fused forms): a bucket's
# chain of additions sequenced exactly as the kernel does, every product by the generated statement under the interpreter, against
# affine big-integer addition (Buckets::sum adds with `+=` on the curve, arithmetic.rs:29-58).  Asserts what curve9.cuh's header argues:
# every intermediate a valid operand (32-bit limbs, column accumulator below 2^63) and X, Y, ZZ, ZZZ normalised after every addition,
# with no carry pass anywhere.
@pytest.mark.parametrize("field", [0, 1])
def test_mixed_addition_limb_model_against_affine_addition(field):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import pasta
    p = P[field]
    rng = random.Random(0xADD + field)
    mul, sqr = _statement("field9_mul.inc"), _statement("field9_sqr.inc")
    dot2, sqr_minus = _statement("field9_dot2.inc"), _statement("field9_sqr_minus.inc")
    R9 = pow(2, 261, p)
    peak_all = 0

    def run(stmt, ops):
        nonlocal peak_all
        r, peak = _run(stmt, {k: _i32(v) for k, v in ops.items()}, field)
        peak_all = max(peak_all, peak)
        assert all(0 <= x < (1 << 29) for x in r[1:8]) and 1 <= r[0] <= (1 << 29), r          # every result leaves normalised
        return r

    def m9(v):                                                      # a coordinate in M9 form, normalised limbs (a table entry)
        v = v * R9 % p
        return [(v >> (29 * i)) & M29 for i in range(8)] + [v >> 232]

    add = lambda x, y: [s + t for s, t in zip(x, y)]
    sub = lambda x, y: [s - t for s, t in zip(x, y)]
    g = ((p - 1) % p, 2)                                            # (-1, 2): on y^2 = x^3 + 5 over both base fields
    assert pasta.on_curve(g, p)
    pts = [pasta.ec_mul(rng.randrange(1, 1 << 64), g, p) for _ in range(24)]
    pts[5] = pasta.ec_neg(pts[2], p)                                # a negated earlier point further down the chain (not adjacent: no P - P)
    acc, want = None, None
    for q in pts:
        qx, qy = m9(q[0]), m9(q[1])
        want = pasta.ec_add(want, q, p)
        if acc is None:                                             # first entry of a bucket: the point itself, ZZ = ZZZ = one
            acc = {"x": qx, "y": qy, "zz": m9(1), "zzz": m9(1)}
            continue
        u2, s2 = run(mul, {"a": qx, "b": acc["zz"]}), run(mul, {"a": qy, "b": acc["zzz"]})
        pd, r = sub(u2, acc["x"]), sub(s2, acc["y"])
        assert _value(pd) % p != 0                                  # (the rare P = +-Q branch is not what this test is about)
        pp = run(sqr, {"a": pd})
        ppp = run(mul, {"a": pd, "b": pp})
        qq = run(mul, {"a": acc["x"], "b": pp})
        x3 = run(sqr_minus, {"a": r, "s": add([2 * v for v in qq], ppp)})
        y3 = run(dot2, {"a": r, "b": sub(qq, x3), "c": [-v for v in acc["y"]], "d": ppp})
        acc = {"x": x3, "y": y3, "zz": run(mul, {"a": acc["zz"], "b": pp}), "zzz": run(mul, {"a": acc["zzz"], "b": ppp})}
        for c in acc.values():
            assert abs(_value(c)) < 1 << 258                        # a valid operand of the next addition (field9.cuh)
        zz, zzz = _value(acc["zz"]) % p, _value(acc["zzz"]) % p
        got = (_value(acc["x"]) * pow(zz, -1, p) % p, _value(acc["y"]) * pow(zzz, -1, p) % p)
        assert got == want
    assert peak_all < 1 << 63


@pytest.mark.parametrize("field", [0, 1])
def test_full_addition_limb_model_against_affine_addition(field):
    """csrc/curve9.cuh, xyzz9_add (add-2008-s, 12M + 2S: the bucket fold's addition -- fold9_finish, the line sums, the bit planes): a
    binary tree of XYZZ + XYZZ additions over partial sums that are themselves results of mixed / full additions, sequenced as the kernel
    does and multiplied by the generated statements, against affine big-integer addition."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import pasta
    p = P[field]
    rng = random.Random(0xF011 + field)
    mul, sqr = _statement("field9_mul.inc"), _statement("field9_sqr.inc")
    dot2, sqr_minus = _statement("field9_dot2.inc"), _statement("field9_sqr_minus.inc")
    R9 = pow(2, 261, p)
    peak_all = 0

    def run(stmt, ops):
        nonlocal peak_all
        r, peak = _run(stmt, {k: _i32(v) for k, v in ops.items()}, field)
        peak_all = max(peak_all, peak)
        assert all(0 <= x < (1 << 29) for x in r[1:8]) and 1 <= r[0] <= (1 << 29), r
        return r

    def m9(v):
        v = v * R9 % p
        return [(v >> (29 * i)) & M29 for i in range(8)] + [v >> 232]

    add = lambda x, y: [s + t for s, t in zip(x, y)]
    sub = lambda x, y: [s - t for s, t in zip(x, y)]

    def full_add(a, b):
        u1, u2 = run(mul, {"a": a["x"], "b": b["zz"]}), run(mul, {"a": b["x"], "b": a["zz"]})
        s1, s2 = run(mul, {"a": a["y"], "b": b["zzz"]}), run(mul, {"a": b["y"], "b": a["zzz"]})
        pd, r = sub(u2, u1), sub(s2, s1)
        assert _value(pd) % p != 0
        pp = run(sqr, {"a": pd})
        ppp = run(mul, {"a": pd, "b": pp})
        qq = run(mul, {"a": u1, "b": pp})
        x3 = run(sqr_minus, {"a": r, "s": add([2 * v for v in qq], ppp)})
        y3 = run(dot2, {"a": r, "b": sub(qq, x3), "c": [-v for v in s1], "d": ppp})
        return {"x": x3, "y": y3, "zz": run(mul, {"a": run(mul, {"a": a["zz"], "b": b["zz"]}), "b": pp}),
                "zzz": run(mul, {"a": run(mul, {"a": a["zzz"], "b": b["zzz"]}), "b": ppp})}

    g = ((p - 1) % p, 2)
    pts = [pasta.ec_mul(rng.randrange(1, 1 << 64), g, p) for _ in range(16)]
    level = [({"x": m9(q[0]), "y": m9(q[1]), "zz": m9(1), "zzz": m9(1)}, q) for q in pts]       # leaves: affine points as XYZZ
    while len(level) > 1:
        nxt = []
        for (a, wa), (b, wb) in zip(level[0::2], level[1::2]):
            c, wc = full_add(a, b), pasta.ec_add(wa, wb, p)
            zz, zzz = _value(c["zz"]) % p, _value(c["zzz"]) % p
            assert (_value(c["x"]) * pow(zz, -1, p) % p, _value(c["y"]) * pow(zzz, -1, p) % p) == wc
            rinv = pow(R9, -1, p)                                   # (limbs hold M9 forms: ZZ R and ZZZ R)
            assert pow(zz * rinv, 3, p) == pow(zzz * rinv, 2, p)    # the XYZZ invariant ZZ^3 = ZZZ^2
            nxt.append((c, wc))
        level = nxt
    assert peak_all < 1 << 63


# ---- two passes and the vector between them ----------------------------------------------------------------------------------------------
def _unpack9(words):                                               # field9.cuh fe9_unpack: nine 29-bit fields of a 256-bit word
    v = sum(w << (32 * i) for i, w in enumerate(words))
    return [(v >> (29 * i)) & M29 for i in range(9)]


def _pack9(limbs):                                                 # field9.cuh fe9_pack: shifts and ors of the 32-bit limb words
    words = []
    for w in range(8):
        x = 0
        for i in range(9):
            lo = 29 * i - 32 * w
            if -29 < lo < 32:
                u = limbs[i] & 0xFFFFFFFF
                x |= (u << lo) & 0xFFFFFFFF if lo >= 0 else u >> (-lo)
        words.append(x)
    return words


@pytest.mark.parametrize("field", [0])
def test_ntt_two_pass_limb_model_with_signed_intermediate(field):
    """A 2^12-point transform as TWO passes of six stages, as ntt_run plans larger ones (2^20 = 10 + 10): the first pass ends in the fold and
    ntt_pack_signed9 (the folded value, possibly NEGATIVE, as a 256-bit two's-complement word), the second starts from
    ntt_unpack_signed9 (limb 8 by an arithmetic shift) and takes no carry pass in its first round.  Same checks as the one-pass model."""
    p = P[field]
    L, n, R = 12, 1 << 12, 6
    rng = random.Random(0x2A55)
    mul = _statement("field9_mul.inc")
    omega = rng.randrange(2, p)
    data = [rng.randrange(0, p) for _ in range(n)]
    want = _reference_network(data, omega, L, p)
    R9 = pow(2, 261, p)
    plimbs = [(p >> (29 * i)) & M29 for i in range(8)] + [p >> 232]
    tw_cache = {}

    def tw(e):
        if e not in tw_cache:
            v = pow(omega, e, p) * R9 % p
            tw_cache[e] = [(v >> (29 * i)) & M29 for i in range(8)] + [v >> 232]
        return tw_cache[e]

    def fmul(a, b):
        r, _ = _run(mul, {"a": _i32(a), "b": b}, field)
        return r

    add = lambda x, y: [s + t for s, t in zip(x, y)]
    sub = lambda x, y: [s - t for s, t in zip(x, y)]

    def fold(v):
        q = (v[8] + (1 << 21)) >> 22
        assert -64 <= q <= 64
        qp, c = [], 0
        for i in range(8):
            tq = q * plimbs[i] + c
            qp.append(tq & M29)
            c = tq >> 29
        qp.append(q * plimbs[8] + c)
        return _norm(_i32(sub(v, qp)))

    rev = lambda k: int(format(k, f"0{L}b")[::-1], 2)
    x = [_unpack9([(data[rev(k)] >> (32 * i)) & 0xFFFFFFFF for i in range(8)]) for k in range(n)]
    negatives = 0
    for s0 in (0, R):
        for u in range(0, R, 2):
            t = s0 + u
            nxt = list(x)
            for base in range(n):
                if base & (3 << t):
                    continue
                low = base & ((1 << t) - 1)
                i00, i01, i10, i11 = base, base + (1 << t), base + (2 << t), base + (3 << t)
                from_memory = u == 0                               # a pass's first round: unpacked (normalised) inputs, no carry pass
                e0, e1, e2, e3 = (x[i00], x[i01], x[i10], x[i11]) if from_memory else (_norm(x[i00]), x[i01], _norm(x[i10]), x[i11])
                if t > 0:
                    wA = tw(low << (L - t - 1))
                    e1, e3 = fmul(e1, wA), fmul(e3, wA)
                a0, a1 = add(e0, e1), sub(e0, e1)
                a2 = _norm(add(e2, e3)) if t == 0 else fmul(add(e2, e3), tw(low << (L - t - 2)))
                a3 = fmul(sub(e2, e3), tw((low + (1 << t)) << (L - t - 2)))
                nxt[i00], nxt[i10] = _i32(add(a0, a2)), _i32(sub(a0, a2))
                nxt[i01], nxt[i11] = _i32(add(a1, a3)), _i32(sub(a1, a3))
            x = nxt
        if s0 == 0:                                                # between the passes: fold, pack as two's complement, unpack with the sign
            mid = []
            for v in x:
                f = fold(v)
                negatives += f[8] < 0
                words = _pack9(f)
                back = _unpack9(words)
                back[8] = _s32(words[7]) >> 8                      # ntt_unpack_signed9
                assert back == f, (f, back)
                mid.append(back)
            x = mid
    assert negatives > n // 8                                      # the signed path is really exercised
    got = []
    for v in x:
        f = fold(v)
        g = _norm([f[i] + (plimbs[i] if f[8] < 0 else 0) for i in range(9)])
        val = _value(g)
        assert 0 <= val < p
        got.append(val)
    assert got == want
