"""The generated 8 x 32 Montgomery multiplier (halo2_amd/csrc/field_mul_sched.inc, written by gen_field_mul.py; R = 2^256, the reference's
memory form -- used wherever data is converted or a kernel is rare: table builds, the sort's helpers, output conversion) interpreted on the
CPU, instruction by instruction: `v_mad_u64_u32` with its carry-out, the carry-chained `v_addc_co_u32` / `v_sub_co_u32` on SGPR pairs and
vcc.  The product is congruent to a b 2^-256 and stays below 2p for canonical operands (fe_mul_sched subtracts p at most once) and below
2p + 2^130 for the lazy operands of fe_mul_lazy (field.cuh); and the schedule respects the gfx940 / gfx950 hazard the generator exists for --
a carry written to an SGPR by a VALU instruction is read no sooner than three instructions later (hipcc pads its own code, not asm text).
The device checks the same statement against the C oracle (tests/native/field_check.hip); this needs no GPU."""
import os
import random
import re

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "halo2_amd", "csrc")
P = {0: 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001,
     1: 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001}
M32, M64 = (1 << 32) - 1, (1 << 64) - 1


def _instructions():
    text = open(os.path.join(CSRC, "field_mul_sched.inc")).read()
    asm = re.search(r'asm\("(.*?)"\n\s*:', text, re.S).group(1)
    return [line.split(None, 1) for line in asm.split("\\n\\t")]


def _run(ins, a, b, p):
    reg = {f"%[a{i}]": (a >> (32 * i)) & M32 for i in range(8)}
    reg.update({f"%[b{i}]": (b >> (32 * i)) & M32 for i in range(8)})
    reg.update({"%[k1]": (p >> 32) & M32, "%[k2]": (p >> 64) & M32, "%[k3]": (p >> 96) & M32, "%[k7]": p >> 224})
    flag = {}                                       # carry registers (SGPR pairs, vcc): one bit in this single-lane model

    def rd(tok):
        return reg[tok] if tok in reg else int(tok, 0) & M32

    def rd64(tok):
        if tok.startswith("v["):
            lo, hi = re.match(r"v\[(\d+):(\d+)\]", tok).groups()
            return reg[f"v{lo}"] | (reg[f"v{hi}"] << 32)
        return int(tok, 0) & M64

    for op, rest in ins:
        t = [x.strip() for x in rest.split(",")]
        if op == "s_nop":
            continue
        if op == "v_mov_b32":
            reg[t[0]] = rd(t[1])
        elif op == "v_mad_u64_u32":                 # D = S0 * S1 + S2 (64-bit), carry-out to the SGPR pair
            val = rd(t[2]) * rd(t[3]) + rd64(t[4])
            lo, hi = re.match(r"v\[(\d+):(\d+)\]", t[0]).groups()
            reg[f"v{lo}"], reg[f"v{hi}"] = val & M32, (val >> 32) & M32
            flag[t[1]] = val >> 64
            assert flag[t[1]] in (0, 1)
        elif op == "v_sub_co_u32":                  # D = S0 - S1, borrow-out
            s0, s1 = rd(t[2]), rd(t[3])
            reg[t[0]] = (s0 - s1) & M32
            flag[t[1]] = 1 if s1 > s0 else 0
        elif op == "v_addc_co_u32":                 # D = S0 + S1 + carry-in, carry-out
            val = rd(t[2]) + rd(t[3]) + flag[t[4]]
            reg[t[0]] = val & M32
            flag[t[1]] = val >> 32
        else:
            raise AssertionError(f"instruction form the model does not know: {op} {rest}")
    return sum(reg[f"%[r{i}]"] << (32 * i) for i in range(8))


@pytest.mark.parametrize("field", [0, 1])
def test_generated_8x32_multiplier_against_big_integers(field):
    p = P[field]
    ins = _instructions()
    assert sum(op == "v_mad_u64_u32" for op, _ in ins) == 96
    rinv = pow(1 << 256, -1, p)
    rng = random.Random(0x8832 + field)
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (1 << 254), (1 << 254) - 1, (1 << 255) - 1 if (1 << 255) - 1 < p else p - 3]
    pairs = [(x, y) for x in edge for y in edge] + [(rng.randrange(p), rng.randrange(p)) for _ in range(300)]
    for a, b in pairs:
        r = _run(ins, a, b, p)
        assert r % p == a * b * rinv % p
        assert r < 2 * p                                           # fe_mul_sched: one conditional subtraction is enough
    # lazy operands (fe_mul_lazy: no subtraction after the product): a, b in [0, 2p + d) give a result below 2p + d again (d ~ 2^130)
    d = 1 << 130
    for _ in range(300):
        a, b = rng.randrange(2 * p + d), rng.randrange(2 * p + d)
        r = _run(ins, a, b, p)
        assert r % p == a * b * rinv % p and r < 2 * p + d


def test_generated_8x32_multiplier_respects_the_sgpr_hazard():
    """VALU writes SGPR / vcc as a carry -> a VALU instruction reading it as a carry needs two wait states in between on gfx940 / gfx950:
    the generator schedules independent instructions (or s_nop) there.  Checked on the text: the distance from a carry's last writer to
    every reader is at least three instructions."""
    ins = _instructions()
    last_write = {}
    closest = 99
    for idx, (op, rest) in enumerate(ins):
        t = [x.strip() for x in rest.split(",")] if op != "s_nop" else []
        reads, writes = [], []
        if op == "v_mad_u64_u32":
            writes = [t[1]]
        elif op == "v_sub_co_u32":
            writes = [t[1]]
        elif op == "v_addc_co_u32":
            reads, writes = [t[4]], [t[1]]
        for r_ in reads:
            assert r_ in last_write, (idx, op, rest)
            gap = idx - last_write[r_]
            closest = min(closest, gap)
            assert gap >= 3, (idx, op, rest, gap)
        for w in writes:
            last_write[w] = idx
    assert closest == 3                                            # the schedule is tight somewhere: the rule is exercised, not vacuous
