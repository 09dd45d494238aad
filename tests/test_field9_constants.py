"""The constants of the carry-free field layer as SHIPPED (halo2_amd/csrc/field9_consts.inc, written by gen_field9_consts.py), parsed out
of the include file and checked against big integers on the CPU: the modulus limbs the generated multipliers multiply by, 1 / R^2 and the
form-conversion factors in M9 form (R9 = 2^261), 32 in the reference's Montgomery form, and the shifted moduli the canonicalisation
subtracts.  The device checks the arithmetic built on them (tests/native/field_check.hip); this needs no GPU."""
import os
import re

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "halo2_amd", "csrc")
SRC = open(os.path.join(CSRC, "field9_consts.inc")).read()
P = {"FP": 0x40000000000000000000000000000000224698fc094cf91b992d30ed00000001,
     "FQ": 0x40000000000000000000000000000000224698fc0994a8dd8c46eb2100000001}


def _v9(limbs):
    assert len(limbs) == 9 and all(0 <= x < (1 << 29) for x in limbs[:8])          # normalised: what "N" means in field9.cuh
    return sum(x << (29 * i) for i, x in enumerate(limbs))


def _lists(text):
    return [[int(x, 16) for x in re.findall(r"0x[0-9a-f]+", body)] for body in re.findall(r"\{\{(.*?)\}\}", text)]


def _function(name):
    return re.search(r"%s\(\w*\s*\w*\) \{(.*?)\n\}" % name, SRC, re.S).group(1)


@pytest.mark.parametrize("field", ["FP", "FQ"])
def test_field9_constants(field):
    p = P[field]
    k = 0 if field == "FP" else 1
    limbs = [(p >> (29 * i)) & ((1 << 29) - 1) for i in range(8)] + [p >> 232]
    assert limbs[0] == 1 and limbs[5:8] == [0, 0, 0] and limbs[8] == 1 << 22       # the sparse shape the multiplier's reduction relies on
    m = re.search(r"struct Mod9<%s> \{\s*static constexpr i32 P1 = (0x[0-9a-f]+), P2 = (0x[0-9a-f]+), P3 = (0x[0-9a-f]+), P4 = (0x[0-9a-f]+);" % field, SRC)
    assert [int(x, 16) for x in m.groups()] == limbs[1:5]
    assert (-pow(p, -1, 1 << 29)) % (1 << 29) == (1 << 29) - 1                     # -p^-1 = -1 mod 2^29: the quotient digit is the negated low limb
    want = {"fe9_k_in": pow(2, 266, p), "fe9_k_out": pow(2, 256, p), "fe9_one": pow(2, 261, p), "fe9_r2": pow(2, 522, p), "fe9_p16": 16 * p}
    for name, value in want.items():
        assert _v9(_lists(_function(name))[k]) == value, name
    # the materialised one of the hot loop: nine v_mov literals per field
    movs = [int(x, 16) for x in re.findall(r"v_mov_b32 %0, (0x[0-9a-f]+)", _function("fe9_one_here"))]
    assert _v9(movs[9 * k:9 * k + 9]) == pow(2, 261, p)
    # 32 in the reference's Montgomery form, 8 x 32 limbs
    k32 = [int(x, 16) for x in re.findall(r"0x[0-9a-f]+", re.findall(r"\{\{(.*?)\}\}", _function("fe_k32"))[k])]
    assert sum(x << (32 * i) for i, x in enumerate(k32)) == (32 << 256) % p
    # p << sh for sh = 4 .. 0: (FP, FQ) pairs in that order
    shl = _lists(_function("fe9_p_shl"))
    assert len(shl) == 10
    for row, sh in enumerate((4, 3, 2, 1, 0)):
        assert _v9(shl[2 * row + k]) == p << sh, sh
