"""Shared by the CPU and GPU plonk tests: a constraint system shaped like the reference's own test circuit
(halo2_proofs/tests/plonk_api.rs:25-400) in the lowered form halo2_amd/plonk.py takes, and satisfying witnesses for it."""
from halo2_amd.plonk import ConstraintSystem

SA, SB, SC, SM, SP, SL = range(6)
A, B, C_ = range(3)


def make_cs(variant="full"):
    if variant == "gates_only":                 # no lookup, no permutation argument, no instance column
        return ConstraintSystem(
            num_fixed_columns=6, num_advice_columns=3, num_instance_columns=0,
            gates=[lambda q: q.advice(A) * q.fixed(SA) + q.advice(B) * q.fixed(SB) + q.advice(A) * q.advice(B) * q.fixed(SM)
                   - q.advice(C_) * q.fixed(SC)],
            advice_queries=[(A, 0), (B, 0), (C_, 0)], instance_queries=[], fixed_queries=[(c, 0) for c in range(4)],
            degree=3, blinding_factors=5)
    if variant == "two_lookups":                # a second, degree-2 tuple lookup (a, a^2) in (sl, sl^2): constraint degree 6
        cs = make_cs()
        cs.lookups = cs.lookups + [([lambda q: q.advice(A), lambda q: q.advice(A) * q.advice(A)],
                                    [lambda q: q.fixed(SL), lambda q: q.fixed(SL) * q.fixed(SL)])]
        cs.degree = 6
        return cs
    return ConstraintSystem(
        num_fixed_columns=6, num_advice_columns=3, num_instance_columns=1,
        gates=[lambda q: q.advice(A) * q.fixed(SA) + q.advice(B) * q.fixed(SB) + q.advice(A) * q.advice(B) * q.fixed(SM)
               - q.advice(C_) * q.fixed(SC),                                     # plonk_api.rs:281-296 without the d * e term
               lambda q: q.fixed(SP) * (q.advice(A) - q.instance(0))],           # plonk_api.rs:298-305
        advice_queries=[(A, 0), (B, 0), (C_, 0)], instance_queries=[(0, 0)], fixed_queries=[(c, 0) for c in range(6)],
        permutation_columns=[("advice", A), ("advice", B), ("advice", C_)],
        lookups=[([lambda q: q.advice(A)], [lambda q: q.fixed(SL)])],            # plonk_api.rs:276-279
        degree=4, blinding_factors=5)


def make_witness(rnd, m, n, usable, break_gate=False):
    table_vals = [rnd.randrange(m) for _ in range(8)]
    fixed = [[0] * n for _ in range(6)]
    a, b, c = [0] * n, [0] * n, [0] * n
    groups = {}                                    # value classes that get copy constraints
    for r in range(usable):
        fixed[SL][r] = table_vals[r % 8]
        a[r] = rnd.choice(table_vals)
        b[r] = c[r - 1] if r and r % 3 == 0 else rnd.randrange(m)          # every third row reuses the previous output
        if r % 2:
            fixed[SM][r], fixed[SC][r] = 1, 1
            c[r] = a[r] * b[r] % m
        else:
            fixed[SA][r], fixed[SB][r], fixed[SC][r] = 1, 1, 1
            c[r] = (a[r] + b[r]) % m
        if r and r % 3 == 0:
            groups.setdefault(("chain", r), []).extend([(C_, r - 1), (B, r)])
        groups.setdefault(("a", a[r]), []).append((A, r))                    # equal `a` cells are tied together
    fixed[SP][0] = 1
    if break_gate:
        c[5] = (c[5] + 1) % m
    mapping = [[(col, r) for r in range(n)] for col in range(3)]
    for cells in groups.values():
        if len(cells) > 1:
            for i, (col, r) in enumerate(cells):
                mapping[col][r] = cells[(i + 1) % len(cells)]
    return fixed, [a, b, c], mapping, [[a[0]]]


