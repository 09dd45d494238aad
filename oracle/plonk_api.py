"""TEST INFRASTRUCTURE ONLY (oracle).  `Params::new` and the keygen of the reference's own test circuit, restated, so that
the verifying key the reference PINS (halo2_proofs/tests/plonk_api.rs:585-984) can be recomputed here: the seven
`fixed_commitments` and twelve permutation commitments are `commit_lagrange` outputs over hash-to-curve generators,
i.e. reference-held golden values for the MSM / commitment path.

  * `params_new`           poly/commitment.rs:38-114 (g by hash-to-curve, g_lagrange by the point iFFT, w, u)
  * `PermutationAssembly`  plonk/permutation/keygen.rs:16-103 (the cycle-merging copy, sizes and all)
  * `keygen_columns`       tests/plonk_api.rs:229-400 laid out as SimpleFloorPlanner does (circuit/floor_planner/
                           single_pass.rs: a region starts at the max of its columns' next free rows; a table column's
                           unused usable rows are filled with its first value, :167-181) for K = 5
  * `pinned_commitments`   plonk/keygen.rs:229-243, permutation/keygen.rs:105-150
"""
from __future__ import annotations

from . import hash_to_curve as h2c
from . import pasta as o

CURVE_ID = {0: "pallas", 1: "vesta", "pallas": "pallas", "vesta": "vesta"}
DELTA = {m: pow(o.GENERATOR, 1 << o.S, m) for m in (o.P, o.Q)}            # ff::PrimeField::DELTA = GENERATOR^(2^S)


def params_new(curve, k: int, with_lagrange: bool = True):
    """Params::new(k) -> (g, g_lagrange, w, u), affine integer pairs."""
    cid = CURVE_ID[curve]
    bm, sm = o.CURVES[cid]
    n = 1 << k
    hasher = h2c.hash_to_curve(cid, "Halo2-Parameters")                   # commitment.rs:52
    g = [hasher(b"\x00" + i.to_bytes(4, "little")) for i in range(n)]     # :57-60: message[1..5] = i as LE u32
    g_lagrange = None
    if with_lagrange:
        # :77-88: best_fft over points with alpha_inv = omega^-1, then * 2^-k.  Definition form (k is small here).
        omega_inv = pow(o.omega_for(sm, k), -1, sm)
        minv = pow(n, -1, sm)
        g_lagrange = []
        for i in range(n):
            acc = None
            for j in range(n):
                acc = o.ec_add(acc, o.ec_mul(pow(omega_inv, i * j, sm) * minv % sm, g[j], bm), bm)
            g_lagrange.append(acc)
    return g, g_lagrange, hasher(b"\x01"), hasher(b"\x02")                # :102-104


class PermutationAssembly:
    """plonk/permutation/keygen.rs:16-103."""

    def __init__(self, n: int, n_columns: int):
        self.mapping = [[(i, j) for j in range(n)] for i in range(n_columns)]
        self.aux = [[(i, j) for j in range(n)] for i in range(n_columns)]
        self.sizes = [[1] * n for _ in range(n_columns)]

    def copy(self, lc: int, lr: int, rc: int, rr: int) -> None:
        left, right = self.aux[lc][lr], self.aux[rc][rr]
        if left == right:
            return
        if self.sizes[left[0]][left[1]] < self.sizes[right[0]][right[1]]:
            left, right = right, left
        self.sizes[left[0]][left[1]] += self.sizes[right[0]][right[1]]
        i = right
        while True:
            self.aux[i[0]][i[1]] = left
            i = self.mapping[i[0]][i[1]]
            if i == right:
                break
        self.mapping[lc][lr], self.mapping[rc][rr] = self.mapping[rc][rr], self.mapping[lc][lr]


K = 5
# configure() order (plonk_api.rs:240-260): advice e=0 a=1 b=2 c=3 d=4; fixed sf=0 sm=1 sa=2 sb=3 sc=4 sp=5 sl=6
SF, SM, SA, SB, SC, SP, SL = range(7)
# enable_equality order (:246-248, :307-315): a, b, c, sf, e, d, p, sm, sa, sb, sc, sp
PERM_A, PERM_B, PERM_C = 0, 1, 2
N_PERM_COLUMNS = 12
BLINDING_FACTORS = 5                                                      # plonk/circuit.rs blinding_factors(): max(3, 1) + 2


def keygen_columns(m: int = o.P):
    """The fixed columns and the permutation mapping `keygen_vk(&params, &empty_circuit)` builds for plonk_api at K = 5."""
    n = 1 << K
    usable = n - (BLINDING_FACTORS + 1)
    fixed = [[0] * n for _ in range(7)]
    perm = PermutationAssembly(n, N_PERM_COLUMNS)
    fixed[SP][0] = 1                                                      # public_input region at row 0 (:229-246)
    row = 1
    for _ in range(10):                                                   # synthesize (:380-394)
        mul, add = row, row + 1                                           # raw_multiply then raw_add, one row each
        fixed[SC][mul] = fixed[SM][mul] = 1                               # :143-146
        fixed[SA][add] = fixed[SB][add] = fixed[SC][add] = 1              # :194-203
        for _twice in range(2):                                           # copy() constrains twice (:214-219)
            perm.copy(PERM_A, mul, PERM_A, add)                           # cs.copy(a0, a1)
        for _twice in range(2):
            perm.copy(PERM_B, add, PERM_C, mul)                           # cs.copy(b1, c0)
        row += 2
    a = 2834758237 * o.zeta(m) % m                                        # :402  Fp::from(2834758237) * Fp::ZETA
    table = [2, a, a, 0]                                                  # :403-404
    for i, v in enumerate(table):
        fixed[SL][i] = v
    for r in range(len(table), usable):                                   # single_pass.rs:176-181
        fixed[SL][r] = table[0]
    return fixed, perm.mapping


def pinned_commitments(g_lagrange, w, bm: int, sm: int):
    """(fixed_commitments, permutation commitments) as affine integer pairs."""
    fixed, mapping = keygen_columns(sm)
    n = 1 << K
    commit_lagrange = lambda col: o.commit(g_lagrange, w, col, 1, bm)     # Blind::default() = 1 (commitment.rs:226-230)
    omega = o.omega_for(sm, K)
    om = [pow(omega, j, sm) for j in range(n)]
    sigmas = [[pow(DELTA[sm], mapping[i][j][0], sm) * om[mapping[i][j][1]] % sm for j in range(n)]
              for i in range(N_PERM_COLUMNS)]
    return [commit_lagrange(c) for c in fixed], [commit_lagrange(s) for s in sigmas]
