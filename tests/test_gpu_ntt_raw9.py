"""The RAW9 intermediate form of the NTT passes (csrc/ntt.hip: raw9_load / raw9_store, H2_NTT_RAW9 = 1 folded, 2 unfolded where
the rounds allow) against the C oracle, elementwise.  The switch is read once per process, so every case runs in a child
process; H2_NTT_MAXR (a sweep knob of the pass planner) forces many short passes at small sizes, which reaches what the plans
of the measured sizes do not: passes of ONE stage (no fused load / store), odd stage counts, and -- in mode 2 -- a boundary
where the 11-round budget is spent and the value is folded after all.  `best_fft` halo2_proofs/src/arithmetic.rs:192-295,
`ifft` / `coeff_to_extended` / `extended_to_coeff` poly/domain.rs:375-383, 241-255, 303-325."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
import numpy as np
import halo2_amd as h
from halo2_amd import fields
from oracle import c_oracle as co
from oracle import pasta as o

def mont(field, v):
    return fields.scalar_limbs(v, field, True)

sizes = [int(x) for x in sys.argv[1].split(",")]
domains = [tuple(int(y) for y in x.split(":")) for x in sys.argv[2].split(",") if x]
for field in (h.FP, h.FQ):
    m = fields.MODULUS[field]
    for log_n in sizes:
        a = co.random_field(field, 9100 + log_n, 1 << log_n)
        omega = mont(field, o.omega_for(m, log_n))
        assert np.array_equal(h.best_fft(a.copy(), omega, log_n, field), co.best_fft(field, a, omega, log_n)), ("best_fft", field, log_n)
        w = co.random_field(field, 9200 + log_n, 1)[0]          # a random, non-root omega (benches/fft.rs:17)
        assert np.array_equal(h.best_fft(a.copy(), w, log_n, field), co.best_fft(field, a, w, log_n)), ("non-root", field, log_n)
    for j, k in domains:
        dom = h.EvaluationDomain(j, k, field)
        ref = o.EvaluationDomain(j, k, m)
        a = co.random_field(field, 9300 + 7 * k + j, dom.n)
        coeff_want = co.ifft(field, a, mont(field, ref.omega_inv), k, mont(field, ref.ifft_divisor))
        coeff = dom.lagrange_to_coeff(a.copy())
        assert np.array_equal(coeff, coeff_want), ("ifft", field, k)
        ext_want = co.coeff_to_extended(field, coeff_want, k, ref.extended_k, mont(field, ref.g_coset), mont(field, ref.g_coset_inv),
                                        mont(field, ref.extended_omega))
        assert np.array_equal(dom.coeff_to_extended(coeff), ext_want), ("coeff_to_extended", field, k)
        e = co.random_field(field, 9400 + k + j, 1 << ref.extended_k)
        back_want = co.extended_to_coeff(field, e, ref.extended_k, mont(field, ref.g_coset), mont(field, ref.g_coset_inv),
                                         mont(field, ref.extended_omega_inv), mont(field, ref.extended_ifft_divisor))
        assert np.array_equal(dom.extended_to_coeff(e.copy()), back_want[: dom.n * dom.quotient_poly_degree]), ("extended_to_coeff", field, k)
print("RAW9_OK")
"""


def run_child(mode, maxr, sizes, domains):
    env = dict(os.environ, H2_NTT_RAW9=str(mode), PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    if maxr:
        env["H2_NTT_MAXR"] = str(maxr)
    r = subprocess.run([sys.executable, "-c", CHILD, ",".join(str(s) for s in sizes), ",".join(f"{j}:{k}" for j, k in domains)],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RAW9_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


@pytest.mark.parametrize("mode", [1, 2])
def test_raw9_measured_plans(mode):
    """the plans of the measured sizes: 2^20 (10 + 10 stages), 2^21 / 2^22 (three passes), and the domain transforms of config 4
    (fused load / store factors around the RAW9 vector): 2^20 -> 2^21"""
    run_child(mode, 0, [20, 21, 22], [(3, 20)])


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("maxr,sizes", [(1, [11, 13]), (2, [12]), (3, [13, 14]), (5, [14])])
def test_raw9_short_passes(mode, maxr, sizes):
    """many short passes: one-stage passes (13 of them spend the 11-round budget: mode 2 folds at a boundary), two-stage passes
    (fused load and store in the same round), odd stage counts (radix-2 tail, unfused store), uneven splits"""
    run_child(mode, maxr, sizes, [(3, max(sizes) - 1)])
