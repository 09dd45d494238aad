"""halo2_amd: MI355X-native (gfx950) implementation of the halo2_proofs prover hot path --
Pasta MSM (`best_multiexp`, `Params::commit*`) and NTT (`best_fft`, `EvaluationDomain`) -- behind a
C ABI (include/halo2_mi355x.h, halo2_amd/libhalo2_mi355x.so).  This package is the thin host-side mirror
of the reference's interface; all compute is hand-written HIP in halo2_amd/csrc/."""
from ._lib import (FORM_CANONICAL, FORM_MONTGOMERY, FP, FQ, LIB_PATH, PALLAS, VESTA, H2Error, lib)  # noqa: F401
from .arithmetic import (batch_invert, best_fft, best_fft_batch, best_multiexp, compute_inner_product, eval_polynomial,  # noqa: F401
                         fold_scalars, grand_product, kate_division, msm_window_bits, parallel_generator_collapse,
                         points_sum, powers, scale_add, small_multiexp)
from .commitment import Blind, Params, hash_to_curve, lagrange_basis, points_from_bytes, points_to_bytes  # noqa: F401
from .domain import EvaluationDomain  # noqa: F401
from .poly import Coeff, ExtendedLagrangeCoeff, LagrangeCoeff, Polynomial  # noqa: F401
