"""ctypes loader for libhalo2_mi355x.so (the C ABI declared in include/halo2_mi355x.h).

The product path has no CPU fallback: if the shared library is missing this raises, and every
entry point returns H2_ERR_NODEV (surfaced as RuntimeError) when no gfx950 device is present."""
from __future__ import annotations

import ctypes as C
import os

H2_OK, H2_ERR_ARGS, H2_ERR_HIP, H2_ERR_NODEV, H2_ERR_HANDLE, H2_ERR_DECODE, H2_ERR_LOOKUP, H2_ERR_PEER = 0, 1, 2, 3, 4, 5, 6, 7
FP, FQ = 0, 1
PALLAS, VESTA = 0, 1
FORM_CANONICAL, FORM_MONTGOMERY = 0, 1
OUT_JACOBIAN, OUT_AFFINE = 0, 1

_HERE = os.path.dirname(os.path.abspath(__file__))
# H2_LIB_PATH: another build of the same sources -- the laboratory build with its A/B switches live (build/ab/libhalo2_mi355x_ab.so,
# `make -C halo2_amd/csrc ab`), which the A/B parity tests and bench/tools load in child processes.  The product path never sets it.
LIB_PATH = os.environ.get("H2_LIB_PATH") or os.path.join(_HERE, "libhalo2_mi355x.so")
_lib = None

u64p = C.POINTER(C.c_uint64)
vp = C.c_void_p
IPA_SWITCH_DEFAULT = 0xFFFFFFFF
IPA_WRITE_POINT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64))     # h2_ipa_write_point_fn
IPA_SQUEEZE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64))         # h2_ipa_squeeze_fn

# every symbol include/halo2_mi355x.h declares: name -> (argtypes, restype)
SIGNATURES = {
    "h2_device_count": ([], C.c_int),
    "h2_current_device": ([], C.c_int),
    "h2_init": ([C.c_int], C.c_int),
    "h2_last_error": ([], C.c_char_p),
    "h2_trim": ([], C.c_int),
    "h2_msm_window_bits": ([C.c_size_t], C.c_int),
    "h2_set_option": ([C.c_char_p, C.c_double], C.c_int),
    "h2_msm": ([C.c_int, u64p, u64p, C.c_size_t, C.c_int, C.c_int, u64p], C.c_int),
    "h2_bases_register": ([C.c_int, u64p, C.c_size_t, C.c_int, C.POINTER(C.c_uint64)], C.c_int),
    "h2_bases_register_ex": ([C.c_int, u64p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_uint64)], C.c_int),
    "h2_commit_column_window_bits": ([C.c_size_t], C.c_int),
    "h2_bases_set_blind_base": ([C.c_uint64, u64p, C.c_int], C.c_int),
    "h2_bases_info": ([C.c_uint64, C.POINTER(C.c_size_t), C.POINTER(C.c_int), C.POINTER(C.c_int)], C.c_int),
    "h2_bases_blind_base_set": ([C.c_uint64], C.c_int),
    "h2_bases_free": ([C.c_uint64], C.c_int),
    "h2_commit": ([C.c_uint64, u64p, C.c_size_t, u64p, u64p, C.c_int, C.c_int, u64p], C.c_int),
    "h2_ntt": ([C.c_int, u64p, C.c_uint, u64p, C.c_int], C.c_int),
    "h2_ifft": ([C.c_int, u64p, C.c_uint, u64p, u64p, C.c_int], C.c_int),
    "h2_coeff_to_extended": ([C.c_int, u64p, u64p, C.c_uint, C.c_uint, u64p, u64p, u64p, C.c_int], C.c_int),
    "h2_extended_to_coeff": ([C.c_int, u64p, C.c_uint, u64p, u64p, u64p, u64p, C.c_int], C.c_int),
    "h2_divide_by_vanishing_poly": ([C.c_int, u64p, C.c_uint, u64p, C.c_size_t, C.c_int], C.c_int),
    "h2_divide_by_vanishing_poly_device": ([C.c_int, vp, C.c_uint, u64p, C.c_size_t, C.c_int, vp], C.c_int),
    "h2_msm_device": ([C.c_int, vp, vp, C.c_size_t, C.c_int, C.c_int, vp, vp], C.c_int),
    "h2_commit_device": ([C.c_uint64, vp, C.c_size_t, vp, vp, C.c_int, C.c_int, vp, vp], C.c_int),
    "h2_commit_batch_device": ([C.c_uint64, C.POINTER(vp), C.c_size_t, C.c_size_t, vp, C.POINTER(vp), C.c_int, C.c_int,
                                C.POINTER(vp), vp], C.c_int),
    "h2_msm_batch_device": ([C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_size_t), C.c_size_t, C.c_int, C.c_int,
                             C.POINTER(vp), vp], C.c_int),
    "h2_ntt_device": ([C.c_int, vp, C.c_uint, u64p, C.c_int, vp], C.c_int),
    "h2_ifft_device": ([C.c_int, vp, C.c_uint, u64p, u64p, C.c_int, vp], C.c_int),
    "h2_ntt_batch_device": ([C.c_int, C.POINTER(vp), C.c_size_t, C.c_uint, u64p, C.c_int, vp], C.c_int),
    "h2_ifft_batch_device": ([C.c_int, C.POINTER(vp), C.c_size_t, C.c_uint, u64p, u64p, C.c_int, vp], C.c_int),
    "h2_coeff_to_extended_device": ([C.c_int, vp, vp, C.c_uint, C.c_uint, u64p, u64p, u64p, C.c_int, vp], C.c_int),
    "h2_extended_to_coeff_device": ([C.c_int, vp, C.c_uint, u64p, u64p, u64p, u64p, C.c_int, vp], C.c_int),
    "h2_points_sum": ([C.c_int, u64p, C.c_size_t, u64p], C.c_int),
    "h2_generator_collapse": ([C.c_int, u64p, C.c_size_t, u64p, C.c_int], C.c_int),
    "h2_generator_collapse_device": ([C.c_int, vp, C.c_size_t, u64p, C.c_int, vp], C.c_int),
    "h2_fold_scalars": ([C.c_int, u64p, C.c_size_t, u64p, C.c_int], C.c_int),
    "h2_fold_scalars_device": ([C.c_int, vp, C.c_size_t, u64p, C.c_int, vp], C.c_int),
    "h2_ipa_round_scalars_device": ([C.c_int, vp, C.c_uint, C.c_uint, u64p, C.c_int, vp, vp, vp], C.c_int),
    "h2_ipa_rounds_device": ([C.c_int, C.c_uint, C.c_uint, C.c_uint64, C.c_int, vp, vp, u64p, u64p, u64p, vp, vp, IPA_WRITE_POINT_FN,
                             IPA_SQUEEZE_FN, vp, u64p, u64p, vp], C.c_int),
    "h2_ipa_rounds": ([C.c_int, C.c_uint, C.c_uint, C.c_uint64, C.c_int, u64p, u64p, u64p, u64p, u64p, IPA_WRITE_POINT_FN, IPA_SQUEEZE_FN, vp,
                      u64p, u64p], C.c_int),
    "h2_ipa_default_switch_rounds": ([C.c_uint, C.c_int], C.c_uint),
    "h2_open_device": ([C.c_int, C.c_uint, C.c_uint64, C.c_uint64, C.c_int, C.c_uint, u64p, vp, u64p, u64p, vp, u64p, u64p, IPA_WRITE_POINT_FN,
                       IPA_SQUEEZE_FN, vp, u64p, u64p, vp], C.c_int),
    "h2_open_device_host_s": ([C.c_int, C.c_uint, C.c_uint64, C.c_uint64, C.c_int, C.c_uint, u64p, vp, u64p, u64p, u64p, u64p, u64p, IPA_WRITE_POINT_FN,
                              IPA_SQUEEZE_FN, vp, u64p, u64p, vp], C.c_int),
    "h2_open": ([C.c_int, C.c_uint, C.c_uint64, C.c_uint64, C.c_int, C.c_uint, u64p, u64p, u64p, u64p, u64p, u64p, u64p, IPA_WRITE_POINT_FN,
                IPA_SQUEEZE_FN, vp, u64p, u64p], C.c_int),
    "h2_transcript_new": ([C.c_int, C.POINTER(C.c_uint64)], C.c_int),
    "h2_transcript_free": ([C.c_uint64], C.c_int),
    "h2_transcript_common_point": ([C.c_uint64, u64p, C.c_int], C.c_int),
    "h2_transcript_write_point": ([C.c_uint64, u64p, C.c_int], C.c_int),
    "h2_transcript_common_scalar": ([C.c_uint64, u64p], C.c_int),
    "h2_transcript_write_scalar": ([C.c_uint64, u64p], C.c_int),
    "h2_transcript_squeeze_challenge": ([C.c_uint64, u64p], C.c_int),
    "h2_transcript_bytes": ([C.c_uint64, C.POINTER(C.c_uint8), C.c_size_t, C.POINTER(C.c_size_t)], C.c_int),
    "h2_transcript_cb_write_point": ([vp, u64p], C.c_int),
    "h2_transcript_cb_squeeze": ([vp, u64p], C.c_int),
    "h2_ipa_collapsed_generators_device": ([C.c_uint64, C.c_uint, C.c_uint, u64p, C.c_int, vp, vp], C.c_int),
    "h2_bases_register_device": ([C.c_int, vp, C.c_size_t, C.c_int, C.POINTER(C.c_uint64)], C.c_int),
    "h2_lagrange_basis": ([C.c_int, u64p, u64p, C.c_uint, C.c_int], C.c_int),
    "h2_lagrange_basis_device": ([C.c_int, vp, vp, C.c_uint, C.c_int, vp], C.c_int),
    "h2_eval_polynomial": ([C.c_int, u64p, C.c_size_t, u64p, C.c_int, u64p], C.c_int),
    "h2_eval_polynomial_device": ([C.c_int, vp, C.c_size_t, u64p, C.c_int, vp, vp], C.c_int),
    "h2_inner_product": ([C.c_int, u64p, u64p, C.c_size_t, C.c_int, u64p], C.c_int),
    "h2_inner_product_device": ([C.c_int, vp, vp, C.c_size_t, C.c_int, vp, vp], C.c_int),
    "h2_kate_division": ([C.c_int, u64p, C.c_size_t, u64p, C.c_int, u64p], C.c_int),
    "h2_kate_division_device": ([C.c_int, vp, C.c_size_t, u64p, C.c_int, vp, vp], C.c_int),
    "h2_powers": ([C.c_int, u64p, C.c_size_t, C.c_int, u64p], C.c_int),
    "h2_powers_device": ([C.c_int, u64p, C.c_size_t, C.c_int, vp, vp], C.c_int),
    "h2_scale_add": ([C.c_int, u64p, u64p, u64p, C.c_size_t, C.c_int], C.c_int),
    "h2_scale_add_device": ([C.c_int, vp, u64p, vp, C.c_size_t, C.c_int, vp], C.c_int),
    "h2_batch_invert": ([C.c_int, u64p, C.c_size_t, C.c_int], C.c_int),
    "h2_batch_invert_device": ([C.c_int, vp, C.c_size_t, C.c_int, vp], C.c_int),
    "h2_grand_product": ([C.c_int, u64p, C.c_size_t, u64p, C.c_int, u64p], C.c_int),
    "h2_grand_product_device": ([C.c_int, vp, C.c_size_t, u64p, C.c_int, vp, vp], C.c_int),
    "h2_evaluate_device": ([C.c_int, C.c_int, C.POINTER(C.c_uint32), C.c_size_t, u64p, C.c_size_t, C.POINTER(vp), C.c_size_t, C.c_uint,
                            u64p, vp, vp], C.c_int),
    "h2_sort_device": ([C.c_int, vp, C.c_size_t, C.c_int, vp], C.c_int),
    "h2_permute_expression_pair_device": ([C.c_int, vp, vp, C.c_size_t, C.c_int, vp, vp, vp], C.c_int),
    "h2_points_compress": ([C.c_int, u64p, C.c_size_t, C.c_int, C.POINTER(C.c_uint8)], C.c_int),
    "h2_points_compress_device": ([C.c_int, vp, C.c_size_t, C.c_int, vp, vp], C.c_int),
    "h2_points_decompress": ([C.c_int, C.POINTER(C.c_uint8), C.c_size_t, C.c_int, u64p], C.c_int),
    "h2_points_decompress_device": ([C.c_int, vp, C.c_size_t, C.c_int, vp, vp], C.c_int),
    "h2_profile_enable": ([C.c_int], C.c_int),
    "h2_points_sum_device": ([C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p], C.c_int),
    "h2_commit_batch_multi": ([C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, C.c_void_p,
                               C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(C.c_void_p)], C.c_int),
    "h2_msm_split_multi": ([C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_void_p], C.c_int),
    "h2_rccl_unique_id": ([C.c_void_p], C.c_int),
    "h2_rccl_init": ([C.c_void_p, C.c_int, C.c_int], C.c_int),
    "h2_rccl_finalize": ([], C.c_int),
    "h2_commit_split_rccl_device": ([C.c_uint64, C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p], C.c_int),
    "h2_commit_range_device": ([C.c_uint64, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p], C.c_int),
    "h2_msm_split_rccl_device": ([C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p], C.c_int),
    "h2_hash_to_curve": ([C.c_int, C.c_char_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p], C.c_int),
    "h2_hash_to_curve_device": ([C.c_int, C.c_char_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p], C.c_int),
    "h2_commit_window_bits": ([C.c_size_t], C.c_int),
    "h2_commit_pair_supported": ([C.c_size_t], C.c_int),
    "h2_commit_pair_device": ([C.c_uint64, C.c_void_p, C.c_size_t, C.c_uint, C.c_int, C.c_int, C.c_void_p, C.c_void_p], C.c_int),
    "h2_profile_read": ([C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)], C.c_int),
    "h2_profile_read_busy": ([C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)], C.c_int),
    "h2_debug_timeline": ([C.POINTER(C.c_ulonglong), C.c_uint], C.c_int),
}


def _share_hip_runtime_with_torch():
    """One process must hold ONE HIP runtime.  The PyTorch wheel bundles its own libamdhip64.so
    (SONAME libamdhip64.so.7, same as /opt/rocm's); if this library pulled in /opt/rocm's copy first,
    torch would later load its own second copy and find no GPU.  Pre-loading torch's copy makes the
    dynamic loader resolve our NEEDED libamdhip64.so.7 to it.  No torch installed -> /opt/rocm's."""
    if os.environ.get("H2_NO_TORCH_RUNTIME"):
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C halo2_amd/csrc). There is no CPU fallback.")
        _share_hip_runtime_with_torch()
        _lib = C.CDLL(LIB_PATH)
        for name, (args, res) in SIGNATURES.items():
            fn = getattr(_lib, name)
            fn.argtypes = args
            fn.restype = res
    return _lib


class H2Error(RuntimeError):
    pass


class ConstraintSystemFailure(ValueError):
    """plonk::Error::ConstraintSystemFailure (plonk/error.rs): the witness does not satisfy the circuit."""


def check(rc: int, what: str):
    if rc == H2_OK:
        return
    if rc == H2_ERR_ARGS:
        # the reference panics (assert_eq!) on bad lengths: arithmetic.rs:144, :205
        raise ValueError(f"{what}: bad arguments")
    if rc == H2_ERR_LOOKUP:
        raise ConstraintSystemFailure(f"{what}: an input value does not occur in the table")    # lookup/prover.rs:609-611
    if rc == H2_ERR_DECODE:
        raise ValueError(f"{what}: invalid point encoding")      # the reference returns io::Error (commitment.rs:193-198)
    msg = lib().h2_last_error().decode()
    names = {H2_ERR_HIP: "HIP failure", H2_ERR_NODEV: "no MI355X device", H2_ERR_HANDLE: "bad handle", H2_ERR_PEER: "another rank failed"}
    raise H2Error(f"{what}: {names.get(rc, rc)}: {msg}")
