"""ctypes binding of libmmrec_hip.so (include/mmrec_hip.h).

There is NO fallback: if the library is missing or its ABI version differs, importing the ops fails
loudly.  `import torch` happens first on purpose: torch loads its bundled libamdhip64.so.7 and the
dynamic linker then resolves our DT_NEEDED libamdhip64.so.7 to that same runtime, so device
pointers and streams are shared between torch and the kernels.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int32, c_int64, c_size_t, c_uint64, c_void_p

import torch  # noqa: F401  (must precede CDLL, see module docstring)

_PKG = os.path.dirname(os.path.abspath(__file__))
# MMREC_HIP_LIB: load another build of the same library (kernel A/B measurements: tools/prof_topk_filter.py)
LIB_PATH = os.environ.get("MMREC_HIP_LIB") or os.path.join(_PKG, "lib", "libmmrec_hip.so")
ABI_VERSION = 14

_P = c_void_p  # every device/host pointer travels as void*

# name -> (restype, argtypes); mirrors include/mmrec_hip.h one to one
SIGNATURES = {
    "mmrec_abi_version": (c_int32, []),
    "mmrec_error_string": (c_char_p, [c_int32]),
    "mmrec_spmm_csr_f32": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_float,
                                     c_float, c_float, c_int32, _P, _P, c_int32, c_int32, _P, _P, _P]),
    "mmrec_spmm_csr_f32_layergcn": (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32,
                                              _P, _P, c_int32, c_int32, _P, _P, _P]),
    "mmrec_spmm_plan_count": (c_int32, [_P, c_int32, c_int32, _P, _P]),
    "mmrec_spmm_plan_fill": (c_int32, [_P, c_int32, c_int32, _P, _P]),
    "mmrec_cos_scale_fwd_f32": (c_int32, [_P, _P, _P, _P, _P, c_int32, c_int32, _P]),
    "mmrec_cos_scale_bwd_f32": (c_int32, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, _P]),
    "mmrec_bpr_workspace_bytes": (c_size_t, [c_int32]),
    "mmrec_bpr_fwd_f32": (c_int32, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_float, _P,
                                    _P, _P, _P]),
    "mmrec_spmm_rows_f32": (c_int32, [_P, _P, _P, _P, _P, c_int32, _P, c_int32, c_int32, c_int32, _P, _P]),
    "mmrec_spmm_rows_any_f32": (c_int32, [_P, _P, _P, _P, _P, c_int32, _P, c_int32, c_int32, c_int32, c_int32, _P, _P]),
    "mmrec_spmm_push_rows_f32": (c_int32, [_P, _P, _P, _P, c_float, _P, c_int32, c_int32, _P, _P, _P]),
    "mmrec_bpr_dots_f32": (c_int32, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, _P, _P]),
    "mmrec_bpr_loss_from_dots_f32": (c_int32, [_P, c_int32, c_int32, c_float, _P, _P, _P, _P]),
    "mmrec_bpr_bwd_f32": (c_int32, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, _P, _P, c_float, _P, _P,
                                    _P, _P]),
    "mmrec_gather_sqnorm_fwd_f32": (c_int32, [_P, _P, c_int32, c_int32, _P, _P, _P]),
    "mmrec_gather_scale_add_bwd_f32": (c_int32, [_P, _P, c_int32, c_int32, _P, _P, _P]),
    "mmrec_infonce_workspace_bytes": (c_size_t, [c_int32]),
    "mmrec_infonce_fwd_f32": (c_int32, [_P, _P, _P, c_int32, c_int32, c_float, _P, _P, _P]),
    "mmrec_infonce_bwd_f32": (c_int32, [_P, c_int32, c_int32, c_float, _P, _P, _P, _P, _P]),
    "mmrec_linear_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "mmrec_linear_fwd_f32": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int32, _P, _P]),
    "mmrec_linear_fwd_split_f32": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int32, _P, _P]),
    "mmrec_linear_bwd_split_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "mmrec_linear_bwd_split_f32": (c_int32, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, _P, _P]),
    "mmrec_linear_bwd_w_f32": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int32, _P, _P]),
    "mmrec_linear_bwd_x_f32": (c_int32, [_P, _P, _P, c_int32, c_int32, c_int32, _P]),
    "mmrec_gemm_nt_f32": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, _P]),
    "mmrec_topk_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32, c_int32]),
    "mmrec_score_topk_f32": (c_int32, [_P, _P, c_int32, c_int32, c_int32, _P, _P, c_int32, _P, _P, _P,
                                       c_int32, _P]),
    "mmrec_scatter_add_rows_sorted_f32": (c_int32, [_P, _P, _P, c_int32, c_int32, _P, _P]),
    "mmrec_topk_prepared_bytes": (c_size_t, [c_int32, c_int32]),
    "mmrec_topk_prepare_f32": (c_int32, [_P, c_int32, c_int32, _P, _P]),
    "mmrec_score_topk_prepared_f32": (c_int32, [_P, _P, _P, c_int32, c_int32, c_int32, _P, _P, c_int32, _P, _P, _P,
                                                c_int32, _P]),
    "mmrec_score_topk_hinted_f32": (c_int32, [_P, _P, _P, c_int32, c_int32, c_int32, _P, _P, c_int32, _P, c_int32, _P,
                                              _P, _P, _P, _P, c_int32, _P]),
    "mmrec_degree_count_i32": (c_int32, [_P, c_int64, _P, c_int32, _P]),
    "mmrec_edge_norm_f32": (c_int32, [_P, _P, c_int64, _P, _P, _P, _P]),
    "mmrec_bipartite_expand": (c_int32, [_P, _P, _P, c_int64, c_int32, _P, _P, _P, _P]),
    "mmrec_coo_to_csr_workspace_bytes": (c_size_t, [c_int64, c_int32]),
    "mmrec_coo_to_csr": (c_int32, [_P, _P, _P, c_int64, c_int32, _P, _P, _P, _P, _P]),
    "mmrec_sample_negatives_i64": (c_int32, [_P, c_int32, _P, _P, _P, c_int32, c_uint64, c_uint64, _P, _P]),
    "mmrec_topk_metrics_f64": (c_int32, [_P, c_int32, c_int32, _P, _P, _P, _P, _P, c_int32, _P, _P, _P]),
    "mmrec_adam_step_f32": (c_int32, [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float,
                                      c_int64, _P]),
    "mmrec_adam_prepare": (c_int32, [_P, _P, c_float, c_float, _P, _P]),
    "mmrec_adam_step_dev_f32": (c_int32, [_P, _P, _P, _P, c_int64, _P, c_float, c_float, c_float, c_float, _P]),
    "mmrec_adam_multi_step_f32": (c_int32, [_P, _P, _P, _P, _P, c_int32, _P, _P, c_float, c_float, c_float,
                                            c_float, _P]),
    "mmrec_adam_multi_step_dev_f32": (c_int32, [_P, _P, _P, _P, _P, c_int32, _P, c_float, c_float, c_float,
                                                c_float, _P]),
    "mmrec_host_sample_negatives": (c_int32, [_P, _P, _P, c_int32, _P, _P, _P, c_int32, _P]),
    "mmrec_host_random_sample_range": (c_int32, [_P, _P, c_int32, c_int32, _P, _P]),
    "mmrec_cosine_workspace_bytes": (c_size_t, [c_int32]),
    "mmrec_cosine_fwd_f32": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, c_float, _P, _P, _P, _P]),
    "mmrec_cosine_bwd_f32": (c_int32, [_P, _P, _P, _P, c_int32, c_int32, _P, _P, c_float, _P, _P]),
    "mmrec_bpr_multi_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "mmrec_bpr_multi_fwd_f32": (c_int32, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, c_int32, c_float, _P, _P, _P, _P, _P]),
    "mmrec_bpr_multi_bwd_f32": (c_int32, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, c_int32, _P, _P, c_float, _P, _P, _P]),
    "mmrec_cosine_multi_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "mmrec_cosine_multi_fwd_f32": (c_int32, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, _P, _P, _P, _P]),
    "mmrec_cosine_multi_bwd_f32": (c_int32, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, _P, _P, _P, _P]),
    "mmrec_row_normalize_fwd_f32": (c_int32, [_P, c_int64, c_int32, c_float, _P, _P, _P]),
    "mmrec_row_normalize_bwd_f32": (c_int32, [_P, _P, _P, c_int64, c_int32, _P, _P]),
    "mmrec_cat_leaky_fwd_f32": (c_int32, [_P, _P, _P, c_int64, c_int32, c_int32, c_float, _P, _P]),
    "mmrec_cat_leaky_bwd_f32": (c_int32, [_P, _P, _P, c_int64, c_int32, c_int32, c_float, _P, _P, _P, _P]),
    "mmrec_rows_reg_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "mmrec_rows_reg_fwd_f32": (c_int32, [_P, _P, _P, c_int32, c_int32, c_int32, c_float, _P, _P, _P, _P]),
    "mmrec_rows_reg_bwd_f32": (c_int32, [_P, _P, _P, c_int32, c_int32, _P, _P, _P, _P]),
    "mmrec_adam_hist_set": (c_int32, [_P, c_int32, c_float, c_float, c_float, _P]),
    "mmrec_adam_rows_owner": (c_int32, [_P, c_int32, _P, _P]),
    "mmrec_adam_rows_catchup_f32": (c_int32, [_P, _P, _P, _P, _P, c_int32, c_int32, c_int32, _P, _P, c_int32, c_float,
                                              c_float, c_float, c_float, _P]),
    "mmrec_adam_rows_step_f32": (c_int32, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, _P, c_int32, c_float, c_float,
                                           c_float, c_float, c_float, c_int32, _P]),
    "mmrec_adam_hist_set_dev": (c_int32, [_P, c_int32, _P, _P, _P, _P]),
    "mmrec_adam_rows_catchup_dev_f32": (c_int32, [_P, _P, _P, _P, _P, c_int32, c_int32, c_int32, _P, _P, c_int32, _P, c_float,
                                                  c_float, c_float, c_float, _P]),
    "mmrec_adam_rows_step_dev_f32": (c_int32, [_P, _P, _P, _P, _P, _P, c_int32, c_int32, _P, c_int32, _P, _P, c_float, c_float,
                                               c_float, c_float, c_int32, _P]),
    "mmrec_adam_rows_fastforward_f32": (c_int32, [_P, _P, _P, _P, _P, c_int32, c_int32, c_int32, _P, _P, c_int32, c_float,
                                                  c_float, c_float, c_float, _P]),
    "mmrec_adam_rows_fastforward_dev_f32": (c_int32, [_P, _P, _P, _P, _P, c_int32, c_int32, c_int32, _P, _P, c_int32, _P,
                                                      c_float, c_float, c_float, c_float, _P]),
}

_lib = None


class MMRecHipError(RuntimeError):
    pass


def load():
    """Load (once) and type the library.  Raises if it is absent -- build it with
    `python -m mmrec_amd.build` (or `__graft_entry__.build()`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MMRecHipError(
            "libmmrec_hip.so not found at %s: run `python -m mmrec_amd.build`; there is no CPU or "
            "eager fallback for the hot path" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    got = lib.mmrec_abi_version()
    if got != ABI_VERSION:
        raise MMRecHipError("libmmrec_hip.so ABI %d != expected %d: rebuild" % (got, ABI_VERSION))
    _lib = lib
    return lib


def check(err: int, what: str):
    if err != 0:
        msg = load().mmrec_error_string(int(err))
        raise MMRecHipError("%s failed: [%d] %s" % (what, err, msg.decode() if msg else "?"))
