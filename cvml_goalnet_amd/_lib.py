"""ctypes binding of libgoalnet_hip.so (C ABI: include/goalnet_hip.h).

The product path has no CPU or eager-PyTorch fallback: if the shared library is missing or a call
fails, an exception is raised (task rule ③: "the product path must fail loudly when the HIP extension
is missing").
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GOALNET_LIB_PATH") or os.path.join(_HERE, "libgoalnet_hip.so")   # override: A/B builds of the kernels
ABI_VERSION = 4
STAT_PARTS = 1024

P = c_void_p  # device pointers and the stream travel as void*


class RowCopy(ctypes.Structure):
    """goalnet_rowcopy (include/goalnet_hip.h)"""
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("row_bytes", c_int64), ("nrows", c_int), ("gather", c_int), ("cursor", c_void_p),
                ("cursor_bias", c_int64)]


# name -> (restype, [argtypes])   — one row per entry point declared in include/goalnet_hip.h
PROTOTYPES = {
    "goalnet_abi_version": (c_int, []),
    "goalnet_last_error": (c_char_p, []),
    "goalnet_fill_uniform": (c_int, [P, c_int64, c_uint64, c_uint32, c_float, c_float, P]),
    "goalnet_dropout_mask": (c_int, [P, c_int64, c_uint64, c_uint32, c_float, P]),
    "goalnet_transpose_inner": (c_int, [P, P, c_int64, c_int64, c_int64, P]),
    "goalnet_conv3x3_weight_flip": (c_int, [P, P, c_int, c_int, P]),
    "goalnet_conv3x3_weight_flip2": (c_int, [P, P, c_int, c_int, P, P, c_int, c_int, P]),
    "goalnet_conv1_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "goalnet_conv1_wgrad_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "goalnet_conv1_wgrad": (c_int, [P, P, P, P, P, c_size_t, c_int, c_int, c_int, P]),
    "goalnet_stat_parts": (c_int, [c_int64]),
    "goalnet_pool_bnstats_fwd": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "goalnet_pool_bnstats_fwd_p16": (c_int, [P, c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "goalnet_bn_finalize": (c_int, [P, c_int, P, P, P, P, c_float, c_float, c_int64, c_int, P, P, P, P, P]),
    "goalnet_bn_bwd_reduce": (c_int, [P, P, P, P, P, c_int, c_int64, c_int, P]),
    "goalnet_bn_bwd_reduce_t": (c_int, [P, c_int, P, c_int, P, P, P, c_int, c_int64, c_int, c_int, P]),
    "goalnet_bn_bwd_finalize": (c_int, [P, c_int, P, P, P, c_int64, c_int, P, P, P, P]),
    "goalnet_bnpool_bwd": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "goalnet_bnpool_bwd_bf16p": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "goalnet_bnpool_bwd_bf16p_t": (c_int, [P, c_int, P, c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "goalnet_bn_small_ws_bytes": (c_size_t, [c_int]),
    "goalnet_pool_bn_fwd_small": (c_int, [P, P, P, P, P, P, P, c_float, c_float, P, P, P, P, P, c_size_t, P, c_int, c_int, c_int, c_int, P]),
    "goalnet_bn_bwd_reduce_small": (c_int, [P, P, P, P, P, P, P, P, P, c_size_t, P, c_int, c_int, c_int, c_int, P]),
    "goalnet_bnpool_bwd_small": (c_int, [P, P, P, P, P, P, P, c_size_t, P, c_int, c_int, c_int, c_int, P]),
    "goalnet_partials_sum": (c_int, [P, c_int, c_int64, c_int, P, P]),
    "goalnet_partials_sum2": (c_int, [P, c_int, c_int, P, P, c_int, c_int, P, P]),
    "goalnet_partials_sum_f64": (c_int, [P, c_int, c_int64, c_int, P, P]),
    "goalnet_conv3x3_fwd_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "goalnet_conv3x3_fwd": (c_int, [P, P, P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P, c_int, P]),
    "goalnet_conv3x3_fwd_kernel_name": (c_char_p, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "goalnet_conv3x3_fwd_bf16p_kernel_name": (c_char_p, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "goalnet_conv3x3_wgrad_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "goalnet_conv3x3_wgrad_codes_bytes": (c_size_t, [c_int, c_int, c_int]),
    "goalnet_conv3x3_wgrad_codes": (c_int, [P, c_int, c_int, c_int, P]),
    "goalnet_conv3x3_wgrad": (c_int, [P, P, P, P, P, P, c_size_t, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "goalnet_cast_bf16": (c_int, [P, P, c_int64, c_int, P]),
    "goalnet_cast_f32": (c_int, [P, P, c_int64, c_int, P]),
    "goalnet_bn_apply_bf16": (c_int, [P, P, P, P, c_int64, c_int, c_int, P]),
    "goalnet_bn_apply_bf16_p16": (c_int, [P, P, P, P, c_int64, c_int, c_int, P]),
    "goalnet_conv3x3_fwd_bf16": (c_int, [P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "goalnet_linear_fwd_bf16_ws_bytes": (c_size_t, [c_int, c_int64, c_int]),
    "goalnet_linear_fwd_bf16": (c_int, [P, c_int64, P, P, c_int, P, c_int64, P, c_int64, P, c_int64, c_int, c_int64, c_int, P, c_size_t, c_int, P]),
    "goalnet_bf16_padded_layout": (c_int, [c_int, c_int, c_int, c_int, ctypes.POINTER(c_int64), ctypes.POINTER(c_int64)]),
    "goalnet_to_bf16_padded": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "goalnet_to_bf16_padded_p16": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "goalnet_conv3x3_fwd_bf16p_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "goalnet_conv3x3_fwd_bf16p": (c_int, [P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, P, c_size_t, c_int, P]),
    "goalnet_conv3x3_wgrad_bf16_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "goalnet_conv3x3_wgrad_bf16": (c_int, [P, P, P, P, c_size_t, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "goalnet_absmax": (c_int, [P, c_int64, P, P, c_int, c_int64, c_int64, P, P]),
    "goalnet_split_scales": (c_int, [P, P, P, P]),
    "goalnet_split_padded": (c_int, [c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "goalnet_split_rows": (c_int, [c_int, P, c_int64, P, P, c_int, P, P, c_int64, c_int64, P]),
    "goalnet_conv3x3_fwd_split": (c_int, [c_int, P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, P, P]),
    "goalnet_conv3x3_wgrad_split_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "goalnet_conv3x3_wgrad_split": (c_int, [c_int, P, P, P, P, c_size_t, c_int, c_int, c_int, c_int, c_int, P, P]),
    "goalnet_linear_split_ok": (c_int, [c_int, c_int, c_int64, c_int]),
    "goalnet_linear_fwd_split_ws_bytes": (c_size_t, [c_int, c_int, c_int64, c_int]),
    "goalnet_linear_fwd_split": (c_int, [c_int, P, P, P, c_int, P, c_int64, P, c_int64, P, c_int64, c_int, c_int64, c_int, P, c_size_t, P, P]),
    "goalnet_linear_bwd_dx_split": (c_int, [c_int, P, P, P, c_int64, c_int, c_int64, c_int, P, P]),
    "goalnet_linear_bwd_dw_split": (c_int, [c_int, P, P, P, c_int, c_int64, c_int, P, P]),
    "goalnet_linear_bwd_dx_bf16": (c_int, [P, c_int64, P, P, c_int64, P, c_int64, c_int, c_int64, c_int, c_int, P]),
    "goalnet_linear_bwd_dx_bf16_o16_ok": (c_int, [c_int, c_int64, c_int]),
    "goalnet_linear_bwd_dx_bf16_o16": (c_int, [P, c_int64, P, P, c_int64, c_int, c_int64, c_int, c_int, P]),
    "goalnet_conv3x3_fwd_bf16p_o16_ok": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "goalnet_conv3x3_fwd_bf16p_o16": (c_int, [P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "goalnet_linear_bwd_dw_bf16": (c_int, [P, c_int64, P, c_int64, P, c_int, c_int64, c_int, c_int, P]),
    "goalnet_linear_fwd_ws_bytes": (c_size_t, [c_int, c_int64, c_int]),
    "goalnet_linear_fwd": (c_int, [P, c_int64, P, P, c_int, P, P, c_int, P, c_int64, P, c_int64, P, c_int64,
                                   c_int, c_int64, c_int, P, c_size_t, P]),
    "goalnet_linear_bwd_dx": (c_int, [P, c_int64, P, P, c_int64, P, c_int64, c_int, c_int64, c_int, P]),
    "goalnet_linear_bwd_dw": (c_int, [P, c_int64, P, c_int64, P, P, c_int, P, P, c_int, c_int64, c_int, P]),
    "goalnet_colsum": (c_int, [P, c_int64, c_int, c_int, P, P]),
    "goalnet_mul": (c_int, [P, c_int64, P, c_int64, P, c_int64, c_int, c_int, P]),
    "goalnet_conv1d_fwd": (c_int, [P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "goalnet_conv1d_bwd_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "goalnet_conv1d_bwd": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, c_size_t, P]),
    "goalnet_conv1d_bwd_small": (c_int, [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "goalnet_relu_bwd": (c_int, [P, P, P, c_int64, P]),
    "goalnet_mlp_blocks": (c_int, []),
    "goalnet_mlp_fwd": (c_int, [P, c_int64, c_int, P, P, P, P, P, P, P, P, P, P, P, c_int, P, P]),
    "goalnet_mlp_bwd_ws_bytes": (c_size_t, [c_int]),
    "goalnet_mlp_bwd": (c_int, [P, P, P, c_int64, P, c_int64, P, P, P, P, c_int64, P, c_int, c_int, c_int, P, c_size_t, P, P]),
    "goalnet_head_fwd": (c_int, [P, c_int64, P, P, P, P, c_int, c_int, P]),
    "goalnet_head_bwd": (c_int, [P, P, P, c_int64, P, P, c_int64, P, c_int64, P, P, c_int, c_int, P]),
    "goalnet_mse_bcast": (c_int, [P, P, c_int, P, P, P]),
    "goalnet_cubic_resample": (c_int, [P, P, P, c_int64, c_int, c_int, P]),
    "goalnet_logmel_slots": (c_int, [P, P, P, c_int, c_int, P, P, P, P, P, P, P, P]),
    "goalnet_mfcc_from_logmel": (c_int, [P, P, c_int, c_int, P, P, P, P, c_int, c_int, c_double, P]),
    "goalnet_cls_head_fwd": (c_int, [P, c_int64, P, P, P, P, c_int, c_int, c_int, P]),
    "goalnet_cross_entropy": (c_int, [P, P, P, P, c_int, c_int, P]),
    "goalnet_cls_head_bwd": (c_int, [P, P, P, c_int64, P, P, c_int64, P, c_int64, P, P, c_int, c_int, c_int, P]),
    "goalnet_argmax_plus1": (c_int, [P, P, c_int, c_int, P]),
    "goalnet_adam_step": (c_int, [P, P, P, P, c_int64, c_double, c_double, c_double, c_double, c_int, c_float, P]),
    "goalnet_counter_add": (c_int, [P, c_int64, P]),
    "goalnet_dropout_masks_dev": (c_int, [P, c_int, ctypes.POINTER(c_int), c_int, c_uint64, c_uint32, c_uint32, P, c_float, c_int64, P]),
    "goalnet_adam_step_dev": (c_int, [P, P, P, P, c_int64, c_double, c_double, c_double, c_double, P, c_int64, c_float, P]),
    "goalnet_adam_step_dev_blocks": (c_int, [P, P, P, P, c_int64, c_double, c_double, c_double, c_double, P, c_int64, c_float, c_int, P]),
    "goalnet_adam_step_dev_shadow": (c_int, [P, P, P, P, c_int64, c_double, c_double, c_double, c_double, P, c_int64, c_float, P, c_int64,
                                             c_int64, c_int, P]),
    "goalnet_adam_step_dev_guarded": (c_int, [P, P, P, P, c_int64, c_double, c_double, c_double, c_double, P, c_int64, c_float, P, c_int64,
                                              c_int64, c_int, P, P]),
    "goalnet_scale": (c_int, [P, c_int64, c_float, P]),
    "goalnet_grad_finite_check": (c_int, [P, c_int64, P, c_int64, P, P, P]),
    "goalnet_counters_add4": (c_int, [P, c_int64, c_int64, c_int64, c_int64, P]),
    "goalnet_counters_add4_guarded": (c_int, [P, c_int64, c_int64, c_int64, c_int64, P, P]),
    "goalnet_rows_scatter_tick": (c_int, [ctypes.POINTER(RowCopy), c_int, P, c_int64, c_int64, c_int64, c_int64, P, P]),
    "goalnet_rows_copy_batch": (c_int, [ctypes.POINTER(RowCopy), c_int, P]),
    "goalnet_frames_preprocess": (c_int, [P, c_int, c_int, c_int, P, c_int, c_int, P, P]),
    "goalnet_knapsack_ws_bytes": (c_size_t, [c_int, c_int]),
    "goalnet_knapsack": (c_int, [P, P, c_int, c_int, P, P, c_size_t, P]),
    "goalnet_fscore": (c_int, [P, P, c_int, c_int, P, P, P]),
    "goalnet_postprocess_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "goalnet_postprocess": (c_int, [P, c_int, c_int, c_int, P, c_int, c_int, c_int, P, c_int, P, P, P, P, P, P, P, c_size_t, P]),
    "goalnet_rows_gather": (c_int, [P, P, c_int64, c_int, P, P]),
    "goalnet_rows_scatter": (c_int, [P, P, c_int64, c_int, P, P]),
}

_lib = None


class GoalnetError(RuntimeError):
    pass


def load():
    """Load libgoalnet_hip.so (built by __graft_entry__.build()). Raises if absent — no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GoalnetError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
            "There is no CPU / eager fallback for the AVM hot path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    v = lib.goalnet_abi_version()
    if v != ABI_VERSION:
        raise GoalnetError(f"libgoalnet_hip.so ABI {v} != expected {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().goalnet_last_error()
        raise GoalnetError(f"{what}: rc={rc}: {msg.decode() if msg else ''}")
