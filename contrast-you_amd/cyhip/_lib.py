"""ctypes binding of libcontrastyou_hip.so (C ABI declared in include/contrastyou_hip.h).

The library is the only compute backend of this package: there is no CPU or eager
fallback.  If it is missing, or a call is made without a GPU tensor, the error is loud.
"""
from __future__ import annotations

import ctypes as C
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_long, c_size_t, c_void_p
from pathlib import Path

CY_F32, CY_BF16, CY_F16 = 0, 1, 2
CY_SRC_DIRECT, CY_SRC_POOL2, CY_SRC_UP2 = 0, 1, 2
ABI_VERSION = 12

_ERRORS = {-1: "CY_ERR_ARG (bad/NULL argument)", -2: "CY_ERR_SHAPE (unsupported shape)",
           -3: "CY_ERR_DTYPE (unsupported dtype)", -4: "CY_ERR_LAUNCH (HIP launch failed)",
           -5: "CY_ERR_WORKSPACE (workspace too small)"}

LIB_PATH = Path(__file__).resolve().parent.parent / "lib" / "libcontrastyou_hip.so"


class HipExtensionMissing(RuntimeError):
    pass


class HipKernelError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    """mirror of cy_conv_desc"""
    _fields_ = [(n, c_int32) for n in (
        "N", "H", "W", "C1", "C2", "Cout", "mode1", "prologue", "in_dtype", "out_dtype",
        "ld1", "ld2", "ldo", "split_c", "ldo2")]


class PackItem(C.Structure):
    """mirror of cy_pack_item"""
    _fields_ = [("w", c_void_p), ("off_f", C.c_longlong), ("off_d", C.c_longlong), ("first", C.c_longlong),
                ("Cout", c_int32), ("Cin", c_int32), ("co_pad", c_int32), ("ci_pad", c_int32),
                ("ci_pad2", c_int32), ("co_pad2", c_int32), ("off_ff", C.c_longlong), ("off_fd", C.c_longlong)]


class ConvPlan(C.Structure):
    """mirror of cy_conv_plan"""
    _fields_ = [(n, c_int32) for n in ("kernel", "th", "tw", "bn", "ksplit", "one_per_cu", "partials", "workgroups")]


class WgradPlan(C.Structure):
    """mirror of cy_wgrad_plan"""
    _fields_ = [(n, c_int32) for n in ("twelve", "wco", "wci", "wk", "th", "tw", "splits", "workgroups")]


class BnAcc(C.Structure):
    """mirror of cy_bn_acc"""
    _fields_ = [("acc", c_void_p), ("R", c_int32), ("C", c_int32)]


class BnFold(C.Structure):
    """mirror of cy_bn_fold"""
    _fields_ = [("acc", c_void_p), ("R", c_int32), ("C", c_int32), ("gamma", c_void_p), ("beta", c_void_p),
                ("count", c_double), ("eps", c_float), ("reserved", c_int32), ("coef", c_void_p)]


class BnBwdIn(C.Structure):
    """mirror of cy_bn_bwd_in"""
    _fields_ = [("y", c_void_p), ("coef", c_void_p), ("acc", POINTER(BnAcc)), ("count", c_double),
                ("batch_stats", c_int32), ("accumulate", c_int32), ("dgamma", c_void_p), ("dbeta", c_void_p),
                ("dy", c_void_p)]


class BnDzOut(C.Structure):
    """mirror of cy_bn_dz_out"""
    _fields_ = [("y", c_void_p), ("coef", c_void_p), ("acc", POINTER(BnAcc)), ("c0", c_int32), ("C", c_int32)]


class BnRunItem(C.Structure):
    """mirror of cy_bn_run_item"""
    _fields_ = [("coef", c_void_p), ("running_mean", c_void_p), ("running_var", c_void_p), ("C", c_int32),
                ("momentum", c_float)]


class MatLayout(C.Structure):
    """mirror of cy_mat_layout: element (i, j) of batch (b1, b2) at b1*s1 + b2*s2 + i*rs + j*cs"""
    _fields_ = [(n, c_long) for n in ("rs", "cs", "s1", "s2")]


_P = c_void_p
_PCD = POINTER(ConvDesc)
_PML = POINTER(MatLayout)
_PBA = POINTER(BnAcc)
_PBF = POINTER(BnFold)

# name -> (restype, argtypes).  restype c_int functions are status-checked.
_SIGS = {
    "cy_abi_version": (c_int, []),
    "cy_build_arch": (c_char_p, []),
    "cy_stream_capture_id": (C.c_ulonglong, [_P]),
    "cy_debug_stamp": (c_int, [_P, c_int, _P]),
    "cy_debug_spin": (c_int, [c_int, _P]),
    "cy_debug_event_create": (c_int, [POINTER(_P)]),
    "cy_debug_event_record": (c_int, [_P, _P]),
    "cy_debug_event_elapsed_us": (c_int, [_P, _P, POINTER(C.c_float)]),
    "cy_debug_event_destroy": (c_int, [_P]),
    "cy_stream_wait_value": (c_int, [_P, _P, c_int]),
    "cy_conv3x3_packed_dims": (c_int, [c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    "cy_conv3x3_packed_elems": (C.c_longlong, [c_int, c_int, c_int]),
    "cy_conv3x3_pack_weights": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "cy_conv3x3_pack_weights_batched": (c_int, [_P, c_int, C.c_longlong, _P, _P, c_int, _P]),
    "cy_conv3x3_num_partials": (c_int, [_PCD]),
    "cy_conv3x3_plan": (c_int, [_PCD, POINTER(ConvPlan)]),
    "cy_conv3x3_wgrad_plan": (c_int, [_PCD, c_int, POINTER(WgradPlan)]),
    "cy_conv3x3_fwd_ws_bytes": (c_size_t, [_PCD]),
    "cy_conv3x3_fwd": (c_int, [_PCD, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "cy_debug_wgrad_stamps": (c_int, [_P]),
    "cy_debug_conv_stamps": (c_int, [_P]),
    "cy_conv3x3_wgrad_ws_bytes": (c_size_t, [_PCD]),
    "cy_conv3x3_wgrad": (c_int, [_PCD, _P, _P, _P, _P, _P, _P, c_int, _P, c_size_t, _P]),
    "cy_conv3x3_wgrad_pair_ws_bytes": (c_size_t, [_PCD, c_int]),
    "cy_conv3x3_wgrad_pair": (c_int, [_PCD, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, c_int, _P,
                                      c_size_t, _P]),
    "cy_bn_acc_replicas": (c_int, [c_int, c_int]),
    "cy_bn_acc_bytes": (c_size_t, [c_int, c_int]),
    "cy_conv3x3_stat_workgroups": (c_int, [_PCD]),
    "cy_conv3x3_fwd_bn": (c_int, [_PCD, _P, _P, _PBF, _P, _P, _P, _P, _P, _P, _PBA, _P, c_size_t, _P]),
    "cy_conv3x3_first_fwd_acc": (c_int, [_P, _P, _P, _PBA, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "cy_conv3x3_dgrad_bn_ok": (c_int, [_PCD]),
    "cy_conv3x3_dgrad_bn": (c_int, [_PCD, _P, POINTER(BnBwdIn), _P, _P, _P, _P, c_size_t, _P]),
    "cy_conv3x3_dgrad_dz_ok": (c_int, [_PCD, c_int, c_int]),
    "cy_conv3x3_dgrad_dz": (c_int, [_PCD, _P, _P, _P, _P, POINTER(BnDzOut), _P, c_size_t, _P]),
    "cy_bn_fold_coef": (c_int, [_PBF, _P]),
    "cy_bn_relu_apply_fold": (c_int, [_P, _PBF, _P, c_long, c_int, c_int, _P]),
    "cy_bn_relu_apply_pool_fold": (c_int, [_P, _PBF, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "cy_bn_running_update": (c_int, [POINTER(BnRunItem), c_int, _P]),
    "cy_bn_relu_bwd_reduce_acc": (c_int, [_P, c_int, _P, _P, _PBA, c_long, c_int, c_int, _P]),
    "cy_bn_relu_bwd_workgroups": (c_int, [c_long, c_int]),
    "cy_maxpool2_bwd_bn_acc": (c_int, [_P, _P, _P, c_int, _P, _P, _P, _PBA, c_int, c_int, c_int, c_int, c_int, _P]),
    "cy_upsample2_bwd_bn_workgroups": (c_int, [c_int, c_int, c_int, c_int]),
    "cy_upsample2_bwd_bn_acc": (c_int, [_P, c_int, _P, _P, _P, _PBA, c_int, c_int, c_int, c_int, c_int, _P]),
    "cy_bn_relu_bwd_apply_fold": (c_int, [_P, c_int, _P, _P, _PBA, c_double, c_int, _P, _P, c_int, _P, c_long, c_int,
                                          c_int, _P]),
    "cy_conv3x3_first_num_partials": (c_int, [c_int, c_int, c_int, c_int]),
    "cy_conv3x3_first_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "cy_conv3x3_first_wgrad_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "cy_conv3x3_first_wgrad": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P,
                                       c_size_t, _P]),
    "cy_bn_finalize": (c_int, [_P, c_int, c_int, c_double, _P, _P, _P, _P, c_float, c_float, c_int,
                               c_int, _P, _P, _P, _P, _P]),
    "cy_bn_relu_apply": (c_int, [_P, _P, _P, _P, c_long, c_int, c_int, c_int, _P]),
    "cy_bn_relu_apply_pool": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "cy_bn_bwd_num_partials": (c_int, [c_long, c_int]),
    "cy_bn_relu_bwd_reduce": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, c_long, c_int, c_int, _P]),
    "cy_bn_bwd_finalize": (c_int, [_P, c_int, c_int, _P, _P, _P, c_double, c_int, _P, _P, c_int, _P, _P]),
    "cy_bn_relu_bwd_apply": (c_int, [_P, c_int, _P, _P, _P, _P, _P, c_long, c_int, c_int, _P]),
    "cy_maxpool2_bwd": (c_int, [_P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "cy_maxpool2_bwd_bn_num_partials": (c_int, [c_int, c_int, c_int, c_int]),
    "cy_maxpool2_bwd_bn": (c_int, [_P, _P, _P, c_int] + [_P] * 7 + [c_int] * 5 + [_P]),
    "cy_upsample2_bwd": (c_int, [_P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "cy_head1x1_fwd": (c_int, [_P, _P, _P, _P, c_long, c_int, c_int, c_int, _P]),
    "cy_head1x1_bwd_ws_bytes": (c_size_t, [c_long, c_int, c_int]),
    "cy_head1x1_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_long, c_int, c_int, c_int, _P, c_size_t,
                               _P]),
    "cy_head1x1_bwd_into": (c_int, [_P, _P, _P, _P, _P, _P, c_long, c_int, c_int, c_int, _P, c_size_t,
                               _P]),
    "cy_softmax_kl_ws_bytes": (c_size_t, [c_long]),
    "cy_softmax_kl_fwd": (c_int, [_P, _P, _P, c_long, c_int, c_float, _P, c_size_t, _P]),
    "cy_softmax_kl_bwd": (c_int, [_P, _P, _P, _P, c_long, c_int, c_float, _P]),
    "cy_avgpool_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "cy_avgpool_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "cy_linear_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "cy_linear_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "cy_linear_bwd_into": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "cy_proj_head_fwd": (c_int, [_P] * 10 + [c_int] * 5 + [c_float, c_float, c_int, _P]),
    "cy_proj_head_bwd_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "cy_proj_head_bwd": (c_int, [_P] * 12 + [c_int, _P, c_size_t] + [c_int] * 5 + [c_float, c_float, c_int, _P]),
    "cy_l2norm_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_float, _P]),
    "cy_l2norm_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, _P]),
    "cy_supcon_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_float, _P]),
    "cy_supcon_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_float, _P]),
    "cy_supcon_excl_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_float, _P]),
    "cy_supcon_excl_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_float, _P]),
    "cy_supcon_fused_ws_bytes": (c_size_t, [c_int, c_int]),
    "cy_supcon_fused_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_size_t, c_int, c_int, c_float, _P]),
    "cy_supcon_fused_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_size_t, c_int, c_int, c_float, _P]),
    "cy_supcon_matrices": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, _P]),
    "cy_sgemm": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_float, c_int, _P]),
    "cy_affine_nearest_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "cy_affine_nearest_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "cy_dice_counts": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "cy_ema_update": (c_int, [_P, _P, c_long, c_float, c_float, _P]),
    "cy_softmax_mse_ws_bytes": (c_size_t, [c_long]),
    "cy_softmax_mse_fwd": (c_int, [_P, _P, _P, c_long, c_int, _P, c_size_t, _P]),
    "cy_softmax_mse_bwd": (c_int, [_P, _P, _P, _P, _P, c_long, c_int, _P]),
    "cy_radam_step": (c_int, [_P, _P, _P, _P, c_long, c_float, c_float, c_float, c_float, c_float,
                              c_long, _P]),
    "cy_dense_proj_fwd": (c_int, [_P, _P, _P, _P, c_int, _P] + [c_int] * 8 + [c_float, c_int, _P]),
    "cy_dense_proj_bwd_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "cy_dense_proj_bwd": (c_int, [_P, _P, _P, _P, c_int, _P, _P, _P, _P, c_int] + [c_int] * 8 +
                          [c_float, c_int, _P, c_size_t, _P]),
    "cy_adaptive_avgpool_fwd": (c_int, [_P, _P, c_int, _P] + [c_int] * 8 + [_P]),
    "cy_adaptive_avgpool_bwd": (c_int, [_P, _P] + [c_int] * 8 + [_P]),
    "cy_adaptive_maxpool_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "cy_adaptive_maxpool_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "cy_gather_rows_fwd": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "cy_gather_rows_bwd": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "cy_cluster_head_fwd": (c_int, [_P, _P, _P, _P, c_long, c_int, c_int, c_int, c_int, c_float, c_int, _P]),
    "cy_cluster_head_bwd_ws_bytes": (c_size_t, [c_long, c_int]),
    "cy_cluster_head_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_long, c_int, c_int, c_int, c_int, c_float, c_int, _P,
                                    c_size_t, _P]),
    "cy_group_softmax_fwd": (c_int, [_P, _P, c_long, c_int, c_int, c_float, _P]),
    "cy_group_softmax_bwd": (c_int, [_P, _P, _P, c_long, c_int, c_int, c_float, _P]),
    "cy_joint_ws_bytes": (c_size_t, [c_int] * 5),
    "cy_joint_fwd": (c_int, [_P, _P, _P] + [c_int] * 6 + [_P, c_size_t, _P]),
    "cy_joint_bwd": (c_int, [_P, _P, _P, _P, _P, _P] + [c_int] * 6 + [_P]),
    "cy_iid_loss_ws_bytes": (c_size_t, [c_int, c_int]),
    "cy_iid_loss": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, _P, c_size_t, _P]),
    "cy_gn_ws_bytes": (c_size_t, [c_int, c_int]),
    "cy_gn_silu_fwd": (c_int, [_P, c_int, _P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_float,
                               c_int, _P, c_size_t, _P]),
    "cy_gn_silu_bwd": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, _P, _P, c_int, _P, _P, _P, c_int, c_int,
                               c_int, c_int, c_int, c_int, _P, c_size_t, _P]),
    "cy_gn_silu_mod_fwd": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_float,
                                   c_int, _P, c_size_t, _P]),
    "cy_gn_silu_mod_bwd": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, c_int,
                                   c_int, c_int, c_int, c_int, c_int, _P, c_size_t, _P]),
    "cy_bilinear_fwd": (c_int, [_P, _P] + [c_int] * 7 + [_P]),
    "cy_gemm_strided_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "cy_gemm_strided": (c_int, [_P, _PML, _P, _PML, _P, _PML, _P, c_int, c_int, c_int, c_int, c_int, c_float, c_int,
                                c_int, _P, c_size_t, _P]),
    "cy_im2col": (c_int, [_P, _P] + [c_int] * 8 + [_P]),
    "cy_col2im": (c_int, [_P, _P, _P] + [c_int] * 8 + [_P]),
    "cy_colsum_ws_bytes": (c_size_t, [c_long, c_int]),
    "cy_colsum": (c_int, [_P, _P, c_long, c_int, c_int, _P, c_size_t, _P]),
    "cy_chan_layernorm_fwd": (c_int, [_P, _P, _P, _P, c_long, c_int, c_float, _P]),
    "cy_chan_layernorm_bwd_ws_bytes": (c_size_t, [c_long, c_int]),
    "cy_chan_layernorm_bwd": (c_int, [_P, _P, _P, _P, _P, _P, c_long, c_int, c_float, _P, c_size_t, _P]),
    "cy_head_softmax_fwd": (c_int, [_P, c_int, c_int, _P, c_long, c_int, c_int, c_float, _P]),
    "cy_head_softmax_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_long, c_int, c_int, c_float, _P]),
    "cy_col_softmax_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "cy_col_softmax_fwd": (c_int, [_P, c_int, c_int, _P, c_int, c_int, c_int, _P, c_size_t, _P]),
    "cy_col_softmax_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "cy_row_softmax_fwd": (c_int, [_P, c_long, c_int, _P]),
    "cy_row_softmax_bwd": (c_int, [_P, _P, c_long, c_int, _P]),
    "cy_sinusoidal_emb": (c_int, [_P, _P, c_int, c_int, _P]),
    "cy_act_fwd": (c_int, [_P, _P, c_long, c_int, _P]),
    "cy_act_bwd": (c_int, [_P, _P, _P, c_long, c_int, _P]),
}

# functions whose int return is a count / size, not a status
_COUNT_FUNCS = {"cy_abi_version", "cy_conv3x3_num_partials", "cy_conv3x3_first_num_partials",
                "cy_bn_bwd_num_partials", "cy_bn_acc_replicas", "cy_conv3x3_stat_workgroups",
                "cy_bn_relu_bwd_workgroups", "cy_conv3x3_dgrad_bn_ok", "cy_conv3x3_dgrad_dz_ok"}
# (cy_maxpool2_bwd_bn_num_partials returns a count or a negative status: the caller tests the sign itself)

_lib = None


def exported_names():
    return sorted(_SIGS)


def load():
    """Load (once) and return the ctypes library with typed signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise HipExtensionMissing(
            f"{LIB_PATH} not found: build it with `python contrast-you_amd/build.py` "
            "(there is no CPU fallback for the hot path)")
    lib = C.CDLL(str(LIB_PATH))
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    ver = lib.cy_abi_version()
    if ver != ABI_VERSION:
        raise HipExtensionMissing(f"ABI mismatch: library {ver}, binding {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc: int, name: str) -> int:
    if rc < 0:
        raise HipKernelError(f"{name} failed: {_ERRORS.get(rc, rc)}")
    return rc


def call(name: str, *args) -> int:
    """Call a status-returning entry point and raise on error."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if name in _COUNT_FUNCS or _SIGS[name][0] is c_int:
        return check(rc, name)
    return rc
