"""ctypes binding of libnpp_hip.so (include/npp_hip.h).

The product path has NO fallback: if the library is missing or a call fails this module
raises.  Build with `python -c "import __graft_entry__ as g; g.build()"` or
`npp_amd/csrc/build.sh`.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnpp_hip.so")

NPP_F32, NPP_BF16, NPP_MASK8 = 0, 1, 2
NPP_E_SHAPE, NPP_E_DTYPE, NPP_E_ALIGN, NPP_E_HIP, NPP_E_UNSUPPORTED, NPP_E_NULL, NPP_E_RCCL = -1, -2, -3, -4, -5, -6, -7
STAT_REPLICAS = 16   # NPP_STAT_REPLICAS in include/npp_hip.h
FAM = {"none": 0, "conv_igemm": 1, "conv_wgrad": 2, "dwconv": 3, "bn": 4, "eltwise": 5, "pool": 6,
       "bilinear": 7, "loss": 8, "conv_s1": 9, "conv_g8": 10, "conv_g4": 11}


class NppP2pSeg(C.Structure):
    _fields_ = [("slabs", C.c_void_p), ("len", C.c_int64), ("split", C.c_int64), ("out0", C.c_void_p), ("out0_dup", C.c_void_p),
                ("out1", C.c_void_p), ("out2", C.c_void_p), ("nrep", C.c_int32), ("zero_rest", C.c_int32), ("out_all", C.c_void_p)]


class NppTensor(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("n", C.c_int64), ("h", C.c_int64), ("w", C.c_int64), ("c", C.c_int64),
                ("ld", C.c_int64), ("dtype", C.c_int32), ("_pad", C.c_int32)]


class NppConvGeom(C.Structure):
    _fields_ = [("kh", C.c_int32), ("kw", C.c_int32), ("sh", C.c_int32), ("sw", C.c_int32), ("ph", C.c_int32),
                ("pw", C.c_int32), ("dh", C.c_int32), ("dw", C.c_int32), ("uph", C.c_int32), ("upw", C.c_int32), ("relu_in", C.c_int32)]


class NppBnSumsArgs(C.Structure):
    _fields_ = [("ya", NppTensor), ("yb", NppTensor), ("mi_a", C.c_void_p), ("mi_b", C.c_void_p), ("sums", C.c_void_p),
                ("two", C.c_int32), ("_pad", C.c_int32)]


class NppWgradItem(C.Structure):
    _fields_ = [("x", NppTensor), ("dy", NppTensor), ("dw_packed", C.c_void_p), ("g", NppConvGeom), ("nslabs", C.c_int32)]


class NppDwWgradItem(C.Structure):
    _fields_ = [("x", NppTensor), ("dy", NppTensor), ("dw", C.c_void_p), ("ws", C.c_void_p), ("g", NppConvGeom), ("_pad", C.c_int32)]


class NppSeGradItem(C.Structure):
    _fields_ = [("pooled", C.c_void_p), ("hidden", C.c_void_p), ("dz", C.c_void_p), ("dw1", C.c_void_p), ("db1", C.c_void_p),
                ("dw2", C.c_void_p), ("db2", C.c_void_p), ("n", C.c_int32), ("c", C.c_int32)]


class NppLossTerm(C.Structure):
    _fields_ = [("acc", C.c_void_p), ("num_idx", C.c_int32), ("den_idx", C.c_int32), ("coef", C.c_float), ("stage", C.c_int32)]


class NppBnFinalizeArgs(C.Structure):
    _fields_ = [("stats", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("running_mean", C.c_void_p),
                ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p), ("scale_shift", C.c_void_p),
                ("mean_invstd", C.c_void_p), ("count", C.c_double), ("nrep", C.c_int32), ("momentum", C.c_float),
                ("eps", C.c_float), ("stats_c", C.c_int32)]


class NppAffineAddJob(C.Structure):
    _fields_ = [("out", NppTensor), ("a", NppTensor), ("b", NppTensor), ("fin_a", NppBnFinalizeArgs), ("fin_b", NppBnFinalizeArgs),
                ("relu", C.c_int32), ("_pad", C.c_int32), ("mask_bits", C.c_void_p), ("ld_mask", C.c_int64)]


class NppBnBwdJob(C.Structure):
    _fields_ = [("dout", NppTensor), ("ya", NppTensor), ("yb", NppTensor), ("relu_out", NppTensor), ("dya", NppTensor), ("dyb", NppTensor),
                ("mi_a", C.c_void_p), ("mi_b", C.c_void_p), ("gamma_a", C.c_void_p), ("gamma_b", C.c_void_p),
                ("dgamma_a", C.c_void_p), ("dbeta_a", C.c_void_p), ("dgamma_b", C.c_void_p), ("dbeta_b", C.c_void_p),
                ("sums", C.c_void_p), ("count", C.c_double)]


class NppSeFwdJob(C.Structure):
    _fields_ = [("y", NppTensor), ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
                ("pooled", C.c_void_p), ("hidden", C.c_void_p), ("gate", C.c_void_p)]


class NppSeBwdJob(C.Structure):
    _fields_ = [("dout", NppTensor), ("w1", C.c_void_p), ("w2", C.c_void_p), ("hidden", C.c_void_p), ("gate", C.c_void_p),
                ("dz", C.c_void_p)]


class NppMixSide(C.Structure):
    _fields_ = [("x", NppTensor), ("dx", NppTensor), ("stats", C.c_void_p), ("mean_invstd", C.c_void_p), ("running_mean", C.c_void_p),
                ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p), ("momentum", C.c_float), ("eps", C.c_float)]


class NppAdamJob(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_int64), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("_pad", C.c_int32)]


class NppPackJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("out", C.c_void_p), ("cout", C.c_int32), ("cin", C.c_int32), ("kh", C.c_int32),
                ("kw", C.c_int32), ("for_dgrad", C.c_int32), ("dtype", C.c_int32), ("first_block", C.c_int64),
                ("co_off", C.c_int32), ("co_total", C.c_int32)]


_lib = None

_P = C.c_void_p
_T = C.POINTER(NppTensor)
_G = C.POINTER(NppConvGeom)
_SIGS = {
    "npp_prof_begin": [C.c_int, C.c_int],
    "npp_prof_end": [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)],
    "npp_pack_weight": [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P],
    "npp_pack_weights_batched": [_P, C.c_int, C.c_int64, _P],
    "npp_pack_weights_batched_map": [_P, C.c_int, _P, C.c_int64, _P],
    "npp_conv_fwd": [_T, _P, _P, _T, _T, _P, _G, _P],
    "npp_conv_dgrad_sums": [_T, _P, _T, _T, _G, _P, _P],
    "npp_conv_fwd_ws": [_T, _P, _P, _T, _T, _P, _G, _P, C.c_int64, _P],
    "npp_conv_wgrad": [_T, _T, _P, _G, _P],
    "npp_unpack_wgrad": [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P],
    "npp_conv_wgrad_slabs": [_T, _T, _P, C.c_int, _G, _P],
    "npp_unpack_wgrad_batched": [_P, _P, C.c_int64, _P],
    "npp_unpack_wgrad_sum": [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P],
    "npp_sum_replicas": [_P, C.c_int, C.c_int, _P, _P],
    "npp_dwconv_fwd": [_T, _P, _T, _G, _P],
    "npp_dwconv_bwd_data": [_T, _P, _T, _T, _G, _P],
    "npp_dwconv_bwd_weight": [_T, _T, _P, _P, _G, _P],
    "npp_channel_stats": [_T, _P, _P],
    "npp_channel_sum": [_T, _P, _P],
    "npp_bn_finalize": [_P, C.c_int, C.c_double, _P, _P, _P, _P, _P, C.c_float, C.c_float, _P, _P, C.c_int, _P],
    "npp_bn_eval_coeffs": [_P, _P, _P, _P, C.c_float, _P, C.c_int, _P],
    "npp_affine_add": [_T, _T, _P, _T, _P, C.c_int, _P],
    "npp_affine_add_m": [_T, _T, _P, _T, _P, C.c_int, _P, C.c_int64, _P],
    "npp_bn_bwd_reduce": [_T, _T, _T, _P, _P, C.c_int, _P],

    "npp_bn_bwd_coeffs": [_P, C.c_int, C.c_double, _P, _P, _P, _P, _P, C.c_int, _P],
    "npp_bn_bwd_sum": [_P, C.c_int, _P, _P, _P, C.c_int, _P],
    "npp_bn_finalize2": [_P, _P, C.c_int, _P],
    "npp_bn_bwd_reduce2": [_T, _T, _T, _T, _P, _P, _P, C.c_int, _P],
    "npp_bn_bwd_coeffs2": [_P, C.c_int, C.c_double, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, _P],
    "npp_bn_bwd_apply2": [_T, _T, _T, _T, _P, _P, _T, _T, _P],
    "npp_bn_bwd_apply": [_T, _T, _T, _P, _T, _P],
    "npp_scale_mask": [_T, _P, _T, _T, _P],
    "npp_bn_fused_ok": [_T],
    "npp_se_fwd_multi": [_T, _P, C.c_int, _P, _P],
    "npp_se_bwd_multi": [_T, _P, C.c_int, _T, _P, C.c_int, _P],
    "npp_affine_add_fin_multi": [_P, C.c_int, _P],
    "npp_affine_add_fin_multi_x": [_P, C.c_int, C.c_int, _P],
    "npp_bn_bwd_apply_multi_x": [_P, C.c_int, C.c_int, _P],
    "npp_affine_add_fin_x": [_T, _T, _P, _T, _P, C.c_int, _P, C.c_int64, C.c_int, _P],
    "npp_bn_bwd_apply_fin_x": [_T, _T, _T, _P, C.c_int, C.c_double, _P, _P, _P, _P, _T, C.c_int, _P],
    "npp_bn_bwd_apply2_fin_x": [_T, _T, _T, _T, _P, C.c_int, C.c_double, _P, _P, _P, _P, _P, _P, _P, _P, _T, _T, C.c_int, _P],
    "npp_bn_bwd_reduce_multi": [_P, C.c_int, C.c_int, _P],
    "npp_bn_bwd_apply_multi": [_P, C.c_int, _P],
    "npp_mix_bn_fwd": [_P, C.c_int, _P, _T, _P],
    "npp_mix_bn_bwd": [_P, C.c_int, _P, _T, _P, _P, _P],
    "npp_mix_bn_fwd_n": [_P, C.c_int, _P, _T, C.c_double, _P],
    "npp_mix_bn_bwd_reduce": [_P, C.c_int, _T, _P, _P],
    "npp_mix_bn_bwd_apply": [_P, C.c_int, _P, _T, _P, C.c_double, _P, _P, _P],
    "npp_affine_add_fin": [_T, _T, _P, _T, _P, C.c_int, _P, C.c_int64, _P],
    "npp_bn_bwd_reduce_acc": [_T, _T, _T, _P, _P, C.c_int, _P],
    "npp_bn_bwd_reduce2_acc": [_T, _T, _T, _T, _P, _P, _P, C.c_int, _P],
    "npp_bn_bwd_apply_fin": [_T, _T, _T, _P, C.c_int, C.c_double, _P, _P, _P, _P, _T, _P],
    "npp_bn_bwd_apply2_fin": [_T, _T, _T, _T, _P, C.c_int, C.c_double, _P, _P, _P, _P, _P, _P, _P, _P, _T, _T, _P],
    "npp_bn_bwd_one_blocks": [C.c_int64, C.c_int64, C.c_int, C.c_int],
    "npp_bn_bwd_one": [_T, _T, _P, C.c_double, _P, _P, _P, _P, _T, _P, _P],
    "npp_bn_bwd_one2": [_T, _T, _T, _P, C.c_double, _P, _P, _P, _P, _P, _P, _P, _P, _T, _T, _P, _P],
    "npp_pool3x3_fwd": [_T, _T, _P, C.c_int, C.c_int, _P, _P],
    "npp_pool3x3_bwd": [_T, _P, _T, C.c_int, C.c_int, _P],
    "npp_pool3x3_bwd_acc": [_T, _P, _T, C.c_int, C.c_int, C.c_int, _P],
    "npp_pool2x2_fwd": [_T, _T, C.c_int, _P, _P],
    "npp_pool2x2_bwd": [_T, _T, _T, C.c_int, _P],
    "npp_global_avgpool": [_T, _P, _P],
    "npp_se_gate_fwd": [_P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P],
    "npp_se_gate_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P],
    "npp_scale_channels": [_T, _P, _T, _P],
    "npp_se_bwd_reduce": [_T, _T, _P, _P],
    "npp_se_bwd_apply": [_T, _P, _P, _T, _P],
    "npp_se_supported": [C.c_int],
    "npp_se_fwd": [_T, _P, _P, _P, _P, _T, _P, _P, _P, _P, _P],
    "npp_se_bwd": [_T, _T, _P, _P, _P, _P, _T, _P, _P, _P],
    "npp_se_bwd_acc": [_T, _T, _P, _P, _P, _P, _T, _P, _P, C.c_int, _P],
    "npp_se_param_grads": [_P, _P],
    "npp_se_param_grads_batched": [_P, C.c_int, _P, _P, C.c_int64, _P],
    "npp_bilinear_fwd": [_T, _T, _P],
    "npp_bilinear_bwd": [_T, _T, _P],
    "npp_bilinear_fwd_ac": [_T, _T, C.c_int, _P],
    "npp_bilinear_bwd_ac": [_T, _T, C.c_int, _P],
    "npp_copy": [_T, _T, _P],
    "npp_add_n": [_P, C.c_int, _T, _P],
    "npp_concat": [_P, C.c_int, _T, _P],
    "npp_concat_m": [_P, C.c_int, _T, _P, C.c_int64, _P],
    "npp_pose_targets": [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _P, _P],
    "npp_edge_target": [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P],
    "npp_normalize_image": [_P, C.c_int, C.c_int, C.c_int, _P, _P, _T, _P],
    "npp_nchw_to_nhwc": [_P, C.c_int, C.c_int, C.c_int, C.c_int, _T, _P],
    "npp_nhwc_to_nchw": [_T, _P, _P],
    "npp_nearest": [_T, _T, C.c_float, C.c_float, C.c_int, _P],
    "npp_weighted_sum_fwd": [_P, C.c_int, _P, _T, _P],
    "npp_weighted_sum_bwd": [_P, _P, C.c_int, _P, _T, _P, _P],
    "npp_interleave2": [_T, _T, _T, C.c_int, _T, _T, _P],
    "npp_mse_fwd": [_T, _P, _P, _P],
    "npp_mse_bwd": [_T, _P, _P, _T, _P],
    "npp_mse_w_fwd": [_T, _P, _P, _P, _P],
    "npp_mse_w_bwd": [_T, _P, _P, _P, _T, _P],
    "npp_ce_pixel_fwd": [_T, _P, C.c_int, C.c_int, _P, C.c_int, _P, _P, _P],
    "npp_kth_smallest": [_P, C.c_int64, C.c_int64, _P, _P, _P],
    "npp_ce_reduce": [_P, _P, _P, _P, C.c_int, C.c_int64, _P, C.c_float, C.c_int, _P, _P],
    "npp_ce_pixel_bwd": [_T, _P, C.c_int, C.c_int, _P, C.c_int, _P, _P, C.c_float, C.c_int, _P, _T, _P],
    "npp_ce_pixel_grad_up": [_T, _P, C.c_int, C.c_int, _P, C.c_int, _P, _P, C.c_float, C.c_int, _P, _P, _P],
    "npp_ce_pixel_grad_up_t": [_T, _P, _P, C.c_int, _P, _P, C.c_float, C.c_int, _P, _T, _P],
    "npp_edge_weights": [_P, C.c_int64, _P, _P],
    "npp_stamp": [_P, C.c_int, _P],
    "npp_loss_tail_fwd": [_P, C.c_int, _P, C.c_int, _P, _P, _P, _P],
    "npp_loss_tail_bwd": [_P, _P, _P, C.c_int, C.c_int, _P, _P, _P],
    "npp_edge_class_weights": [_P, C.c_int64, _P, _P, _P],
    "npp_adam_step": [_P, _P, C.c_int, _P, _P],
    "npp_bilinear_bwd_ws": [_T, _T, C.c_int, _P, C.c_int64, _P],
    "npp_conv_wgrad_batchable": [_T, _T, _G],
    "npp_conv_wgrad_batched_splits": [_T, _T, _G],
    "npp_conv_wgrad_batched_slabs": [_T, _T, _G],
    "npp_dwconv_bwd_weight_batchable": [_T, _T, _G],
    "npp_dwconv_bwd_weight_batched": [_P, C.c_int, _P, _P, C.c_int64, _P],
    "npp_conv_wgrad_batched": [_P, C.c_int, _P, _P, C.c_int64, _P],
    "npp_comm_unique_id": [_P],
    "npp_comm_init": [_P, C.c_int, C.c_int],
    "npp_comm_world": [],
    "npp_comm_destroy": [],
    "npp_allreduce_bucket": [_P, C.c_int64, C.c_int, C.c_int, _P],
    "npp_syncbn_exchange": [_P, C.c_int64, _P],
    "npp_p2p_handle_bytes": [],
    "npp_p2p_alloc": [C.c_int, C.c_int, C.c_int64, C.c_int, _P],
    "npp_p2p_open": [_P],
    "npp_p2p_channels": [],
    "npp_p2p_set_mode": [C.c_int],
    "npp_p2p_alloc_kind": [],
    "npp_p2p_exchange": [_P, C.c_int64, C.c_int, _P],
    "npp_p2p_exchange_slabs": [_P, C.c_int, C.c_int, _P],
    "npp_p2p_exchange_folded_test": [_P, C.c_int64, C.c_int, _P],
    "npp_p2p_status": [],
    "npp_p2p_reset_errors": [],
    "npp_p2p_close": [],
    "npp_parsing_confusion": [_T, _T, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P],
}
EXPORTS = sorted(list(_SIGS) + ["npp_version", "npp_last_error", "npp_clear_hip_error", "npp_packed_weight_elems", "npp_pack_job_blocks", "npp_reduce_blocks",
                                 "npp_dwconv_bwd_weight_ws", "npp_dwconv_bwd_weight_ws_zeroed", "npp_conv_fwd_ws_bytes", "npp_adam_chunk_elems",
                                 "npp_conv_wgrad_splits", "npp_debug_nonfinite", "npp_conv_wgrad_batched_ws", "npp_dwconv_bwd_weight_batched_ws", "npp_bilinear_bwd_ws_bytes",
                                 "npp_se_ws_floats", "npp_se_param_grads_batched_ws", "npp_p2p_capacity", "npp_p2p_set_timeout_ms", "npp_unpack_job_blocks"])


def kernel_source_hash() -> str:
    """sha256 over the HIP sources and headers the library is built from (sorted by name): profiles/*_pmc_traffic.json records
    it, and bench.py prints a measured `traffic` only when the kernels are still the ones the counters were collected on."""
    import hashlib
    h = hashlib.sha256()
    here = os.path.dirname(os.path.abspath(__file__))
    files = [os.path.join(here, "csrc", f) for f in sorted(os.listdir(os.path.join(here, "csrc"))) if f.endswith((".hip", ".h"))]
    files.append(os.path.join(os.path.dirname(here), "include", "npp_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def built_source_hash() -> str:
    """The source hash compiled into the loaded library (npp_version: '... src <hash>'); 'unknown' for a library built without it."""
    v = lib().npp_version().decode()
    return v.rsplit(" src ", 1)[1] if " src " in v else "unknown"


def lib():
    """Load (once) and return the C library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is not built. There is no CPU/eager fallback; "
                f"run npp_amd/csrc/build.sh (or __graft_entry__.build()).")
        L = C.CDLL(LIB_PATH)
        L.npp_version.restype = C.c_char_p
        L.npp_last_error.restype = C.c_char_p
        L.npp_packed_weight_elems.restype = C.c_int64
        L.npp_packed_weight_elems.argtypes = [C.c_int] * 5
        L.npp_pack_job_blocks.restype = C.c_int64
        L.npp_pack_job_blocks.argtypes = [C.c_int] * 5
        L.npp_dwconv_bwd_weight_ws.restype = C.c_int64
        L.npp_dwconv_bwd_weight_ws.argtypes = [_T, _G]
        L.npp_adam_chunk_elems.restype = C.c_int
        L.npp_adam_chunk_elems.argtypes = []
        L.npp_conv_fwd_ws_bytes.restype = C.c_int64
        L.npp_conv_fwd_ws_bytes.argtypes = [_T, _T, _G]
        L.npp_dwconv_bwd_weight_ws_zeroed.restype = C.c_int
        L.npp_dwconv_bwd_weight_ws_zeroed.argtypes = [_T, _G]
        L.npp_conv_wgrad_splits.restype = C.c_int
        L.npp_conv_wgrad_splits.argtypes = [_T, _T, _G]
        L.npp_reduce_blocks.restype = C.c_int
        L.npp_reduce_blocks.argtypes = [C.c_int64, C.c_int64, C.c_int]
        L.npp_bilinear_bwd_ws_bytes.restype = C.c_int64
        L.npp_bilinear_bwd_ws_bytes.argtypes = [_T, _T]
        L.npp_dwconv_bwd_weight_batched_ws.restype = C.c_int64
        L.npp_dwconv_bwd_weight_batched_ws.argtypes = [C.c_int]
        L.npp_conv_wgrad_batched_ws.restype = C.c_int64
        L.npp_conv_wgrad_batched_ws.argtypes = [C.c_int]
        L.npp_debug_nonfinite.restype = C.c_int64
        L.npp_debug_nonfinite.argtypes = [_T, _P]
        L.npp_se_ws_floats.restype = C.c_int64
        L.npp_se_ws_floats.argtypes = [C.c_int, C.c_int]
        L.npp_se_param_grads_batched_ws.restype = C.c_int64
        L.npp_se_param_grads_batched_ws.argtypes = [_P, C.c_int]
        L.npp_p2p_capacity.restype = C.c_int64
        L.npp_p2p_capacity.argtypes = []
        L.npp_unpack_job_blocks.restype = C.c_int64
        L.npp_unpack_job_blocks.argtypes = [C.c_int, C.c_int, C.c_int]
        L.npp_p2p_set_timeout_ms.restype = C.c_int64
        L.npp_p2p_set_timeout_ms.argtypes = [C.c_int64]
        for name, sig in _SIGS.items():
            f = getattr(L, name)
            f.restype = C.c_int
            f.argtypes = sig
        _lib = _NanChecked(L) if os.environ.get("NPP_NAN_CHECK") else L
    return _lib


class _NanChecked:
    """Debugging aid (NPP_NAN_CHECK=1): after every library call, count the NaN / Inf elements of each NppTensor argument
    (inputs and outputs alike) and report the first calls that see any -- which kernel made a NaN, or was handed one."""

    def __init__(self, L):
        self._L, self._left, self._calls = L, int(os.environ.get("NPP_NAN_CHECK_REPORTS", "12")), 0

    def _conv_detail(self, args, stream):
        """A convolution whose output holds NaN: the row padding of its inputs and its packed weights, and the geometry."""
        x, y, g = args[0]._obj, args[4]._obj, args[6]._obj
        wp = args[1] if isinstance(args[1], int) else getattr(args[1], "value", None)

        def count(ptr, c, ld, pixels, dtype):
            t = NppTensor(ptr, 1, 1, pixels, c, ld, dtype, 0)
            return self._L.npp_debug_nonfinite(C.byref(t), stream)
        msg = [f"geom k{g.kh}x{g.kw} s{g.sh} p{g.ph} d{g.dh} up{g.uph} relu_in{g.relu_in}",
               f"x {x.n}x{x.c}x{x.h}x{x.w} ld {x.ld}: whole rows (with padding) non-finite {count(x.ptr, x.ld, x.ld, x.n * x.h * x.w, x.dtype)}"]
        m = getattr(args[3], "_obj", None)
        if isinstance(m, NppTensor) and m.ptr and m.dtype in (NPP_F32, NPP_BF16):
            msg.append(f"mask c {m.c} ld {m.ld}: whole rows non-finite {count(m.ptr, m.ld, m.ld, m.n * m.h * m.w, m.dtype)}")
        if wp:
            cp = (x.c + 7) // 8 * 8
            kpad = (g.kh * g.kw * cp + 63) // 64 * 64
            rows = (y.c + 31) // 32 * 32
            msg.append(f"packed weights [{rows}][{kpad}]: non-finite {count(wp, kpad, kpad, rows, x.dtype)}; "
                       f"first {y.c} rows {count(wp, kpad, kpad, y.c, x.dtype)}; K columns below {g.kh * g.kw * cp}: "
                       f"{count(wp, g.kh * g.kw * cp, kpad, rows, x.dtype)}")
        sys.stderr.write("[npp nan-check]    " + " | ".join(msg) + "\n")

    def __getattr__(self, name):
        f = getattr(self._L, name)
        if name not in _SIGS or name.startswith("npp_comm"):
            return f
        sig = _SIGS[name]

        def checked(*args):
            rc = f(*args)
            self._calls += 1
            if self._left > 0:
                stream = args[-1] if sig and sig[-1] is _P and not hasattr(args[-1], "_obj") else None
                for k, a in enumerate(args):
                    obj = getattr(a, "_obj", None)
                    if isinstance(obj, NppTensor) and obj.ptr and obj.dtype in (NPP_F32, NPP_BF16):
                        bad = self._L.npp_debug_nonfinite(a, stream)
                        if bad:
                            self._left -= 1
                            sys.stderr.write(f"[npp nan-check] call #{self._calls} {name}: argument {k} "
                                             f"({obj.n}x{obj.c}x{obj.h}x{obj.w} ld {obj.ld}) has {bad} non-finite elements\n")
                            if name in ("npp_conv_fwd", "npp_conv_fwd_ws") and k == 4:
                                self._conv_detail(args, stream)
            return rc
        return checked


def check(rc: int, what: str = ""):
    if rc != 0:
        raise RuntimeError(f"libnpp_hip {what} failed ({rc}): {lib().npp_last_error().decode()}")


def npp_dtype(t: torch.dtype) -> int:
    if t == torch.float32:
        return NPP_F32
    if t == torch.bfloat16:
        return NPP_BF16
    raise TypeError(f"libnpp_hip supports float32 / bfloat16 activations, got {t}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream_ptr() -> int:
    """Raw handle of torch's current HIP stream.  Called once per launch (~5000 times per step): the two C calls cost
    ~0.3 us where `torch.cuda.current_stream().cuda_stream` costs ~8 (device-index and availability checks, an env read)."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def new_nhwc(n: int, c: int, h: int, w: int, dtype, device, zero: bool = False, ld: int = 0) -> torch.Tensor:
    """Logical [n,c,h,w] tensor stored NHWC (optionally with a wider pixel stride ld)."""
    if not ld:
        ld = c if c % 8 == 0 else (c + 7) // 8 * 8    # odd channel counts (3, 6, 20, 2): 16-byte rows, zero pad
        zero = zero or ld != c
    buf = (torch.zeros if zero else torch.empty)((n, h, w, ld), dtype=dtype, device=device)
    t = buf.permute(0, 3, 1, 2)
    return t[:, :c] if ld != c else t


def nhwc_ld(t: torch.Tensor):
    """Pixel stride (elements) if `t` is a logical-NCHW tensor with NHWC memory (channel stride 1, dense pixels,
    possibly a channel slice of a wider buffer); None otherwise."""
    if t.dim() != 4:
        return None
    n, c, h, w = t.shape
    sn, sc, sh, sw = t.stride()
    if sc == 1 and sw >= c and sh == w * sw and sn == h * sh and w > 1:      # the common case first
        return sw
    if c > 1 and sc != 1:
        return None
    if w > 1:
        ld = sw
    elif h > 1:
        ld = sh
    elif n > 1:
        ld = sn
    else:
        ld = c
    if ld < c:
        return None
    if h > 1 and w > 1 and sh != w * ld:
        return None
    if n > 1 and (h > 1 or w > 1) and sn != h * w * ld:
        return None
    return ld


def is_nhwc(t: torch.Tensor) -> bool:
    return nhwc_ld(t) is not None


def desc(t: torch.Tensor) -> NppTensor:
    """NppTensor view of a logical-NCHW tensor whose memory is NHWC (channel stride 1)."""
    if t.dim() == 4:
        # fast path (called ~8000 times per step): dense NHWC, or a channel slice of a wider NHWC buffer
        n, c, h, w = t.shape
        sn, sc, sh, sw = t.stride()
        if sc == 1 and sw >= c and sh == w * sw and sn == h * sh:
            dt = t.dtype
            return NppTensor(t.data_ptr(), n, h, w, c, sw, NPP_BF16 if dt == torch.bfloat16 else npp_dtype(dt), 0)
    ld = nhwc_ld(t)
    if ld is None:
        raise ValueError(f"tensor is not NHWC-strided: shape {tuple(t.shape)} strides {t.stride()}")
    n, c, h, w = t.shape
    return NppTensor(t.data_ptr(), n, h, w, c, ld, npp_dtype(t.dtype), 0)


def to_nhwc(t: torch.Tensor) -> torch.Tensor:
    """Return t if it already is NHWC-strided, else a channels-last copy (made by our own copy kernel when
    possible)."""
    if is_nhwc(t):
        return t
    return t.contiguous(memory_format=torch.channels_last)


def geom(kh, kw, sh, sw, ph, pw, dh, dw, up=1, relu_in=0) -> NppConvGeom:
    uph, upw = (up, up) if isinstance(up, int) else up
    return NppConvGeom(kh, kw, sh, sw, ph, pw, dh, dw, uph, upw, int(relu_in))


def ptr(t):
    return None if t is None else t.data_ptr()


def tref(t):
    return None if t is None else C.byref(desc(t))
