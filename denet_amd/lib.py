"""ctypes binding of libdenet_hip.so (include/denet_hip.h).

This is the only door from the Python host side to the HIP kernels: plain pointers, sizes and a stream.
There is deliberately NO fallback: if the library is missing or a call fails the caller gets an exception.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdenet_hip.so")

P = ctypes.c_void_p
I = ctypes.c_int
L = ctypes.c_long
F = ctypes.c_float
Z = ctypes.c_size_t
U64 = ctypes.c_uint64
D = ctypes.c_double

# name -> (restype, argtypes); mirrors include/denet_hip.h declaration by declaration
SIGNATURES = {
    "denet_last_error": (ctypes.c_char_p, []),
    "denet_abi_version": (I, []),
    "denet_device_info": (I, [I, P, P, P, I]),
    "denet_host_py_random_sample": (I, [P, P, I, I, P, P]),
    "denet_host_edit_samples": (I, [P, P, P, P, I, I, I, P, P, I, P, P, P, P]),
    "denet_host_mt_prefetch": (I, [P, P, L, P, P, P, I, P]),
    "denet_host_edit_samples_stream": (I, [P, L, P, P, P, P, I, I, I, P, P, I, P, P, P, P]),
    "denet_host_handoff_stream": (I, [P, L, P, P, P, P, P, I, I, I, I, I, P, P, I, P, P, P, P, P]),
    "denet_host_handoff_boxes_stream": (I, [P, L, P, P, P, P, I, I, I, I, I, P, P, I, P, P]),
    "denet_host_handoff_boxes_stream_u": (I, [P, L, P, P, P, P, I, I, I, I, I, P, P, I, P, P, P]),
    "denet_host_mt_uniforms": (I, [P, L, P]),
    "denet_host_detect_targets": (I, [P, P, P, P, I, I, I, I, I, I, ctypes.c_double, ctypes.c_double, P, P, P, P]),
    "denet_conv_fwd": (I, [P, P, P, P, P] + [I] * 12 + [P]),
    "denet_conv_fwd_act": (I, [P, P, P, P, P] + [I] * 13 + [P]),
    "denet_spin": (I, [I, P]),
    "denet_tune_export": (I, [P, I]),
    "denet_tune_import": (I, [P, I]),
    "denet_tune_clear": (I, []),
    "denet_conv_fwd_stats": (I, [P, P, P, P, P, P, Z, P] + [I] * 12 + [P]),
    "denet_conv_dgrad": (I, [P, P, P, P] + [I] * 12 + [P]),
    "denet_conv_wgrad_workspace_bytes": (Z, [I] * 7),
    "denet_conv_wino_workspace_bytes": (Z, [I] * 6),
    "denet_conv_wino_tune": (I, [P, Z, P, Z] + [I] * 6 + [P]),
    "denet_conv_wino_wgrad": (I, [P, P, P, P, P, Z, P, Z] + [I] * 6 + [P]),
    "denet_conv_wino_filter": (I, [P, P, I, I, I, I, P]),
    "denet_conv_wino_fwd": (I, [P] * 8 + [Z] + [I] * 6 + [P]),
    "denet_conv_wino_fwd_act": (I, [P] * 7 + [I, P, Z] + [I] * 6 + [P]),
    "denet_conv_wino_fwd_stats": (I, [P] * 8 + [Z, P, P, Z] + [I] * 6 + [P]),
    "denet_conv_wino_fwd_stats_up": (I, [P] * 8 + [Z, P, P, Z] + [I] * 6 + [P]),
    "denet_conv_wino_dgrad": (I, [P] * 6 + [Z] + [I] * 6 + [P]),
    "denet_gemm_bf16x3_ok": (I, [I] * 3),
    "denet_gemm_bf16x3_nt": (I, [P] * 4 + [I] * 3 + [P]),
    "denet_transpose_f32": (I, [P, P, I, I, P]),
    "denet_conv_wino2f_ok": (I, [I] * 5),
    "denet_conv_wino2f": (I, [P] * 5 + [I, P, Z, P] + [I] * 5 + [P]),
    "denet_conv_wino2f_wgrad_ok": (I, [I] * 5),
    "denet_conv_wino2f_wgrad_workspace_bytes": (Z, [I] * 3),
    "denet_conv_wino2f_wgrad": (I, [P] * 4 + [Z] + [I] * 5 + [P]),
    "denet_conv_stem_ok": (I, [I] * 13),
    "denet_conv_stem_fwd": (I, [P] * 5 + [Z, P] + [I] * 3 + [P]),
    "denet_conv_stem_fwd_from": (I, [P, I] + [P] * 4 + [Z, P] + [I] * 3 + [P]),
    "denet_conv_stem_wgrad_from": (I, [P, I] + [P] * 3 + [Z] + [I] * 3 + [P]),
    "denet_conv_stem_fwd_act": (I, [P, I] + [P] * 3 + [I, P, Z, P] + [I] * 3 + [P]),
    "denet_conv_stem_wgrad_workspace_bytes": (Z, []),
    "denet_conv_stem_wgrad": (I, [P] * 4 + [Z] + [I] * 3 + [P]),
    "denet_conv_tune": (I, [I, P, P, P, P, P, P, Z] + [I] * 12 + [P]),
    "denet_conv_tuned": (I, [I] * 11 + [P, P, P]),
    "denet_conv_last_config": (I, [P] * 5),
    "denet_conv_wino4f_mode": (I, [I]),
    "denet_conv_wino4g_mode": (I, [I]),
    "denet_conv_profile": (I, [I]),
    "denet_conv_profile_count": (I, []),
    "denet_conv_profile_read": (I, [I] + [P] * 5),
    "denet_conv_wgrad": (I, [P, P, P, P, Z] + [I] * 12 + [P]),
    "denet_bn_final_arm_stats": (I, [L, I, F, F, P, P, P, P, P, I]),
    "denet_bn_final_arm_sums": (I, [L, I, P, P, P, P, I]),
    "denet_bn_final_disarm": (I, []),
    "denet_bn_final_mode": (I, [I]),
    "denet_bn_workspace_bytes": (Z, [L, I]),
    "denet_bn_fwd_train": (I, [P] * 10 + [L, I, F, F, I, P]),
    "denet_bn_fwd_train_pre": (I, [P] * 10 + [I, L, I, F, F, I, P]),
    "denet_bn_relu_pool_fwd_train": (I, [P] * 10 + [I, P] + [I] * 9 + [F, F, P]),
    "denet_bn_relu_pool_bwd": (I, [P] * 11 + [I] * 9 + [P]),
    "denet_bn_relu_pool_fwd_train_xhat": (I, [P] * 11 + [I, P] + [I] * 9 + [F, F, P]),
    "denet_bn_relu_pool_bwd_sums": (I, [P] * 9 + [I] * 6 + [P]),
    "denet_bn_relu_pool_bwd_apply": (I, [P] * 9 + [I] * 9 + [P]),
    "denet_bn_fold": (I, [P] * 6 + [F, P, P, I, L, P]),
    "denet_bn_stats_final": (I, [P, I, L, I, F, F, P, P, P, P, P]),
    "denet_bn_apply": (I, [P] * 7 + [L, I, I, P]),
    "denet_bn_bwd_sums": (I, [P] * 11 + [L, I, I, P]),
    "denet_bn_bwd_apply": (I, [P] * 10 + [L, I, I, P]),
    "denet_conv_wino_fwd_fold": (I, [P] * 7 + [I, P, Z, P, P, Z] + [I] * 6 + [P]),
    "denet_conv_wino_dgrad_fold": (I, [P] * 8 + [Z, P, P, Z] + [I] * 6 + [P, P]),
    "denet_conv_wino_dgrad_sums": (I, [P] * 7 + [Z, P, P, Z] + [I] * 6 + [P]),
    "denet_conv_dgrad_sums": (I, [P] * 6 + [Z, P] + [I] * 12 + [P]),
    "denet_conv_dgrad_1x1t": (I, [P] * 6 + [Z, P] + [I] * 5 + [P]),
    "denet_conv_dgrad_t": (I, [P] * 6 + [Z, P] + [I] * 12 + [P]),
    "denet_conv_wino2f_sums": (I, [P] * 5 + [I, P, Z, P, P] + [I] * 5 + [P]),
    "denet_conv_dgrad_s2_ok": (I, [I] * 5),
    "denet_conv_dgrad_s2_stats_rows": (I, [I] * 3),
    "denet_conv_dgrad_s2_pack": (I, [P, P, I, I, P]),
    "denet_conv_dgrad_s2": (I, [P] * 6 + [Z, P] + [I] * 5 + [P]),
    "denet_conv_wino4t_ok": (I, [I] * 5),
    "denet_conv_wino4t_stats_rows": (I, [I] * 3),
    "denet_conv_wino4t_pack": (I, [P, P, I, I, P]),
    "denet_conv_wino4t_sums": (I, [P] * 5 + [I, P, Z, P, P] + [I] * 5 + [P]),
    "denet_bn_bwd_final": (I, [P, I, L, I, P, P, P, P]),
    "denet_conv_wino_wgrad_dm": (I, [P, P, P, P, P, Z, P, Z] + [I] * 6 + [P]),
    "denet_bn_fwd_test": (I, [P] * 8 + [I, L, I, F, I, P]),
    "denet_bn_bwd": (I, [P] * 12 + [L, I, I, P]),
    "denet_maxpool_fwd": (I, [P, P, P] + [I] * 9 + [P]),
    "denet_maxpool_bwd": (I, [P, P, P] + [I] * 9 + [P]),
    "denet_avgpool_fwd": (I, [P, P] + [I] * 9 + [P]),
    "denet_avgpool_bwd": (I, [P, P] + [I] * 9 + [P]),
    "denet_pool_inv_fwd": (I, [P, P] + [I] * 6 + [P]),
    "denet_pool_inv_bwd": (I, [P, P] + [I] * 6 + [P]),
    "denet_host_resample_coeffs": (I, [I, D, D, I, I, P, P, L]),
    "denet_image_crop": (I, [P, P] + [I] * 9 + [P]),
    "denet_image_reduce": (I, [P, P, I, I, I, I, P]),
    "denet_image_resample_pass": (I, [P, P, I, I, I, I, P, P, I, P]),
    "denet_image_render_batch": (I, [I] + [P] * 13 + [I, I, P, P, Z, P, P, P, Z, P, P]),
    "denet_image_finish": (I, [P, P, I, I, I, I, P, P, P, P, I, P, P]),
    "denet_border_fwd": (I, [P, P] + [I] * 8 + [P]),
    "denet_border_bwd": (I, [P, P] + [I] * 8 + [P]),
    "denet_crop_mirror_fwd": (I, [P, P] + [I] * 6 + [F, F, I, U64, P]),
    "denet_crop_mirror_bwd": (I, [P, P] + [I] * 6 + [F, F, I, U64, P]),
    "denet_dropout": (I, [P, P, I, I, I, I, F, U64, P]),
    "denet_concat_fwd": (I, [P, P, P, L] + [I] * 5 + [P]),
    "denet_concat_bwd": (I, [P, P, P, L] + [I] * 5 + [P]),
    "denet_add_bias": (I, [P, P, P, L, I, P]),
    "denet_nchw_to_nhwc": (I, [P, P] + [I] * 5 + [P]),
    "denet_nhwc_to_nchw": (I, [P, P] + [I] * 5 + [P]),
    "denet_add": (I, [P, P, P, L, I, P]),
    "denet_relu_fwd": (I, [P, P, L, P]),
    "denet_relu_bwd": (I, [P, P, P, L, P]),
    "denet_colsum_workspace_bytes": (Z, [L, I]),
    "denet_colsum": (I, [P, P, P, L, I, P]),
    "denet_solver_step": (I, [P, P, P, L, L, F, F, I, F, F, I, P]),
    "denet_solver_adam": (I, [P, P, P, P, L, L, F, F, F, I, F, F, P]),
    "denet_scale": (I, [P, L, F, P]),
    "denet_corner_fwd": (I, [P, P] + [I] * 5 + [P]),
    "denet_loss_workspace_bytes": (Z, []),
    "denet_corner_loss": (I, [P] * 5 + [I] * 5 + [F, P]),
    "denet_sparse_fwd": (I, [P, P, P, P] + [I] * 10 + [P]),
    "denet_sparse_sort_workspace_bytes": (Z, [I] * 5),
    "denet_sparse_sort_is_single": (I, [I] * 5),
    "denet_sparse_sort": (I, [P, P, Z] + [I] * 5 + [P]),
    "denet_sparse_bwd": (I, [P, P, P, Z, P] + [I] * 10 + [P]),
    "denet_detect_loss": (I, [P] * 9 + [I] * 6 + [F, F, F, I, P]),
    "denet_detect_decode": (I, [P] * 5 + [I] * 6 + [F, P]),
    "denet_detect_nms": (I, [P] * 5 + [I, I, I, F, F, P]),
    "denet_soft_nms_batch_host": (L, [P, P, P, P, I, I, I, F, F, P, P, P, P, L]),
    "denet_soft_nms_host": (I, [P, P, I, F, P, P, P]),
    "denet_soft_nms_workspace_bytes": (Z, [I, I, I]),
    "denet_soft_nms_batch": (I, [P, P, P, P, I, I, I, F, F, P, P, P, P, P, P, Z, P]),
    "denet_build_samples_workspace_bytes": (Z, [I] * 6),
    "denet_build_samples": (I, [P, P, P, P, P, Z] + [I] * 4 + [F, I, I, I, P]),
    "denet_build_samples_stats": (I, [P, Z] + [I] * 6 + [P, P, P]),
    "denet_edit_samples_device": (I, [P, P, I, I, P, L, L, P, P] + [I] * 4 + [P, P, P]),
    "denet_host_cluster_samples": (I, [P, I, F, I, P, P]),
    "denet_samples_finish_host": (I, [P, P, P, I, I, I, I, P]),
}


class DenetHipError(RuntimeError):
    pass


_lib = None


def load():
    """Loads the library (once). Raises if it has not been built: there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DenetHipError(
            "libdenet_hip.so is missing (%s): run `python -m denet_amd.build` / __graft_entry__.build() first. "
            "The DeNet hot path has no CPU fallback." % LIB_PATH)
    # torch first: its wheel carries the HIP runtime (libamdhip64) that owns the streams and the memory this library is handed;
    # loaded the other way round the dynamic linker would bind the library to a second copy of the runtime (/opt/rocm) in which
    # torch's streams do not exist ("no ROCm-capable device" at the first launch)
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().denet_last_error().decode("utf-8", "replace")
        raise DenetHipError("%s failed (rc=%d): %s" % (what or "denet call", rc, msg))


def ptr(t):
    """device (or host) pointer of a torch tensor / None"""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
