"""ctypes binding of ``libhands_hip.so`` (declared in include/hands_hip.h).

The product path has no CPU fallback: if the shared library is missing or cannot be loaded,
:func:`lib` raises and every op of :mod:`hands_amd` fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HANDS_HIP_LIB") or os.path.join(_HERE, "libhands_hip.so")   # HANDS_HIP_LIB: developer A/B builds only

c_float_p = C.c_void_p  # raw device pointers travel as integers


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "B", "H", "W", "Cin", "Ho", "Wo", "Cout", "KH", "KW", "stride", "pad",
        "in_pix_stride", "out_pix_stride", "res_pix_stride", "Kpad", "act")]


ACT_NONE, ACT_RELU, ACT_GELU, ACT_LEAKY_RELU = 0, 1, 2, 3


class ManoConsts(C.Structure):
    _fields_ = [("pose_mean", C.c_void_p), ("J_template", C.c_void_p), ("J_shapedirs", C.c_void_p),
                ("lbs_weights", C.c_void_p), ("tip_ids", C.c_void_p)]


class ManoOut(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("joints3d", C.c_void_p), ("v3d_cam", C.c_void_p),
                ("j3d_cam", C.c_void_p), ("j2d_norm", C.c_void_p), ("cam_t", C.c_void_p)]


class ManoSide(C.Structure):
    _fields_ = [("consts", ManoConsts), ("blend_w", C.c_void_p), ("blend_bias", C.c_void_p), ("rot", C.c_void_p),
                ("betas", C.c_void_p), ("cam_wp", C.c_void_p), ("out", ManoOut)]


class PackedDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("Cin", "Cout", "Cout_pad", "Kpad")]


class EvalIn(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "pred_j3d_r", "pred_j3d_l", "gt_j3d_r", "gt_j3d_l", "pred_j2d_r", "pred_j2d_l", "gt_j2d_r", "gt_j2d_l",
        "is_valid", "right_valid", "left_valid", "joints_valid_r", "joints_valid_l")]


class EvalOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "mpjpe_ra_h", "mpjpe_pa_ra_r", "mpjpe_pa_ra_l", "mpjpe_pa_ra_h", "mrrpe_rl", "pix_err_r", "pix_err_l")]


# name -> argtypes; every function returns int (0 = ok) except the two noted below
_P, _I, _F = C.c_void_p, C.c_int, C.c_float
class ConvJob(C.Structure):
    """hands_conv_job (include/hands_hip.h): one member of a grouped pointwise launch."""
    _fields_ = [("desc", C.POINTER(ConvDesc)), ("in_", C.c_void_p), ("w_packed", C.c_void_p), ("bias", C.c_void_p),
                ("residual", C.c_void_p), ("out", C.c_void_p), ("pre_scale", C.c_void_p), ("pre_shift", C.c_void_p)]


SIGNATURES = {
    "hands_conv2d_nhwc_f32": [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P],
    "hands_conv3x3_winograd_supported": [C.POINTER(ConvDesc)],
    "hands_conv3x3_winograd_f32": [C.POINTER(ConvDesc), _P, _P, _P, _P, _P],
    "hands_conv3x3_winograd4_supported": [C.POINTER(ConvDesc)],
    "hands_conv3x3_winograd4_f32": [C.POINTER(ConvDesc), _P, _P, _P, _P, _P],
    "hands_conv2d_splitk_factor": [C.POINTER(ConvDesc)],
    "hands_conv2d_group_class": [C.POINTER(ConvDesc), _I],
    "hands_conv2d_group_f32": [C.POINTER(ConvJob), _I, _P],
    "hands_conv2d_nhwc_splitk_f32": [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, C.c_longlong, _P],
    "hands_conv2d_nhwc_splitk_fused_f32": [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _I, _P, C.c_longlong, _P, C.c_longlong, _P],
    "hands_conv2d_streamk_grid": [C.POINTER(ConvDesc)],
    "hands_conv2d_nhwc_streamk_f32": [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, C.c_longlong, _I, _P],
    "hands_conv1x1_dual_nhwc_f32": [C.POINTER(ConvDesc), _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "hands_conv2d_nhwc_splitk_n_f32": [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _I, _P, C.c_longlong, _P],
    "hands_conv2d_nhwc_pre_f32": [C.POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, _I, _P, C.c_longlong, _P],
    "hands_stem_conv_maxpool_nhwc_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "hands_stem_conv_maxpool_nchw_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "hands_nchw3_to_nhwc4_f32": [_P, _P, _I, _I, _I, _P],
    "hands_maxpool3x3s2_nhwc_f32": [_P, _P, _I, _I, _I, _I, _P],
    "hands_sumpool_nhwc_f32": [_P, _P, _I, _I, _I, _I, _P],
    "hands_avgpool_nhwc_f32": [_P, _P, _I, _I, _I, _I, _P],
    "hands_image_posenc_nhwc_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "hands_frontend_dense_maps_f32": [_P, _P, _P, _P, _I, _I, _I, _P],
    "hands_dense_posenc_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "hands_concat_nhwc_f32": [_P, _I, _I, _P, _I, _P, C.c_longlong, _I, _P, _I, _I, _I, _I, _P],
    "hands_upsample_bilinear_ac_f32": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "hands_rot_leftmul_f32": [_P, _P, _I, _P],
    "hands_perspective_correction_f32": [_P, _P, _P, _P, _I, _P],
    "hands_kpe_concat_f32": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "hands_hmr_init_f32": [_P, _P, _I, _I, _I, _P],
    "hands_rot6d_to_matrix_f32": [_P, _I, _P, _I, _P],
    "hands_flip_swap_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "hands_matrix_to_axis_angle_f32": [_P, _P, C.c_longlong, _P],
    "hands_axis_angle_to_matrix_f32": [_P, _P, C.c_longlong, _P],
    "hands_grasp_input_f32": [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P],
    "hands_mano_pose_f32": [C.POINTER(ManoConsts), _P, _P, _I, _P, _I, _P, _P, _I, _P],
    "hands_mano_heads_f32": [C.POINTER(ManoSide), _I, _P, _I, _F, _F, _I, _I, _P],
    "hands_mano_skin_f32": [C.POINTER(ManoConsts), _P, _I, _P, _P, _P, _P, _F, _F, C.POINTER(ManoOut), _I, _P],
    "hands_resize_crop_nchw3_to_nhwc4_f32": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "hands_layernorm_f32": [_P, _P, _P, _P, _P, _I, _I, _I, _F, _P],
    "hands_add_pos_f32": [_P, _P, _P, _I, _I, _I, _P],
    "hands_kpe_encode_f32": [_P, _P, _P, _I, _I, _I, _P],
    "hands_attention_f32": [_P, _P, _I, _I, _I, _I, _F, _P],
    "hands_cross_attention_1q_f32": [_P, _P, _P, _I, _I, _I, _I, _F, _P],
    "hands_rot6d_to_matrix_cols_f32": [_P, _I, _P, _I, _P],
    "hands_upsample_bilinear_add_f32": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "hands_pool2x2_nhwc_f32": [_P, _P, _I, _I, _I, _I, _I, _P],
    "hands_channel_pool_f32": [_P, _P, C.c_longlong, _I, _P],
    "hands_gate_apply_f32": [_P, _P, _I, _P, _P, C.c_longlong, _I, _P],
    "hands_add_embed2_f32": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "hands_add_rowvec_f32": [_P, _P, _P, _I, _I, _I, _P],
    "hands_token_sum_f32": [_P, _P, _I, _I, _I, _P],
    "hands_bn_leaky_f32": [_P, _P, _P, _P, C.c_longlong, _I, _P],
    "hands_upsample_nearest2x_add_f32": [_P, _P, _P, _I, _I, _I, _I, _P],
    "hands_spatial_softmax_f32": [_P, _I, _P, _P, _I, _I, _I, _I, _P],
    "hands_flash_attention_f32": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P],
    "hands_ceiling_hbm_read_f32": [_P, C.c_longlong, _P, _P],
    "hands_eval_metrics_f32": [C.POINTER(EvalIn), C.POINTER(EvalOut), _I, _P],
    "hands_mano_pose_aa_f32": [C.POINTER(ManoConsts), _P, _P, _I, _P, _I, _P, _P, _I, _P],
    "hands_gt_targets_f32": [_P, _P, _P, _P, _F, _P, _P, _P, _I, _I, _P],
    "hands_unnormalize_kp2d_f32": [_P, _P, C.c_longlong, _F, _P],
    "hands_frontend_boxes_f32": [_P, _P, _I, _P, _I, _I, _I, C.c_double] + [_P] * 10 + [_P],
    "hands_warp_affine_cubic_norm_f32": [_P, _P, _P, _I, _I, _I, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_float), _P],
    # host-side packing (csrc/pack.cpp): HOST pointers
    "hands_pack_conv_dims": [_I, _I, _I, _I, _I, C.POINTER(PackedDims)],
    "hands_fold_bn_f32": [_I, C.c_longlong, _P, _P, _P, _P, _P, C.c_double, _P, _P],
    "hands_pack_conv_f64": [_I, _I, _I, _I, _I, _P, _P, _P, _P],
    "hands_pack_linear_f64": [_I, _I, _P, _P, _P, _I, _P, _I, _P, _P],
    "hands_pack_conv1x1_dual_f64": [_I, _I, _I, _P, _P, _P, _P, _P, _P],
    "hands_pack_mano_f32": [_P] * 10,
    "hands_pack_conv3x3_winograd_f64": [_I, _I, _P, _P],
    "hands_pack_conv3x3_winograd4_f64": [_I, _I, _P, _P],
}
EXTRA_SYMBOLS = ("hands_abi_version", "hands_error_string", "hands_conv2d_workspace_floats", "hands_pack_conv3x3_winograd_floats", "hands_conv3x3_winograd_executed_macs",
                 "hands_pack_conv3x3_winograd4_floats", "hands_conv3x3_winograd4_executed_macs",
                 "hands_conv2d_streamk_workspace_bytes", "hands_stream_is_capturing", "hands_csrc_sha16", "hands_ceiling_mfma_f32")

ABI_VERSION = 5      # HANDS_ABI_VERSION of include/hands_hip.h this wrapper was written against
_lib = None


def lib():
    """Load (once) and return the shared library; raise if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(
            f"hands_amd: {LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C hands_amd/csrc`. There is no CPU fallback.")
    h = C.CDLL(LIB_PATH)
    h.hands_abi_version.restype = C.c_int
    if h.hands_abi_version() != ABI_VERSION:
        # a stale build (e.g. an A/B variant through $HANDS_HIP_LIB) would otherwise fail late with an AttributeError
        raise RuntimeError(f"hands_amd: {LIB_PATH} has ABI version {h.hands_abi_version()}, this package needs {ABI_VERSION}: "
                           "rebuild it (`make -C hands_amd/csrc`)")
    for name, argtypes in SIGNATURES.items():
        fn = getattr(h, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    h.hands_conv2d_workspace_floats.restype = C.c_longlong
    h.hands_conv2d_workspace_floats.argtypes = [C.POINTER(ConvDesc), C.c_int]
    h.hands_conv2d_streamk_workspace_bytes.restype = C.c_longlong
    h.hands_conv2d_streamk_workspace_bytes.argtypes = []
    h.hands_pack_conv3x3_winograd_floats.restype = C.c_longlong
    h.hands_pack_conv3x3_winograd_floats.argtypes = [C.c_int, C.c_int]
    h.hands_conv3x3_winograd_executed_macs.restype = C.c_longlong
    h.hands_conv3x3_winograd_executed_macs.argtypes = [C.POINTER(ConvDesc)]
    h.hands_pack_conv3x3_winograd4_floats.restype = C.c_longlong
    h.hands_pack_conv3x3_winograd4_floats.argtypes = [C.c_int, C.c_int]
    h.hands_conv3x3_winograd4_executed_macs.restype = C.c_longlong
    h.hands_conv3x3_winograd4_executed_macs.argtypes = [C.POINTER(ConvDesc)]
    h.hands_abi_version.restype = C.c_int
    h.hands_stream_is_capturing.restype = C.c_int
    h.hands_stream_is_capturing.argtypes = [C.c_void_p]
    h.hands_csrc_sha16.restype = C.c_char_p
    h.hands_csrc_sha16.argtypes = []
    h.hands_ceiling_mfma_f32.restype = C.c_longlong
    h.hands_ceiling_mfma_f32.argtypes = [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p]
    h.hands_error_string.restype = C.c_char_p
    h.hands_error_string.argtypes = [C.c_int]
    _lib = h
    return h


def check(code: int, what: str = ""):
    if code != 0:
        msg = lib().hands_error_string(code).decode()
        raise RuntimeError(f"hands_amd: {what} failed with code {code}: {msg}")


def ptr(t, offset_elems: int = 0):
    """Device pointer of a torch tensor (optionally offset by elements), or None."""
    if t is None:
        return None
    return t.data_ptr() + offset_elems * t.element_size()
