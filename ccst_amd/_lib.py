"""ctypes binding of libccst_hip.so (the C ABI declared in include/ccst_hip.h).

The product path has no CPU or eager-PyTorch fallback: if the library is
missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_uint32, c_void_p

import torch  # noqa: F401  -- must be imported first so libamdhip64.so.7 resolves to torch's copy

_HERE = os.path.dirname(os.path.abspath(__file__))
# CCST_HIP_LIB points at an alternative build of the same ABI (kernel A/B experiments, tools/build_variant.sh)
LIB_PATH = os.environ.get("CCST_HIP_LIB") or os.path.join(_HERE, "csrc", "libccst_hip.so")

CONV_RELU, CONV_POOL2, CONV_UPS2, CONV_REFLECT, CONV_ACCUM, CONV_FLIP = 1, 2, 4, 8, 16, 32


class CcstConvDesc(Structure):
    _fields_ = [("n", c_int32), ("ho", c_int32), ("wo", c_int32), ("hi", c_int32), ("wi", c_int32),
                ("cin", c_int32), ("cout", c_int32), ("cout_pad", c_int32), ("nky", c_int32), ("nkx", c_int32),
                ("ay", c_int32), ("by", c_int32), ("cy", c_int32), ("ax", c_int32), ("bx", c_int32), ("cx", c_int32),
                ("tap_base", c_int32), ("tap_sy", c_int32), ("tap_sx", c_int32),
                ("xsN", c_int64), ("xsH", c_int32), ("xsW", c_int32),
                ("y_off", c_int64), ("ysN", c_int64), ("ysH", c_int32), ("ysW", c_int32), ("ysC", c_int32),
                ("flags", c_uint32)]


class CcstImageXform(Structure):
    _fields_ = [("src_off", c_int64), ("src_w", c_int32), ("crop_i", c_int32), ("crop_j", c_int32), ("crop_h", c_int32),
                ("crop_w", c_int32), ("flip", c_int32), ("kx", c_int32), ("ky", c_int32),
                ("bounds_x", c_int32), ("coefs_x", c_int32), ("bounds_y", c_int32), ("coefs_y", c_int32)]


_P = c_void_p
# name -> argtypes (restype is int unless listed in _RESTYPES).  Every symbol of include/ccst_hip.h.
_SIGNATURES = {
    "ccst_abi_version": [],
    "ccst_last_error": [],
    "ccst_conv2d_igemm_f32": [POINTER(CcstConvDesc), _P, _P, _P, _P, _P],
    "ccst_conv2d_igemm_stats_f32": [POINTER(CcstConvDesc), _P, _P, _P, _P, _P, _P],
    "ccst_conv2d_igemm_stats_groups": [c_int, c_int, c_int, c_int],
    "ccst_conv2d_stream_ok": [POINTER(CcstConvDesc)],
    "ccst_conv2d_pointwise_half_f32": [POINTER(CcstConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "ccst_pack_conv_weight_split_f32": [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "ccst_conv2d_igemm_half_f32": [POINTER(CcstConvDesc), _P, _P, _P, _P, _P, _P],
    "ccst_pack_conv_weights_split_batch_f32": [_P, c_int, _P],
    "ccst_conv2d_pointwise_ok": [POINTER(CcstConvDesc)],
    "ccst_conv2d_igemm_accum_masked_f32": [POINTER(CcstConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "ccst_conv2d_igemm_bn_relu_bwd_f32": [POINTER(CcstConvDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "ccst_bn_relu_maxpool_train_fwd_f32": [_P, _P, _P, _P, _P, c_float, c_float, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P,
                                           c_int64, _P, _P],
    "ccst_bn_relu_maxpool_train_bwd_f32": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int64, _P, _P],
    "ccst_bn_train_bwd_partials_f32": [_P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, c_int, c_int64, c_int, _P, c_int64, _P, _P],
    "ccst_conv3x3_halo_f32": [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_uint32, _P],
    "ccst_absmax_f32": [_P, c_int64, _P, _P],
    "ccst_absmax_batch_f32": [_P, c_int, _P, _P],
    "ccst_absmax_samples_f32": [_P, c_int, c_int64, _P, _P],
    "ccst_conv3x3_halo_split_f32": [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_uint32, _P, _P],
    "ccst_pack_conv_weight_f43_f32": [_P, _P, c_int, c_int, c_int, _P, _P],
    "ccst_conv3x3_f43_f32": [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_uint32, _P, _P, _P, _P],
    "ccst_adain_fold_affine_f32": [_P, c_int, _P, _P, c_int, c_float, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, _P, _P, _P],
    "ccst_conv3x3_f43_workgroups": [c_int, c_int, c_int, c_int],
    "ccst_conv3x3_f43_tiles": [c_int, c_int, c_int],
    "ccst_conv3x3_halo_split_tiles": [c_int, c_int, c_int],
    "ccst_pack_conv_weight_halo_split_f32": [_P, _P, c_int, c_int, c_int, _P, c_int, _P],
    "ccst_pack_conv_weights_halo_split_batch_f32": [_P, c_int, _P],
    "ccst_conv3x3_halo_train_split_f32": [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_uint32, _P, _P, _P, _P, _P, _P, _P],
    "ccst_conv3x3_halo_narrow": [c_int, c_int, c_int, c_int],
    "ccst_wino_weight_floats": [c_int, c_int],
    "ccst_pack_conv_weight_wino_f32": [_P, _P, c_int, c_int, c_int, _P],
    "ccst_pack_stem3_weight_f32": [_P, _P, _P, c_int, _P],
    "ccst_conv3x3_stem3_f32": [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P],
    "ccst_chan_sums_finalize_f32": [_P, c_int, c_int, c_int, _P, _P, _P],
    "ccst_pack_conv_weight_wino_bwd_f32": [_P, _P, c_int, c_int, c_int, _P],
    "ccst_conv3x3_wino_train_f32": [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_uint32, _P],
    "ccst_conv3x3_wino_stats_groups": [c_int, c_int, c_int],
    "ccst_pack_conv_weights_wino_batch_f32": [_P, c_int, _P],
    "ccst_conv3x3_halo_train_f32": [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_uint32, _P],
    "ccst_conv3x3_halo_stats_groups": [c_int, c_int, c_int],
    "ccst_conv2d_igemm_tile": [c_int, c_int, c_int, c_int, c_int],
    "ccst_pack_conv_weights_batch_f32": [_P, c_int, _P],
    "ccst_conv3x3_zform_weight_floats": [c_int],
    "ccst_pack_conv_weight_zform_f32": [_P, _P, _P, c_int, c_int, _P],
    "ccst_conv3x3_zform_f32": [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "ccst_pack_conv_weight_f32": [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "ccst_nchw_to_nhwc4_pad_f32": [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "ccst_nhwc_layer_f32": [c_int, _P, _P, c_int, c_int, c_int, c_int, c_int, _P],
    "ccst_nchw_to_nhwc_f32": [_P, _P, c_int, c_int, c_int, c_int, _P],
    "ccst_nhwc_to_nchw_f32": [_P, _P, c_int, c_int, c_int, c_int, _P],
    "ccst_quantize_u8_hwc_f32": [_P, _P, c_int, c_int, c_int, _P],
    "ccst_resize_bilinear_nchw_f32": [_P, _P, c_int, c_int, c_int, c_int, c_int, _P],
    "ccst_image_plan": [c_int, _P, c_int, c_int, _P, c_int64],
    "ccst_crop_resize_norm_u8_f32": [_P, _P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P],
    "ccst_conv2d_bwd_weight_f32": [POINTER(CcstConvDesc), _P, _P, _P, c_int, c_int, _P, c_int64, _P],
    "ccst_conv2d_bwd_weight_splits": [c_int, c_int, c_int, c_int],
    "ccst_conv2d_bwd_weight_split_f32": [POINTER(CcstConvDesc), _P, _P, _P, _P, _P, c_int, c_int, _P, c_int64, _P],
    "ccst_conv2d_bwd_weight_split_splits": [c_int, c_int, c_int, c_int],
    "ccst_calc_mean_std_f32": [_P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, c_int64, _P],
    "ccst_adain_tile_sums_f32": [_P, _P, c_int, c_int, _P, _P, c_int, c_float, _P, c_int, c_int, c_int, c_float, _P, _P, _P, _P],
    "ccst_interp_blend_f32": [_P, _P, _P, c_int, c_int64, c_float, c_float, _P, _P],
    "ccst_adain_f32": [_P, _P, _P, c_int, c_float, _P, c_int, c_int, c_int, c_int, c_float, _P, c_int64, _P, _P],
    "ccst_chan_sums_f32": [_P, _P, _P, c_int, c_int, c_int, c_int, _P, c_int64, _P],
    "ccst_stats_workspace_bytes": [c_int, c_int, c_int],
    "ccst_bn_eval_fwd_f32": [_P, _P, _P, _P, _P, c_float, _P, c_int, _P, c_int64, c_int, _P, _P],
    "ccst_bn_train_fwd_mask_f32": [_P, _P, _P, _P, _P, c_float, c_float, _P, c_int, _P, _P, _P, _P, c_int64, c_int, _P, c_int, _P, c_int64, _P, _P],
    "ccst_bn_train_bwd_mask_f32": [_P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, c_int, c_int64, c_int, _P, c_int64, _P, _P],
    "ccst_bn_workspace_bytes": [c_int64, c_int],
    "ccst_maxpool3s2_fwd_f32": [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "ccst_maxpool3s2_bwd_f32": [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "ccst_avgpool_fwd_f32": [_P, _P, c_int, c_int, c_int, _P],
    "ccst_avgpool_bwd_f32": [_P, _P, c_int, c_int, c_int, _P],
    "ccst_linear_fwd_f32": [_P, _P, _P, _P, c_int, c_int, c_int, _P],
    "ccst_linear_bwd_f32": [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P],
    "ccst_softmax_ce_f32": [_P, _P, _P, _P, _P, c_int, c_int, _P],
    "ccst_sgd_f32": [_P, _P, c_float, c_int64, _P],
    "ccst_scale_f32": [_P, c_float, c_int64, _P],
    "ccst_fedavg_f32": [_P, _P, _P, c_int, c_int64, _P],
    "ccst_fill_f32": [_P, c_float, c_int64, _P],
    "ccst_add_i64": [_P, c_int64, c_int, _P],
    "ccst_mul_scalar_f32": [_P, _P, _P, c_int64, _P],
    "ccst_stem_grad_unfold_f32": [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P],
}
_RESTYPES = {"ccst_last_error": c_char_p, "ccst_stats_workspace_bytes": c_int64, "ccst_bn_workspace_bytes": c_int64,
             "ccst_wino_weight_floats": c_int64, "ccst_image_plan": c_int64,
             "ccst_conv3x3_zform_weight_floats": c_int64}
EXPORTS = tuple(_SIGNATURES)

_lib = None


def load():
    """Load the library once; raise loudly if it is absent (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "ccst_amd: %s not found. The HIP extension is required (there is no CPU/PyTorch fallback); "
            "build it with `python -m ccst_amd.build`." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the .so lacks a declared symbol
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    if lib.ccst_abi_version() != 2:
        raise RuntimeError("ccst_amd: ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().ccst_last_error()
        raise RuntimeError("ccst_amd: %s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def raw_stream(index=None):
    """hipStream_t (as an int) of torch's current stream on device `index` (default: the current device).  The
    private C entry points cost ~0.3 us; torch.cuda.current_stream() builds a Stream object (~8 us), which at
    ~600 launches per ResNet train step was 2 ms of host time."""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device() if index is None else index)
    return torch.cuda.current_stream(index).cuda_stream


def stream_ptr():
    return c_void_p(raw_stream())


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)
