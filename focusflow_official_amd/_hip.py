"""ctypes binding of libfocusflow_hip.so (the C ABI in include/focusflow_hip.h).

There is NO fallback: if the library is missing, or a call is made without a
HIP device, this raises.  PyTorch is only used by callers for device memory and
streams; every pointer handed over here is ``tensor.data_ptr()``.
"""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "lib", "libfocusflow_hip.so")
# lab builds (tools/build_lab.sh: in-kernel stamps, timing-only ablations) live in their own libraries and are loaded
# only when named explicitly; bench.py refuses to run with the variable set
if os.environ.get("FF_LAB_LIB"):
    LIB_PATH = os.path.join(_PKG, "lib", os.environ["FF_LAB_LIB"])

ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, ACT_LEAKY = 0, 1, 2, 3, 4
W_F32, W_F16X3, W_F16 = 0, 1, 2
MAX_SEG = 3
_fp = C.c_void_p
_ll = C.c_longlong


class FFFusionPair(C.Structure):
    """include/focusflow_hip.h: FFFusionPair (ff_fusion_pair_fwd)."""
    _fields_ = [
        ("x", _fp * 2), ("x_ld", C.c_int * 2), ("xres", _fp * 2), ("xres_ld", C.c_int * 2), ("scale", _fp * 2), ("shift", _fp * 2),
        ("in_act", C.c_int), ("w_frag", _fp * 2), ("bias", _fp * 2), ("w_format", C.c_int), ("y", _fp * 2), ("y_ld", C.c_int * 2),
        ("B", C.c_int), ("HW", C.c_int), ("C", C.c_int),
    ]


class FFConvParams(C.Structure):
    _fields_ = [
        ("x", _fp * MAX_SEG), ("x_ld", C.c_int * MAX_SEG), ("x_c", C.c_int * MAX_SEG),
        ("x_gstride", _ll * MAX_SEG), ("groups", C.c_int), ("B", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("w", _fp), ("w_gstride", _ll), ("bias", _fp), ("ch_scale", _fp), ("ch_shift", _fp),
        ("out_scale", C.c_float), ("res", _fp), ("res_ld", C.c_int), ("y", _fp), ("y_ld", C.c_int),
        ("y_gstride", _ll), ("Ho", C.c_int), ("Wo", C.c_int), ("Cout", C.c_int),
        ("KH", C.c_int), ("KW", C.c_int), ("stride", C.c_int), ("pad_h", C.c_int), ("pad_w", C.c_int),
        ("act", C.c_int), ("act_res", C.c_int), ("w_format", C.c_int),
        ("dil_h", C.c_int), ("dil_w", C.c_int), ("x_amax", _fp),
        ("in_scale", _fp), ("in_shift", _fp), ("in_act", C.c_int),
        ("res2", _fp), ("res2_ld", C.c_int), ("res_split", C.c_int),
        ("splitk_ws", _fp), ("splitk", C.c_int),
        ("ep_mode", C.c_int), ("ep_split", C.c_int), ("ep_a", _fp), ("ep_a_ld", C.c_int), ("ep_b", _fp), ("ep_b_ld", C.c_int),
        ("stats_part", _fp),
        ("x_fmt", C.c_int * MAX_SEG), ("y_fmt", C.c_int), ("y_fmt_from", C.c_int), ("y2", _fp), ("y2_ld", C.c_int),
        ("w_frag", _fp),
    ]


PACK_MAX_MEMBERS = 4    # include/focusflow_hip.h: FF_PACK_MAX_MEMBERS / FF_PACK_MAX_SLICES
PACK_MAX_SLICES = 4


class FFPackJob(C.Structure):
    _fields_ = [
        ("w", _fp * PACK_MAX_MEMBERS), ("bias", _fp * PACK_MAX_MEMBERS),
        ("cout_m", C.c_int * PACK_MAX_MEMBERS), ("off", C.c_int * PACK_MAX_MEMBERS), ("nmem", C.c_int), ("cout", C.c_int),
        ("cin_src", C.c_int), ("nslice", C.c_int), ("slice_lo", C.c_int * PACK_MAX_SLICES), ("slice_hi", C.c_int * PACK_MAX_SLICES),
        ("cin", C.c_int), ("cin_pad", C.c_int), ("KH", C.c_int), ("KW", C.c_int),
        ("fwd", _fp), ("bias_dst", _fp), ("dgrad", _fp),
        ("fwd_format", C.c_int), ("dgrad_format", C.c_int), ("cout_pad", C.c_int), ("reserved", C.c_int),
        ("items_fwd", _ll), ("items_dgrad", _ll), ("block0", _ll),
    ]


# name -> argtypes; every function returns int (0 = ok) except the two below
_SIGS = {
    "ff_pack_job_check": [C.POINTER(FFPackJob)],
    "ff_pack_weights_table": [_fp, C.c_int, _ll, _fp],
    "ff_unpack_wgrad_group": [_fp, _fp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int,
                              C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, _fp, _fp],
    "ff_conv2d_fwd": [C.POINTER(FFConvParams), _fp],
    "ff_pack_conv_weight": [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_int, _fp],
    "ff_pack_split_f16": [_fp, _fp, _ll, C.c_int, _fp],
    "ff_pack_frag16": [_fp, _fp, C.c_int, C.c_int, _fp],
    "ff_norm_stats": [_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp],
    "ff_norm_stats_finish": [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp],
    "ff_mask_upsample_pack": [_fp, _fp, _fp],
    "ff_mask_upsample_fwd": [_fp, C.c_int, _fp, C.c_int, _fp, C.c_float, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, _fp],
    "ff_probe_memory_kernel": [_fp, _ll, _fp, _ll, C.c_int, C.c_int, C.c_int, C.c_uint, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), _fp],
    "ff_launch_timing_begin": [C.c_int],
    "ff_launch_timing_end": [C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "ff_norm_apply": [_fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_float,
                      _fp, _fp, C.c_int, _fp, C.c_int, _fp],
    "ff_norm_coeffs": [_fp, C.c_int, C.c_int, _ll, C.c_float, _fp, _fp, _fp, _fp, _fp],
    "ff_bn_fold": [_fp, _fp, _fp, _fp, C.c_float, _fp, _fp, C.c_int, _fp],
    "ff_bn_update_running": [_fp, _ll, C.c_float, _fp, _fp, C.c_int, _fp],
    "ff_prep_input": [_fp, C.c_int, C.c_float, _fp, C.c_int, C.c_int, C.c_int, _fp],
    "ff_corr_pyramid": [_fp, _fp, _fp, _fp, _ll, C.c_int, C.c_int, _fp],
    "ff_corr_lookup_fwd": [C.POINTER(_fp), C.c_int, C.c_int, _fp, _ll, C.c_int, C.c_int, _fp, C.c_int, _fp, _fp],
    "ff_corr_build": [_fp, _fp, C.POINTER(_fp), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp],
    "ff_corr_retile": [_fp, _fp, _ll, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp],
    "ff_corr_tile_rows": [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp],
    "ff_corr_lookup_tiled_fwd": [C.POINTER(_fp), C.c_int, _fp, _ll, C.c_int, C.c_int, _fp, C.c_int, _fp, _fp],
    "ff_corr_lookup_tiled_bwd_all": [_fp, C.POINTER(_fp), C.POINTER(_fp), C.c_int, C.c_int, _ll, C.c_int, C.c_int, _fp],
    "ff_corr_lookup_tiled_bwd": [C.POINTER(_fp), _fp, _fp, C.c_int, _ll, C.c_int, C.c_int, _fp],
    "ff_corr_pyramid_tiled_bwd": [_fp, _fp, _fp, _fp, _ll, C.c_int, C.c_int, _fp],
    "ff_act_copy": [_fp, C.c_int, _fp, C.c_int, _ll, C.c_int, C.c_int, _fp],
    "ff_range_probe": [_fp, C.c_int, _ll, C.c_int, _fp, _fp],
    "ff_split_copy": [_fp, C.c_int, _fp, C.c_int, _ll, C.c_int, C.c_int, C.c_int, _fp],
    "ff_coords_init": [_fp, _fp, C.c_int, C.c_int, C.c_int, _fp],
    "ff_coords_step": [_fp, _fp, C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp],
    "ff_gru_pass": [C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, _fp, _fp, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int,
                    C.c_int, C.c_int, C.c_int, _fp],
    "ff_gru_pass_rec": [C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, _fp, _fp, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int,
                        _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp],
    "ff_fusion_pair_fwd": [C.POINTER(FFFusionPair), _fp],
    "ff_gru_rh": [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _ll, C.c_int, _fp],
    "ff_gru_blend": [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _ll, C.c_int, _fp],
    "ff_upsample_flow": [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, _fp],
    "ff_nhwc_to_nchw": [_fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp],
    # backward
    "ff_conv2d_wgrad": [C.POINTER(FFConvParams), _fp, _ll, _fp, _fp],
    "ff_unpack_conv_wgrad": [_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp],
    "ff_pack_conv_weight_dgrad": [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_int, _fp],
    "ff_act_bwd": [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _ll, C.c_int, C.c_int, C.c_int, C.c_float, _fp, _fp],
    "ff_deconv4x4s2_small": [_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_int, _fp, C.c_int, _fp],
    "ff_dilate2": [_fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp],
    "ff_norm_bwd": [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_float, _fp, _fp,
                    C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp],
    "ff_gru_rh_bwd": [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _ll, C.c_int, _fp],
    "ff_gru_blend_bwd": [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int,
                         _fp, C.c_int, _ll, C.c_int, _fp],
    "ff_upsample_flow_bwd": [_fp, _fp, C.c_int, _fp, C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp],
    "ff_upsample_flow_bwd_ex": [_fp, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_float, _fp, C.c_int, C.c_int, C.c_int, _fp],
    "ff_gru_bwd_blend": [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int,
                         _fp, _fp, _ll, C.c_int, _fp],
    "ff_gru_bwd_rh": [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int, _fp, _ll, C.c_int, _fp],
    "ff_gru_bwd_out": [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int, _fp, _ll, C.c_int, _fp],
    "ff_sum_stack": [_fp, C.c_int, _ll, _fp, _fp],
    # FF-PWC native component
    "ff_pwc_costvolume_fwd": [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp],
    "ff_pwc_costvolume_fwd_ex": [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp],
    "ff_pwc_costvolume_bwd": [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp],
    "ff_pwc_gout_transpose": [_fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp],
    "ff_pwc_backwarp": [_fp, C.c_int, _fp, C.c_int, C.c_float, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp],
    # fused sequence loss
    "ff_loss_prepare": [_fp, _fp, _fp, _fp, C.c_int, C.c_float, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp],
    "ff_loss_accumulate": [_fp, _fp, _fp, _fp, _fp, C.c_float, C.c_float, C.c_float, _fp, _fp, C.c_int, C.c_int,
                           C.c_int, _fp],
    "ff_epe_metric": [_fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp],
    "ff_nchw_to_nhwc4": [_fp, C.c_int, C.c_float, _fp, C.c_int, C.c_int, C.c_int, _fp],
    "ff_resize_bilinear": [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _fp],
    "ff_pwc_backwarp_bwd": [_fp, C.c_int, _fp, C.c_int, C.c_float, _fp, C.c_int, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int,
                            C.c_int, _fp],
    "ff_pwc_loss_mask": [_fp, _fp, C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp],
    "ff_pwc_loss_scale": [_fp, _fp, _fp, _fp, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _fp, _fp, C.c_int,
                          C.c_int, C.c_int, C.c_int, C.c_int, _fp],
    "ff_pwc_epe_mean": [_fp, _fp, C.c_int, C.c_float, C.c_float, _fp, C.c_int, C.c_int, C.c_int, _fp],
    "ff_pwc_loss_scale_sparse": [_fp, _fp, _fp, _fp, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _fp, _fp, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_int, _fp],
    "ff_pwc_epe_mean_sparse": [_fp, _fp, C.c_int, C.c_float, C.c_float, _fp, C.c_int, C.c_int, C.c_int, _fp],
    "ff_resize_to_nhwc4": [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_int, C.c_int, _fp],
    "ff_png_unfilter": [_fp, _ll, C.c_int, C.c_int, C.c_int, _fp],
    "ff_mask_prepare": [C.c_int, _fp, _fp, _fp, C.c_int, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp],
    "ff_chan_stats_fwd": [_fp, C.c_int, C.c_int, _ll, _fp, C.c_int, _fp, _fp],
    "ff_chan_stats_bwd": [_fp, C.c_int, _fp, C.c_int, _ll, _fp, C.c_int, _fp],
    "ff_spatial_stats_fwd": [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp],
    "ff_spatial_stats_bwd": [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp],
    "ff_scale_add_fwd": [_fp, C.c_int, _fp, C.c_int, _fp, _fp, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp],
    "ff_scale_add_bwd": [_fp, C.c_int, _fp, C.c_int, _fp, C.c_int, _fp, _fp, C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_int,
                         C.c_int, _fp],
}
EXPORTS = sorted(list(_SIGS) + ["ff_last_error", "ff_abi_version", "ff_corr_plane_elems", "ff_conv2d_splitk_hint", "ff_conv2d_stats_parts", "ff_fusion_pair_tile"])

ABI_VERSION = 7      # include/focusflow_hip.h: FF_ABI_VERSION
_lib = None


class FocusFlowHipError(RuntimeError):
    pass


# Timing-only ablations (WRONG results by design) live in the separate lab build (tools/): the product refuses to load
# while one of their switches is set, so that a stray variable in a shell cannot silently corrupt flow.
_ABLATION_VARS = ("FF_LOOKUP_ABLATE", "FF_LOOKUP_ABLATE3", "FF_CORR_BUILD_ABLATE", "FF_PATCH_ABLATE")
# tile-shape overrides that the product library still reads (the tests walk the tile variants through them): results stay right,
# measurements change - bench.py refuses these too
_TUNING_VARS = ("FF_DMA_TILE", "FF_GRU_PASS_TH", "FF_LAB_LIB")      # (every other tuning override exists in the lab build only: csrc/ff_common.h tune_env)


def lab_variables_set():
    """Names of lab switches present in the environment (ablations first)."""
    return [v for v in _ABLATION_VARS + _TUNING_VARS if os.environ.get(v) is not None]


def load():
    """Load the library once.  Raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    abl = [v for v in _ABLATION_VARS if os.environ.get(v) is not None]
    if abl and not os.environ.get("FF_LAB_LIB"):
        raise FocusFlowHipError(f"{', '.join(abl)} set in the environment: timing-only ablations return wrong results and are not part of "
                                "libfocusflow_hip.so; unset them (the lab build under tools/ is where they live)")
    if not os.path.exists(LIB_PATH):
        raise FocusFlowHipError(
            f"{LIB_PATH} is missing: build it with `python -m focusflow_official_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU/PyTorch fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.ff_last_error.restype = C.c_char_p
    lib.ff_last_error.argtypes = []
    lib.ff_abi_version.restype = C.c_int
    lib.ff_abi_version.argtypes = []
    lib.ff_corr_plane_elems.restype = C.c_int
    lib.ff_corr_plane_elems.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    lib.ff_conv2d_splitk_hint.restype = C.c_int
    lib.ff_conv2d_splitk_hint.argtypes = [C.POINTER(FFConvParams)]
    lib.ff_fusion_pair_tile.restype = C.c_int
    lib.ff_fusion_pair_tile.argtypes = [C.c_int]
    lib.ff_conv2d_stats_parts.restype = C.c_int
    lib.ff_conv2d_stats_parts.argtypes = [C.POINTER(FFConvParams)]
    got = lib.ff_abi_version()
    if got != ABI_VERSION:
        raise FocusFlowHipError(f"{LIB_PATH} speaks ABI version {got}, these bindings (FFConvParams layout, argument lists) "
                                f"version {ABI_VERSION}: rebuild with `python -m focusflow_official_amd.build`")
    _lib = lib
    return lib


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise FocusFlowHipError(f"{name} failed ({rc}): {lib.ff_last_error().decode()}")
