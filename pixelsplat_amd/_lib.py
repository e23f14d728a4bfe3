"""ctypes binding of libpixelsplat_hip.so (the C ABI of include/pixelsplat_hip.h).

The product path has NO fallback: if the shared library is missing or a symbol is absent
this raises -- there is no eager/PyTorch or CPU path behind these ops.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# PIXELSPLAT_HIP_LIB: another build of the same library (A/B runs of kernel variants)
LIB_PATH = os.environ.get("PIXELSPLAT_HIP_LIB") or os.path.join(HERE, "libpixelsplat_hip.so")

PS_SH_GK3, PS_SH_G3K = 0, 1
PS_COV_6, PS_COV_33 = 0, 1
PS_FLAG_BWD_TEMP_ZEROED = 1
PS_FLAG_DEFER_SH_COLORS = 2
PS_FLAG_DETERMINISTIC = 4
PS_VIEW_STRIDE = 48
PS_VIEW_VIEWMATRIX, PS_VIEW_PROJMATRIX, PS_VIEW_CAMPOS = 0, 16, 32
PS_VIEW_TANFOVX, PS_VIEW_TANFOVY, PS_VIEW_BG, PS_VIEW_SCALE = 35, 36, 37, 40


class PsRasterDesc(C.Structure):
    _fields_ = [
        ("n_scenes", C.c_int32), ("views_per_scene", C.c_int32), ("n_gaussians", C.c_int32),
        ("height", C.c_int32), ("width", C.c_int32), ("sh_degree", C.c_int32),
        ("sh_coeffs", C.c_int32), ("sh_layout", C.c_int32), ("cov_layout", C.c_int32),
        ("flags", C.c_int32),
        ("near_cull", C.c_float), ("guard", C.c_float), ("lowpass", C.c_float),
        ("w_eps", C.c_float), ("lambda_floor", C.c_float), ("alpha_max", C.c_float),
        ("alpha_min", C.c_float), ("t_min", C.c_float), ("det2_eps", C.c_float),
    ]


class PsRasterStateLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in (
        "records", "rects", "sorted_idx", "sorted_rect", "n_vis", "final_T", "n_contrib",
        "tile_end", "tile_ranges", "num_rendered", "tile_order", "clamp_bits", "checkpoint",
        "cell_windows", "total")]


class PsEpipolarDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("b", "v", "h", "w", "s", "c", "heads", "octaves",
                                         "ld_q", "ld_u", "ld_e", "ld_f", "ld_p", "ld_a",
                                         "hs_in", "hs_out", "tail_pad_in", "tail_pad_out")]


class PsDepthSamplerDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_views", "rays_per_view", "buckets", "surfaces", "spp",
                                         "deterministic", "use_transmittance")] + [
        ("opacity_exponent", C.c_float), ("opacity_scale", C.c_float)]


class PsFoldDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("heads", "head_dim", "kv_dim", "q_dim", "out_dim",
                                         "octaves", "other_views")]


class PsDepthLossDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_images", "height", "width", "channels",
                                         "use_second_derivative", "use_sigma")] + [
        ("sigma_image", C.c_float), ("weight", C.c_float)]


# every symbol include/pixelsplat_hip.h declares
EXPORTS = [
    "ps_raster_default_desc", "ps_raster_state_bytes", "ps_raster_temp_bytes",
    "ps_raster_backward_temp_bytes",
    "ps_raster_state_layout", "ps_raster_forward", "ps_raster_forward_plan",
    "ps_raster_forward_render", "ps_raster_forward_colors", "ps_raster_forward_bins", "ps_raster_forward_tiles", "ps_raster_backward", "ps_raster_backward_prepare",
    "ps_raster_check", "ps_camera_setup", "ps_epipolar_geometry", "ps_epipolar_gather",
    "ps_epipolar_attention_forward", "ps_epipolar_attention_backward", "ps_status_string", "ps_build_info", "ps_roctx_available",
    "ps_gemm_tn_workspace_bytes", "ps_gemm_tn_f32", "ps_gemm_tn_colsum_f32", "ps_invert_cameras", "ps_epipolar_feature_grad", "ps_epipolar_token_grad_floats", "ps_epipolar_ray_box_words", "ps_epipolar_feature_grad_two_pass", "ps_epipolar_feature_bins", "ps_epipolar_feature_grad_binned", "ps_gaussian_adapter_views",
    "ps_gaussian_adapter_forward", "ps_gaussian_adapter_backward",
    "ps_gaussian_head_forward", "ps_gaussian_head_backward",
    "ps_depth_sampler_forward", "ps_depth_sampler_backward",
    "ps_layer_norm_workspace_floats", "ps_layer_norm_forward", "ps_layer_norm_backward",
    "ps_fold_scratch_floats", "ps_fold_attention_weights", "ps_fold_attention_weights_backward",
    "ps_image_mse_workspace_bytes", "ps_image_mse", "ps_depth_smoothness_workspace_bytes",
    "ps_depth_smoothness_forward", "ps_depth_smoothness_backward",
    "ps_profile_enable", "ps_profile_group_count", "ps_profile_group_name", "ps_profile_collect",
    "ps_abi_version",
]
PS_ABI_VERSION = 7      # include/pixelsplat_hip.h

_lib = None


class HipExtensionMissing(RuntimeError):
    pass


def load():
    """Loads the library (cached).  Raises HipExtensionMissing -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipExtensionMissing(
            f"{LIB_PATH} not found: build it with `python -m pixelsplat_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no non-HIP fallback for this path.")
    # torch ships its own libamdhip64; it has to be in the process BEFORE this library is
    # dlopen-ed so that both resolve to the same HIP runtime (otherwise kernels registered with
    # one runtime are launched on streams of the other: "HIP launch failed")
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise HipExtensionMissing(f"{LIB_PATH} does not export {name}")
    lib.ps_abi_version.restype = C.c_int
    if lib.ps_abi_version() != PS_ABI_VERSION:
        raise HipExtensionMissing(
            f"{LIB_PATH} has descriptor layout version {lib.ps_abi_version()}, this package expects "
            f"{PS_ABI_VERSION}: rebuild it with `python -m pixelsplat_amd.build`")
    vp = C.c_void_p
    lib.ps_raster_default_desc.argtypes = [C.POINTER(PsRasterDesc)]
    lib.ps_raster_default_desc.restype = None
    lib.ps_raster_state_bytes.argtypes = [C.POINTER(PsRasterDesc)]
    lib.ps_raster_state_bytes.restype = C.c_size_t
    lib.ps_raster_temp_bytes.argtypes = [C.POINTER(PsRasterDesc)]
    lib.ps_raster_temp_bytes.restype = C.c_size_t
    lib.ps_raster_backward_temp_bytes.argtypes = [C.POINTER(PsRasterDesc), C.c_size_t]
    lib.ps_raster_backward_temp_bytes.restype = C.c_size_t
    lib.ps_raster_state_layout.argtypes = [C.POINTER(PsRasterDesc), C.POINTER(PsRasterStateLayout)]
    lib.ps_raster_state_layout.restype = C.c_int
    lib.ps_raster_forward.argtypes = [C.POINTER(PsRasterDesc)] + [vp] * 8 + [
        vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t, vp]
    lib.ps_raster_forward.restype = C.c_int
    lib.ps_raster_forward_plan.argtypes = [C.POINTER(PsRasterDesc)] + [vp] * 7 + [
        vp, C.c_size_t, vp, C.c_size_t, vp]
    lib.ps_raster_forward_plan.restype = C.c_int
    lib.ps_raster_forward_render.argtypes = [C.POINTER(PsRasterDesc), vp, vp, vp, C.c_size_t, vp,
                                             C.c_size_t, vp, C.c_size_t, vp]
    lib.ps_raster_forward_render.restype = C.c_int
    lib.ps_raster_backward.argtypes = [C.POINTER(PsRasterDesc)] + [vp] * 8 + [
        vp, C.c_size_t, vp, C.c_size_t, vp, C.c_size_t] + [vp] * 6 + [vp]
    lib.ps_raster_backward.restype = C.c_int
    lib.ps_raster_forward_colors.argtypes = [C.POINTER(PsRasterDesc), vp, vp, vp, vp, vp, C.c_size_t,
                                             vp, C.c_size_t, vp]
    lib.ps_raster_forward_colors.restype = C.c_int
    lib.ps_raster_forward_bins.argtypes = [C.POINTER(PsRasterDesc), vp, C.c_size_t, vp, C.c_size_t,
                                           vp, C.c_size_t, vp]
    lib.ps_raster_forward_bins.restype = C.c_int
    lib.ps_raster_forward_tiles.argtypes = [C.POINTER(PsRasterDesc), vp, vp, vp, C.c_size_t, vp,
                                            C.c_size_t, vp, C.c_size_t, vp]
    lib.ps_raster_forward_tiles.restype = C.c_int
    lib.ps_raster_backward_prepare.argtypes = [C.POINTER(PsRasterDesc), vp, C.c_size_t, C.c_size_t, vp]
    lib.ps_raster_backward_prepare.restype = C.c_int
    lib.ps_camera_setup.argtypes = [C.c_int32, vp, vp, vp, vp, vp, C.c_int32, vp, vp]
    lib.ps_camera_setup.restype = C.c_int
    lib.ps_epipolar_geometry.argtypes = [C.c_int32] * 5 + [vp] * 13 + [vp]
    lib.ps_epipolar_geometry.restype = C.c_int
    pe = C.POINTER(PsEpipolarDesc)
    lib.ps_epipolar_gather.argtypes = [pe, vp, vp, vp, vp, vp]
    lib.ps_epipolar_gather.restype = C.c_int
    lib.ps_epipolar_attention_forward.argtypes = [pe] + [vp] * 7 + [C.c_float] + [vp] * 5
    lib.ps_epipolar_attention_forward.restype = C.c_int
    lib.ps_epipolar_attention_backward.argtypes = [pe] + [vp] * 12 + [C.c_float] + [vp] * 7
    lib.ps_epipolar_attention_backward.restype = C.c_int
    lib.ps_epipolar_feature_grad.argtypes = [pe, C.c_int32] + [vp] * 9
    lib.ps_epipolar_feature_grad.restype = C.c_int
    lib.ps_epipolar_token_grad_floats.argtypes = [pe]
    lib.ps_epipolar_token_grad_floats.restype = C.c_size_t
    lib.ps_epipolar_ray_box_words.argtypes = [pe]
    lib.ps_epipolar_ray_box_words.restype = C.c_size_t
    lib.ps_epipolar_feature_grad_two_pass.argtypes = [pe, C.c_int32] + [vp] * 10
    lib.ps_epipolar_feature_grad_two_pass.restype = C.c_int
    lib.ps_epipolar_feature_bins.argtypes = [pe, vp, vp, vp, vp]
    lib.ps_epipolar_feature_bins.restype = C.c_int
    lib.ps_epipolar_feature_grad_binned.argtypes = [pe, C.c_int32] + [vp] * 10
    lib.ps_epipolar_feature_grad_binned.restype = C.c_int
    lib.ps_gaussian_adapter_views.argtypes = [C.c_int32] * 4 + [vp] * 5
    lib.ps_gaussian_adapter_views.restype = C.c_int
    lib.ps_gaussian_adapter_forward.argtypes = [C.c_int32] * 4 + [C.c_float] * 3 + [vp] * 8
    lib.ps_gaussian_adapter_forward.restype = C.c_int
    lib.ps_gaussian_adapter_backward.argtypes = [C.c_int32] * 4 + [C.c_float] * 3 + [vp] * 11
    lib.ps_gaussian_adapter_backward.restype = C.c_int
    lib.ps_gaussian_head_forward.argtypes = [C.c_int32] * 6 + [C.c_float] * 3 + [vp] * 7
    lib.ps_gaussian_head_forward.restype = C.c_int
    lib.ps_gaussian_head_backward.argtypes = [C.c_int32] * 6 + [C.c_float] * 3 + [vp] * 9
    lib.ps_gaussian_head_backward.restype = C.c_int
    pd = C.POINTER(PsDepthSamplerDesc)
    lib.ps_depth_sampler_forward.argtypes = [pd] + [vp] * 8
    lib.ps_depth_sampler_forward.restype = C.c_int
    lib.ps_depth_sampler_backward.argtypes = [pd] + [vp] * 8
    lib.ps_depth_sampler_backward.restype = C.c_int
    lib.ps_layer_norm_workspace_floats.argtypes = [C.c_int32, C.c_int32]
    lib.ps_layer_norm_workspace_floats.restype = C.c_size_t
    lib.ps_layer_norm_forward.argtypes = [C.c_int32, C.c_int32, C.c_float] + [vp] * 7
    lib.ps_layer_norm_forward.restype = C.c_int
    lib.ps_layer_norm_backward.argtypes = [C.c_int32, C.c_int32] + [vp] * 11
    lib.ps_layer_norm_backward.restype = C.c_int
    pf = C.POINTER(PsFoldDesc)
    lib.ps_fold_scratch_floats.argtypes = [pf]
    lib.ps_fold_scratch_floats.restype = C.c_size_t
    lib.ps_fold_attention_weights.argtypes = [pf] + [vp] * 12
    lib.ps_fold_attention_weights.restype = C.c_int
    lib.ps_fold_attention_weights_backward.argtypes = [pf] + [vp] * 20
    lib.ps_fold_attention_weights_backward.restype = C.c_int
    lib.ps_image_mse_workspace_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.ps_image_mse_workspace_bytes.restype = C.c_size_t
    lib.ps_image_mse.argtypes = [C.c_int32, C.c_int32, vp, vp, C.c_float, vp, vp, vp, vp,
                                 C.c_size_t, vp]
    lib.ps_image_mse.restype = C.c_int
    pl = C.POINTER(PsDepthLossDesc)
    lib.ps_depth_smoothness_workspace_bytes.argtypes = [pl]
    lib.ps_depth_smoothness_workspace_bytes.restype = C.c_size_t
    lib.ps_depth_smoothness_forward.argtypes = [pl, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.ps_depth_smoothness_forward.restype = C.c_int
    lib.ps_depth_smoothness_backward.argtypes = [pl] + [vp] * 7
    lib.ps_depth_smoothness_backward.restype = C.c_int
    lib.ps_invert_cameras.argtypes = [C.c_int32, vp, vp, vp, vp, vp]
    lib.ps_invert_cameras.restype = C.c_int
    lib.ps_gemm_tn_workspace_bytes.argtypes = [C.c_int32] * 3
    lib.ps_gemm_tn_workspace_bytes.restype = C.c_size_t
    lib.ps_gemm_tn_f32.argtypes = [C.c_int32] * 3 + [vp, C.c_int32, vp, C.c_int32, vp, vp, C.c_size_t, vp]
    lib.ps_gemm_tn_f32.restype = C.c_int
    lib.ps_gemm_tn_colsum_f32.argtypes = [C.c_int32] * 3 + [vp, C.c_int32, vp, C.c_int32, vp, vp, vp, C.c_size_t, vp]
    lib.ps_gemm_tn_colsum_f32.restype = C.c_int
    lib.ps_raster_check.argtypes = [C.POINTER(PsRasterDesc), vp, C.c_size_t,
                                    C.POINTER(C.c_uint64), vp]
    lib.ps_raster_check.restype = C.c_int
    lib.ps_profile_enable.argtypes = [C.c_int]
    lib.ps_profile_group_count.restype = C.c_int
    lib.ps_profile_group_name.argtypes = [C.c_int]
    lib.ps_profile_group_name.restype = C.c_char_p
    lib.ps_profile_collect.argtypes = [vp, vp]
    lib.ps_status_string.argtypes = [C.c_int]
    lib.ps_status_string.restype = C.c_char_p
    lib.ps_build_info.argtypes = []
    lib.ps_build_info.restype = C.c_char_p
    _lib = lib
    return lib


def check(status: int, what: str) -> None:
    if status != 0:
        msg = load().ps_status_string(status).decode()
        raise RuntimeError(f"{what} failed: {msg} (PsStatus {status})")


def default_desc() -> PsRasterDesc:
    d = PsRasterDesc()
    load().ps_raster_default_desc(C.byref(d))
    return d
