"""Host-side mirror of the reference decoder API on top of the batched HIP rasterizer.

Same names, argument meaning and return shapes as

  DecoderSplattingCUDA.forward / .render_depth
      /root/reference/src/model/decoder/decoder_splatting_cuda.py:35-91
  render_cuda / render_cuda_orthographic / render_depth_cuda / get_projection_matrix
      /root/reference/src/model/decoder/cuda_splatting.py:17-269
  Decoder / DecoderOutput
      /root/reference/src/model/decoder/decoder.py:19-48

but (i) one library call renders all b*v views, (ii) the Gaussians are NOT repeated per
view (`views_per_scene` tells the kernel which scene a view reads), (iii) nothing is pulled
to the host (no `.item()`), (iv) the SH tensor is consumed in the reference's own
[G,3,d_sh] layout and the covariance as [G,3,3] (no transpose / triu copies), (v) the
scale-invariant renorm (cuda_splatting.py:64-71) is folded into the kernels.
"""
from __future__ import annotations

from dataclasses import dataclass
from math import isqrt
from typing import Literal

import torch
from torch import Tensor, nn

from ._lib import PS_COV_33, PS_SH_G3K
from .geometry import get_fov, get_projection_matrix  # noqa: F401  (re-exported name)
from .raster import RasterConfig, forward_with_state, pack_view_params, rasterize  # noqa: F401

DepthRenderingMode = Literal["depth", "log", "disparity", "relative_disparity"]


@dataclass
class DecoderOutput:
    color: Tensor          # [batch, view, 3, height, width]
    depth: Tensor | None   # [batch, view, height, width]


def camera_setup(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor,
                 background_color: Tensor, scale_invariant: bool = True) -> Tensor:
    """[V,4,4] c2w, [V,3,3], [V], [V], [V,3] -> packed view parameters [V,48] in ONE HIP
    launch (ps_camera_setup): the renorm, field of view, projection, transposed view /
    full-projection matrices and camera position of cuda_splatting.py:64-87,110."""
    import ctypes as C

    from . import _lib
    from .raster import _p, _stream

    lib = _lib.load()
    v = extrinsics.shape[0]
    e = extrinsics.contiguous().float()
    k = intrinsics.contiguous().float()
    n, f = near.contiguous().float(), far.contiguous().float()
    bg = background_color.contiguous().float()
    if not e.is_cuda:
        raise RuntimeError("pixelsplat_amd needs GPU tensors (no CPU fallback)")
    out = torch.empty((v, 48), dtype=torch.float32, device=e.device)
    _lib.check(lib.ps_camera_setup(C.c_int32(v), _p(e), _p(k), _p(n), _p(f), _p(bg),
                                   C.c_int32(int(scale_invariant)), _p(out), _stream()),
               "ps_camera_setup")
    return out


def _view_params_torch(extrinsics, near, far, fov_x, fov_y, tan_fov, background_color, scale):
    """PyTorch composition of the same block (only the orthographic visualisation path uses
    it: its field of view is prescribed, not derived from intrinsics)."""
    projection = get_projection_matrix(near, far, fov_x, fov_y).transpose(1, 2)
    view = torch.linalg.inv(extrinsics).transpose(1, 2)
    full = view @ projection
    return pack_view_params(view.contiguous(), full.contiguous(), extrinsics[:, :3, 3], tan_fov,
                            background_color, scale)


def _render(vp, image_shape, gaussian_means, gaussian_covariances, gaussian_sh_coefficients,
            gaussian_opacities, use_sh, views_per_scene, return_aux, list_capacity=0, deterministic=None):
    assert use_sh or gaussian_sh_coefficients.shape[-1] == 1
    v_total = vp.shape[0]
    s, g, _ = gaussian_means.shape
    assert s * views_per_scene == v_total, "views must be grouped by scene"
    h, w = image_shape
    n = gaussian_sh_coefficients.shape[-1]
    degree = isqrt(n) - 1
    cfg = RasterConfig(n_scenes=s, views_per_scene=views_per_scene, n_gaussians=g, height=h,
                       width=w, sh_degree=degree if use_sh else 0, sh_coeffs=n if use_sh else 0,
                       sh_layout=PS_SH_G3K, cov_layout=PS_COV_33, list_capacity=int(list_capacity),
                       deterministic=deterministic)
    if use_sh:
        sh, colors = gaussian_sh_coefficients, None
    else:
        sh, colors = None, gaussian_sh_coefficients[..., 0]  # [V, G, 3] per-view colours
        assert colors.shape[0] == v_total
    if return_aux:
        image, radii = rasterize(cfg, gaussian_means, gaussian_covariances, gaussian_opacities,
                                 vp, sh=sh, colors=colors)
        # second (no-grad) pass only to expose the saved state to tests
        with torch.no_grad():
            res, lay = forward_with_state(
                cfg, gaussian_means.detach(), gaussian_covariances.detach(),
                gaussian_opacities.detach(), vp, sh=None if sh is None else sh.detach(),
                colors=None if colors is None else colors.detach())
        return image, dict(cfg=cfg, radii=radii, state=res.state, layout=lay, view_params=vp,
                           point_list=res.point_list, num_rendered=res.num_rendered)
    image, _ = rasterize(cfg, gaussian_means, gaussian_covariances, gaussian_opacities, vp,
                         sh=sh, colors=colors)
    return image


def render_cuda(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor,
                image_shape: tuple[int, int], background_color: Tensor, gaussian_means: Tensor,
                gaussian_covariances: Tensor, gaussian_sh_coefficients: Tensor,
                gaussian_opacities: Tensor, scale_invariant: bool = True, use_sh: bool = True,
                views_per_scene: int = 1, return_aux: bool = False,
                view_params: Tensor | None = None, list_capacity: int = 0,
                deterministic: bool | None = None):
    """[B,4,4] c2w, [B,3,3], [B], [B], (h,w), [B,3], Gaussians [B/vps, G, ...] -> [B,3,h,w].

    With views_per_scene == 1 the call is argument-for-argument the reference's
    render_cuda (Gaussians given once per view).  `view_params` ([B,48], the packed block of
    `camera_setup` / `pack_view_params`) replaces the camera arguments when given: callers that
    render the same cameras repeatedly skip the set-up launch, and the parity tests feed the
    settings recorded from the reference's own host code.  `list_capacity` > 0 fixes the size of
    the tile point list up front (entries, all views together): no host synchronisation in the
    call, which is what hipGraph capture of a training step needs; an overflow raises from
    backward() (or from `pixelsplat_amd.raster.captured_overflow_flags()` after a graph replay).
    `deterministic=True`: the backward sums the gradients of Gaussians over more than four tiles in a
    fixed order instead of through float atomics (PS_FLAG_DETERMINISTIC; default: the
    PIXELSPLAT_DETERMINISTIC environment variable, off)."""
    vp = view_params if view_params is not None else camera_setup(
        extrinsics, intrinsics, near, far, background_color, scale_invariant)
    return _render(vp, image_shape, gaussian_means, gaussian_covariances,
                   gaussian_sh_coefficients, gaussian_opacities, use_sh, views_per_scene,
                   return_aux, list_capacity, deterministic)


def render_cuda_orthographic(extrinsics: Tensor, width: Tensor, height: Tensor, near: Tensor,
                             far: Tensor, image_shape: tuple[int, int], background_color: Tensor,
                             gaussian_means: Tensor, gaussian_covariances: Tensor,
                             gaussian_sh_coefficients: Tensor, gaussian_opacities: Tensor,
                             fov_degrees: float = 0.1, use_sh: bool = True,
                             dump: dict | None = None, views_per_scene: int = 1) -> Tensor:
    """Fake orthographic projection: camera moved far back with a tiny field of view
    (cuda_splatting.py:130-220)."""
    b = extrinsics.shape[0]
    fov_x = torch.tensor(fov_degrees, device=extrinsics.device).deg2rad()
    tan_fov_x = (0.5 * fov_x).tan()
    distance_to_near = (0.5 * width) / tan_fov_x
    tan_fov_y = 0.5 * height / distance_to_near
    fov_y = (2 * tan_fov_y).atan()
    near = near + distance_to_near
    far = far + distance_to_near
    move_back = torch.eye(4, dtype=torch.float32, device=extrinsics.device).repeat(b, 1, 1)
    move_back[:, 2, 3] = -distance_to_near
    extrinsics = extrinsics @ move_back
    if dump is not None:
        dump["extrinsics"] = extrinsics
        dump["fov_x"] = fov_x
        dump["fov_y"] = fov_y
        dump["near"] = near
        dump["far"] = far
    fov_xb = fov_x.expand(b)
    tan_fov = torch.stack((tan_fov_x.expand(b), tan_fov_y.expand(b)), dim=-1)
    vp = _view_params_torch(extrinsics, near, far, fov_xb, fov_y.expand(b), tan_fov,
                            background_color, torch.ones_like(near))
    return _render(vp, image_shape, gaussian_means, gaussian_covariances,
                   gaussian_sh_coefficients, gaussian_opacities, use_sh, views_per_scene, False)


def render_depth_cuda(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor,
                      image_shape: tuple[int, int], gaussian_means: Tensor,
                      gaussian_covariances: Tensor, gaussian_opacities: Tensor,
                      scale_invariant: bool = True, mode: DepthRenderingMode = "depth",
                      views_per_scene: int = 1) -> Tensor:
    """Depth-as-colour render (cuda_splatting.py:226-269) -> [B,h,w]."""
    b = extrinsics.shape[0]
    w2c = torch.linalg.inv(extrinsics)
    means_v = gaussian_means.repeat_interleave(views_per_scene, dim=0) \
        if views_per_scene > 1 else gaussian_means
    fake_color = (torch.einsum("bj,bgj->bg", w2c[:, 2, :3], means_v) + w2c[:, 2, 3:4])
    if mode == "disparity":
        fake_color = 1 / fake_color
    elif mode == "relative_disparity":
        eps = 1e-10
        n_, f_ = near[:, None], far[:, None]
        disp_near, disp_far = 1 / (n_ + eps), 1 / (f_ + eps)
        fake_color = 1 - (1 / (fake_color + eps) - disp_far) / (disp_near - disp_far + eps)
    elif mode == "log":
        fake_color = fake_color.minimum(near[:, None]).maximum(far[:, None]).log()
    result = render_cuda(
        extrinsics, intrinsics, near, far, image_shape,
        torch.zeros((b, 3), dtype=fake_color.dtype, device=fake_color.device), gaussian_means,
        gaussian_covariances, fake_color[:, :, None, None].expand(-1, -1, 3, 1),
        gaussian_opacities, scale_invariant=scale_invariant, use_sh=False,
        views_per_scene=views_per_scene)
    return result.mean(dim=1)


@dataclass
class DecoderSplattingCUDACfg:
    name: Literal["splatting_cuda"]


class Decoder(nn.Module):
    def __init__(self, cfg, dataset_cfg) -> None:
        super().__init__()
        self.cfg = cfg
        self.dataset_cfg = dataset_cfg


class DecoderSplattingCUDA(Decoder):
    """Drop-in for the reference class of the same name (registry key "splatting_cuda",
    /root/reference/src/model/decoder/__init__.py:5-13).  No parameters, one non-persistent
    buffer `background_color` -- released checkpoints load unchanged."""

    background_color: Tensor

    def __init__(self, cfg, dataset_cfg) -> None:
        super().__init__(cfg, dataset_cfg)
        self.register_buffer(
            "background_color",
            torch.tensor(dataset_cfg.background_color, dtype=torch.float32),
            persistent=False,
        )

    def forward(self, gaussians, extrinsics: Tensor, intrinsics: Tensor, near: Tensor,
                far: Tensor, image_shape: tuple[int, int],
                depth_mode: DepthRenderingMode | None = None) -> DecoderOutput:
        b, v, _, _ = extrinsics.shape
        color = render_cuda(
            extrinsics.reshape(b * v, 4, 4), intrinsics.reshape(b * v, 3, 3),
            near.reshape(b * v), far.reshape(b * v), image_shape,
            self.background_color.expand(b * v, 3), gaussians.means, gaussians.covariances,
            gaussians.harmonics, gaussians.opacities, views_per_scene=v)
        color = color.reshape(b, v, *color.shape[1:])
        return DecoderOutput(
            color,
            None if depth_mode is None else self.render_depth(
                gaussians, extrinsics, intrinsics, near, far, image_shape, depth_mode),
        )

    def render_depth(self, gaussians, extrinsics: Tensor, intrinsics: Tensor, near: Tensor,
                     far: Tensor, image_shape: tuple[int, int],
                     mode: DepthRenderingMode = "depth") -> Tensor:
        b, v, _, _ = extrinsics.shape
        result = render_depth_cuda(
            extrinsics.reshape(b * v, 4, 4), intrinsics.reshape(b * v, 3, 3),
            near.reshape(b * v), far.reshape(b * v), image_shape, gaussians.means,
            gaussians.covariances, gaussians.opacities, mode=mode, views_per_scene=v)
        return result.reshape(b, v, *result.shape[1:])
