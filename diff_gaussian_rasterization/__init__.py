"""Drop-in replacement for the third-party module `diff_gaussian_rasterization`
(git+https://github.com/dcharatan/diff-gaussian-rasterization-modified, the CUDA rasterizer
pixelSplat imports at /root/reference/src/model/decoder/cuda_splatting.py:5-8) backed by
the hand-written gfx950 kernels in libpixelsplat_hip.so.

Put the repository root on PYTHONPATH and the reference's unmodified `cuda_splatting.py`
(render_cuda :99-124, render_cuda_orthographic :192-217, render_depth_cuda :226-269) runs
on MI355X.  Same two public names, same keyword signature, same return values
(color [3,H,W] float32, radii [G] int32), gradients to means3D / means2D (screen-space, NDC
units) / shs|colors_precomp / opacities / cov3D_precomp.  SH degree 0..4.

There is no CPU or PyTorch fallback: without the HIP library (or without a GPU tensor) the
call raises.
"""
from __future__ import annotations

from typing import NamedTuple

import torch
from torch import Tensor, nn

from pixelsplat_amd._lib import PS_COV_6, PS_SH_GK3
from pixelsplat_amd.raster import RasterConfig, pack_view_params, rasterize

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: Tensor
    scale_modifier: float
    viewmatrix: Tensor
    projmatrix: Tensor
    sh_degree: int
    campos: Tensor
    prefiltered: bool
    debug: bool


def _cov3d_from_scale_rotation(scales: Tensor, rotations: Tensor, modifier: float) -> Tensor:
    """[G,3], [G,4] (r,x,y,z) -> [G,6]; Sigma = R S S^T R^T, plain torch (autograd carries the
    gradient; the reference never takes this branch, it passes cov3D_precomp)."""
    r, x, y, z = rotations.unbind(-1)
    rot = torch.stack((
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)), -1).reshape(-1, 3, 3)
    m = rot * (modifier * scales)[:, None, :]
    sigma = m @ m.transpose(1, 2)
    row, col = torch.triu_indices(3, 3)
    return sigma[:, row, col]


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: Tensor) -> Tensor:
        """Frustum test of the upstream module (view-space z > 0.2)."""
        rs = self.raster_settings
        hom = torch.cat((positions, torch.ones_like(positions[:, :1])), -1)
        return (hom @ rs.viewmatrix)[:, 2] > 0.2

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None,
                rotations=None, cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (
                shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception(
                "Please provide exactly one of either scale/rotation pair or precomputed 3D "
                "covariance!")
        if cov3D_precomp is None:
            cov3D_precomp = _cov3d_from_scale_rotation(scales, rotations, rs.scale_modifier)

        g = means3D.shape[0]
        dev = means3D.device
        # floats in render_cuda (:102-103, .item()), 0-d / [1] tensors in render_cuda_orthographic
        # (:196-197)
        tanfov = torch.stack(
            (torch.as_tensor(rs.tanfovx, dtype=torch.float32, device=dev).reshape(-1)[0],
             torch.as_tensor(rs.tanfovy, dtype=torch.float32, device=dev).reshape(-1)[0]))
        vp = pack_view_params(rs.viewmatrix.reshape(1, 4, 4).contiguous(),
                              rs.projmatrix.reshape(1, 4, 4).contiguous(),
                              rs.campos.reshape(1, 3), tanfov.reshape(1, 2),
                              rs.bg.reshape(1, 3))
        k = 0 if shs is None else shs.shape[1]
        cfg = RasterConfig(n_scenes=1, views_per_scene=1, n_gaussians=g,
                           height=int(rs.image_height), width=int(rs.image_width),
                           sh_degree=int(rs.sh_degree) if shs is not None else 0, sh_coeffs=k,
                           sh_layout=PS_SH_GK3, cov_layout=PS_COV_6)
        color, radii = rasterize(
            cfg, means3D.reshape(1, g, 3), cov3D_precomp.reshape(1, g, 6),
            opacities.reshape(1, g), vp,
            sh=None if shs is None else shs.reshape(1, g, k, 3),
            colors=None if colors_precomp is None else colors_precomp.reshape(1, g, 3),
            means2d=None if means2D is None else means2D.reshape(1, g, 3))
        return color[0], radii[0]
