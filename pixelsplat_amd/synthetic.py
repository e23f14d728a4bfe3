"""Synthetic re10k-shaped inputs for the hot path (no dataset / checkpoint is available
offline).  Follows the recipe of SURVEY.md section 8d; every formula restates the
reference code that would have produced the tensor in a real run:

  cameras / bounds     src/dataset/shims/bounds_shim.py:9-37, encoder_epipolar.py:215-230
  depth from disparity src/model/encoder/epipolar/conversions.py:5-14
  world rays           src/geometry/projection.py:74-114
  scales / covariance  src/model/encoder/common/gaussian_adapter.py:62-80,97-108
                       src/model/encoder/common/gaussians.py:8-41
  SH mask              src/model/encoder/common/gaussian_adapter.py:40-46
  opacity              src/model/encoder/encoder_epipolar.py:170
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
from torch import Tensor


@dataclass
class Cameras:
    extrinsics: Tensor  # [b, v, 4, 4] camera-to-world
    intrinsics: Tensor  # [b, v, 3, 3] normalised
    near: Tensor        # [b, v]
    far: Tensor         # [b, v]


@dataclass
class SceneGaussians:
    """Same fields/layout as the reference `Gaussians` (src/model/types.py:7-12)."""
    means: Tensor        # [b, G, 3]
    covariances: Tensor  # [b, G, 3, 3]
    harmonics: Tensor    # [b, G, 3, d_sh]
    opacities: Tensor    # [b, G]


def _yaw(angle: Tensor) -> Tensor:
    c, s = angle.cos(), angle.sin()
    r = torch.zeros(angle.shape + (4, 4), dtype=torch.float32)
    r[..., 0, 0] = c
    r[..., 0, 2] = s
    r[..., 1, 1] = 1
    r[..., 2, 0] = -s
    r[..., 2, 2] = c
    r[..., 3, 3] = 1
    return r


def make_cameras(b: int, v_ctx: int, v_tgt: int, hw: tuple[int, int], gen: torch.Generator,
                 fx: float = 0.89, max_yaw_deg: float = 5.0) -> tuple[Cameras, Cameras]:
    """Context cameras on the x axis over a unit baseline, targets in between."""
    h, w = hw

    def poses(xs: Tensor) -> Tensor:
        yaw = (torch.rand(xs.shape, generator=gen) * 2 - 1) * math.radians(max_yaw_deg)
        m = _yaw(yaw)
        m[..., 0, 3] = xs
        return m

    ctx_x = torch.linspace(0, 1, v_ctx).expand(b, v_ctx).contiguous()
    tgt_x = torch.rand((b, v_tgt), generator=gen)
    k = torch.eye(3, dtype=torch.float32)
    k[0, 0] = fx
    k[1, 1] = fx
    k[0, 2] = 0.5
    k[1, 2] = 0.5
    mean_pixel = 0.5 * (1 / (w * fx) + 1 / (h * fx))
    baseline = 1.0
    near = baseline / (3.0 * min(h, w) * mean_pixel)
    far = baseline / (0.5 * mean_pixel)

    def cams(xs, v):
        return Cameras(poses(xs), k.expand(b, v, 3, 3).contiguous(),
                       torch.full((b, v), near), torch.full((b, v), far))

    return cams(ctx_x, v_ctx), cams(tgt_x, v_tgt)


def _quat_to_matrix(q: Tensor, eps: float = 1e-8) -> Tensor:
    i, j, k, r = q.unbind(-1)
    two_s = 2 / ((q * q).sum(-1) + eps)
    o = torch.stack((
        1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
        two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
        two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


# Scene distributions ("a trained model keeps most in frame ... Measure, don't assume").
# Every variant draws the SAME random numbers in the same order as "survey" and only maps them
# differently, so the cameras, pixel jitter and colours of a seed are shared across scenes.
#   survey  the SURVEY 8d recipe: depth uniform in disparity over [near, far] (most Gaussians close to the
#           context cameras: ~40 % of (target view, Gaussian) pairs in frame), opacity U(0, 1/3)
#   dense   depth restricted to the far part of the disparity range, where the parallax between the
#           context and the in-between target cameras is small: >= 80 % of the pairs in frame (D/G >= 2)
#   opaque  opacity U(0.5, 1) (encoder_epipolar.py:170 after training): pixels saturate and stop early
#   large   scale multiplier x 3 (gaussian_adapter.py:62-69 at the top of its range): most Gaussians
#           cover more than 4 tiles -> the tile backward's atomic path
SCENES = ("survey", "dense", "opaque", "large")
_DENSE_U_MIN = 0.85


def make_gaussians(ctx: Cameras, hw: tuple[int, int], gen: torch.Generator,
                   per_pixel: int = 3, sh_degree: int = 4, scene: str = "survey") -> SceneGaussians:
    """G = v_ctx * h * w * per_pixel Gaussians per scene, footprint statistics as an
    untrained pixelSplat encoder would emit them (`scene`: see SCENES)."""
    if scene not in SCENES:
        raise ValueError(f"scene must be one of {SCENES}")
    h, w = hw
    b, v = ctx.near.shape
    n = h * w * per_pixel
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    xy = torch.stack(((xs + 0.5) / w, (ys + 0.5) / h), -1).reshape(1, 1, h * w, 1, 2)
    jitter = (torch.rand((b, v, h * w, per_pixel, 2), generator=gen) - 0.5) / torch.tensor(
        [w, h], dtype=torch.float32)
    xy = (xy + jitter).reshape(b, v, n, 2)

    u = torch.rand((b, v, n), generator=gen)
    if scene == "dense":
        u = _DENSE_U_MIN + (1 - _DENSE_U_MIN) * u
    near, far = ctx.near[..., None], ctx.far[..., None]
    eps = 1e-10
    disp_near, disp_far = 1 / (near + eps), 1 / (far + eps)
    depth = 1 / ((1 - u) * (disp_near - disp_far) + disp_far + eps)

    k_inv = torch.linalg.inv(ctx.intrinsics)[:, :, None]            # [b,v,1,3,3]
    hom = torch.cat((xy, torch.ones_like(xy[..., :1])), -1)
    d_cam = torch.einsum("bvnij,bvnj->bvni", k_inv.expand(b, v, n, 3, 3), hom)
    d_cam = d_cam / d_cam.norm(dim=-1, keepdim=True)
    rot = ctx.extrinsics[:, :, None, :3, :3]
    d_world = torch.einsum("bvnij,bvnj->bvni", rot.expand(b, v, n, 3, 3), d_cam)
    origin = ctx.extrinsics[:, :, None, :3, 3]
    means = origin + d_world * depth[..., None]

    fx = ctx.intrinsics[..., 0, 0][..., None]
    fy = ctx.intrinsics[..., 1, 1][..., None]
    mult = (0.3 if scene == "large" else 0.1) * (1 / (w * fx) + 1 / (h * fy))
    raw = torch.randn((b, v, n, 3), generator=gen)
    scales = (0.5 + 14.5 * raw.sigmoid()) * depth[..., None] * mult[..., None]
    q = torch.randn((b, v, n, 4), generator=gen)
    q = q / (q.norm(dim=-1, keepdim=True) + 1e-8)
    r = _quat_to_matrix(q)
    s = torch.diag_embed(scales)
    cov = r @ s @ s.transpose(-1, -2) @ r.transpose(-1, -2)
    cov = rot @ cov @ rot.transpose(-1, -2)

    d_sh = (sh_degree + 1) ** 2
    mask = torch.ones(d_sh)
    for deg in range(1, sh_degree + 1):
        mask[deg ** 2:(deg + 1) ** 2] = 0.1 * 0.25 ** deg
    sh = torch.randn((b, v, n, 3, d_sh), generator=gen) * mask
    opacity = torch.rand((b, v, n), generator=gen)
    opacity = 0.5 + 0.5 * opacity if scene == "opaque" else opacity / per_pixel

    return SceneGaussians(
        means.reshape(b, v * n, 3).contiguous(),
        cov.reshape(b, v * n, 3, 3).contiguous(),
        sh.reshape(b, v * n, 3, d_sh).contiguous(),
        opacity.reshape(b, v * n).contiguous(),
    )


def make_workload(b: int, hw: tuple[int, int], v_ctx: int = 2, v_tgt: int = 4, seed: int = 0,
                  per_pixel: int = 3, sh_degree: int = 4, scene: str = "survey"):
    """(context cameras, target cameras, scene Gaussians, target images) on CPU, fp32."""
    gen = torch.Generator().manual_seed(seed)
    ctx, tgt = make_cameras(b, v_ctx, v_tgt, hw, gen)
    gaussians = make_gaussians(ctx, hw, gen, per_pixel, sh_degree, scene)
    target = torch.rand((b, v_tgt, 3, hw[0], hw[1]), generator=gen)
    return ctx, tgt, gaussians, target
