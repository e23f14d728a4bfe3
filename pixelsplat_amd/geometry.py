"""Host-side camera conventions of the rasterizer path (tiny, runs on whatever device the
inputs live on).  Restated from the reference so the C-ABI receives exactly the matrices
the reference would hand to `diff_gaussian_rasterization`:

  get_fov                /root/reference/src/geometry/projection.py:233-247
  get_projection_matrix  /root/reference/src/model/decoder/cuda_splatting.py:17-44
  view / full projection /root/reference/src/model/decoder/cuda_splatting.py:84-87
"""
from __future__ import annotations

import torch
from torch import Tensor


def get_fov(intrinsics: Tensor) -> Tensor:
    """[B,3,3] normalised intrinsics -> [B,2] (fov_x, fov_y) in radians."""
    inv = torch.linalg.inv(intrinsics)

    def ray(v):
        v = torch.tensor(v, dtype=torch.float32, device=intrinsics.device)
        d = torch.einsum("bij,j->bi", inv, v)
        return d / d.norm(dim=-1, keepdim=True)

    left, right = ray([0.0, 0.5, 1.0]), ray([1.0, 0.5, 1.0])
    top, bottom = ray([0.5, 0.0, 1.0]), ray([0.5, 1.0, 1.0])
    fov_x = (left * right).sum(dim=-1).acos()
    fov_y = (top * bottom).sum(dim=-1).acos()
    return torch.stack((fov_x, fov_y), dim=-1)


def get_projection_matrix(near: Tensor, far: Tensor, fov_x: Tensor, fov_y: Tensor) -> Tensor:
    """x,y -> (-1,1), z -> (0,1), w = z.  [B,4,4] (column-vector convention)."""
    tan_x = (0.5 * fov_x).tan()
    tan_y = (0.5 * fov_y).tan()
    top = tan_y * near
    bottom = -top
    right = tan_x * near
    left = -right
    (b,) = near.shape
    m = torch.zeros((b, 4, 4), dtype=torch.float32, device=near.device)
    m[:, 0, 0] = 2 * near / (right - left)
    m[:, 1, 1] = 2 * near / (top - bottom)
    m[:, 0, 2] = (right + left) / (right - left)
    m[:, 1, 2] = (top + bottom) / (top - bottom)
    m[:, 3, 2] = 1
    m[:, 2, 2] = far / (far - near)
    m[:, 2, 3] = -(far * near) / (far - near)
    return m


def camera_matrices(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor):
    """Per-view (tan_fov [B,2], view^T [B,4,4], (P @ view)^T [B,4,4], campos [B,3]).

    `extrinsics` are camera-to-world.  Matrices come back TRANSPOSED (row-vector
    convention) exactly as cuda_splatting.py:84-87 builds them, contiguous.
    """
    fov = get_fov(intrinsics)
    fov_x, fov_y = fov.unbind(dim=-1)
    tan_fov = torch.stack(((0.5 * fov_x).tan(), (0.5 * fov_y).tan()), dim=-1)
    proj_t = get_projection_matrix(near, far, fov_x, fov_y).transpose(1, 2)
    view_t = torch.linalg.inv(extrinsics).transpose(1, 2)
    full_t = view_t @ proj_t
    campos = extrinsics[:, :3, 3]
    return tan_fov.contiguous(), view_t.contiguous(), full_t.contiguous(), campos.contiguous()
