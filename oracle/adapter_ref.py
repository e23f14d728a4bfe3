"""ORACLE (test infrastructure, NOT product code): CPU restatement of the Gaussian adapter,
SURVEY.md section 8(f) rank 2 -- the producer of the rasterizer's inputs:

    /root/reference/src/model/encoder/common/gaussian_adapter.py:48-95   GaussianAdapter.forward
    /root/reference/src/model/encoder/common/gaussians.py:8-41           quaternion -> covariance
    /root/reference/src/geometry/projection.py:65-108                    unproject, get_world_rays
    /root/reference/src/misc/sh_rotation.py:10-31                        rotate_sh

PINNING.  Everything except the two e3nn calls is pinned against the real reference modules
run in the build container (tests/golden/adapter.npz, tests/golden/make_adapter_golden.py).
`rotate_sh` calls `e3nn.o3.matrix_to_angles` and `e3nn.o3.wigner_D`; e3nn is a pip dependency
(`requirements.txt:21`, UNPINNED) that is not installed here and whose source is not under
/root/reference, so those two functions are restated below from e3nn's published algorithm
(e3nn 0.5.x, `e3nn/o3/_rotation.py`, `e3nn/o3/_wigner.py`) and the golden run uses this
restatement in e3nn's place: for the SH rotation the parity is UNPINNED.  What can be checked
without e3nn is (tests/test_oracle_adapter.py): D^1 equals the rotation matrix in e3nn's
(x, y, z) order, every D^l is orthogonal and a homomorphism (D(R1 R2) = D(R1) D(R2)), and
matrix_to_angles inverts angles_to_matrix.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
from torch import Tensor


# ---------------------------------------------------------------------------------------
# e3nn restatement (e3nn/o3/_rotation.py, e3nn/o3/_wigner.py, version 0.5.x)
# ---------------------------------------------------------------------------------------
def matrix_y(angle: Tensor) -> Tensor:
    c, s, o, z = angle.cos(), angle.sin(), torch.ones_like(angle), torch.zeros_like(angle)
    return torch.stack([torch.stack([c, z, s], -1), torch.stack([z, o, z], -1),
                        torch.stack([-s, z, c], -1)], -2)


def matrix_x(angle: Tensor) -> Tensor:
    c, s, o, z = angle.cos(), angle.sin(), torch.ones_like(angle), torch.zeros_like(angle)
    return torch.stack([torch.stack([o, z, z], -1), torch.stack([z, c, -s], -1),
                        torch.stack([z, s, c], -1)], -2)


def angles_to_matrix(alpha: Tensor, beta: Tensor, gamma: Tensor) -> Tensor:
    """e3nn: R = Y(alpha) X(beta) Y(gamma)."""
    return matrix_y(alpha) @ matrix_x(beta) @ matrix_y(gamma)


def xyz_to_angles(xyz: Tensor):
    """e3nn: xyz = Y(alpha) X(beta) (0, 1, 0)."""
    xyz = torch.nn.functional.normalize(xyz, p=2, dim=-1).clamp(-1, 1)
    beta = torch.acos(xyz[..., 1])
    alpha = torch.atan2(xyz[..., 0], xyz[..., 2])
    return alpha, beta


def matrix_to_angles(r: Tensor):
    """e3nn.o3.matrix_to_angles (called at sh_rotation.py:20)."""
    x = r @ r.new_tensor([0.0, 1.0, 0.0])
    a, b = xyz_to_angles(x)
    r = angles_to_matrix(a, b, torch.zeros_like(a)).transpose(-1, -2) @ r
    c = torch.atan2(r[..., 0, 2], r[..., 0, 0])
    return a, b, c


def su2_generators(j: int) -> Tensor:
    m = torch.arange(-j, j, dtype=torch.float64)
    raising = torch.diag(-torch.sqrt(j * (j + 1) - m * (m + 1)), diagonal=-1)
    m = torch.arange(-j + 1, j + 1, dtype=torch.float64)
    lowering = torch.diag(torch.sqrt(j * (j + 1) - m * (m - 1)), diagonal=1)
    m = torch.arange(-j, j + 1, dtype=torch.float64)
    return torch.stack([
        0.5 * (raising + lowering).to(torch.complex128),      # x (usually)
        torch.diag(1j * m),                                     # z (usually)
        -0.5j * (raising - lowering).to(torch.complex128),     # -y (usually)
    ], dim=0)


def change_basis_real_to_complex(l: int) -> Tensor:
    # https://en.wikipedia.org/wiki/Spherical_harmonics#Real_form
    q = torch.zeros((2 * l + 1, 2 * l + 1), dtype=torch.complex128)
    for m in range(-l, 0):
        q[l + m, l + abs(m)] = 1 / math.sqrt(2)
        q[l + m, l - abs(m)] = -1j / math.sqrt(2)
    q[l, l] = 1
    for m in range(1, l + 1):
        q[l + m, l + abs(m)] = (-1) ** m / math.sqrt(2)
        q[l + m, l - abs(m)] = 1j * (-1) ** m / math.sqrt(2)
    return (-1j) ** l * q   # factor that makes the Clebsch-Gordan coefficients real


def so3_generators(l: int) -> Tensor:
    x = su2_generators(l)
    q = change_basis_real_to_complex(l)
    x = torch.conj(q.T) @ x @ q
    assert x.imag.abs().max() < 1e-12
    return x.real


def wigner_D(l: int, alpha: Tensor, beta: Tensor, gamma: Tensor) -> Tensor:
    """e3nn.o3.wigner_D (called at sh_rotation.py:24): the (2l+1)x(2l+1) representation of
    Y(alpha) X(beta) Y(gamma) on e3nn's real spherical harmonics."""
    alpha, beta, gamma = torch.broadcast_tensors(alpha, beta, gamma)
    x = so3_generators(l).to(alpha.dtype)
    a = alpha[..., None, None] % (2 * math.pi)
    b = beta[..., None, None] % (2 * math.pi)
    c = gamma[..., None, None] % (2 * math.pi)
    return torch.matrix_exp(a * x[1]) @ torch.matrix_exp(b * x[0]) @ torch.matrix_exp(c * x[1])


def rotate_sh(sh: Tensor, rotations: Tensor) -> Tensor:
    """sh_rotation.py:10-31."""
    n = sh.shape[-1]
    alpha, beta, gamma = matrix_to_angles(rotations)
    out = []
    for degree in range(math.isqrt(n)):
        d = wigner_D(degree, alpha, beta, gamma).to(sh.dtype)
        out.append(torch.einsum("...ij,...j->...i", d, sh[..., degree ** 2:(degree + 1) ** 2]))
    return torch.cat(out, dim=-1)


# ---------------------------------------------------------------------------------------
# adapter
# ---------------------------------------------------------------------------------------
def quaternion_to_matrix(q: Tensor, eps: float = 1e-8) -> Tensor:
    """gaussians.py:8-30 (xyzw order)."""
    i, j, k, r = torch.unbind(q, dim=-1)
    two_s = 2 / ((q * q).sum(dim=-1) + eps)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)),
                    -1)
    return o.reshape(*q.shape[:-1], 3, 3)


def build_covariance(scale: Tensor, rotation_xyzw: Tensor) -> Tensor:
    """gaussians.py:33-41."""
    s = scale.diag_embed()
    r = quaternion_to_matrix(rotation_xyzw)
    return r @ s @ s.transpose(-1, -2) @ r.transpose(-1, -2)


def world_rays(coordinates: Tensor, extrinsics: Tensor, intrinsics: Tensor):
    """projection.py:65-108."""
    hom = torch.cat([coordinates, torch.ones_like(coordinates[..., :1])], -1)
    d = torch.einsum("...ij,...j->...i", intrinsics.inverse(), hom)
    d = d / d.norm(dim=-1, keepdim=True)
    d = torch.einsum("...ij,...j->...i", extrinsics[..., :3, :3], d)
    o = extrinsics[..., :3, 3].broadcast_to(d.shape)
    return o, d


def sh_mask(sh_degree: int) -> Tensor:
    """gaussian_adapter.py:41-46."""
    m = torch.ones(((sh_degree + 1) ** 2,), dtype=torch.float32)
    for degree in range(1, sh_degree + 1):
        m[degree ** 2:(degree + 1) ** 2] = 0.1 * 0.25 ** degree
    return m


def scale_multiplier(intrinsics: Tensor, pixel_size: Tensor, multiplier: float = 0.1) -> Tensor:
    """gaussian_adapter.py:97-108."""
    xy = multiplier * torch.einsum("...ij,j->...i", intrinsics[..., :2, :2].inverse(), pixel_size)
    return xy.sum(dim=-1)


@dataclass
class Gaussians:
    means: Tensor
    covariances: Tensor
    scales: Tensor
    rotations: Tensor
    harmonics: Tensor
    opacities: Tensor


def adapter_forward(extrinsics, intrinsics, coordinates, depths, opacities, raw_gaussians,
                    image_shape, scale_min: float, scale_max: float, sh_degree: int,
                    eps: float = 1e-8) -> Gaussians:
    """gaussian_adapter.py:48-95, same broadcasting."""
    d_sh = (sh_degree + 1) ** 2
    scales, rotations, sh = raw_gaussians.split((3, 4, 3 * d_sh), dim=-1)
    scales = scale_min + (scale_max - scale_min) * scales.sigmoid()
    h, w = image_shape
    pixel_size = 1 / torch.tensor((w, h), dtype=torch.float32, device=extrinsics.device)
    mult = scale_multiplier(intrinsics, pixel_size)
    scales = scales * depths[..., None] * mult[..., None]
    rotations = rotations / (rotations.norm(dim=-1, keepdim=True) + eps)
    sh = sh.reshape(*sh.shape[:-1], 3, d_sh)
    sh = sh.broadcast_to((*opacities.shape, 3, d_sh)) * sh_mask(sh_degree).to(sh)
    cov = build_covariance(scales, rotations)
    c2w = extrinsics[..., :3, :3]
    cov = c2w @ cov @ c2w.transpose(-1, -2)
    o, d = world_rays(coordinates, extrinsics, intrinsics)
    means = o + d * depths[..., None]
    return Gaussians(means, cov, scales, rotations.broadcast_to((*scales.shape[:-1], 4)),
                     rotate_sh(sh, c2w[..., None, :, :]), opacities)
