"""GaussianAdapter mirror (/root/reference/src/model/encoder/common/gaussian_adapter.py:13-116).

Same constructor `(cfg)`, same `forward(extrinsics, intrinsics, coordinates, depths, opacities,
raw_gaussians, image_shape, eps)` and return type; the whole forward -- scale map, quaternion,
covariance, world-space means, SH mask and e3nn Wigner-D rotation -- and its backward are one
HIP kernel each (csrc/gaussian_adapter.hip), writing the tensors the rasterizer consumes.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass

import torch
from torch import Tensor, nn

from .. import _lib
from ..wigner import conjugation_matrices


@dataclass
class GaussianAdapterCfg:
    gaussian_scale_min: float
    gaussian_scale_max: float
    sh_degree: int


class Gaussians:
    """Fields of the reference's `Gaussians` dataclass (gaussian_adapter.py:13-20).  `scales`
    and `rotations` are only read by the visualisation / ply dumps
    (encoder_epipolar.py:176-183), so they are computed on first access."""

    def __init__(self, means, covariances, harmonics, opacities, scales_fn, rotations_fn):
        self.means, self.covariances = means, covariances
        self.harmonics, self.opacities = harmonics, opacities
        self._scales_fn, self._rotations_fn = scales_fn, rotations_fn
        self._scales = self._rotations = None

    @property
    def scales(self) -> Tensor:
        if self._scales is None:
            self._scales = self._scales_fn()
        return self._scales

    @property
    def rotations(self) -> Tensor:
        if self._rotations is None:
            self._rotations = self._rotations_fn()
        return self._rotations


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_CONJ: dict = {}


def _conj(device) -> Tensor:
    if device not in _CONJ:
        _CONJ[device] = torch.from_numpy(conjugation_matrices().copy()).to(device)
    return _CONJ[device]


class _Adapter(torch.autograd.Function):
    """(coordinates [V,E,2], depths [V,E,spp], raw [V,E,7+3K]) -> means [V,E,spp,3],
    covariances [V,E,spp,3,3], harmonics [V,E,spp,3,K]."""

    @staticmethod
    def forward(ctx, cfg, image_shape, eps, extrinsics, intrinsics, coordinates, depths, raw):
        lib = _lib.load()
        nv, ne, spp = depths.shape
        deg = cfg.sh_degree
        k = (deg + 1) ** 2
        dev = depths.device
        f32 = dict(dtype=torch.float32, device=dev)
        views = torch.empty((nv, 192), **f32)
        _lib.check(lib.ps_gaussian_adapter_views(
            nv, deg, int(image_shape[0]), int(image_shape[1]), _p(extrinsics), _p(intrinsics),
            _p(_conj(dev)), _p(views), _stream()), "ps_gaussian_adapter_views")
        means = torch.empty((nv, ne, spp, 3), **f32)
        cov = torch.empty((nv, ne, spp, 3, 3), **f32)
        harm = torch.empty((nv, ne, spp, 3, k), **f32)
        _lib.check(lib.ps_gaussian_adapter_forward(
            nv, ne, spp, deg, C.c_float(cfg.gaussian_scale_min), C.c_float(cfg.gaussian_scale_max),
            C.c_float(eps), _p(views), _p(coordinates), _p(depths), _p(raw), _p(means), _p(cov),
            _p(harm), _stream()), "ps_gaussian_adapter_forward")
        ctx.cfg, ctx.eps = cfg, eps
        ctx.save_for_backward(views, coordinates, depths, raw)
        return means, cov, harm

    @staticmethod
    def backward(ctx, d_means, d_cov, d_harm):
        lib = _lib.load()
        views, coordinates, depths, raw = ctx.saved_tensors
        nv, ne, spp = depths.shape
        cfg = ctx.cfg
        d_raw, d_depths = torch.empty_like(raw), torch.empty_like(depths)
        d_coords = torch.empty_like(coordinates)
        _lib.check(lib.ps_gaussian_adapter_backward(
            nv, ne, spp, cfg.sh_degree, C.c_float(cfg.gaussian_scale_min),
            C.c_float(cfg.gaussian_scale_max), C.c_float(ctx.eps), _p(views), _p(coordinates),
            _p(depths), _p(raw), _p(d_means.contiguous()), _p(d_cov.contiguous()),
            _p(d_harm.contiguous()), _p(d_raw), _p(d_depths), _p(d_coords), _stream()),
            "ps_gaussian_adapter_backward")
        return None, None, None, None, None, d_coords, d_depths, d_raw


class GaussianAdapter(nn.Module):
    cfg: GaussianAdapterCfg

    def __init__(self, cfg: GaussianAdapterCfg):
        super().__init__()
        self.cfg = cfg
        # same non-persistent buffer as the reference (gaussian_adapter.py:37-46); the kernel
        # carries the same constants
        self.register_buffer("sh_mask", torch.ones((self.d_sh,), dtype=torch.float32),
                             persistent=False)
        for degree in range(1, self.cfg.sh_degree + 1):
            self.sh_mask[degree ** 2:(degree + 1) ** 2] = 0.1 * 0.25 ** degree

    def forward(self, extrinsics: Tensor, intrinsics: Tensor, coordinates: Tensor, depths: Tensor,
                opacities: Tensor, raw_gaussians: Tensor, image_shape: tuple[int, int],
                eps: float = 1e-8) -> Gaussians:
        if not depths.is_cuda:
            raise RuntimeError("pixelsplat_amd GaussianAdapter needs GPU tensors (no CPU fallback)")
        if self.cfg.sh_degree > 4:
            raise NotImplementedError("sh_degree <= 4")
        full = torch.broadcast_shapes(extrinsics.shape[:-2], intrinsics.shape[:-2],
                                      coordinates.shape[:-1], depths.shape, opacities.shape,
                                      raw_gaussians.shape[:-1])
        n = len(full)

        def lead(t, tail):        # batch shape padded on the left to the full rank
            return (1,) * (n - (t.dim() - tail)) + tuple(t.shape[:t.dim() - tail])

        cam = [max(a, b) for a, b in zip(lead(extrinsics, 2), lead(intrinsics, 2))]
        n_view_dims = max((i + 1 for i, s in enumerate(cam) if s != 1), default=0)
        per_entry = [max(a, b) for a, b in zip(lead(coordinates, 1), lead(raw_gaussians, 1))]
        n_sample_dims = 0
        while n_sample_dims < n - n_view_dims and per_entry[n - 1 - n_sample_dims] == 1:
            n_sample_dims += 1
        v_shape, e_shape = full[:n_view_dims], full[n_view_dims:n - n_sample_dims]
        s_shape = full[n - n_sample_dims:]
        nv, ne, spp = math.prod(v_shape), math.prod(e_shape), math.prod(s_shape)
        ext = extrinsics.broadcast_to(*v_shape, *([1] * (n - n_view_dims)), 4, 4)
        ext = ext.reshape(nv, 4, 4).float().contiguous()
        intr = intrinsics.broadcast_to(*v_shape, *([1] * (n - n_view_dims)), 3, 3)
        intr = intr.reshape(nv, 3, 3).float().contiguous()
        ones = [1] * n_sample_dims
        coords = coordinates.broadcast_to(*v_shape, *e_shape, *ones, 2).reshape(nv, ne, 2)
        raw = raw_gaussians.broadcast_to(*v_shape, *e_shape, *ones, raw_gaussians.shape[-1])
        raw = raw.reshape(nv, ne, raw_gaussians.shape[-1])
        if raw.shape[-1] != self.d_in:
            raise ValueError(f"raw_gaussians has {raw.shape[-1]} channels, expected {self.d_in}")
        dep = depths.broadcast_to(full).reshape(nv, ne, spp)
        means, cov, harm = _Adapter.apply(self.cfg, image_shape, eps, ext, intr,
                                          coords.float().contiguous(), dep.float().contiguous(),
                                          raw.float().contiguous())

        def scales_fn():          # gaussian_adapter.py:62-69
            s = raw_gaussians[..., :3].sigmoid()
            s = self.cfg.gaussian_scale_min + (self.cfg.gaussian_scale_max
                                               - self.cfg.gaussian_scale_min) * s
            h, w = image_shape
            px = 1 / torch.tensor((w, h), dtype=torch.float32, device=depths.device)
            return s * depths[..., None] * self.get_scale_multiplier(intrinsics, px)[..., None]

        def rotations_fn():       # gaussian_adapter.py:72, 94
            r = raw_gaussians[..., 3:7]
            r = r / (r.norm(dim=-1, keepdim=True) + eps)
            return r.broadcast_to((*full, 4))

        return Gaussians(means.reshape(*full, 3), cov.reshape(*full, 3, 3),
                         harm.reshape(*full, 3, self.d_sh), opacities.broadcast_to(full),
                         scales_fn, rotations_fn)

    def get_scale_multiplier(self, intrinsics: Tensor, pixel_size: Tensor,
                             multiplier: float = 0.1) -> Tensor:
        xy = multiplier * torch.einsum("...ij,j->...i", intrinsics[..., :2, :2].inverse(),
                                       pixel_size)
        return xy.sum(dim=-1)

    @property
    def d_sh(self) -> int:
        return (self.cfg.sh_degree + 1) ** 2

    @property
    def d_in(self) -> int:
        return 7 + 3 * self.d_sh
