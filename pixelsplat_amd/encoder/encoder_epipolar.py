"""The tail of `EncoderEpipolar.forward` (/root/reference/src/model/encoder/encoder_epipolar.py:
143-214): per-pixel features -> depth samples -> Gaussians in the rasterizer's layout, i.e.
the chain depth predictor -> `to_gaussians` head -> Gaussian adapter
(rank 2).  `EncoderEpipolarHead` owns the reference's submodules under the reference's names
(`depth_predictor`, `to_gaussians`, `gaussian_adapter`, `to_opacity`), so the corresponding
`encoder.*` checkpoint entries load unchanged; the backbone, the epipolar transformer
(pixelsplat_amd.encoder.EpipolarTransformer) and the skip connection stay with the caller.

Fusions relative to the reference's op list:
  * one ReLU shared by the two heads (both `nn.Sequential`s start with the same ReLU);
  * both linear layers use the split-k weight gradient (k = every ray of the batch);
  * softmax / sampling / depth / opacity mapping / 1/gpp: ps_depth_sampler_* (one kernel);
  * xy-offset sigmoid, pixel grid, the [..., 2:] slice and the whole adapter:
    ps_gaussian_head_* on the linear layer's own 84-float rows (no slice copy, and the
    kernel's gradient IS the linear layer's output gradient).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch
from torch import Tensor, nn

from .. import _lib
from ..epipolar import _RayLinear
from .depth_predictor import DepthPredictorMonocular
from .gaussian_adapter import GaussianAdapter, GaussianAdapterCfg, _conj, _p, _stream


@dataclass
class OpacityMappingCfg:      # encoder_epipolar.py:25-29
    initial: float
    final: float
    warm_up: int


@dataclass
class EncoderEpipolarHeadCfg:
    """The fields of EncoderEpipolarCfg (encoder_epipolar.py:32-48) this part reads."""
    d_feature: int
    num_monocular_samples: int
    num_surfaces: int
    predict_opacity: bool
    gaussians_per_pixel: int
    gaussian_adapter: GaussianAdapterCfg
    opacity_mapping: OpacityMappingCfg
    use_transmittance: bool


@dataclass
class Gaussians:              # src/model/types.py:7-12
    means: Tensor             # [batch, gaussian, 3]
    covariances: Tensor       # [batch, gaussian, 3, 3]
    harmonics: Tensor         # [batch, gaussian, 3, d_sh]
    opacities: Tensor         # [batch, gaussian]


class _Head(torch.autograd.Function):
    """(depths [V, E, spp], head rows [V, E, 2 + 7 + 3K]) -> means, covariances, harmonics."""

    @staticmethod
    def forward(ctx, cfg, image_shape, surfaces, eps, extrinsics, intrinsics, depths, rows):
        lib = _lib.load()
        nv, ne, spp = depths.shape
        h, w = int(image_shape[0]), int(image_shape[1])
        k = (cfg.sh_degree + 1) ** 2
        f32 = dict(dtype=torch.float32, device=depths.device)
        views = torch.empty((nv, 192), **f32)
        _lib.check(lib.ps_gaussian_adapter_views(nv, cfg.sh_degree, h, w, _p(extrinsics),
                                                 _p(intrinsics), _p(_conj(depths.device)),
                                                 _p(views), _stream()), "ps_gaussian_adapter_views")
        means = torch.empty((nv, ne, spp, 3), **f32)
        cov = torch.empty((nv, ne, spp, 3, 3), **f32)
        harm = torch.empty((nv, ne, spp, 3, k), **f32)
        _lib.check(lib.ps_gaussian_head_forward(
            nv, h, w, surfaces, spp, cfg.sh_degree, C.c_float(cfg.gaussian_scale_min),
            C.c_float(cfg.gaussian_scale_max), C.c_float(eps), _p(views), _p(depths), _p(rows),
            _p(means), _p(cov), _p(harm), _stream()), "ps_gaussian_head_forward")
        ctx.args = (cfg, h, w, surfaces, eps)
        ctx.save_for_backward(views, depths, rows)
        return means, cov, harm

    @staticmethod
    def backward(ctx, d_means, d_cov, d_harm):
        lib = _lib.load()
        views, depths, rows = ctx.saved_tensors
        cfg, h, w, surfaces, eps = ctx.args
        nv, ne, spp = depths.shape
        d_rows, d_depths = torch.empty_like(rows), torch.empty_like(depths)
        _lib.check(lib.ps_gaussian_head_backward(
            nv, h, w, surfaces, spp, cfg.sh_degree, C.c_float(cfg.gaussian_scale_min),
            C.c_float(cfg.gaussian_scale_max), C.c_float(eps), _p(views), _p(depths), _p(rows),
            _p(d_means.contiguous()), _p(d_cov.contiguous()), _p(d_harm.contiguous()),
            _p(d_rows), _p(d_depths), _stream()), "ps_gaussian_head_backward")
        return None, None, None, None, None, None, d_depths, d_rows


class EncoderEpipolarHead(nn.Module):
    depth_predictor: DepthPredictorMonocular
    to_gaussians: nn.Sequential
    gaussian_adapter: GaussianAdapter

    def __init__(self, cfg: EncoderEpipolarHeadCfg) -> None:
        super().__init__()
        self.cfg = cfg
        self.depth_predictor = DepthPredictorMonocular(
            cfg.d_feature, cfg.num_monocular_samples, cfg.num_surfaces, cfg.use_transmittance)
        self.gaussian_adapter = GaussianAdapter(cfg.gaussian_adapter)
        if cfg.predict_opacity:
            self.to_opacity = nn.Sequential(nn.ReLU(), nn.Linear(cfg.d_feature, 1), nn.Sigmoid())
        self.to_gaussians = nn.Sequential(
            nn.ReLU(),
            nn.Linear(cfg.d_feature, cfg.num_surfaces * (2 + self.gaussian_adapter.d_in)),
        )

    def opacity_exponent(self, global_step: int) -> float:
        """2**x of map_pdf_to_opacity (encoder_epipolar.py:105-107)."""
        m = self.cfg.opacity_mapping
        return 2.0 ** (m.initial + min(global_step / m.warm_up, 1) * (m.final - m.initial))

    def forward(self, features: Tensor, context: dict, global_step: int,
                deterministic: bool = False, eps: float = 1e-8) -> Gaussians:
        """features [b, v, c, h, w] (after the skip connection, encoder_epipolar.py:138-140);
        context: extrinsics [b, v, 4, 4], intrinsics [b, v, 3, 3], near, far [b, v]."""
        if not features.is_cuda:
            raise RuntimeError("pixelsplat_amd EncoderEpipolarHead needs GPU tensors (no CPU fallback)")
        cfg = self.cfg
        b, v, c, h, w = features.shape
        r, srf = h * w, cfg.num_surfaces
        gpp = 1 if deterministic else cfg.gaussians_per_pixel
        # "b v c h w -> b v (h w) c" (:143); free when the producer is channels-last
        rows_in = features.permute(0, 1, 3, 4, 2).reshape(b, v, r, c)
        activated = torch.relu(rows_in).reshape(b * v * r, c)
        depths, opacities = self.depth_predictor._run(
            rows_in, context["near"], context["far"], deterministic, gpp,
            self.opacity_exponent(global_step), 1.0 / cfg.gaussians_per_pixel, activated=activated)
        linear = self.to_gaussians[1]
        head_rows = _RayLinear.apply(activated, linear.weight, linear.bias, None)
        means, cov, harm = _Head.apply(
            cfg.gaussian_adapter, (h, w), srf, eps,
            context["extrinsics"].reshape(b * v, 4, 4).float().contiguous(),
            context["intrinsics"].reshape(b * v, 3, 3).float().contiguous(),
            depths.reshape(b * v, r * srf, gpp), head_rows.view(b * v, r * srf, -1))
        if cfg.predict_opacity:   # :186-190
            opacities = opacities * self.to_opacity(rows_in).view(b, v, r, 1, 1)
        g = v * r * srf * gpp     # "b v r srf spp ... -> b (v r srf spp) ..." (:195-214)
        d_sh = self.gaussian_adapter.d_sh
        return Gaussians(means.view(b, g, 3), cov.view(b, g, 3, 3), harm.view(b, g, 3, d_sh),
                         opacities.reshape(b, g))
