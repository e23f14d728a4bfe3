"""DepthPredictorMonocular mirror
(/root/reference/src/model/encoder/epipolar/depth_predictor_monocular.py:10-81).

Same constructor `(d_in, num_samples, num_surfaces, use_transmittance)`, same parameter names
(`projection.1.weight / bias`: released checkpoints load unchanged) and the same
`forward(features, near, far, deterministic, gaussians_per_pixel) -> (depth, opacity)`.  The
projection stays a library GEMM; everything after it -- bucket softmax, offset sigmoid, the
discrete sampler (discrete_probability_distribution.py:7-33), the gathers, the disparity ->
depth conversion (conversions.py:5-14) and the optional transmittance opacity -- is one HIP
kernel forward and one backward (csrc/depth_sampler.hip).  The uniform numbers are drawn with
the reference's own call, `torch.rand((*batch, num_samples), device=...)`, so a seeded run
consumes the generator identically.

`forward_mapped` additionally folds the encoder's `map_pdf_to_opacity(...) / gpp`
(encoder_epipolar.py:97-110, :170) into the same kernels.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import Tensor, nn

from .. import _lib
from ..epipolar import _RayLinear


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class _DepthSampler(torch.autograd.Function):
    """projected [V, R, 2*S*srf], near/far [V], uniforms [V, R, srf, spp] | None ->
    depth, opacity [V, R, srf, spp]."""

    @staticmethod
    def forward(ctx, projected, near, far, uniforms, buckets, surfaces, spp, use_transmittance,
                exponent, scale):
        lib = _lib.load()
        nv, nr, _ = projected.shape
        desc = _lib.PsDepthSamplerDesc(nv, nr, buckets, surfaces, spp, int(uniforms is None),
                                       int(use_transmittance), float(exponent), float(scale))
        dev = projected.device
        depth = torch.empty((nv, nr, surfaces, spp), dtype=torch.float32, device=dev)
        opacity = torch.empty_like(depth)
        index = torch.empty((nv, nr, surfaces, spp), dtype=torch.int32, device=dev)
        _lib.check(lib.ps_depth_sampler_forward(C.byref(desc), _p(projected), _p(near), _p(far),
                                                _p(uniforms), _p(depth), _p(opacity), _p(index),
                                                _stream()), "ps_depth_sampler_forward")
        ctx.desc = desc
        ctx.save_for_backward(projected, near, far, index)
        ctx.mark_non_differentiable(index)
        ctx.set_materialize_grads(False)    # (no zero-filled "gradient" of the bucket index)
        return depth, opacity, index

    @staticmethod
    def backward(ctx, d_depth, d_opacity, _d_index):
        lib = _lib.load()
        projected, near, far, index = ctx.saved_tensors
        if d_depth is None and d_opacity is None:
            return (None,) * 10
        shape = index.shape
        d_depth = projected.new_zeros(shape) if d_depth is None else d_depth
        d_opacity = projected.new_zeros(shape) if d_opacity is None else d_opacity
        d_projected = torch.empty_like(projected)
        _lib.check(lib.ps_depth_sampler_backward(
            C.byref(ctx.desc), _p(projected), _p(near), _p(far), _p(index),
            _p(d_depth.contiguous()), _p(d_opacity.contiguous()), _p(d_projected), _stream()),
            "ps_depth_sampler_backward")
        return (d_projected,) + (None,) * 9


def sample_depths(projected: Tensor, near: Tensor, far: Tensor, num_surfaces: int,
                  uniforms: Tensor | None, num_samples_drawn: int,
                  use_transmittance: bool = False, opacity_exponent: float = 0.0,
                  opacity_scale: float = 1.0):
    """Functional form over the C ABI.  projected [b, v, r, 2*S*srf]; near, far [b, v];
    uniforms [b, v, r, srf, spp] or None (deterministic top-k).  Returns depth, opacity
    [b, v, r, srf, spp] and the int32 bucket index."""
    if not projected.is_cuda:
        raise RuntimeError("pixelsplat_amd depth sampler needs GPU tensors (no CPU fallback)")
    b, v, r, n = projected.shape
    buckets = n // (2 * num_surfaces)
    if buckets * 2 * num_surfaces != n:
        raise ValueError("projection width must be 2 * num_samples * num_surfaces")
    if uniforms is not None:
        if tuple(uniforms.shape) != (b, v, r, num_surfaces, num_samples_drawn):
            raise ValueError("uniforms must be [b, v, r, srf, spp]")
        uniforms = uniforms.reshape(b * v, r, num_surfaces, num_samples_drawn)
        uniforms = uniforms.to(torch.float32).contiguous()
    proj = projected.reshape(b * v, r, n).to(torch.float32).contiguous()
    nr = near.reshape(b * v).to(torch.float32).contiguous()
    fr = far.reshape(b * v).to(torch.float32).contiguous()
    depth, opacity, index = _DepthSampler.apply(proj, nr, fr, uniforms, buckets, num_surfaces,
                                                num_samples_drawn, use_transmittance,
                                                opacity_exponent, opacity_scale)
    shape = (b, v, r, num_surfaces, num_samples_drawn)
    return depth.view(shape), opacity.view(shape), index.view(shape)


class DepthPredictorMonocular(nn.Module):
    projection: nn.Sequential
    num_samples: int
    num_surfaces: int

    def __init__(self, d_in: int, num_samples: int, num_surfaces: int,
                 use_transmittance: bool) -> None:
        super().__init__()
        self.projection = nn.Sequential(
            nn.ReLU(),
            nn.Linear(d_in, 2 * num_samples * num_surfaces),
        )
        self.num_samples = num_samples
        self.num_surfaces = num_surfaces
        self.use_transmittance = use_transmittance
        # The reference keeps these two modules "for hooks to latch onto"
        # (depth_predictor_monocular.py:34-36; only src/paper/generate_sampling_figure.py does).
        # They are parameter-free; the fused kernel does not call them.
        self.to_pdf = nn.Softmax(dim=-1)
        self.to_offset = nn.Sigmoid()

    def _run(self, features, near, far, deterministic, gaussians_per_pixel, exponent, scale,
             activated=None):
        """`activated` = relu(features) as [b * v * r, c] when the caller already has it (the
        encoder head shares it with `to_gaussians`)."""
        if len(self.to_pdf._forward_hooks) or len(self.to_offset._forward_hooks):
            raise RuntimeError("forward hooks on to_pdf / to_offset are not served by the fused "
                               "depth sampler; read the distribution from `projection` instead")
        # projection = ReLU + Linear (parameters live in self.projection); the weight gradient
        # is a GEMM with k = all rays of the batch, which goes to the split-k kernel
        linear = self.projection[1]
        b, v, r, c = features.shape
        if activated is None:
            activated = torch.relu(features).reshape(b * v * r, c)
        projected = _RayLinear.apply(activated, linear.weight, linear.bias, None).view(b, v, r, -1)
        uniforms = None
        if not deterministic:
            # discrete_probability_distribution.py:20, same shape and device
            uniforms = torch.rand((b, v, r, self.num_surfaces, gaussians_per_pixel),
                                  device=projected.device)
        depth, opacity, _ = sample_depths(projected, near, far, self.num_surfaces, uniforms,
                                          gaussians_per_pixel, self.use_transmittance, exponent,
                                          scale)
        return depth, opacity

    def forward(self, features: Tensor, near: Tensor, far: Tensor, deterministic: bool,
                gaussians_per_pixel: int) -> tuple[Tensor, Tensor]:
        return self._run(features, near, far, deterministic, gaussians_per_pixel, 0.0, 1.0)

    def forward_mapped(self, features: Tensor, near: Tensor, far: Tensor, deterministic: bool,
                       gaussians_per_pixel: int, opacity_exponent: float,
                       opacity_scale: float) -> tuple[Tensor, Tensor]:
        """forward + `map_pdf_to_opacity(densities, step) * opacity_scale` of
        encoder_epipolar.py:97-110/:170 with opacity_exponent = 2**x of :106-107."""
        return self._run(features, near, far, deterministic, gaussians_per_pixel,
                         opacity_exponent, opacity_scale)
