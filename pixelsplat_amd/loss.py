"""Mirrors of the reference's image-side losses and PSNR metric:
`LossMse` (/root/reference/src/loss/loss_mse.py:12-31), `LossDepth`
(src/loss/loss_depth.py:14-60) and `compute_psnr` (src/evaluation/metrics.py:12-19), same
config dataclasses, constructor `(cfg_wrapper)` and
`forward(prediction, batch, gaussians, global_step)`; each runs as one pass over the rendered
images (csrc/image_losses.hip).  `LossMse.forward` leaves dL/dcolor behind, so its backward is a
single scale by the incoming gradient; `mse_and_psnr` returns the loss and the per-view PSNR
from the same pass.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, fields

import torch
from torch import Tensor, nn

from . import _lib


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(t: Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"pixelsplat_amd {what} needs GPU tensors (no CPU fallback)")


def _image_mse(pred: Tensor, target: Tensor, n_images: int, grad_scale: float, want_grad: bool):
    lib = _lib.load()
    elems = pred.numel() // n_images
    dev = pred.device
    sse = torch.empty((2, n_images), dtype=torch.float32, device=dev)
    grad = torch.empty_like(pred) if want_grad else None
    ws_bytes = lib.ps_image_mse_workspace_bytes(n_images, elems)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=dev)
    _lib.check(lib.ps_image_mse(n_images, elems, _p(pred), _p(target), C.c_float(grad_scale),
                                _p(grad), _p(sse[0]), _p(sse[1]), _p(ws), C.c_size_t(ws_bytes),
                                _stream()), "ps_image_mse")
    return sse, grad


class _Mse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, weight, n_images):
        n = pred.numel()
        sse, grad = _image_mse(pred, target, n_images, 2.0 * weight / n, ctx.needs_input_grad[0])
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(sse)
        ctx.set_materialize_grads(False)
        return sse[0].sum() * (weight / n), sse

    @staticmethod
    def backward(ctx, d_loss, _d_sse):
        if d_loss is None:
            return None, None, None, None
        (grad,) = ctx.saved_tensors
        return grad * d_loss, None, None, None


def _mse(color: Tensor, target: Tensor, weight: float):
    _need_gpu(color, "LossMse")
    lead = color.shape[:-3]
    n_images = max(1, int(torch.Size(lead).numel()))
    pred = color.to(torch.float32).contiguous()
    tgt = target.to(torch.float32).expand_as(color).contiguous()
    loss, sse = _Mse.apply(pred, tgt, float(weight), n_images)
    return loss, sse, lead, pred.numel() // n_images


def mse_loss(color: Tensor, target: Tensor, weight: float = 1.0) -> Tensor:
    """weight * mean((color - target)^2) (loss_mse.py:30-31) in one pass."""
    return _mse(color, target, weight)[0]


def mse_and_psnr(color: Tensor, target: Tensor, weight: float = 1.0):
    """color, target [..., c, h, w] -> (weight * mean squared error, PSNR per image [...]),
    both from the same pass."""
    loss, sse, lead, elems = _mse(color, target, weight)
    return loss, (-10 * (sse[1] / elems).log10()).view(lead)


@torch.no_grad()
def compute_psnr(ground_truth: Tensor, predicted: Tensor) -> Tensor:
    """metrics.py:12-19: [batch, c, h, w] x2 -> [batch]."""
    _need_gpu(predicted, "compute_psnr")
    n = predicted.shape[0]
    pred = predicted.to(torch.float32).contiguous()
    sse, _ = _image_mse(pred, ground_truth.to(torch.float32).expand_as(pred).contiguous(), n, 0.0,
                        False)
    return -10 * (sse[1] / (pred.numel() // n)).log10()


class Loss(nn.Module):
    """loss.py:15-26: the config is the single field of its wrapper dataclass."""

    def __init__(self, cfg) -> None:
        super().__init__()
        (field,) = fields(type(cfg))
        self.cfg = getattr(cfg, field.name)
        self.name = field.name


@dataclass
class LossMseCfg:
    weight: float


@dataclass
class LossMseCfgWrapper:
    mse: LossMseCfg


class LossMse(Loss):
    def forward(self, prediction, batch, gaussians, global_step: int) -> Tensor:
        return mse_loss(prediction.color, batch["target"]["image"], self.cfg.weight)


@dataclass
class LossDepthCfg:
    weight: float
    sigma_image: float | None
    use_second_derivative: bool


@dataclass
class LossDepthCfgWrapper:
    depth: LossDepthCfg


class _DepthSmoothness(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, near, far, image, weight, sigma, second):
        lib = _lib.load()
        n, h, w = depth.shape
        desc = _lib.PsDepthLossDesc(n, h, w, 0 if image is None else image.shape[1], int(second),
                                    int(sigma is not None), float(sigma or 0.0), float(weight))
        ws_bytes = lib.ps_depth_smoothness_workspace_bytes(C.byref(desc))
        if ws_bytes == 0:
            raise ValueError("depth maps must be larger than the difference stencil")
        ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=depth.device)
        loss = torch.empty((1,), dtype=torch.float32, device=depth.device)
        _lib.check(lib.ps_depth_smoothness_forward(C.byref(desc), _p(depth), _p(near), _p(far),
                                                   _p(image), _p(loss), _p(ws),
                                                   C.c_size_t(ws_bytes), _stream()),
                   "ps_depth_smoothness_forward")
        ctx.desc = desc
        ctx.save_for_backward(depth, near, far, image)
        return loss.view(())

    @staticmethod
    def backward(ctx, d_loss):
        lib = _lib.load()
        depth, near, far, image = ctx.saved_tensors
        d_depth = torch.empty_like(depth)
        d_loss = d_loss.to(torch.float32).reshape(1).contiguous()
        _lib.check(lib.ps_depth_smoothness_backward(C.byref(ctx.desc), _p(depth), _p(near), _p(far),
                                                    _p(image), _p(d_loss), _p(d_depth), _stream()),
                   "ps_depth_smoothness_backward")
        return d_depth, None, None, None, None, None, None


def depth_smoothness(depth: Tensor, near: Tensor, far: Tensor, weight: float,
                     sigma_image: float | None, use_second_derivative: bool,
                     target_image: Tensor | None = None) -> Tensor:
    """depth [b, v, h, w]; near, far [b, v]; target_image [b, v, c, h, w] (with sigma_image)."""
    _need_gpu(depth, "LossDepth")
    b, v, h, w = depth.shape
    image = None
    if sigma_image is not None:
        image = target_image.to(torch.float32).reshape(b * v, -1, h, w).contiguous()
    return _DepthSmoothness.apply(
        depth.to(torch.float32).reshape(b * v, h, w).contiguous(),
        near.to(torch.float32).reshape(b * v).contiguous(),
        far.to(torch.float32).reshape(b * v).contiguous(), image, weight, sigma_image,
        use_second_derivative)


class LossDepth(Loss):
    def forward(self, prediction, batch, gaussians, global_step: int) -> Tensor:
        t = batch["target"]
        return depth_smoothness(prediction.depth, t["near"], t["far"], self.cfg.weight,
                                self.cfg.sigma_image, self.cfg.use_second_derivative, t["image"])
