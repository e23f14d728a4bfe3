"""GPU parity of the image-side loss kernels (csrc/image_losses.hip) through the LossMse /
LossDepth / compute_psnr mirrors: against the golden vectors of the REAL reference classes
(tests/golden/loss.npz) and against oracle/loss_ref.py at BASELINE.json configs[1]'s image
count.  fp32 sums of up to 5.5 M terms: 2e-6 relative on the scalars, gradients 1e-6 relative
to their largest entry (they are exact products / stencil sums of a few terms)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import loss_ref

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "loss.npz")


def gold(dev=None):
    g = {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(GOLD).items()}
    return g if dev is None else {k: v.to(dev) for k, v in g.items()}


def rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


def test_mse_and_psnr_vs_reference_golden(gpu_device):
    from pixelsplat_amd.loss import LossMse, LossMseCfg, LossMseCfgWrapper, compute_psnr

    g = gold(gpu_device)
    color = g["color"].clone().requires_grad_(True)
    loss = LossMse(LossMseCfgWrapper(LossMseCfg(weight=float(g["mse_weight"]))))
    assert loss.name == "mse"
    batch = {"target": {"image": g["target"]}}
    val = loss(SimpleNamespace(color=color, depth=None), batch, None, 0)
    assert val.shape == ()
    assert rel(val.detach(), g["mse"]) < 2e-6
    (val * 3.0).backward()
    assert rel(color.grad, g["mse_grad"] * 3.0) < 1e-6
    ps = compute_psnr(g["target"].flatten(0, 1), g["color"].flatten(0, 1))
    assert ps.shape == g["psnr"].shape and rel(ps, g["psnr"]) < 2e-6


@pytest.mark.parametrize("tag", ["d1", "d2", "d1s", "d2s"])
def test_depth_loss_vs_reference_golden(gpu_device, tag):
    from pixelsplat_amd.loss import LossDepth, LossDepthCfg, LossDepthCfgWrapper

    g = gold(gpu_device)
    weight, sigma, second = (float(x) for x in g[tag + "_cfg"])
    cfg = LossDepthCfg(weight=weight, sigma_image=None if sigma < 0 else sigma,
                       use_second_derivative=bool(second))
    loss = LossDepth(LossDepthCfgWrapper(cfg))
    depth = g["depth"].clone().requires_grad_(True)
    batch = {"target": {"image": g["target"], "near": g["near"], "far": g["far"]}}
    val = loss(SimpleNamespace(color=None, depth=depth), batch, None, 0)
    assert rel(val.detach(), g[tag + "_loss"]) < 2e-6
    (val * 0.5).backward()
    assert rel(depth.grad, g[tag + "_grad"] * 0.5) < 2e-6


def test_losses_at_full_size_vs_oracle(gpu_device):
    """BASELINE.json configs[1]: 7 scenes x 4 target views of 3 x 256 x 256."""
    from pixelsplat_amd.loss import depth_smoothness, mse_and_psnr

    dev = gpu_device
    torch.manual_seed(3)
    b, v, h, w = 7, 4, 256, 256
    color = torch.rand(b, v, 3, h, w) * 1.2 - 0.1
    target = torch.rand(b, v, 3, h, w)
    c0 = color.clone().requires_grad_(True)
    ref = loss_ref.mse_loss(c0, target, 1.0)
    ref.backward()
    c1 = color.to(dev).requires_grad_(True)
    val, psnr = mse_and_psnr(c1, target.to(dev), 1.0)
    val.backward()
    assert rel(val.detach().cpu(), ref.detach()) < 2e-6
    assert rel(c1.grad.cpu(), c0.grad) < 1e-6
    assert rel(psnr.cpu().flatten(), loss_ref.psnr(target.flatten(0, 1), color.flatten(0, 1))) < 2e-6
    # property: the loss of an image against itself is exactly zero, with zero gradient
    z, zp = mse_and_psnr(c1.detach().requires_grad_(True), c1.detach(), 1.0)
    assert float(z.detach()) == 0.0 and bool(torch.isinf(zp).all())
    # the same pass twice gives the same bits (fixed-order sums)
    val2, _ = mse_and_psnr(color.to(dev), target.to(dev), 1.0)
    assert float(val2) == float(val)

    near, far = torch.rand(b, v) + 0.5, torch.rand(b, v) * 50 + 20
    depth = near.log()[..., None, None] + (far.log() - near.log())[..., None, None] * (
        torch.rand(b, v, h, w) * 1.2 - 0.1)
    for sigma, second in ((None, False), (3.0, True)):
        d0 = depth.clone().requires_grad_(True)
        ref = loss_ref.depth_loss(d0, near, far, 0.25, sigma, second, target)
        ref.backward()
        d1 = depth.to(dev).requires_grad_(True)
        val = depth_smoothness(d1, near.to(dev), far.to(dev), 0.25, sigma, second, target.to(dev))
        val.backward()
        assert rel(val.detach().cpu(), ref.detach()) < 2e-6
        assert rel(d1.grad.cpu(), d0.grad) < 2e-6


def test_losses_refuse_cpu_tensors():
    from pixelsplat_amd.loss import compute_psnr, mse_and_psnr

    with pytest.raises(RuntimeError):
        mse_and_psnr(torch.rand(1, 3, 4, 4), torch.rand(1, 3, 4, 4))
    with pytest.raises(RuntimeError):
        compute_psnr(torch.rand(1, 3, 4, 4), torch.rand(1, 3, 4, 4))
