"""GPU parity of the Gaussian adapter kernels (csrc/gaussian_adapter.hip) through the drop-in
GaussianAdapter module: against the golden vectors of the REAL reference module
(tests/golden/adapter.npz; e3nn part self-referential, see oracle/adapter_ref.py) and against
autograd through the oracle at other shapes / SH degrees.  fp32 tolerances: 2e-6 relative on
outputs, 2e-5 on gradients (sums of up to 75 products)."""
import os

import numpy as np
import pytest
import torch

from oracle import adapter_ref as A
from pixelsplat_amd.synthetic import make_cameras

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "adapter.npz")


def _close(a, b, rel, name):
    scale = max(b.abs().max().item(), 1e-6)
    err = (a - b).abs().max().item() / scale
    assert err < rel, f"{name}: {err:.2e}"


def test_adapter_vs_reference_golden(gpu_device):
    from pixelsplat_amd.encoder import GaussianAdapter, GaussianAdapterCfg

    g = {k: torch.from_numpy(v) for k, v in np.load(GOLD).items() if k != "image_shape"}
    h, w = (int(x) for x in np.load(GOLD)["image_shape"])
    dev = gpu_device
    net = GaussianAdapter(GaussianAdapterCfg(0.5, 15.0, 4)).to(dev)
    leaves = {k: g[k].to(dev).requires_grad_(True)
              for k in ("coordinates", "depths", "opacities", "raw_gaussians")}
    out = net(g["extrinsics"][:, :, None, None, None].to(dev),
              g["intrinsics"][:, :, None, None, None].to(dev), leaves["coordinates"],
              leaves["depths"], leaves["opacities"], leaves["raw_gaussians"], (h, w))
    outs = dict(means=out.means, covariances=out.covariances, harmonics=out.harmonics,
                opacities_out=out.opacities, scales=out.scales, rotations=out.rotations)
    for k, t in outs.items():
        assert t.shape == g[k].shape, k
        _close(t.detach().cpu(), g[k], 2e-6, k)
    sum((outs[k] * g["w_" + k].to(dev)).sum()
        for k in ("means", "covariances", "harmonics", "opacities_out")).backward()
    for k, t in leaves.items():
        _close(t.grad.cpu(), g["grad_" + k], 2e-5, "grad_" + k)


@pytest.mark.parametrize("deg,b,v,r,srf,spp", [(4, 1, 2, 150, 1, 3), (2, 2, 3, 70, 2, 1),
                                               (0, 1, 1, 64, 1, 2), (3, 1, 2, 129, 1, 4)])
def test_adapter_vs_oracle_autograd(gpu_device, deg, b, v, r, srf, spp):
    from pixelsplat_amd.encoder import GaussianAdapter, GaussianAdapterCfg

    torch.manual_seed(deg * 10 + spp)
    ctx, _ = make_cameras(b, v, 4, (64, 64), torch.Generator().manual_seed(3))
    k = (deg + 1) ** 2
    inputs = dict(coordinates=torch.rand(b, v, r, srf, 1, 2),
                  depths=torch.rand(b, v, r, srf, spp) * 4 + 0.3,
                  opacities=torch.rand(b, v, r, srf, spp),
                  raw_gaussians=torch.randn(b, v, r, srf, 1, 7 + 3 * k))
    ext, intr = ctx.extrinsics[:, :, None, None, None], ctx.intrinsics[:, :, None, None, None]
    wts = None

    def run(device):
        nonlocal wts
        lv = {kk: t.clone().to(device).requires_grad_(True) for kk, t in inputs.items()}
        if device == "cpu":
            o = A.adapter_forward(ext, intr, lv["coordinates"], lv["depths"], lv["opacities"],
                                  lv["raw_gaussians"], (48, 64), 0.5, 15.0, deg)
        else:
            net = GaussianAdapter(GaussianAdapterCfg(0.5, 15.0, deg)).to(device)
            o = net(ext.to(device), intr.to(device), lv["coordinates"], lv["depths"],
                    lv["opacities"], lv["raw_gaussians"], (48, 64))
        outs = [o.means, o.covariances, o.harmonics]
        if wts is None:
            wts = [torch.randn_like(t) for t in outs]
        sum((t * wt.to(device)).sum() for t, wt in zip(outs, wts)).backward()
        return [t.detach().cpu() for t in outs], {kk: t.grad.cpu() for kk, t in lv.items()
                                                  if t.grad is not None}

    o_ref, g_ref = run("cpu")
    o_hip, g_hip = run(gpu_device)
    for a, bb, name in zip(o_hip, o_ref, ("means", "covariances", "harmonics")):
        _close(a, bb, 3e-6, name)
    for kk in g_ref:
        _close(g_hip[kk], g_ref[kk], 3e-5, "grad_" + kk)
