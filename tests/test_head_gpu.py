"""GPU parity of the encoder head (EncoderEpipolarHead: ps_depth_sampler_* + ps_gaussian_head_*
+ split-k linear layers) against the golden vectors of the REAL reference modules chained as in
encoder_epipolar.py:143-214 (tests/golden/head.npz) and against oracle/head_ref.py.  Outputs
2e-5 relative to each tensor's largest entry (an fp32 GEMM feeds exp / sigmoid / 1/x chains),
gradients 1e-4."""
import os

import numpy as np
import pytest
import torch

from oracle import head_ref

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "head.npz")
OUTS = ("means", "covariances", "harmonics", "opacities")


def load(tag):
    z = np.load(GOLD)
    return {k[len(tag) + 1:]: torch.from_numpy(np.asarray(z[k])) for k in z.files
            if k.startswith(tag + "_")}


def rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-12)


def make_head(c, s, srf, gpp, x_map, dev, predict_opacity=False):
    from pixelsplat_amd.encoder import (EncoderEpipolarHead, EncoderEpipolarHeadCfg,
                                        GaussianAdapterCfg, OpacityMappingCfg)
    cfg = EncoderEpipolarHeadCfg(
        d_feature=c, num_monocular_samples=s, num_surfaces=srf, predict_opacity=predict_opacity,
        gaussians_per_pixel=gpp, gaussian_adapter=GaussianAdapterCfg(0.5, 15.0, 4),
        opacity_mapping=OpacityMappingCfg(initial=x_map, final=x_map, warm_up=1),
        use_transmittance=False)
    return EncoderEpipolarHead(cfg).to(dev)


def with_uniforms(uniforms, fn):
    real = torch.rand
    torch.rand = lambda *a, **k: uniforms
    try:
        return fn()
    finally:
        torch.rand = real


@pytest.mark.parametrize("tag", ["a", "b"])
def test_head_vs_reference_golden(gpu_device, tag):
    g = load(tag)
    dev = gpu_device
    s, srf, gpp, x_map = int(g["cfg"][0]), int(g["cfg"][1]), int(g["cfg"][2]), float(g["cfg"][3])
    head = make_head(g["features"].shape[2], s, srf, gpp, x_map, dev)
    head.load_state_dict({"depth_predictor.projection.1.weight": g["dp_weight"],
                          "depth_predictor.projection.1.bias": g["dp_bias"],
                          "to_gaussians.1.weight": g["tg_weight"], "to_gaussians.1.bias": g["tg_bias"]})
    feats = g["features"].to(dev).requires_grad_(True)
    ctx = {k: g[k].to(dev) for k in ("extrinsics", "intrinsics", "near", "far")}
    out = with_uniforms(g["uniforms"].to(dev), lambda: head(feats, ctx, global_step=0))
    outs = (out.means, out.covariances, out.harmonics, out.opacities)
    for name, t in zip(OUTS, outs):
        assert t.shape == g[name].shape, name
        assert rel(t.detach().cpu(), g[name]) < 2e-5, f"{name}: {rel(t.detach().cpu(), g[name]):.2e}"
    sum((t * g["w_" + n].to(dev)).sum() for n, t in zip(OUTS, outs)).backward()
    assert rel(feats.grad.cpu(), g["grad_features"]) < 1e-4
    lin = {"dp": head.depth_predictor.projection[1], "tg": head.to_gaussians[1]}
    for k, m in lin.items():
        assert rel(m.weight.grad.cpu(), g[f"{k}_grad_weight"]) < 1e-4, k
        assert rel(m.bias.grad.cpu(), g[f"{k}_grad_bias"]) < 1e-4, k


def test_head_equals_the_separate_modules(gpu_device):
    """The fused head layout against DepthPredictorMonocular + GaussianAdapter called the way
    encoder_epipolar.py:145-173 calls them (slice, coordinate tensor and all)."""
    from pixelsplat_amd.synthetic import make_cameras

    dev = gpu_device
    torch.manual_seed(2)
    b, v, h, w, c, s, srf, gpp = 1, 2, 9, 13, 32, 32, 2, 3
    head = make_head(c, s, srf, gpp, 0.0, dev)
    cam, _ = make_cameras(b, v, 4, (64, 64), torch.Generator().manual_seed(1))
    ctx = {k: getattr(cam, k).to(dev) for k in ("extrinsics", "intrinsics", "near", "far")}
    feats = torch.randn(b, v, c, h, w, device=dev, requires_grad=True)
    uniforms = torch.rand(b, v, h * w, srf, gpp, device=dev)
    out = with_uniforms(uniforms, lambda: head(feats, ctx, 0))
    wts = [torch.randn_like(t) for t in (out.means, out.covariances, out.harmonics, out.opacities)]
    sum((t * x).sum() for t, x in zip((out.means, out.covariances, out.harmonics, out.opacities), wts)).backward()
    g_fused = feats.grad.clone()
    feats.grad = None

    rows = feats.permute(0, 1, 3, 4, 2).reshape(b, v, h * w, c)
    depths, dens = with_uniforms(uniforms, lambda: head.depth_predictor(rows, ctx["near"], ctx["far"], False, gpp))
    gs = head.to_gaussians(rows).view(b, v, h * w, srf, -1)
    xy = head_ref.sample_image_grid(h, w).to(dev).reshape(h * w, 1, 2)
    xy = xy + (gs[..., :2].sigmoid() - 0.5) / torch.tensor((w, h), dtype=torch.float32, device=dev)
    ga = head.gaussian_adapter(ctx["extrinsics"][:, :, None, None, None],
                               ctx["intrinsics"][:, :, None, None, None], xy[..., None, :], depths,
                               dens / gpp, gs[..., None, 2:], (h, w))
    n = v * h * w * srf * gpp
    sep = (ga.means.reshape(b, n, 3), ga.covariances.reshape(b, n, 3, 3),
           ga.harmonics.reshape(b, n, 3, 25), ga.opacities.reshape(b, n))
    for name, a, r in zip(OUTS, (out.means, out.covariances, out.harmonics, out.opacities), sep):
        assert rel(a.detach(), r.detach()) < 2e-6, name
    sum((t * x).sum() for t, x in zip(sep, wts)).backward()
    assert rel(g_fused, feats.grad) < 2e-5


def test_head_vs_oracle_deterministic(gpu_device):
    """deterministic=True (one Gaussian per pixel, opacity still / gpp) against the oracle."""
    from pixelsplat_amd.synthetic import make_cameras

    dev = gpu_device
    torch.manual_seed(4)
    b, v, h, w, c, s, srf, gpp = 2, 2, 8, 8, 24, 16, 1, 3
    head = make_head(c, s, srf, gpp, 1.0, dev)
    cam, _ = make_cameras(b, v, 4, (64, 64), torch.Generator().manual_seed(5))
    ctx_cpu = {k: getattr(cam, k) for k in ("extrinsics", "intrinsics", "near", "far")}
    feats = torch.randn(b, v, c, h, w)
    out = head(feats.to(dev), {k: t.to(dev) for k, t in ctx_cpu.items()}, global_step=5,
               deterministic=True)
    sd = {k: t.detach().cpu() for k, t in head.state_dict().items()}
    ref = head_ref.head_forward(
        feats, ctx_cpu, sd["depth_predictor.projection.1.weight"],
        sd["depth_predictor.projection.1.bias"], sd["to_gaussians.1.weight"],
        sd["to_gaussians.1.bias"], num_surfaces=srf, gaussians_per_pixel=gpp, uniforms=None,
        opacity_exponent=2.0, scale_min=0.5, scale_max=15.0, sh_degree=4)
    assert out.means.shape == (b, v * h * w, 3)
    # a GEMM-rounding tie in the top-1 choice moves a whole Gaussian: compare the rest
    same = (out.opacities.cpu() - ref[3]).abs() < 1e-4 * ref[3].abs() + 1e-7
    assert float(same.float().mean()) > 0.99
    for a, r in zip((out.means, out.covariances, out.harmonics), ref[:3]):
        a, r = a.cpu()[same], r[same]
        assert rel(a, r) < 2e-5
