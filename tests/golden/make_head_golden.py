"""Generates tests/golden/head.npz: the tail of the REFERENCE's EncoderEpipolar.forward
(/root/reference/src/model/encoder/encoder_epipolar.py:143-214) run on the CPU in the build
container with the reference's own DepthPredictorMonocular, GaussianAdapter, sample_image_grid
and nn.Sequential heads.  The glue between them (the method body itself needs the backbone and
the dataset types) is restated here line by line with the reference line numbers.  e3nn is the
oracle's stand-in (oracle/adapter_ref.py).  Seeds are searched for a draw whose uniforms stay
1e-3 away from every CDF edge, so that an fp32 GEMM on another device picks the same buckets.

    python tests/golden/make_head_golden.py
"""
import importlib
import os
import sys

import numpy as np
import torch
from einops import rearrange
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from pixelsplat_amd.synthetic import make_cameras  # noqa: E402


def run(mods, seed, *, b, v, h, w, c, s, srf, gpp, x_map):
    depth_mod, ad, proj = mods
    torch.manual_seed(seed)
    adapter = ad.adapter.GaussianAdapter(ad.adapter.GaussianAdapterCfg(0.5, 15.0, 4))
    depth_predictor = depth_mod.DepthPredictorMonocular(c, s, srf, False)
    to_gaussians = nn.Sequential(nn.ReLU(), nn.Linear(c, srf * (2 + adapter.d_in)))  # :85-91
    with torch.no_grad():
        depth_predictor.projection[1].weight.mul_(4.0)
    ctx, _ = make_cameras(b, v, 4, (64, 64), torch.Generator().manual_seed(seed))
    context = dict(extrinsics=ctx.extrinsics, intrinsics=ctx.intrinsics, near=ctx.near, far=ctx.far)
    features_in = torch.randn(b, v, c, h, w, requires_grad=True)
    drawn, pdfs = [], []
    real_rand = torch.rand

    def recording_rand(*a, **k):
        t = real_rand(*a, **k)
        drawn.append(t.clone())
        return t

    hook = depth_predictor.to_pdf.register_forward_hook(lambda _m, _i, o: pdfs.append(o.detach()))
    torch.rand = recording_rand
    try:
        features = rearrange(features_in, "b v c h w -> b v (h w) c")                       # :143
        depths, densities = depth_predictor.forward(features, context["near"], context["far"],
                                                    False, gpp)                              # :145
        xy_ray, _ = proj.sample_image_grid((h, w), features.device)                         # :154
        xy_ray = rearrange(xy_ray, "h w xy -> (h w) () xy")                                 # :155
        gaussians = rearrange(to_gaussians(features), "... (srf c) -> ... srf c", srf=srf)  # :156
        offset_xy = gaussians[..., :2].sigmoid()                                            # :161
        pixel_size = 1 / torch.tensor((w, h), dtype=torch.float32)                          # :162
        xy_ray = xy_ray + (offset_xy - 0.5) * pixel_size                                    # :163
        exponent = 2 ** x_map                                                               # :106-107
        opac = 0.5 * (1 - (1 - densities) ** exponent + densities ** (1 / exponent))        # :110
        g = adapter.forward(                                                                # :165
            rearrange(context["extrinsics"], "b v i j -> b v () () () i j"),
            rearrange(context["intrinsics"], "b v i j -> b v () () () i j"),
            rearrange(xy_ray, "b v r srf xy -> b v r srf () xy"),
            depths, opac / gpp,
            rearrange(gaussians[..., 2:], "b v r srf c -> b v r srf () c"), (h, w))
    finally:
        torch.rand = real_rand
        hook.remove()
    (uniforms,), (pdf,) = drawn, pdfs
    cdf = (pdf / pdf.sum(-1, keepdim=True)).double().cumsum(-1)
    margin = (cdf[..., None, :] - uniforms.double()[..., None]).abs().min().item()
    out = dict(                                                                             # :195-214
        means=rearrange(g.means, "b v r srf spp xyz -> b (v r srf spp) xyz"),
        covariances=rearrange(g.covariances, "b v r srf spp i j -> b (v r srf spp) i j"),
        harmonics=rearrange(g.harmonics, "b v r srf spp c d_sh -> b (v r srf spp) c d_sh"),
        opacities=rearrange(g.opacities, "b v r srf spp -> b (v r srf spp)"))
    return margin, out, features_in, uniforms, context, depth_predictor, to_gaussians


def main():
    ref_import.setup(2)
    depth_mod = importlib.import_module("src.model.encoder.epipolar.depth_predictor_monocular")
    ad = ref_import.adapter_modules()
    mods = (depth_mod, ad, ad.projection)
    saved = {}
    for tag, kw in (("a", dict(b=1, v=2, h=4, w=6, c=16, s=8, srf=1, gpp=3, x_map=0.5)),
                    ("b", dict(b=2, v=2, h=3, w=5, c=12, s=6, srf=2, gpp=2, x_map=0.0))):
        for seed in range(200):
            margin, out, feats, uniforms, context, dp, tg = run(mods, seed, **kw)
            if margin > 1e-3:
                break
        else:
            raise RuntimeError("no seed with a safe margin")
        weights = {k: torch.randn_like(t) for k, t in out.items()}
        sum((out[k] * weights[k]).sum() for k in out).backward()
        saved.update({f"{tag}_{k}": t.detach() for k, t in out.items()})
        saved.update({f"{tag}_w_{k}": t for k, t in weights.items()})
        saved.update({f"{tag}_features": feats.detach(), f"{tag}_grad_features": feats.grad,
                      f"{tag}_uniforms": uniforms,
                      f"{tag}_cfg": np.array([kw["s"], kw["srf"], kw["gpp"], kw["x_map"], seed, margin])})
        saved.update({f"{tag}_{k}": t for k, t in context.items()})
        for name, mod in (("dp", dp.projection[1]), ("tg", tg[1])):
            saved[f"{tag}_{name}_weight"] = mod.weight.detach()
            saved[f"{tag}_{name}_bias"] = mod.bias.detach()
            saved[f"{tag}_{name}_grad_weight"] = mod.weight.grad
            saved[f"{tag}_{name}_grad_bias"] = mod.bias.grad
        print(tag, "seed", seed, "margin %.2e" % margin)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "head.npz"),
                        **{k: (t.numpy() if isinstance(t, torch.Tensor) else t) for k, t in saved.items()})
    print("wrote head.npz", len(saved), "arrays")


if __name__ == "__main__":
    main()
