"""Generates tests/golden/epipolar_*.npz by running the REFERENCE's own modules (imported
unmodified from /root/reference through oracle/ref_import.py) on CPU with fixed seeds.
Run in the build container:   python tests/golden/make_epipolar_golden.py
The fixtures travel to the GPU box; /root/reference does not.
"""
import os
import sys

import numpy as np
import torch
from einops import rearrange

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_import as RI  # noqa: E402
from pixelsplat_amd.synthetic import make_cameras  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def _chain(m, sampler, feat, ctx, pe, lin, attn_mod, v):
    """The reference's own chain sampler -> get_depth -> clip -> relative disparity -> depth encoding ->
    PreNorm(Attention(x, z = kv)) + residual, in whatever dtype the inputs carry."""
    out = sampler(feat, ctx.extrinsics, ctx.intrinsics, ctx.near, ctx.far)
    depths = m.lines.get_depth(
        rearrange(out.origins, "b v r xyz -> b v () r () xyz"),
        rearrange(out.directions, "b v r xyz -> b v () r () xyz"), out.xy_sample,
        rearrange(sampler.collect(ctx.extrinsics), "b v ov i j -> b v ov () () i j"),
        rearrange(sampler.collect(ctx.intrinsics), "b v ov i j -> b v ov () () i j"))
    nr = rearrange(ctx.near, "b v -> b v () () ()")
    fr = rearrange(ctx.far, "b v -> b v () () ()")
    rel = m.conversions.depth_to_relative_disparity(depths.maximum(nr).minimum(fr), nr, fr)
    kv = out.features + lin(pe(rel[..., None]))
    captured = {}
    hook = attn_mod.fn.attend.register_forward_hook(lambda mod, i, o: captured.__setitem__("attn", o))
    q = rearrange(feat, "b v c h w -> (b v h w) () c")
    y = attn_mod(q, z=rearrange(kv, "b v ov r s c -> (b v r) (s ov) c")) + q
    hook.remove()
    return dict(depths=depths, rel=rel, kv=kv, y=y, attn=captured["attn"], sampled=out.features)


def reference_in_float64(m, v, s, feat, ctx, pe_octaves, lin, attn_mod):
    """The SAME reference modules and weights evaluated in float64 (default dtype switched, so that the
    pixel grids and every constant the reference creates are double too): the yardstick for how much of
    a float32 difference is the reference's own rounding noise -- its 3x3 lstsq per sample, amplified by
    the 2 pi 2^9 gain of the positional encoding."""
    import copy
    from pixelsplat_amd.synthetic import Cameras
    torch.set_default_dtype(torch.float64)
    try:
        ctx64 = Cameras(ctx.extrinsics.double(), ctx.intrinsics.double(), ctx.near.double(), ctx.far.double())
        r = _chain(m, m.sampler.EpipolarSampler(v, s), feat.double(), ctx64,
                   m.pe.PositionalEncoding(pe_octaves), copy.deepcopy(lin).double(),
                   copy.deepcopy(attn_mod).double(), v)
    finally:
        torch.set_default_dtype(torch.float32)
    return {k: t.detach().numpy() for k, t in r.items()}


def one(name, b, v, grid, c, s, seed, d_dot=16, heads=2):
    m = RI.modules(v)
    gen = torch.Generator().manual_seed(seed)
    ctx, _ = make_cameras(b, v, 4, (256, 256), gen)
    h, w = grid
    feat = torch.randn((b, v, c, h, w), generator=gen)
    sampler = m.sampler.EpipolarSampler(v, s)
    out = sampler(feat, ctx.extrinsics, ctx.intrinsics, ctx.near, ctx.far)
    proj = m.lines.project_rays(
        rearrange(out.origins, "b v r xyz -> b v () r xyz"),
        rearrange(out.directions, "b v r xyz -> b v () r xyz"),
        rearrange(sampler.collect(ctx.extrinsics), "b v ov i j -> b v ov () i j"),
        rearrange(sampler.collect(ctx.intrinsics), "b v ov i j -> b v ov () i j"),
        rearrange(ctx.near, "b v -> b v () ()"), rearrange(ctx.far, "b v -> b v () ()"))
    depths = m.lines.get_depth(
        rearrange(out.origins, "b v r xyz -> b v () r () xyz"),
        rearrange(out.directions, "b v r xyz -> b v () r () xyz"), out.xy_sample,
        rearrange(sampler.collect(ctx.extrinsics), "b v ov i j -> b v ov () () i j"),
        rearrange(sampler.collect(ctx.intrinsics), "b v ov i j -> b v ov () () i j"))
    # depth encoding + one cross-attention layer exactly as EpipolarTransformer wires them
    torch.manual_seed(seed)
    pe = m.pe.PositionalEncoding(10)
    lin = torch.nn.Linear(20, c)
    nr = rearrange(ctx.near, "b v -> b v () () ()")
    fr = rearrange(ctx.far, "b v -> b v () () ()")
    dclip = depths.maximum(nr).minimum(fr)
    rel = m.conversions.depth_to_relative_disparity(dclip, nr, fr)
    enc = lin(pe(rel[..., None]))
    kv = out.features + enc
    tfm = m.tfm.Transformer(c, 1, heads, d_dot, 2 * c, selfatt=False, kv_dim=c,
                            feed_forward_layer=lambda dim, hid, dropout=0.0: torch.nn.Identity())
    attn_mod = tfm.layers[0][0]
    captured = {}
    attn_mod.fn.attend.register_forward_hook(lambda mod, i, o: captured.__setitem__("attn", o))
    q = rearrange(feat, "b v c h w -> (b v h w) () c")
    z = rearrange(kv, "b v ov r s c -> (b v r) (s ov) c")
    y = attn_mod(q, z=z) + q
    sd = {f"attn.{k}": t.detach().numpy() for k, t in attn_mod.state_dict().items()}
    r64 = reference_in_float64(m, v, s, feat, ctx, 10, lin, attn_mod)
    # the refactored chain reproduces the float32 numbers stored below bit for bit
    r32 = _chain(m, sampler, feat, ctx, pe, lin, attn_mod, v)
    assert torch.equal(r32["y"], y) and torch.equal(r32["attn"], captured["attn"]) and torch.equal(r32["rel"], rel)
    print(name, "reference fp32 vs fp64: depth rel", float(np.median(np.abs(depths.detach().numpy() - r64["depths"]) / np.abs(r64["depths"]))),
          "rel_disparity", float(np.abs(rel.detach().numpy() - r64["rel"]).max()),
          "attn", float(np.abs(captured["attn"].detach().numpy() - r64["attn"]).max()),
          "out", float(np.abs(y.detach().numpy() - r64["y"]).max()))
    np.savez_compressed(
        os.path.join(HERE, name),
        features_in=feat.numpy(), extrinsics=ctx.extrinsics.numpy(),
        intrinsics=ctx.intrinsics.numpy(), near=ctx.near.numpy(), far=ctx.far.numpy(),
        num_samples=s, heads=heads, d_dot=d_dot,
        origins=out.origins.numpy(), directions=out.directions.numpy(),
        xy_ray=out.xy_ray.numpy(), valid=out.valid.numpy(), xy_sample=out.xy_sample.numpy(),
        xy_sample_near=out.xy_sample_near.numpy(), xy_sample_far=out.xy_sample_far.numpy(),
        sampled=out.features.numpy(),
        t_min=proj["t_min"].numpy(), t_max=proj["t_max"].numpy(), xy_min=proj["xy_min"].numpy(),
        xy_max=proj["xy_max"].numpy(), overlaps=proj["overlaps_image"].numpy(),
        depths=depths.detach().numpy(), rel_disparity=rel.detach().numpy(),
        depth_w=lin.weight.detach().numpy(), depth_b=lin.bias.detach().numpy(),
        kv=kv.detach().numpy(), attn_out=y.detach().numpy(),
        attn_weights=captured["attn"].detach().numpy(),
        depths64=r64["depths"], rel_disparity64=r64["rel"], attn_weights64=r64["attn"],
        attn_out64=r64["y"], sampled64=r64["sampled"],
        index_v=sampler.index_v.numpy(), transpose_v=sampler.transpose_v.numpy(),
        transpose_ov=sampler.transpose_ov.numpy(), **sd)
    print(name, "ok", {k: v.shape for k, v in [("sampled", out.features), ("kv", kv)]})


if __name__ == "__main__":
    one("epipolar_v2.npz", 1, 2, (8, 8), 8, 8, seed=0)
    one("epipolar_v3.npz", 1, 3, (6, 10), 8, 4, seed=1)
