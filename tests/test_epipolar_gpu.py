"""GPU parity of the (A) epipolar kernels against oracle/epipolar_ref.py (itself pinned
bit-for-bit against the imported reference, tests/test_oracle_epipolar.py).  Integer paths
(overlap flags, frame selectors, bilinear corner indices) and xy_sample: bit-exact.  Depth:
the reference solves a 3x3 lstsq in fp32, conditioning ~ 1/(1 - (a.b)^2) -- tolerance set
from that."""
import numpy as np
import pytest
import torch

from oracle import epipolar_ref as E
from pixelsplat_amd.synthetic import make_cameras

pytestmark = pytest.mark.gpu


def _cams(b, v, seed, hw=(256, 256), yaw=5.0):
    gen = torch.Generator().manual_seed(seed)
    ctx, _ = make_cameras(b, v, 4, hw, gen, max_yaw_deg=yaw)
    return ctx


@pytest.mark.parametrize("b,v,grid,s,seed", [(1, 2, (16, 16), 32, 0), (2, 3, (12, 20), 8, 1),
                                             (1, 2, (64, 64), 32, 2), (2, 2, (9, 7), 5, 3),
                                             (1, 3, (64, 64), 32, 4),      # configs[3] rays
                                             (1, 2, (128, 128), 32, 5)])   # configs[4] rays
def test_geometry_bit_exact(gpu_device, b, v, grid, s, seed):
    from pixelsplat_amd.epipolar import sample_geometry

    ctx = _cams(b, v, seed)
    h, w = grid
    feat = torch.zeros((b, v, 1, h, w))
    ref = E.sample(feat, ctx.extrinsics, ctx.intrinsics, ctx.near, ctx.far, s)
    w2c, k_inv = torch.linalg.inv(ctx.extrinsics), torch.linalg.inv(ctx.intrinsics)
    g = sample_geometry(ctx.extrinsics.to(gpu_device), ctx.intrinsics.to(gpu_device),
                        ctx.near.to(gpu_device), ctx.far.to(gpu_device), grid, s,
                        w2c=w2c.to(gpu_device), k_inv=k_inv.to(gpu_device))
    cpu = lambda t: t.cpu()
    assert torch.equal(cpu(g.origins), ref.origins.contiguous())
    assert torch.equal(cpu(g.directions), ref.directions)
    seg = ref.segment
    assert torch.equal(cpu(g.overlaps), seg.overlaps)
    flags = cpu(g.flags)
    assert torch.equal((flags >> 1) & 1, seg.near_valid.to(torch.uint8))
    assert torch.equal((flags >> 2) & 1, seg.far_valid.to(torch.uint8))
    use_fmin = ~seg.near_valid
    use_fmax = ~seg.far_valid
    assert torch.equal(((flags >> 3) & 3)[use_fmin].long(), seg.sel_min[use_fmin])
    assert torch.equal(((flags >> 5) & 3)[use_fmax].long(), seg.sel_max[use_fmax])
    m = seg.overlaps
    for name in ("xy_min", "xy_max", "t_min", "t_max"):
        a, r_ = cpu(getattr(g, name)), getattr(seg, name)
        mm = m if a.dim() == m.dim() else m[..., None].expand_as(a)
        assert torch.equal(a[mm], r_[mm]), name
    assert torch.equal(cpu(g.xy_sample), ref.xy_sample)
    # bilinear corner indices (functions of xy_sample) are therefore identical as well
    x0, y0, _, _, masks = E.bilinear_corners(cpu(g.xy_sample), h, w)
    rx0, ry0, _, _, rmasks = E.bilinear_corners(ref.xy_sample, h, w)
    assert torch.equal(x0, rx0) and torch.equal(y0, ry0) and torch.equal(masks, rmasks)
    # depth (a6): closed-form two-ray depth here, a 3x3 lstsq per sample in the reference.  The arbiter is
    # the oracle's formulas evaluated in float64.  No sample is excluded:
    #   * well-conditioned samples (|cos| < 0.999 between the two rays): 2e-3 relative worst case, 1e-5
    #     median, against the float32 oracle as before AND against float64;
    #   * the near-parallel rest (conditioning ~ 1 / (1 - cos^2)) is COUNTED and must land on the same
    #     side of the [near, far] clamp as the float64 depth (unless that sits within 1e-3 of a bound),
    #     with a relative-disparity error bounded by the conditioning.
    d, rd = cpu(g.depth), ref.depths
    ab = (ref.directions[:, :, None, :, None, :] * E.world_rays(
        ref.xy_sample, ctx.extrinsics[:, E.heterogeneous_index(v)][:, :, :, None, None],
        k_inv[:, E.heterogeneous_index(v)][:, :, :, None, None])[1]).sum(-1)
    d64 = E.sample(feat.double(), ctx.extrinsics.double(), ctx.intrinsics.double(), ctx.near.double(),
                   ctx.far.double(), s).depths
    good = m[..., None] & (ab.abs() < 0.999)
    rel = ((d - rd).abs() / rd.abs().clamp(min=1e-6))[good]
    assert rel.numel() == 0 or rel.max() < 2e-3, rel.max()
    assert rel.numel() == 0 or rel.median() < 1e-5
    rel64 = ((d.double() - d64).abs() / d64.abs().clamp(min=1e-6))[good]
    assert rel64.numel() == 0 or (rel64.max() < 2e-3 and rel64.median() < 1e-5), (rel64.max(), rel64.median())
    nr, fr = ctx.near[:, :, None, None, None], ctx.far[:, :, None, None, None]
    ref_rel = E.relative_disparity(rd.maximum(nr).minimum(fr), nr, fr)
    assert (cpu(g.rel_disparity) - ref_rel)[good].abs().max() < 2e-3
    # the near-parallel samples
    hard = m[..., None] & ~(ab.abs() < 0.999)
    n_hard = int(hard.sum())
    if n_hard:
        side = lambda t: (t > fr.to(t.dtype)).long() - (t < nr.to(t.dtype)).long()
        cond = (1 - ab.double() ** 2).clamp(min=1e-12)
        tol = (2e-3 * (2e-3 / cond).clamp(min=1.0)).clamp(max=1.0)       # ~ float32 eps / (1 - cos^2)
        # (a depth within its own conditioning error of a bound may fall on either side of it)
        band = ((d64 / nr.double() - 1).abs() < tol) | ((d64 / fr.double() - 1).abs() < tol)
        # the reference's own discontinuity: `parallel = dot > 1 - 1e-5` sends the depth to 1e10; a dot
        # product within float32 rounding of that threshold may be flagged either way (by the reference too)
        flag_band = (ab.double() - (1 - 1e-5)).abs() < 1e-6
        n_flag = int((hard & flag_band).sum())
        wrong_side = hard & (side(d) != side(d64)) & ~band & ~flag_band
        assert int(wrong_side.sum()) == 0, (int(wrong_side.sum()), n_hard)
        rel64_disp = E.relative_disparity(d64.maximum(nr.double()).minimum(fr.double()), nr.double(), fr.double())
        err = (cpu(g.rel_disparity).double() - rel64_disp).abs()
        chk = hard & ~flag_band
        assert bool((err[chk] <= tol[chk]).all()), float((err[chk] / tol[chk]).max())
        print(f"\n[depth b={b} v={v} {grid}] near-parallel samples (|cos| >= 0.999): {n_hard} of "
              f"{int((m[..., None].expand_as(hard)).sum())} ({n_flag} within rounding of the reference's parallel "
              f"threshold), all others on the float64 side of the [near, far] clamp; "
              f"max rel-disparity error there {float(err[hard].max()):.2e}")


def test_degenerate_cameras(gpu_device):
    """Identical cameras (rays parallel to their own re-projection) and a camera behind the
    other: nothing overlaps or everything is parallel; flags and samples still match."""
    from pixelsplat_amd.epipolar import sample_geometry

    ctx = _cams(1, 2, 7)
    ctx.extrinsics[:, 1] = ctx.extrinsics[:, 0]
    feat = torch.zeros((1, 2, 1, 8, 8))
    ref = E.sample(feat, ctx.extrinsics, ctx.intrinsics, ctx.near, ctx.far, 4)
    g = sample_geometry(ctx.extrinsics.to(gpu_device), ctx.intrinsics.to(gpu_device),
                        ctx.near.to(gpu_device), ctx.far.to(gpu_device), (8, 8), 4,
                        w2c=torch.linalg.inv(ctx.extrinsics).to(gpu_device),
                        k_inv=torch.linalg.inv(ctx.intrinsics).to(gpu_device))
    assert torch.equal(g.overlaps.cpu(), ref.segment.overlaps)
    assert torch.equal(g.xy_sample.cpu(), ref.xy_sample)
    assert torch.isfinite(g.rel_disparity).all()


def _golden(name):
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name))
    return {k: (torch.from_numpy(z[k]) if z[k].ndim else z[k].item()) for k in z.files}


@pytest.mark.parametrize("name", ["epipolar_v2.npz", "epipolar_v3.npz"])
def test_fused_attention_vs_reference_golden(gpu_device, name):
    """Golden vectors produced by the REAL reference (sampler, get_depth, depth encoding,
    PreNorm(Attention(x, z=kv)) + residual): fused HIP path vs the reference's numbers."""
    from pixelsplat_amd.epipolar import fused_cross_attention, gather_features, sample_geometry

    g = _golden(name)
    dev = gpu_device
    feat = g["features_in"].to(dev)
    b, v, c, h, w = feat.shape
    s, heads = int(g["num_samples"]), int(g["heads"])
    geo = sample_geometry(g["extrinsics"].to(dev), g["intrinsics"].to(dev), g["near"].to(dev),
                          g["far"].to(dev), (h, w), s,
                          w2c=torch.linalg.inv(g["extrinsics"]).to(dev),
                          k_inv=torch.linalg.inv(g["intrinsics"]).to(dev))
    assert torch.equal(geo.xy_sample.cpu(), g["xy_sample"])
    assert torch.equal(geo.overlaps.cpu(), g["overlaps"])
    fmap = feat.permute(0, 1, 3, 4, 2).contiguous()
    sampled = gather_features(fmap, geo)
    assert (sampled.cpu() - g["sampled"]).abs().max() < 5e-6
    x = feat.permute(0, 1, 3, 4, 2).reshape(-1, 1, c)
    xn = torch.nn.functional.layer_norm(x, (c,), g["attn.norm.weight"].to(dev),
                                        g["attn.norm.bias"].to(dev))
    y, attn = fused_cross_attention(
        xn, fmap, geo, w_q=g["attn.fn.to_q.weight"].to(dev), w_kv=g["attn.fn.to_kv.weight"].to(dev),
        w_out=g["attn.fn.to_out.0.weight"].to(dev), b_out=g["attn.fn.to_out.0.bias"].to(dev),
        heads=heads, depth_w=g["depth_w"].to(dev), depth_b=g["depth_b"].to(dev), octaves=10,
        return_attn=True)
    # Yardstick: the golden also holds the SAME reference modules evaluated in
    # float64.  |ref32 - ref64| is the reference's own float32 noise on these inputs -- its per-sample
    # 3x3 lstsq, amplified by the 2 pi 2^9 gain of the positional encoding
    # -- and the product may not be further from the float64 truth than twice that, per tensor, in the
    # worst element, the 99th percentile and the mean.
    def held_to_reference_noise(what, hip, ref32, ref64, floor):
        hip, ref32, ref64 = (np.asarray(t, np.float64) for t in (hip, ref32, ref64))
        e_hip, e_ref = np.abs(hip - ref64).ravel(), np.abs(ref32 - ref64).ravel()
        for stat, f in (("max", np.max), ("p99", lambda e: np.quantile(e, 0.99)), ("mean", np.mean)):
            assert f(e_hip) <= 2.0 * f(e_ref) + floor, \
                (name, what, stat, float(f(e_hip)), float(f(e_ref)))
        return float(e_hip.max()), float(e_ref.max())

    ov_mask = g["overlaps"].numpy()[..., None]
    # a6 (depth): after the [near, far] clip, as relative disparity -- what the encoding consumes
    rd = held_to_reference_noise("rel_disparity", geo.rel_disparity.cpu().numpy() * ov_mask,
                                 g["rel_disparity"].numpy() * ov_mask,
                                 g["rel_disparity64"].numpy() * ov_mask, 1e-7)
    # a7 / a8: attention weights and the layer's output
    aw = held_to_reference_noise("attn_weights", attn.cpu().numpy(), g["attn_weights"].numpy(),
                                 g["attn_weights64"].numpy(), 1e-6)
    ao = held_to_reference_noise("attn_out", (y + x).cpu().numpy(), g["attn_out"].numpy(),
                                 g["attn_out64"].numpy(), 2e-6)
    print(f"\n[{name}] |hip - ref64| vs |ref32 - ref64| (max): rel_disparity {rd[0]:.2e} / {rd[1]:.2e}, "
          f"attention weights {aw[0]:.2e} / {aw[1]:.2e}, output {ao[0]:.2e} / {ao[1]:.2e}")


@pytest.mark.parametrize("dims", [
    (2, 3, 16, 6, 8, 4, 2, 8),       # view embeddings, T = 8
    (1, 2, 64, 5, 7, 13, 3, 16),     # channel class 64, T = 13 (chunk tail), 3 heads, odd sizes
    (1, 2, 256, 4, 4, 9, 4, 8),      # channel class 256
    (1, 3, 36, 3, 5, 40, 1, 12),     # c = 36 (class 64 with idle lanes), T = 80 (> 64), 1 head
    (1, 2, 128, 9, 9, 32, 4, 32),    # the paper's per-ray shape
    # the PAPER shape, one scene of BASELINE configs[1] (config/model/encoder/epipolar.yaml:
    # c 128, 64x64 rays per view, 32 samples, 4 heads x d_dot 128 = inner 512): R = 8192 rays
    (1, 2, 128, 64, 64, 32, 4, 128),
    # one scene of BASELINE configs[3] (3 context views): S' = 64 kv tokens per ray, view
    # embeddings, R = 12 288 rays
    (1, 3, 128, 64, 64, 32, 4, 128),
    # the paper shape with b = 2 scenes in one launch (VERDICT r2 next #1b): scene-strided
    # feature maps, ray indices beyond one scene, the tile-owner gather over 4 source images
    (2, 2, 128, 64, 64, 32, 4, 128),
    # BASELINE configs[4] (512x512 images): 128x128 feature maps -- the feature-gradient kernel's
    # per-tile ray cull at 4x the tiles and rays (narrow channels keep the CPU side cheap)
    (1, 2, 16, 128, 128, 8, 2, 8),
])
def test_fused_attention_vs_oracle_and_autograd(gpu_device, dims):
    """Same rel_disparity on both sides (so the PE noise amplification drops out): forward to
    1e-5 and every gradient (features, q path, depth-encoding weights, to_kv / to_out)
    against torch autograd through the oracle's unfused restatement."""
    from pixelsplat_amd.epipolar import fused_cross_attention, sample_geometry

    torch.manual_seed(0)
    b, v, c, h, w, s, heads, dh = dims
    ctx = _cams(b, v, 5)
    feat = torch.randn(b, v, c, h, w)
    dev = gpu_device
    geo = sample_geometry(ctx.extrinsics.to(dev), ctx.intrinsics.to(dev), ctx.near.to(dev),
                          ctx.far.to(dev), (h, w), s,
                          w2c=torch.linalg.inv(ctx.extrinsics).to(dev),
                          k_inv=torch.linalg.inv(ctx.intrinsics).to(dev))
    inner = heads * dh
    P = dict(w_q=torch.randn(inner, c) * 0.3, w_kv=torch.randn(2 * inner, c) * 0.3,
             w_out=torch.randn(c, inner) * 0.3, b_out=torch.randn(c) * 0.1,
             depth_w=torch.randn(c, 20) * 0.3, depth_b=torch.randn(c) * 0.1,
             view_emb=torch.randn(v - 1, c) * 0.3)
    if c == 128 and v == 2:  # the 2-view configs have no view embedding (epipolar_transformer.py:126)
        del P["view_emb"]
    x = torch.randn(b * v * h * w, 1, c)
    gout = torch.randn(b * v * h * w, 1, c)

    def run(device, fused):
        leaves = {k: t.clone().to(device).requires_grad_(True) for k, t in P.items()}
        f = feat.clone().to(device).requires_grad_(True)
        xx = x.clone().to(device).requires_grad_(True)
        fmap = f.permute(0, 1, 3, 4, 2).contiguous()
        if fused:
            y = fused_cross_attention(xx, fmap, geo, heads=heads, octaves=10, **leaves)
        else:
            xy = geo.xy_sample.cpu()
            sampled = torch.stack([torch.stack([torch.stack([
                E.gather_features(f[bi, int(E.heterogeneous_index(v)[vi, oi])],
                                  xy[bi, vi, oi].reshape(-1, 2)).reshape(h * w, s, c)
                for oi in range(v - 1)]) for vi in range(v)]) for bi in range(b)])
            sampled = sampled * geo.overlaps.cpu()[..., None, None]
            enc = E.positional_encoding(geo.rel_disparity.cpu(), 10) @ leaves["depth_w"].T \
                + leaves["depth_b"]
            kv = sampled + enc
            if "view_emb" in leaves:
                kv = kv + leaves["view_emb"][None, None, :, None, None, :]
            z = kv.permute(0, 1, 3, 4, 2, 5).reshape(b * v * h * w, s * (v - 1), c)
            q = xx @ leaves["w_q"].T
            k_, v_ = (z @ leaves["w_kv"].T).chunk(2, dim=-1)
            sp = lambda t: t.reshape(t.shape[0], -1, heads, dh).transpose(1, 2)
            a = ((sp(q) @ sp(k_).transpose(-1, -2)) * dh ** -0.5).softmax(-1)
            y = (a @ sp(v_)).transpose(1, 2).reshape(-1, 1, inner) @ leaves["w_out"].T \
                + leaves["b_out"]
        (y * gout.to(device)).sum().backward()
        grads = {k: t.grad.cpu() for k, t in leaves.items()}
        grads["feat"], grads["x"] = f.grad.cpu(), xx.grad.cpu()
        return y.detach().cpu(), grads

    y_ref, g_ref = run("cpu", False)
    y_hip, g_hip = run(dev, True)
    assert (y_hip - y_ref).abs().max() < 2e-5 * max(1.0, y_ref.abs().max().item())
    for k in g_ref:
        scale = g_ref[k].abs().max().item()
        assert (g_hip[k] - g_ref[k]).abs().max() < 1e-4 * max(scale, 1e-3), k


def _lstsq_rel_disparity(ctx, grid, s, v, dtype):
    """Relative disparity of every sample by the REFERENCE's route, restated: `intersect_rays`
    (projection.py:176-230) -- parallel test at 1 - 1e-5, normal equations of the two rays, ONE
    torch.linalg.lstsq per sample in the working precision, 1e10 for parallel pairs -- then
    get_depth's norm (epipolar_lines.py:280-292), the [near, far] clip and depth_to_relative_disparity
    (epipolar_transformer.py:100-118).  Returns (rel [b,v,ov,r,s], xy_sample, overlaps) in `dtype`."""
    ext, intr = ctx.extrinsics.to(dtype), ctx.intrinsics.to(dtype)
    near, far = ctx.near.to(dtype), ctx.far.to(dtype)
    h, w = grid
    b = ext.shape[0]
    smp = E.sample(torch.zeros((b, v, 1, h, w), dtype=dtype), ext, intr, near, far, s, with_depth=False)
    idx = E.heterogeneous_index(v)
    k_inv = torch.linalg.inv(intr)
    o2, d2 = E.world_rays(smp.xy_sample, ext[:, idx][:, :, :, None, None], k_inv[:, idx][:, :, :, None, None])
    o1 = smp.origins[:, :, None, :, None, :].expand_as(o2)
    d1 = smp.directions[:, :, None, :, None, :].expand_as(d2)
    parallel = (d1 * d2).sum(-1) > 1 - 1e-5
    eye = torch.eye(3, dtype=dtype)
    n1 = d1[..., :, None] * d1[..., None, :] - eye
    n2 = d2[..., :, None] * d2[..., None, :] - eye
    rhs = (n1 @ o1[..., None]) + (n2 @ o2[..., None])
    p = torch.linalg.lstsq(n1 + n2, rhs).solution[..., 0]
    p = torch.where(parallel[..., None], torch.full_like(p, 1e10), p)
    depth = (p - o1).norm(dim=-1)
    nr, fr = near[:, :, None, None, None], far[:, :, None, None, None]
    return E.relative_disparity(depth.maximum(nr).minimum(fr), nr, fr), smp.xy_sample, smp.segment.overlaps


@pytest.mark.parametrize("dims", [(2, 3, 16, 6, 8, 4, 2, 8), (1, 2, 128, 9, 9, 32, 4, 32),
                                  (1, 2, 128, 64, 64, 32, 4, 128)])     # the last: the paper shape
def test_fused_attention_vs_lstsq_depth_route(gpu_device, dims):
    """The product's closed-form depth meets the reference's lstsq (the strict test
    above feeds the oracle the product's own rel_disparity).  Here the oracle side computes ITS OWN depths
    by the reference's route -- one torch.linalg.lstsq per sample -- twice: in float32 (what the reference
    does) and in float64 (the truth of the same formulas).  The fused layer's output may not be further
    from the float64 result than twice the float32 lstsq route is (its noise is the reference's own:
    lstsq round-off amplified 2 pi 2^9 times by the depth encoding)."""
    from pixelsplat_amd.epipolar import fused_cross_attention, sample_geometry

    torch.manual_seed(0)
    b, v, c, h, w, s, heads, dh = dims
    ctx = _cams(b, v, 5)
    feat = torch.randn(b, v, c, h, w)
    dev = gpu_device
    inner = heads * dh
    P = dict(w_q=torch.randn(inner, c) * 0.3, w_kv=torch.randn(2 * inner, c) * 0.3,
             w_out=torch.randn(c, inner) * 0.3, b_out=torch.randn(c) * 0.1,
             depth_w=torch.randn(c, 20) * 0.3, depth_b=torch.randn(c) * 0.1)
    if v > 2:
        P["view_emb"] = torch.randn(v - 1, c) * 0.3
    x = torch.randn(b * v * h * w, 1, c)

    def unfused(dtype):
        rel, xy, overlaps = _lstsq_rel_disparity(ctx, (h, w), s, v, dtype)
        f = feat.to(dtype)
        W = {k: t.to(dtype) for k, t in P.items()}
        sampled = torch.stack([torch.stack([torch.stack([
            E.gather_features(f[bi, int(E.heterogeneous_index(v)[vi, oi])],
                              xy[bi, vi, oi].reshape(-1, 2)).reshape(h * w, s, c)
            for oi in range(v - 1)]) for vi in range(v)]) for bi in range(b)])
        kv = sampled * overlaps[..., None, None] + E.positional_encoding(rel, 10) @ W["depth_w"].T + W["depth_b"]
        if "view_emb" in W:
            kv = kv + W["view_emb"][None, None, :, None, None, :]
        z = kv.permute(0, 1, 3, 4, 2, 5).reshape(b * v * h * w, s * (v - 1), c)
        q = x.to(dtype) @ W["w_q"].T
        k_, v_ = (z @ W["w_kv"].T).chunk(2, dim=-1)
        sp = lambda t: t.reshape(t.shape[0], -1, heads, dh).transpose(1, 2)
        a = ((sp(q) @ sp(k_).transpose(-1, -2)) * dh ** -0.5).softmax(-1)
        y = (a @ sp(v_)).transpose(1, 2).reshape(-1, 1, inner) @ W["w_out"].T + W["b_out"]
        return y.double(), a.double(), rel.double(), overlaps

    y32, a32, rel32, ov = unfused(torch.float32)
    y64, a64, rel64, _ = unfused(torch.float64)
    geo = sample_geometry(ctx.extrinsics.to(dev), ctx.intrinsics.to(dev), ctx.near.to(dev),
                          ctx.far.to(dev), (h, w), s, w2c=torch.linalg.inv(ctx.extrinsics).to(dev),
                          k_inv=torch.linalg.inv(ctx.intrinsics).to(dev))
    y_hip, a_hip = fused_cross_attention(x.to(dev), feat.permute(0, 1, 3, 4, 2).contiguous().to(dev), geo,
                                         heads=heads, octaves=10, return_attn=True,
                                         **{k: t.to(dev) for k, t in P.items()})
    y_hip, a_hip = y_hip.cpu().double(), a_hip.cpu().double().reshape(a64.shape)
    m5 = ov[..., None].double()
    stats = {}
    for what, hip, r32, r64, floor in (
            ("rel_disparity", geo.rel_disparity.cpu().double() * m5, rel32 * m5, rel64 * m5, 1e-7),
            ("attention weights", a_hip, a32, a64, 1e-6),
            ("output", y_hip, y32, y64, 2e-5 * max(1.0, float(y64.abs().max())))):
        e_hip, e_ref = (hip - r64).abs().flatten(), (r32 - r64).abs().flatten()
        if e_hip.numel() > 4_000_000:        # (torch.quantile's size limit; a fixed stride keeps it exact enough)
            e_hip_q, e_ref_q = e_hip[::8], e_ref[::8]
        else:
            e_hip_q, e_ref_q = e_hip, e_ref
        for stat, f, a_, b_ in (("max", torch.max, e_hip, e_ref),
                                ("p99", lambda e: torch.quantile(e, 0.99), e_hip_q, e_ref_q),
                                ("mean", torch.mean, e_hip, e_ref)):
            assert float(f(a_)) <= 2.0 * float(f(b_)) + floor, (what, stat, float(f(a_)), float(f(b_)))
        stats[what] = (float(e_hip.max()), float(e_ref.max()))
    print(f"\n[lstsq route {dims}] max |hip - f64| / max |lstsq f32 - f64|: "
          + ", ".join(f"{k} {a_:.2e} / {b_:.2e}" for k, (a_, b_) in stats.items()))


@pytest.mark.parametrize("b,v", [(7, 2), (4, 3)])
def test_batched_launch_equals_single_scene_launches(gpu_device, b, v):
    """The benchmarked launch of (A) -- b = 7 scenes x 2 views (57 344 rays; configs[3]: 4 x 3
    views, 49 152 rays, 64 tokens per ray, view term) at the paper shape -- gives, BIT FOR BIT,
    what b single-scene launches give: the fused forward kernel's output and attention weights,
    and in the backward the per-ray gradient rows and the feature-map gradient (the deterministic
    two-pass gather; `epipolar_tile_order_kernel` and the XCD-ordered slices see 14 / 12 source
    images only in this launch).  The kernels are driven directly with the same per-ray operand
    rows `[q~ | u | e]` (the library GEMMs either side may pick different tilings for different
    row counts and are not part of this comparison; the module-level result is held to 2e-6)."""
    from pixelsplat_amd.epipolar import (FeatureGradBatch, _FusedEpipolarAttention,
                                         fused_cross_attention, sample_geometry)

    torch.manual_seed(1)
    c, h, w, s, heads, dh, octaves = 128, 64, 64, 32, 4, 128, 10
    dev = gpu_device
    ctx = _cams(b, v, 11)
    has_e = v > 2
    lh = _FusedEpipolarAttention.head_width(c, octaves, v - 1, has_e)
    R = v * h * w
    feat = torch.randn(b, v, h, w, c)
    qin = torch.randn(b * R, heads * lh) * 0.5
    gout = torch.randn(b * R, heads * lh)

    def geometry(sl):
        return sample_geometry(ctx.extrinsics[sl].to(dev), ctx.intrinsics[sl].to(dev),
                               ctx.near[sl].to(dev), ctx.far[sl].to(dev), (h, w), s,
                               w2c=torch.linalg.inv(ctx.extrinsics[sl]).to(dev),
                               k_inv=torch.linalg.inv(ctx.intrinsics[sl]).to(dev))

    def run(s0, s1):
        geo = geometry(slice(s0, s1))
        n = s1 - s0
        f = feat[s0:s1].clone().to(dev).reshape(n * v, h, w, c).requires_grad_(True)
        q = qin[s0 * R:s1 * R].clone().to(dev).requires_grad_(True)
        out, attn = _FusedEpipolarAttention.apply(
            (n, v, h, w, s, c, heads, octaves), float(dh) ** -0.5, has_e, f, geo.xy_sample,
            geo.flags, geo.rel_disparity, q, FeatureGradBatch())   # the bench's deferred gather
        (out * gout[s0 * R:s1 * R].to(dev)).sum().backward()
        return out.detach(), attn, f.grad.reshape(n, v, h, w, c), q.grad

    out, attn, df, dq = run(0, b)
    for si in range(b):
        o1, a1, df1, dq1 = run(si, si + 1)
        rs = slice(si * R, (si + 1) * R)
        assert torch.equal(out[rs], o1), ("forward rows", si)
        assert torch.equal(attn[rs], a1), ("attention weights", si)
        assert torch.equal(dq[rs], dq1), ("per-ray gradient rows", si)
        assert torch.equal(df[si], df1[0]), ("feature-map gradient", si)

    # the module-level call (library GEMMs either side of the kernels) at the same shape
    P = dict(w_q=torch.randn(heads * dh, c) * 0.3, w_kv=torch.randn(2 * heads * dh, c) * 0.3,
             w_out=torch.randn(c, heads * dh) * 0.3, b_out=torch.randn(c) * 0.1,
             depth_w=torch.randn(c, 2 * octaves) * 0.3, depth_b=torch.randn(c) * 0.1)
    if has_e:
        P["view_emb"] = torch.randn(v - 1, c) * 0.3
    P = {k: t.to(dev) for k, t in P.items()}
    x = torch.randn(b * R, 1, c, device=dev)
    fm = feat.to(dev)
    y = fused_cross_attention(x, fm, geometry(slice(0, b)), heads=heads, octaves=octaves, **P)
    for si in (0, b - 1):
        y1 = fused_cross_attention(x[si * R:(si + 1) * R], fm[si:si + 1], geometry(slice(si, si + 1)),
                                   heads=heads, octaves=octaves, **P)
        assert (y[si * R:(si + 1) * R] - y1).abs().max() <= 2e-6 * y1.abs().max()


@pytest.mark.parametrize("name,v,octaves", [
    ("transformer_v2.npz", 2, 10), ("transformer_v3.npz", 3, 10),
    # num_octaves = 0: kv = sampled features (+ view embeddings), no depth encoding
    # (epipolar_transformer.py:50,100-121; config/experiment/re10k_ablation_no_depth_encoding.yaml)
    ("transformer_v3_no_depth_encoding.npz", 3, 0)])
def test_epipolar_transformer_module_vs_reference_golden(gpu_device, name, v, octaves):
    """The drop-in EpipolarTransformer (HIP sampler + fused attention) loaded with the
    REFERENCE's weights reproduces the reference's forward output and sampling."""
    from pixelsplat_amd.encoder import (EpipolarTransformer, EpipolarTransformerCfg,
                                        ImageSelfAttentionCfg)

    g = _golden(name)
    cfg = EpipolarTransformerCfg(
        self_attention=ImageSelfAttentionCfg(patch_size=2, num_octaves=4, num_layers=1,
                                             num_heads=2, d_token=16, d_dot=8, d_mlp=32),
        num_octaves=octaves, num_layers=2, num_heads=2, num_samples=4, d_dot=8, d_mlp=32, downscale=2)
    net = EpipolarTransformer(cfg, 16, num_context_views=v)
    assert hasattr(net, "depth_encoding") == (octaves > 0)
    net.load_state_dict({k[3:]: t for k, t in g.items() if k.startswith("sd.")}, strict=True)
    net = net.to(gpu_device)
    dev = gpu_device
    args = [g[k].to(dev) for k in ("features_in", "extrinsics", "intrinsics", "near", "far")]
    out, samp = net(*args, materialize_sampling=True,
                    view_shuffle=g["shuffle"].to(dev) if v > 2 else None)
    assert torch.equal(samp.valid.cpu(), g["valid"])
    # the GPU inverts the camera matrices itself here (torch.linalg.inv on device): xy to 1e-6
    assert (samp.xy_sample.cpu() - g["xy_sample"]).abs().max() < 1e-5
    assert (samp.xy_sample_near.cpu() - g["xy_sample_near"]).abs().max() < 1e-5
    assert (samp.xy_sample_far.cpu() - g["xy_sample_far"]).abs().max() < 1e-5
    assert torch.equal(samp.xy_ray.cpu(), g["xy_ray"])
    assert (samp.features.cpu() - g["sampled"]).abs().max() < 1e-4
    # as the reference's EncoderEpipolar calls it (no materialize_sampling, no hooks): the fused path
    # does not build the sampled features, a reader of `sampling.features` (the visualiser, through the
    # visualization_dump) gets them on first access -- the same bits
    out_l, samp_l = net(*args, view_shuffle=g["shuffle"].to(dev) if v > 2 else None)
    assert torch.equal(out_l, out) and not samp_l.features_materialized
    assert torch.equal(samp_l.features, samp.features) and samp_l.features_materialized
    assert not samp_l.features.requires_grad
    scale = g["out"].abs().max().item()
    assert (out.cpu() - g["out"]).abs().max() < 5e-3 * max(scale, 1.0)
    # the hooked (unfused) fallback gives the same result and exposes the attention weights
    seen = []
    hook = net.transformer.layers[0][0].fn.attend.register_forward_hook(
        lambda m, i, o: seen.append(o))
    out2, _ = net(*args, view_shuffle=g["shuffle"].to(dev) if v > 2 else None)
    assert len(seen) == 1 and seen[0].shape[-1] == 4 * (v - 1)
    assert (out2 - out).abs().max() < 1e-4 * max(scale, 1.0)
    # and it trains: gradients reach every parameter
    out.sum().backward()
    missing = [n for n, p in net.named_parameters() if p.grad is None]
    assert missing == []
    # the side-stream weight folding (forward and, through autograd, backward) gives the same
    # gradients as folding inline on the main stream
    hook.remove()
    g_side = {n: p.grad.clone() for n, p in net.named_parameters()}
    net.zero_grad()
    net.fold_layers = lambda view_emb=None: [None] * len(net.transformer.layers)
    out3, _ = net(*args, view_shuffle=g["shuffle"].to(dev) if v > 2 else None)
    assert torch.equal(out3, out)
    out3.sum().backward()
    for n, p in net.named_parameters():
        ref = p.grad
        assert (g_side[n] - ref).abs().max() <= 1e-5 * max(ref.abs().max().item(), 1e-6), n


@pytest.mark.parametrize("m,n,k", [(512, 128, 4096), (128, 512, 1000), (80, 128, 777), (128, 80, 258),
                                   (4, 4, 1), (130, 260, 513), (576, 128, 57344), (1300, 644, 300),
                                   (128, 592, 12289)])
def test_gemm_tn_splitk(gpu_device, m, n, k):
    """ps_gemm_tn_f32 (fp32 MFMA, split-k, fixed-order reduction) against a float64 product:
    within fp32 accumulation error, and bit-reproducible."""
    from pixelsplat_amd.epipolar import gemm_tn

    gen = torch.Generator().manual_seed(m * 7 + n)
    a = torch.randn(k, m, generator=gen)
    b = torch.randn(k, n, generator=gen)
    ref = (a.double().T @ b.double())
    c1 = gemm_tn(a.to(gpu_device), b.to(gpu_device))
    c2 = gemm_tn(a.to(gpu_device), b.to(gpu_device))
    assert torch.equal(c1, c2)
    err = (c1.cpu().double() - ref).abs().max().item()
    assert err < 2e-6 * k ** 0.5 * 4, err
    # strided operand (a column block of a wider matrix), as autograd hands them over
    wide = torch.randn(k, m + 8, generator=gen).to(gpu_device)
    c3 = gemm_tn(wide[:, 4:4 + m], b.to(gpu_device))
    ref3 = wide[:, 4:4 + m].cpu().double().T @ b.double()
    assert (c3.cpu().double() - ref3).abs().max().item() < 2e-6 * k ** 0.5 * 4
    # the column-sum by-product (the bias gradient when a = dY): same product bit for bit, the
    # sums against float64, reproducible
    c4, s4 = gemm_tn(a.to(gpu_device), b.to(gpu_device), colsum=True)
    c5, s5 = gemm_tn(a.to(gpu_device), b.to(gpu_device), colsum=True)
    assert torch.equal(c4, c1) and torch.equal(s4, s5) and s4.shape == (m,)
    assert (s4.cpu().double() - a.double().sum(0)).abs().max().item() < 2e-6 * k ** 0.5 * 4


@pytest.mark.parametrize("v,c,octaves", [(3, 32, 10), (2, 32, 9), (4, 16, 10)])
def test_attention_kernels_zero_the_head_padding(gpu_device, v, c, octaves):
    """The head stride of the row-of-heads matrices is rounded up to a multiple of 4; the kernels
    fill the 1 ... 3 floats behind each head's last block with zeros themselves (PsEpipolarDesc.
    tail_pad_out / tail_pad_in) so that the caller passes uninitialised matrices to the consuming
    GEMMs.  Both outputs are poisoned with NaN before the launch here."""
    import ctypes as C
    from pixelsplat_amd import _lib
    from pixelsplat_amd.epipolar import _FusedEpipolarAttention as F, sample_geometry

    torch.manual_seed(3)
    lib = _lib.load()
    b, h, w, s, heads = 2, 6, 5, 8, 4
    dev = gpu_device
    ctx = _cams(b, v, 7)
    geo = sample_geometry(ctx.extrinsics.to(dev), ctx.intrinsics.to(dev), ctx.near.to(dev),
                          ctx.far.to(dev), (h, w), s)
    has_e = v > 2
    dims = (b, v, h, w, s, c, heads, octaves)
    d, lh = F._desc(dims, has_e)
    used = c + 2 * octaves + ((v - 1) if has_e else 0)
    assert d.tail_pad_out == lh - used and 1 <= lh - used <= 3
    R, T, P = b * v * h * w, s * (v - 1), 2 * octaves
    fmap = torch.randn(b * v, h, w, c, device=dev)
    qin = torch.randn(R, heads * lh, device=dev)
    out = torch.full((R, heads * lh), float("nan"), device=dev)
    attn = torch.empty((R, heads, T), device=dev)
    p = lambda t, off=0: C.c_void_p(t.data_ptr() + 4 * off)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.ps_epipolar_attention_forward(
        C.byref(d), p(fmap), p(geo.xy_sample), p(geo.flags), p(geo.rel_disparity), p(qin), p(qin, c),
        p(qin, c + P) if has_e else None, C.c_float(0.25), p(out), p(out, c), p(out, c + P), p(attn), st),
        "ps_epipolar_attention_forward")
    dout = torch.randn(R, heads * lh, device=dev)
    dqin = torch.full((R, heads * lh), float("nan"), device=dev)
    ds = torch.empty((R, heads, T), device=dev)
    _lib.check(lib.ps_epipolar_attention_backward(
        C.byref(d), p(fmap), p(geo.xy_sample), p(geo.flags), p(geo.rel_disparity), p(qin), p(attn),
        p(out), p(out, c), p(out, c + P) if has_e else None, p(dout), p(dout, c),
        p(dout, c + P) if has_e else None, C.c_float(0.25), p(dqin), p(dqin, c),
        p(dqin, c + P) if has_e else None, p(ds), None, None, st), "ps_epipolar_attention_backward")
    for m in (out, dqin):
        m3 = m.view(R, heads, lh)
        assert torch.isfinite(m3).all()
        assert (m3[:, :, used:] == 0).all()
        assert m3[:, :, :used].abs().sum() > 0


@pytest.mark.parametrize("v,c", [(2, 128), (3, 32)])
def test_feature_grad_batch_matches_per_layer(gpu_device, v, c):
    """Two chained attention layers on one feature map: the deferred one-pass scatter of both
    layers' feature-map gradients (FeatureGradBatch / ps_epipolar_feature_grad, n_layers = 2)
    equals the sum of the per-layer scatters, for every gradient."""
    from pixelsplat_amd.epipolar import FeatureGradBatch, fused_cross_attention, sample_geometry

    torch.manual_seed(1)
    b, h, w, s, heads, dh = 2, 9, 7, 12, 4, 8
    dev = gpu_device
    ctx = _cams(b, v, 5)
    geo = sample_geometry(ctx.extrinsics.to(dev), ctx.intrinsics.to(dev), ctx.near.to(dev),
                          ctx.far.to(dev), (h, w), s)
    inner = heads * dh
    mk = lambda *shape, sc=0.3: (torch.randn(*shape) * sc).to(dev)
    layers = [dict(w_q=mk(inner, c), w_kv=mk(2 * inner, c), w_out=mk(c, inner), b_out=mk(c, sc=0.1),
                   depth_w=mk(c, 20), depth_b=mk(c, sc=0.1),
                   view_emb=(mk(v - 1, c) if v > 2 else None)) for _ in range(2)]
    feat0 = torch.randn(b, v, h, w, c, device=dev)

    def run(batched):
        feat = feat0.clone().requires_grad_(True)
        leaves = [{k: (t.clone().requires_grad_(True) if t is not None else None)
                   for k, t in lay.items()} for lay in layers]
        batch = FeatureGradBatch() if batched else None
        x = feat.reshape(-1, 1, c)
        for lay in leaves:
            x = fused_cross_attention(torch.tanh(x), feat, geo, heads=heads, octaves=10,
                                      batch=batch, **lay) + x
        x.square().mean().backward()
        grads = [feat.grad] + [t.grad for lay in leaves for t in lay.values() if t is not None]
        return x.detach(), grads

    y0, g0 = run(False)
    y1, g1 = run(True)
    assert torch.equal(y0, y1)
    for a, bb in zip(g0, g1):
        assert (a - bb).abs().max() <= 2e-6 * max(a.abs().max().item(), 1e-6)


def test_invert_cameras(gpu_device):
    """ps_invert_cameras (double-precision cofactors, one launch, no host sync) against
    torch.linalg.inv in float64."""
    import ctypes as C

    from pixelsplat_amd import _lib

    lib = _lib.load()
    ctx = _cams(3, 3, 5)
    c2w = ctx.extrinsics.reshape(-1, 4, 4).contiguous().to(gpu_device)
    k = ctx.intrinsics.reshape(-1, 3, 3).contiguous().to(gpu_device)
    w2c, k_inv = torch.empty_like(c2w), torch.empty_like(k)
    p = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(lib.ps_invert_cameras(C.c_int32(c2w.shape[0]), p(c2w), p(k), p(w2c), p(k_inv),
                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)),
               "ps_invert_cameras")
    ref_w = torch.linalg.inv(c2w.double().cpu())
    ref_k = torch.linalg.inv(k.double().cpu())
    assert (w2c.cpu().double() - ref_w).abs().max() < 1e-6 * ref_w.abs().max()
    assert (k_inv.cpu().double() - ref_k).abs().max() < 1e-6 * ref_k.abs().max()


@pytest.mark.parametrize("heads,dh,c,d,d_out,octaves,ov,bias", [
    (4, 32, 128, 128, 128, 10, 0, True), (2, 8, 20, 12, 16, 3, 2, True), (3, 5, 17, 9, 17, 1, 1, False),
    (4, 128, 128, 128, 128, 10, 0, True),     # the paper's layer: inner = 512 (configs[1])
    (4, 128, 128, 128, 128, 10, 2, True)])    # + view embeddings (configs[3], 3 context views)
def test_fold_weights_kernels_vs_torch(gpu_device, heads, dh, c, d, d_out, octaves, ov, bias):
    """ps_fold_attention_weights (+ backward) against the torch statement of the same algebra
    (fold_attention_weights_torch, itself checked against the unfused attention on the CPU)."""
    from pixelsplat_amd.epipolar import fold_attention_weights, fold_attention_weights_torch

    torch.manual_seed(heads * 100 + c)
    dev = gpu_device
    inner = heads * dh
    mk = lambda *s: (torch.randn(*s, device=dev) * 0.3).requires_grad_(True)
    args = dict(w_q=mk(inner, d), w_kv=mk(2 * inner, c), w_out=mk(d_out, inner),
                b_out=mk(d_out) if bias else None, depth_w=mk(c, 2 * octaves), depth_b=mk(c),
                view_emb=mk(ov, c) if ov else None)
    ref = fold_attention_weights_torch(heads=heads, **args)
    wts = [torch.randn_like(t) for t in ref]
    sum((a * b).sum() for a, b in zip(ref, wts)).backward()
    g_ref = {k: t.grad.clone() for k, t in args.items() if t is not None}
    for t in args.values():
        if t is not None:
            t.grad = None
    out = fold_attention_weights(heads=heads, **args)
    for a, b in zip(out, ref):
        assert a.shape == b.shape
        assert (a - b).abs().max().item() < 2e-6 * max(b.abs().max().item(), 1.0)
    sum((a * b).sum() for a, b in zip(out, wts)).backward()
    for k, t in args.items():
        if t is not None:
            err = (t.grad - g_ref[k]).abs().max().item() / max(g_ref[k].abs().max().item(), 1e-9)
            assert err < 5e-6, f"{k}: {err:.2e}"


@pytest.mark.parametrize("rows,dim", [(57344, 128), (1000, 128), (37, 20), (513, 512), (3, 260)])
def test_layer_norm_kernels_vs_torch(gpu_device, rows, dim):
    """ps_layer_norm_* against torch.nn.functional.layer_norm in float64 (the plain fp32
    reference of this floating-point kernel): 2e-6 on y, 1e-5 on the gradients relative to
    their largest entry; d_gamma / d_beta bit-reproducible run to run."""
    from pixelsplat_amd.epipolar import layer_norm

    torch.manual_seed(rows + dim)
    dev = gpu_device
    norm = torch.nn.LayerNorm(dim).to(dev)
    with torch.no_grad():
        norm.weight.uniform_(0.5, 1.5)
        norm.bias.uniform_(-0.5, 0.5)
    x = (torch.randn(rows, 1, dim, device=dev) * 2 + 0.7).requires_grad_(True)
    w = torch.randn(rows, 1, dim, device=dev)
    y = layer_norm(x, norm)
    (y * w).sum().backward()
    got = (y.detach(), x.grad.clone(), norm.weight.grad.clone(), norm.bias.grad.clone())
    x.grad = norm.weight.grad = norm.bias.grad = None
    y2 = layer_norm(x, norm)
    (y2 * w).sum().backward()
    assert torch.equal(norm.weight.grad, got[2]) and torch.equal(norm.bias.grad, got[3])
    xd = x.detach().double().requires_grad_(True)
    gd, bd = norm.weight.detach().double().requires_grad_(True), norm.bias.detach().double().requires_grad_(True)
    yd = torch.nn.functional.layer_norm(xd, (dim,), gd, bd, norm.eps)
    (yd * w.double()).sum().backward()
    for name, a, b, tol in (("y", got[0], yd.detach(), 2e-6), ("dx", got[1], xd.grad, 1e-5),
                            ("dgamma", got[2], gd.grad, 1e-5), ("dbeta", got[3], bd.grad, 1e-5)):
        err = (a.double() - b).abs().max().item() / max(b.abs().max().item(), 1e-12)
        assert err < tol, f"{name}: {err:.2e}"


def test_bench_gemm_table_keeps_parity(gpu_device):
    """bench.py turns on the committed TunableOp table for the library GEMMs next to the HIP
    kernels (pixelsplat_amd/gemm_tuning).  Same fused layer at the bench's GEMM shapes
    ([57 344 x 128] x [128 x 592] and back) with the table on and off: the library picks other
    kernels, the results must agree to fp32 GEMM round-off, forward and every gradient
    (the benchmarked GEMM kernels were never under a parity test)."""
    from pixelsplat_amd import gemm_tuning
    from pixelsplat_amd.epipolar import fused_cross_attention, sample_geometry

    dev = gpu_device
    torch.manual_seed(3)
    b, v, c, h, w, s, heads, dh = 7, 2, 128, 64, 64, 32, 4, 128
    ctx = _cams(b, v, 11)
    geo = sample_geometry(ctx.extrinsics.to(dev), ctx.intrinsics.to(dev), ctx.near.to(dev),
                          ctx.far.to(dev), (h, w), s)
    inner = heads * dh
    P = dict(w_q=torch.randn(inner, c) * 0.1, w_kv=torch.randn(2 * inner, c) * 0.1,
             w_out=torch.randn(c, inner) * 0.1, b_out=torch.randn(c) * 0.1,
             depth_w=torch.randn(c, 20) * 0.1, depth_b=torch.randn(c) * 0.1)
    feat = torch.randn(b, v, h, w, c, device=dev)
    x = torch.randn(b * v * h * w, 1, c, device=dev)
    gout = torch.randn(b * v * h * w, 1, c, device=dev)

    def run():
        leaves = {k: t.clone().to(dev).requires_grad_(True) for k, t in P.items()}
        f = feat.clone().requires_grad_(True)
        xx = x.clone().requires_grad_(True)
        y = fused_cross_attention(xx, f, geo, heads=heads, octaves=10, **leaves)
        (y * gout).sum().backward()
        return y.detach(), [f.grad, xx.grad] + [t.grad for t in leaves.values()]

    y0, g0 = run()
    on = gemm_tuning.enable()
    try:
        y1, g1 = run()
    finally:
        torch.cuda.tunable.enable(False)
    if not on:
        pytest.skip("the committed TunableOp table was not accepted by this library build")
    assert (y1 - y0).abs().max() <= 2e-5 * max(1.0, y0.abs().max().item())
    for a, bb in zip(g1, g0):
        assert (a - bb).abs().max() <= 2e-5 * max(bb.abs().max().item(), 1e-6)
