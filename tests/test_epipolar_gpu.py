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
                                             (1, 2, (64, 64), 32, 2), (2, 2, (9, 7), 5, 3)])
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
    # depth: well-conditioned samples to 1e-4 relative, everything after the [near, far] clip
    d, rd = cpu(g.depth), ref.depths
    ab = (ref.directions[:, :, None, :, None, :] * E.world_rays(
        ref.xy_sample, ctx.extrinsics[:, E.heterogeneous_index(v)][:, :, :, None, None],
        k_inv[:, E.heterogeneous_index(v)][:, :, :, None, None])[1]).sum(-1)
    good = m[..., None] & (ab.abs() < 0.999)
    rel = ((d - rd).abs() / rd.abs().clamp(min=1e-6))[good]
    assert rel.numel() == 0 or rel.max() < 2e-3, rel.max()
    assert rel.numel() == 0 or rel.median() < 1e-5
    nr, fr = ctx.near[:, :, None, None, None], ctx.far[:, :, None, None, None]
    ref_rel = E.relative_disparity(rd.maximum(nr).minimum(fr), nr, fr)
    assert (cpu(g.rel_disparity) - ref_rel)[good].abs().max() < 2e-3


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
