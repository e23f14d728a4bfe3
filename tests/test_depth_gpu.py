"""GPU parity of the depth-sampler kernels (csrc/depth_sampler.hip) through the C ABI
(sample_depths / DepthPredictorMonocular): against the golden vectors of the REAL reference
module (tests/golden/depth.npz) and against oracle/depth_ref.py at other shapes and at
BASELINE.json configs[1]'s full size.

The bucket index is integer work: it must equal the reference's wherever the uniform draw is
not within 2e-6 of a CDF edge (or, deterministic, the top two probabilities are not within
1e-6) -- there the choice depends on the summation order of an fp32 cumsum, which already
differs between the reference's own CPU and GPU runs.  Rows with the same index: depth and
opacity to 2e-6 relative, gradients 2e-5."""
import os

import numpy as np
import pytest
import torch

from oracle import depth_ref

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "depth.npz")


def load(tag):
    z = np.load(GOLD)
    g = {k[len(tag) + 1:]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith(tag + "_")}
    s, srf, spp, det, trans = (int(x) for x in g["cfg"])
    return g, s, srf, spp, bool(det), bool(trans)


def edge_rows(projected, srf, uniforms, index_ref, tol=2e-6):
    """Rows whose choice is decided within fp32 rounding (excluded from the exact check)."""
    pdf_raw, _ = depth_ref.split_projection(projected.double(), srf)
    pdf = pdf_raw.softmax(-1)
    if uniforms is None:
        top = pdf.topk(min(index_ref.shape[-1] + 1, pdf.shape[-1]), dim=-1).values
        return ((top[..., :-1] - top[..., 1:]).min(-1).values < tol)[..., None].expand_as(index_ref)
    cdf = (pdf / pdf.sum(-1, keepdim=True)).cumsum(-1)
    return ((cdf[..., None, :] - uniforms.double()[..., None]).abs().min(-1).values < tol)


def check_against(ref_depth, ref_opacity, ref_index, depth, opacity, index, edge, rel=2e-6):
    same = index.cpu().long() == ref_index
    assert bool((same | edge).all()), f"{int((~(same | edge)).sum())} index mismatches off the edges"
    assert float(edge.float().mean()) < 0.01
    # depth = 1 / ((1 - rd) (1/near - 1/far) + 1/far): one ulp of rd is ~ depth / near ulps
    # of depth, up to ~1e-5 relative at far = 60 near
    for name, a, b, tol in (("depth", depth, ref_depth, max(rel, 1e-5)),
                            ("opacity", opacity, ref_opacity, rel)):
        err = ((a.cpu() - b).abs() / b.abs().clamp_min(1e-3))[same]
        assert float(err.max()) < tol, f"{name}: {float(err.max()):.2e}"
    return same


@pytest.mark.parametrize("tag", ["train", "det", "srf2"])
def test_sampler_vs_reference_golden(gpu_device, tag):
    from pixelsplat_amd.encoder import sample_depths

    g, s, srf, spp, det, trans = load(tag)
    dev = gpu_device
    if det:
        spp = 1
    projected = g["projected"].to(dev).requires_grad_(True)
    uniforms = None if det else g["uniforms"].to(dev)
    depth, opacity, index = sample_depths(projected, g["near"].to(dev), g["far"].to(dev), srf,
                                          uniforms, spp, trans)
    # index of the reference (not stored by it): recomputed by the pinned oracle
    _, _, ref_index = depth_ref.depth_sampler_forward(g["projected"], g["near"], g["far"], srf,
                                                      None if det else g["uniforms"], trans)
    edge = edge_rows(g["projected"], srf, None if det else g["uniforms"], ref_index)
    # transmittance divides by 1 - cumsum: rounding of the sum order is amplified by 1/den
    same = check_against(g["depth"], g["opacity"], ref_index, depth.detach(), opacity.detach(),
                         index, edge, rel=5e-5 if trans else 2e-6)
    assert bool(same.all())  # the committed vectors have no edge rows; keeps the grads comparable
    # mapped opacity + gradients: the loss of make_depth_golden.py
    exponent = float(g["exponent"])
    projected.grad = None
    depth, mapped, _ = sample_depths(projected, g["near"].to(dev), g["far"].to(dev), srf, uniforms,
                                     spp, trans, opacity_exponent=exponent, opacity_scale=1.0)
    # 1 - (1 - p)^E cancels for small p: absolute 1e-6 on values in [0, 1]
    err = ((mapped.detach().cpu() - g["mapped"]).abs() / (1e-6 + 5e-6 * g["mapped"].abs())).max().item()
    assert err < (20 if trans else 1), f"mapped: {err:.2e} x (1e-6 + 5e-6 |ref|)"
    ((depth * g["w_depth"].to(dev)).sum() + (mapped * g["w_opacity"].to(dev)).sum()).backward()
    gref = g["grad_projected"]
    gerr = (projected.grad.cpu() - gref).abs().max().item() / gref.abs().max().item()
    assert gerr < 2e-5, f"grad_projected: {gerr:.2e}"


def test_module_vs_reference_golden(gpu_device):
    """Whole module (ReLU + Linear on the GPU library GEMM, then the kernels) with the
    reference's weights and the reference's generator stream."""
    from pixelsplat_amd.encoder import DepthPredictorMonocular

    g, s, srf, spp, det, trans = load("train")
    dev = gpu_device
    net = DepthPredictorMonocular(g["features"].shape[-1], s, srf, trans).to(dev)
    net.load_state_dict({"projection.1.weight": g["weight"], "projection.1.bias": g["bias"]})
    features = g["features"].to(dev).requires_grad_(True)
    real_rand = torch.rand
    torch.rand = lambda *a, **k: g["uniforms"].to(dev)   # the numbers the reference drew
    try:
        depth, opacity = net(features, g["near"].to(dev), g["far"].to(dev), False, spp)
    finally:
        torch.rand = real_rand
    assert depth.shape == g["depth"].shape and opacity.shape == g["opacity"].shape
    close = (depth.detach().cpu() - g["depth"]).abs() < 1e-4 * g["depth"].abs()
    assert float(close.float().mean()) > 0.99   # GEMM rounding may flip an edge row
    keep = close.to(dev).float()
    ((depth * g["w_depth"].to(dev) * keep).sum()).backward()
    assert torch.isfinite(features.grad).all()


def test_module_draws_like_the_reference(gpu_device):
    from pixelsplat_amd.encoder import DepthPredictorMonocular

    dev = gpu_device
    net = DepthPredictorMonocular(16, 32, 1, False).to(dev)
    x = torch.randn(1, 2, 50, 16, device=dev)
    near, far = torch.full((1, 2), 0.7, device=dev), torch.full((1, 2), 30.0, device=dev)
    torch.manual_seed(5)
    d1, o1 = net(x, near, far, False, 3)
    torch.manual_seed(5)
    u = torch.rand((1, 2, 50, 1, 3), device=dev)      # the reference's call, same generator state
    proj = net.projection(x)
    d2, o2, _ = depth_ref.depth_sampler_forward(proj.cpu(), near.cpu(), far.cpu(), 1, u.cpu())
    assert float(((d1.cpu() - d2).abs() < 1e-4 * d2).float().mean()) > 0.98
    assert d1.shape == (1, 2, 50, 1, 3)
    dd, od = net(x, near, far, True, 1)
    assert dd.shape == (1, 2, 50, 1, 1)
    with pytest.raises(RuntimeError):
        net(x.cpu(), near.cpu(), far.cpu(), True, 1)


@pytest.mark.parametrize("s,srf,spp,det,trans,exponent", [
    (32, 1, 3, False, False, 0.0), (32, 1, 1, True, False, 2.0), (64, 1, 2, False, True, 0.0),
    (5, 3, 4, False, False, 1.7), (17, 2, 3, True, False, 0.0), (8, 1, 5, False, True, 1.0)])
def test_sampler_vs_oracle_autograd(gpu_device, s, srf, spp, det, trans, exponent):
    from pixelsplat_amd.encoder import sample_depths

    dev = gpu_device
    torch.manual_seed(s * 7 + spp)
    b, v, r = 2, 2, 131
    projected = (torch.randn(b, v, r, 2 * s * srf) * 2.5).requires_grad_(True)
    near = torch.rand(b, v) + 0.3
    far = near + torch.rand(b, v) * 50 + 2
    uniforms = None if det else torch.rand(b, v, r, srf, spp)
    rd, ro, ri = depth_ref.depth_sampler_forward(projected, near, far, srf, uniforms, trans)
    if det and spp > 1:   # the oracle's top-1 generalised: torch.topk order, as the reference
        pdf = depth_ref.split_projection(projected, srf)[0].softmax(-1)
        ri = pdf.topk(spp, dim=-1).indices
        rd, ro, _ = _oracle_at(projected, near, far, srf, ri, trans)
    scale = 1.0 / 3
    rm = (depth_ref.map_pdf_to_opacity(ro, exponent) if exponent else ro) * scale
    pg = projected.detach().to(dev).requires_grad_(True)
    depth, opacity, index = sample_depths(pg, near.to(dev), far.to(dev), srf,
                                          None if det else uniforms.to(dev), spp, trans,
                                          opacity_exponent=exponent, opacity_scale=scale)
    edge = edge_rows(projected.detach(), srf, uniforms, ri)
    if trans:  # transmittance opacities near the tail divide by ~0: compare where well-posed
        sane = (ro.detach().abs() < 10) & torch.isfinite(rm.detach())
    else:
        sane = torch.ones_like(edge)
    same = (index.cpu().long() == ri)
    assert bool((same | edge).all())
    ok = same & sane
    for name, a, ref in (("depth", depth, rd), ("opacity", opacity, rm)):
        err = ((a.detach().cpu() - ref.detach()).abs() / ref.detach().abs().clamp_min(1e-3))[ok]
        tol = 1e-4 if trans else (1e-5 if name == "depth" else 5e-6)
        assert float(err.max()) < tol, f"{name}: {float(err.max()):.2e}"
    wd = torch.randn_like(rd) * ok
    wo = torch.randn_like(rd) * ok
    ((rd * wd).sum() + (torch.where(ok, rm, torch.zeros_like(rm)) * wo).sum()).backward()
    ((depth * wd.to(dev)).sum() + (torch.where(ok.to(dev), opacity, torch.zeros_like(opacity))
                                   * wo.to(dev)).sum()).backward()
    gerr = (pg.grad.cpu() - projected.grad).abs().max().item() / projected.grad.abs().max().item()
    assert gerr < 5e-5, f"grad: {gerr:.2e}"


def _oracle_at(projected, near, far, srf, index, trans):
    """The oracle's arithmetic at given indices (top-k with k > 1)."""
    pdf_raw, offset_raw = depth_ref.split_projection(projected, srf)
    s = pdf_raw.shape[-1]
    pdf = pdf_raw.softmax(-1)
    normalized = pdf / (depth_ref.F32_EPS + pdf.sum(-1, keepdim=True))
    off = offset_raw.sigmoid().gather(-1, index)
    depth = depth_ref.relative_disparity_to_depth((index + off) / s, near[:, :, None, None, None],
                                                  far[:, :, None, None, None])
    if trans:
        partial = pdf.cumsum(-1) - pdf
        op = (pdf / (1 - partial + 1e-10)).gather(-1, index)
    else:
        op = normalized.gather(-1, index)
    return depth, op, index


def test_full_size_vs_oracle(gpu_device):
    """BASELINE.json configs[1]: 14 context views x 65 536 rays x 32 buckets, 3 samples."""
    from pixelsplat_amd.encoder import sample_depths

    dev = gpu_device
    torch.manual_seed(11)
    b, v, r, s, spp = 7, 2, 256 * 256, 32, 3
    projected = torch.randn(b, v, r, 2 * s) * 2
    near = torch.rand(b, v) + 0.5
    far = near + 60
    uniforms = torch.rand(b, v, r, 1, spp)
    rd, ro, ri = depth_ref.depth_sampler_forward(projected, near, far, 1, uniforms)
    depth, opacity, index = sample_depths(projected.to(dev), near.to(dev), far.to(dev), 1,
                                          uniforms.to(dev), spp)
    edge = edge_rows(projected, 1, uniforms, ri)
    check_against(rd, ro, ri, depth, opacity, index, edge)
    # size-independent properties: every depth inside its view's [near, far]; the index in range
    assert int(index.min()) >= 0 and int(index.max()) < s
    lo, hi = near.to(dev)[:, :, None, None, None], far.to(dev)[:, :, None, None, None]
    assert bool(((depth >= lo * (1 - 1e-6)) & (depth <= hi * (1 + 1e-6))).all())
    # deterministic top-1 = argmax of the logits, exactly
    d1, o1, i1 = sample_depths(projected.to(dev), near.to(dev), far.to(dev), 1, None, 1)
    arg = projected.view(b, v, r, s, 2)[..., 0].argmax(-1)
    assert float((i1.view(b, v, r).cpu() == arg).float().mean()) > 0.9999
