"""CPU: the ORACLES chained against the connected golden of the REAL reference (tests/golden/connected.npz,
tests/golden/make_connected_golden.py): from the reference transformer's output on, oracle/head_ref.py (depth
sampler + `to_gaussians` head + adapter) -> oracle/raster_ref.c per target view through the host glue of
tests/cases.oracle_view_inputs -> oracle/loss_ref.py must reproduce the Gaussians, the image and the loss the
reference chain produced, and the oracle rasterizer's backward chained into torch autograd must reproduce the
reference chain's gradient with respect to the transformer output.  Pins the composition of the oracles the GPU
tests compare the product with (the golden's rasterizer IS oracle/raster_ref.c, so the image comparison pins the
host glue around it -- renorm, SH transpose, matrices -- and the head / adapter restatements, not the rasterizer's
arithmetic, which stays unpinned: DESIGN.md 2)."""
import os
from types import SimpleNamespace

import numpy as np
import torch

from oracle import head_ref, raster_ref as R
from tests.cases import oracle_view_inputs

GOLD = os.path.join(os.path.dirname(__file__), "golden", "connected.npz")


class _OracleRaster(torch.autograd.Function):
    """One target view through oracle/raster_ref.c, forward and backward, on the SCALED per-view inputs (what
    the reference's render_cuda hands to the rasterizer)."""

    @staticmethod
    def forward(ctx, means, cov6, opacity, sh_gk3, inp):
        st = R.forward(means=means.detach().numpy(), cov6=cov6.detach().numpy(),
                       opacity=opacity.detach().numpy(), sh=sh_gk3.detach().numpy(), **inp)
        ctx.st = st
        return torch.from_numpy(st.image.copy())

    @staticmethod
    def backward(ctx, d_img):
        g = R.backward(ctx.st, d_img.numpy().astype(np.float32))
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32))
        return f(g["means3D"]), f(g["cov6"]), f(g["opacity"]), f(g["sh"]), None


def test_oracle_chain_reproduces_the_reference_chain():
    g = np.load(GOLD)
    t = lambda k: torch.from_numpy(g[k])
    feats = t("transformer_out").clone().requires_grad_(True)
    ctx = {k: t("ctx_" + k) for k in ("extrinsics", "intrinsics", "near", "far")}
    means, cov, sh, op = head_ref.head_forward(
        feats, ctx, t("sd.dp.projection.1.weight"), t("sd.dp.projection.1.bias"), t("sd.tg.1.weight"),
        t("sd.tg.1.bias"), num_surfaces=1, gaussians_per_pixel=3, uniforms=t("uniforms"), opacity_exponent=1.0,
        scale_min=0.5, scale_max=15.0, sh_degree=4)
    rel = lambda a, b: float((a.detach() - b).abs().max() / b.abs().max())
    assert rel(means, t("g_means")) < 2e-5 and rel(cov, t("g_cov")) < 2e-5 and rel(op, t("g_op")) < 2e-5
    assert rel(sh[:, :2048], t("g_sh_first_2048")) < 2e-5

    tgt = SimpleNamespace(**{k: t("tgt_" + k) for k in ("extrinsics", "intrinsics", "near", "far")})
    gs = SimpleNamespace(means=means.detach(), covariances=cov.detach(), harmonics=sh.detach(), opacities=op.detach())
    b, v = tgt.near.shape
    h, w = g["target"].shape[-2:]
    row, col = torch.triu_indices(3, 3)
    images = []
    for vi in range(v):
        inp = oracle_view_inputs(gs, tgt, 0, vi)          # cuda_splatting.py:64-124 for this view
        scale = 1 / tgt.near[0, vi]
        keep = {k: inp[k] for k in ("view", "proj", "campos", "bg", "tanfovx", "tanfovy", "sh_degree")}
        images.append(_OracleRaster.apply(means[0] * scale, (cov[0] * scale ** 2)[:, row, col], op[0],
                                          sh[0].permute(0, 2, 1).contiguous(), dict(H=h, W=w, **keep)))
    img = torch.stack(images)[None]
    err = (img.detach() - t("image")).abs()
    assert float(err.max()) < 1e-6, float(err.max())        # measured: 0.0 (same rasterizer, same glue, same Gaussians)
    loss = ((img - t("target")) ** 2).mean()          # loss_mse.py:30-31, weight 1
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5 * float(g["loss"])
    loss.backward()
    x, y = feats.grad.double().flatten(), t("grad_transformer_out").double().flatten()
    cos = float(torch.dot(x, y) / (x.norm() * y.norm()))
    l2 = float((x - y).norm() / y.norm())
    print(f"\noracle chain vs reference chain: image p99.9 {float(err.quantile(0.999)):.1e} max {float(err.max()):.1e}, "
          f"d(transformer output): 1 - cos {1 - cos:.1e}, relative L2 {l2:.1e}")
    assert cos > 1 - 1e-9 and l2 < 1e-5, (cos, l2)          # measured: 1 - cos 3e-14, relative L2 1.3e-7
