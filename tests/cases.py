"""Shared test-case builders (test infrastructure).

`oracle_view_inputs` is the host glue of the reference's per-view rasterizer call
(/root/reference/src/model/decoder/cuda_splatting.py:64-124) written out for the oracle:
scale-invariant renorm, SH [G,3,K]->[G,K,3], cov 3x3 -> upper triangle, transposed matrices.
"""
from __future__ import annotations

import numpy as np
import torch

from pixelsplat_amd.geometry import camera_matrices
from pixelsplat_amd.synthetic import make_workload  # noqa: F401  (re-export)


def oracle_view_inputs(g, cams, b: int, v: int, use_sh: bool = True, bg=None,
                       scale_invariant: bool = True, dtype=np.float32, view_params=None) -> dict:
    """`view_params` (one [48] row of the product's packed per-view block) overrides the
    matrices so the oracle sees bit-identical camera inputs (CPU and GPU `linalg.inv` may
    differ in the last bit)."""
    if view_params is not None:
        vp = np.asarray(view_params, np.float32)
        scale = torch.tensor(vp[40])
        means = g.means[b] * scale
        cov = g.covariances[b] * scale ** 2
        row, col = torch.triu_indices(3, 3)
        out = dict(
            means=means.numpy().astype(dtype), cov6=cov[:, row, col].numpy().astype(dtype),
            opacity=g.opacities[b].numpy().astype(dtype), view=vp[0:16].astype(dtype),
            proj=vp[16:32].astype(dtype), campos=vp[32:35].astype(dtype),
            bg=vp[37:40].astype(dtype), tanfovx=float(vp[35]), tanfovy=float(vp[36]))
        if use_sh:
            d_sh = g.harmonics.shape[-1]
            out["sh"] = g.harmonics[b].permute(0, 2, 1).contiguous().numpy().astype(dtype)
            out["sh_degree"] = int(round(d_sh ** 0.5)) - 1
        return out
    scale = 1 / cams.near[b, v] if scale_invariant else torch.tensor(1.0)
    ext = cams.extrinsics[b, v].clone()
    ext[:3, 3] = ext[:3, 3] * scale
    means = g.means[b] * scale
    cov = g.covariances[b] * scale ** 2
    near = cams.near[b, v] * scale
    far = cams.far[b, v] * scale
    tanfov, view_t, full_t, campos = camera_matrices(
        ext[None], cams.intrinsics[b, v][None], near[None], far[None])
    row, col = torch.triu_indices(3, 3)
    out = dict(
        means=means.numpy().astype(dtype), cov6=cov[:, row, col].numpy().astype(dtype),
        opacity=g.opacities[b].numpy().astype(dtype), view=view_t[0].numpy().astype(dtype),
        proj=full_t[0].numpy().astype(dtype), campos=campos[0].numpy().astype(dtype),
        bg=np.zeros(3, dtype) if bg is None else np.asarray(bg, dtype),
        tanfovx=float(tanfov[0, 0]), tanfovy=float(tanfov[0, 1]))
    if use_sh:
        d_sh = g.harmonics.shape[-1]
        out["sh"] = g.harmonics[b].permute(0, 2, 1).contiguous().numpy().astype(dtype)
        out["sh_degree"] = int(round(d_sh ** 0.5)) - 1
    else:
        out["colors"] = g.harmonics[b][:, :, 0].contiguous().numpy().astype(dtype)
        out["sh_degree"] = 0
    return out


_GOLDEN_DECODER = None


def decoder_golden():
    """tests/golden/decoder.npz: the reference's decoder host glue run unmodified over a recording
    rasterizer stand-in (tests/golden/make_decoder_golden.py)."""
    global _GOLDEN_DECODER
    if _GOLDEN_DECODER is None:
        import os
        _GOLDEN_DECODER = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                               "golden", "decoder.npz"))
    return _GOLDEN_DECODER


def reference_cameras(name: str):
    """(make_workload kwargs, view block [V,48]) of a full-size parity configuration.  The view
    block holds the settings the REFERENCE's render_cuda (cuda_splatting.py:64-110) built for the
    synthetic target cameras of that configuration -- tanfov, transposed view / full-projection
    matrices, campos, bg, 1/near -- so that oracle and product are both fed by the reference's
    host glue and not by the product's own camera kernel."""
    z = decoder_golden()
    b, h, w, v_ctx, v_tgt, seed = (int(x) for x in z[f"cam_{name}_def"])
    return dict(b=b, hw=(h, w), v_ctx=v_ctx, v_tgt=v_tgt, seed=seed), z[f"cam_{name}"].copy()


def small_scene(n: int = 48, hw=(32, 32), seed: int = 0, dtype=np.float64, sh_degree: int = 4,
                opacity_hi: float = 0.6):
    """A few well-conditioned Gaussians in front of one camera (for gradient checks)."""
    rng = np.random.default_rng(seed)
    h, w = hw
    means = np.stack([rng.uniform(-0.8, 0.8, n), rng.uniform(-0.8, 0.8, n),
                      rng.uniform(2.0, 4.0, n)], -1)
    a = rng.normal(size=(n, 3, 3)) * 0.12
    cov = a @ a.transpose(0, 2, 1) + 0.004 * np.eye(3)
    cov6 = np.stack([cov[:, 0, 0], cov[:, 0, 1], cov[:, 0, 2], cov[:, 1, 1], cov[:, 1, 2],
                     cov[:, 2, 2]], -1)
    k = (sh_degree + 1) ** 2
    sh = rng.normal(size=(n, k, 3)) * 0.3
    sh[:, 0] += 0.8
    opacity = rng.uniform(0.1, opacity_hi, n)
    ext = torch.eye(4)[None]
    ext[0, 0, 3] = 0.1
    intr = torch.tensor([[[0.9, 0, 0.5], [0, 0.9, 0.5], [0, 0, 1.0]]])
    tanfov, view_t, full_t, campos = camera_matrices(
        ext, intr, torch.tensor([1.0]), torch.tensor([100.0]))
    return dict(means=means.astype(dtype), cov6=cov6.astype(dtype), opacity=opacity.astype(dtype),
                sh=sh.astype(dtype), sh_degree=sh_degree,
                view=view_t[0].numpy().astype(dtype), proj=full_t[0].numpy().astype(dtype),
                campos=campos[0].numpy().astype(dtype), bg=np.array([0.2, 0.5, 0.7], dtype),
                tanfovx=float(tanfov[0, 0]), tanfovy=float(tanfov[0, 1]), H=h, W=w)
