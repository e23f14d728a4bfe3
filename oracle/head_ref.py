"""ORACLE (test infrastructure, NOT product code): CPU restatement of the tail of the
reference's EncoderEpipolar.forward (src/model/encoder/encoder_epipolar.py:143-214): features
-> depth predictor -> `to_gaussians` head -> Gaussian adapter -> flattened Gaussians, built
from the other oracles (depth_ref, adapter_ref).

Pinned: tests/test_oracle_head.py checks it against tests/golden/head.npz
(tests/golden/make_head_golden.py: the reference's own modules chained in the build container;
e3nn part self-referential, see adapter_ref.py).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import adapter_ref, depth_ref


def sample_image_grid(h: int, w: int):
    """src/geometry/projection.py:111-140: pixel centres in [0, 1], (x, y) order, [h, w, 2]."""
    ys = (torch.arange(h, dtype=torch.float32) + 0.5) / h
    xs = (torch.arange(w, dtype=torch.float32) + 0.5) / w
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    return torch.stack((gx, gy), dim=-1)


def head_forward(features, context, w_depth, b_depth, w_gauss, b_gauss, *, num_surfaces: int,
                 gaussians_per_pixel: int, uniforms, opacity_exponent: float, scale_min: float,
                 scale_max: float, sh_degree: int, use_transmittance: bool = False):
    """features [b, v, c, h, w]; uniforms [b, v, h*w, srf, spp] or None (deterministic).
    Returns means [b, G, 3], covariances [b, G, 3, 3], harmonics [b, G, 3, d_sh], opacities [b, G]."""
    b, v, c, h, w = features.shape
    srf = num_surfaces
    rows = features.permute(0, 1, 3, 4, 2).reshape(b, v, h * w, c)                 # :143
    projected = F.linear(rows.relu(), w_depth, b_depth)
    depths, densities, _ = depth_ref.depth_sampler_forward(
        projected, context["near"], context["far"], srf, uniforms, use_transmittance)  # :145-151
    xy = sample_image_grid(h, w).reshape(h * w, 1, 2)                              # :154-155
    gaussians = F.linear(rows.relu(), w_gauss, b_gauss).reshape(b, v, h * w, srf, -1)  # :156-160
    offset_xy = gaussians[..., :2].sigmoid()                                       # :161
    pixel_size = 1 / torch.tensor((w, h), dtype=torch.float32)                     # :162
    xy = xy + (offset_xy - 0.5) * pixel_size                                       # :163
    opac = depth_ref.map_pdf_to_opacity(densities, opacity_exponent) / gaussians_per_pixel  # :170
    g = adapter_ref.adapter_forward(
        context["extrinsics"][:, :, None, None, None], context["intrinsics"][:, :, None, None, None],
        xy[..., None, :], depths, opac, gaussians[..., None, 2:], (h, w), scale_min, scale_max,
        sh_degree)                                                                  # :165-173
    n = g.means.shape[1:5].numel()
    return (g.means.reshape(b, n, 3), g.covariances.reshape(b, n, 3, 3),            # :195-214
            g.harmonics.reshape(b, n, 3, -1), g.opacities.reshape(b, n))
