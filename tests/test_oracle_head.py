"""Pins oracle/head_ref.py (the chain depth predictor -> to_gaussians -> adapter of
encoder_epipolar.py:143-214) against tests/golden/head.npz, generated from the REAL reference
modules (tests/golden/make_head_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import head_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden", "head.npz")
OUTS = ("means", "covariances", "harmonics", "opacities")


def load(tag):
    z = np.load(GOLD)
    return {k[len(tag) + 1:]: torch.from_numpy(np.asarray(z[k])) for k in z.files
            if k.startswith(tag + "_")}


def oracle_run(g, leaves):
    s, srf, gpp, x_map = int(g["cfg"][0]), int(g["cfg"][1]), int(g["cfg"][2]), float(g["cfg"][3])
    ctx = {k: g[k] for k in ("extrinsics", "intrinsics", "near", "far")}
    return head_ref.head_forward(
        leaves["features"], ctx, leaves["dp_weight"], leaves["dp_bias"], leaves["tg_weight"],
        leaves["tg_bias"], num_surfaces=srf, gaussians_per_pixel=gpp, uniforms=g["uniforms"],
        opacity_exponent=2 ** x_map, scale_min=0.5, scale_max=15.0, sh_degree=4)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_head_oracle_matches_reference_chain(tag):
    g = load(tag)
    names = ("features", "dp_weight", "dp_bias", "tg_weight", "tg_bias")
    leaves = {k: g[k].clone().requires_grad_(True) for k in names}
    outs = oracle_run(g, leaves)
    for name, t in zip(OUTS, outs):
        assert t.shape == g[name].shape, name
        torch.testing.assert_close(t, g[name], rtol=2e-5, atol=2e-6, msg=lambda m: f"{name}: {m}")
    sum((t * g["w_" + n]).sum() for n, t in zip(OUTS, outs)).backward()
    for k in names:
        ref = g["grad_" + k] if k == "features" else g[k.replace("_", "_grad_", 1)]
        err = (leaves[k].grad - ref).abs().max().item() / ref.abs().max().item()
        assert err < 2e-5, f"grad {k}: {err:.2e}"
