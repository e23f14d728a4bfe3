"""Pins oracle/depth_ref.py against tests/golden/depth.npz (the REAL reference's
DepthPredictorMonocular run in the build container, tests/golden/make_depth_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import depth_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden", "depth.npz")
CASES = ["train", "det", "srf2"]


def load(tag):
    z = np.load(GOLD)
    g = {k[len(tag) + 1:]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith(tag + "_")}
    s, srf, spp, det, trans = (int(x) for x in g["cfg"])
    return g, s, srf, spp, bool(det), bool(trans)


@pytest.mark.parametrize("tag", CASES)
def test_oracle_matches_reference_outputs_and_gradients(tag):
    g, s, srf, spp, det, trans = load(tag)
    projected = g["projected"].clone().requires_grad_(True)
    depth, opacity, index = depth_ref.depth_sampler_forward(
        projected, g["near"], g["far"], srf, None if det else g["uniforms"], trans)
    assert depth.shape == g["depth"].shape
    assert torch.equal(depth, g["depth"])          # same ops in the same order on the same CPU
    assert torch.equal(opacity, g["opacity"])
    mapped = depth_ref.map_pdf_to_opacity(opacity, float(g["exponent"]))
    torch.testing.assert_close(mapped, g["mapped"], rtol=1e-6, atol=1e-7)
    ((depth * g["w_depth"]).sum() + (mapped * g["w_opacity"]).sum()).backward()
    torch.testing.assert_close(projected.grad, g["grad_projected"], rtol=1e-5, atol=1e-7)
    assert index.dtype == torch.int64 and int(index.max()) < s


def test_projection_split_is_the_reference_rearrange():
    from einops import rearrange
    x = torch.randn(2, 3, 5, 2 * 7 * 3)
    a, b = rearrange(x, "... (dpt srf c) -> c ... srf dpt", c=2, srf=3)
    pa, pb = depth_ref.split_projection(x, 3)
    assert torch.equal(a, pa) and torch.equal(b, pb)


def test_opacity_exponent_schedule():
    assert depth_ref.opacity_exponent(0.0, 0.0, 1, 10) == 1.0
    assert depth_ref.opacity_exponent(0.0, 2.0, 100, 50) == 2.0
    assert depth_ref.opacity_exponent(0.0, 2.0, 100, 500) == 4.0
    p = torch.rand(100)
    torch.testing.assert_close(depth_ref.map_pdf_to_opacity(p, 1.0), p)
