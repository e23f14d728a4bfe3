"""Pins oracle/loss_ref.py against tests/golden/loss.npz (the REAL reference's LossMse,
LossDepth and compute_psnr run in the build container, tests/golden/make_loss_golden.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import loss_ref

GOLD = os.path.join(os.path.dirname(__file__), "golden", "loss.npz")


def gold():
    return {k: torch.from_numpy(np.asarray(v)) for k, v in np.load(GOLD).items()}


def test_mse_and_psnr_match_reference():
    g = gold()
    color = g["color"].clone().requires_grad_(True)
    val = loss_ref.mse_loss(color, g["target"], float(g["mse_weight"]))
    assert torch.equal(val, g["mse"])
    val.backward()
    assert torch.equal(color.grad, g["mse_grad"])
    assert torch.equal(loss_ref.psnr(g["target"].flatten(0, 1), g["color"].flatten(0, 1)), g["psnr"])


@pytest.mark.parametrize("tag", ["d1", "d2", "d1s", "d2s"])
def test_depth_loss_matches_reference(tag):
    g = gold()
    weight, sigma, second = (float(x) for x in g[tag + "_cfg"])
    depth = g["depth"].clone().requires_grad_(True)
    val = loss_ref.depth_loss(depth, g["near"], g["far"], weight, None if sigma < 0 else sigma,
                              bool(second), g["target"])
    torch.testing.assert_close(val, g[tag + "_loss"], rtol=1e-6, atol=0)
    val.backward()
    torch.testing.assert_close(depth.grad, g[tag + "_grad"], rtol=1e-6, atol=1e-9)
