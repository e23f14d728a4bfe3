"""Generates tests/golden/depth.npz by running the REFERENCE's DepthPredictorMonocular
(/root/reference/src/model/encoder/epipolar/depth_predictor_monocular.py) and the encoder's
opacity mapping (encoder_epipolar.py:97-110) on the CPU in the build container.  The uniform
numbers drawn inside sample_discrete_distribution are recorded by wrapping torch.rand.

    python tests/golden/make_depth_golden.py
"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402


def run_case(mod, tag, out, *, b, v, r, d_in, s, srf, spp, deterministic, transmittance, seed):
    torch.manual_seed(seed)
    net = mod.DepthPredictorMonocular(d_in, s, srf, transmittance)
    with torch.no_grad():
        net.projection[1].weight.mul_(6.0)  # peaky distributions as well as flat ones
    features = torch.randn(b, v, r, d_in, requires_grad=True)
    near = torch.rand(b, v) * 0.5 + 0.5
    far = near + torch.rand(b, v) * 20 + 1
    drawn, captured = [], []
    real_rand = torch.rand

    def recording_rand(*a, **k):
        t = real_rand(*a, **k)
        drawn.append(t.clone())
        return t

    def keep_projection(_module, _inputs, output):
        output.retain_grad()
        captured.append(output)

    handle = net.projection.register_forward_hook(keep_projection)
    torch.rand = recording_rand
    try:
        depth, opacity = net.forward(features, near, far, deterministic, 1 if deterministic else spp)
    finally:
        torch.rand = real_rand
        handle.remove()
    projected, = captured
    # transmittance opacities can round above 1, where a fractional power is NaN in the
    # reference too; that case keeps the config's default exponent 2**0
    exponent = 1.0 if transmittance else 2 ** 0.75
    mapped = 0.5 * (1 - (1 - opacity) ** exponent + opacity ** (1 / exponent))  # encoder_epipolar.py:109
    w_d, w_o = torch.randn_like(depth), torch.randn_like(opacity)
    ((depth * w_d).sum() + (mapped * w_o).sum()).backward()
    out.update({
        f"{tag}_weight": net.projection[1].weight.detach(), f"{tag}_bias": net.projection[1].bias.detach(),
        f"{tag}_features": features.detach(), f"{tag}_near": near, f"{tag}_far": far,
        f"{tag}_projected": projected.detach(), f"{tag}_depth": depth.detach(),
        f"{tag}_opacity": opacity.detach(), f"{tag}_mapped": mapped.detach(),
        f"{tag}_w_depth": w_d, f"{tag}_w_opacity": w_o,
        f"{tag}_grad_projected": projected.grad, f"{tag}_grad_features": features.grad,
        f"{tag}_cfg": np.array([s, srf, spp, int(deterministic), int(transmittance)]),
        f"{tag}_exponent": np.array(exponent),
    })
    if drawn:
        assert len(drawn) == 1
        out[f"{tag}_uniforms"] = drawn[0]


def main():
    ref_import.setup(2)
    mod = importlib.import_module("src.model.encoder.epipolar.depth_predictor_monocular")
    out = {}
    run_case(mod, "train", out, b=2, v=2, r=37, d_in=16, s=32, srf=1, spp=3,
             deterministic=False, transmittance=False, seed=1)
    run_case(mod, "det", out, b=1, v=2, r=29, d_in=16, s=32, srf=1, spp=3,
             deterministic=True, transmittance=False, seed=2)
    run_case(mod, "srf2", out, b=1, v=3, r=23, d_in=12, s=12, srf=2, spp=2,
             deterministic=False, transmittance=True, seed=3)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "depth.npz"),
                        **{k: (t.numpy() if isinstance(t, torch.Tensor) else t) for k, t in out.items()})
    print("wrote depth.npz", {k: tuple(np.shape(t)) for k, t in out.items()})


if __name__ == "__main__":
    main()
