"""Generates tests/golden/loss.npz by running the REFERENCE's LossMse, LossDepth and
compute_psnr (/root/reference/src/loss/loss_mse.py, loss_depth.py,
src/evaluation/metrics.py) on the CPU in the build container (oracle/ref_import.loss_modules).

    python tests/golden/make_loss_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402


def main():
    m = ref_import.loss_modules()
    torch.manual_seed(0)
    b, v, h, w = 2, 3, 19, 23
    out = {}
    color = (torch.rand(b, v, 3, h, w) * 1.3 - 0.15).requires_grad_(True)   # some values outside [0, 1]
    target = torch.rand(b, v, 3, h, w)
    batch = {"target": {"image": target, "near": torch.rand(b, v) + 0.5,
                        "far": torch.rand(b, v) * 50 + 20}}
    pred = m.decoder.DecoderOutput(color=color, depth=None)
    loss = m.mse.LossMse(m.mse.LossMseCfgWrapper(m.mse.LossMseCfg(weight=0.7)))
    val = loss.forward(pred, batch, None, 0)
    val.backward()
    out.update(color=color.detach(), target=target, mse_weight=np.array(0.7), mse=val.detach(),
               mse_grad=color.grad.clone(),
               psnr=m.metrics.compute_psnr(target.flatten(0, 1), color.detach().flatten(0, 1)))
    out.update(near=batch["target"]["near"], far=batch["target"]["far"])
    # depth rendered in "log" mode lives between log(near) and log(far); overshoot both ends
    lo, hi = batch["target"]["near"].log(), batch["target"]["far"].log()
    base = lo[..., None, None] + (hi - lo)[..., None, None] * (torch.rand(b, v, h, w) * 1.2 - 0.1)
    out["depth"] = base
    for tag, sigma, second in (("d1", None, False), ("d2", None, True), ("d1s", 4.0, False),
                               ("d2s", 2.5, True)):
        depth = base.clone().requires_grad_(True)
        pred = m.decoder.DecoderOutput(color=color.detach(), depth=depth)
        cfg = m.depth.LossDepthCfg(weight=0.25, sigma_image=sigma, use_second_derivative=second)
        val = m.depth.LossDepth(m.depth.LossDepthCfgWrapper(cfg)).forward(pred, batch, None, 0)
        val.backward()
        out[tag + "_loss"] = val.detach()
        out[tag + "_grad"] = depth.grad.clone()
        out[tag + "_cfg"] = np.array([0.25, -1.0 if sigma is None else sigma, float(second)])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "loss.npz"),
                        **{k: (t.numpy() if isinstance(t, torch.Tensor) else t) for k, t in out.items()})
    print("wrote loss.npz", {k: tuple(np.shape(t)) for k, t in out.items()})


if __name__ == "__main__":
    main()
