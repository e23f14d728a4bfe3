"""Generates tests/golden/adapter.npz by running the REFERENCE's GaussianAdapter
(/root/reference/src/model/encoder/common/gaussian_adapter.py) on the CPU in the build
container (oracle/ref_import.adapter_modules).  e3nn is absent and unpinned: its two functions
are oracle/adapter_ref.py's restatement (see that file's header) -- the SH rotation part of
these vectors is therefore self-referential, everything else is the reference's own code.

    python tests/golden/make_adapter_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from pixelsplat_amd.synthetic import make_cameras  # noqa: E402


def main():
    m = ref_import.adapter_modules()
    adapter = m.adapter.GaussianAdapter(m.adapter.GaussianAdapterCfg(0.5, 15.0, 4))
    torch.manual_seed(0)
    b, v, h, w, srf, spp = 2, 2, 6, 5, 2, 3
    r = h * w
    ctx, _ = make_cameras(b, v, 4, (64, 64), torch.Generator().manual_seed(0))
    ext = ctx.extrinsics[:, :, None, None, None]
    intr = ctx.intrinsics[:, :, None, None, None]
    leaves = dict(coordinates=torch.rand(b, v, r, srf, 1, 2),
                  depths=torch.rand(b, v, r, srf, spp) * 5 + 0.5,
                  opacities=torch.rand(b, v, r, srf, spp),
                  raw_gaussians=torch.randn(b, v, r, srf, 1, 7 + 75))
    for t in leaves.values():
        t.requires_grad_(True)
    g = adapter.forward(ext, intr, leaves["coordinates"], leaves["depths"], leaves["opacities"],
                        leaves["raw_gaussians"], (h, w))
    outs = dict(means=g.means, covariances=g.covariances, harmonics=g.harmonics,
                opacities_out=g.opacities, scales=g.scales, rotations=g.rotations)
    weights = {k: torch.randn_like(t) for k, t in outs.items()
               if k in ("means", "covariances", "harmonics", "opacities_out")}
    sum((outs[k] * wt).sum() for k, wt in weights.items()).backward()
    out = dict(extrinsics=ctx.extrinsics, intrinsics=ctx.intrinsics, image_shape=np.array([h, w]))
    out.update({k: t.detach() for k, t in leaves.items()})
    out.update({k: t.detach() for k, t in outs.items()})
    out.update({"w_" + k: t for k, t in weights.items()})
    out.update({"grad_" + k: t.grad for k, t in leaves.items()})
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "adapter.npz"),
                        **{k: (t.numpy() if isinstance(t, torch.Tensor) else t) for k, t in out.items()})
    print("wrote adapter.npz", {k: tuple(np.shape(t)) for k, t in out.items()})


if __name__ == "__main__":
    main()
