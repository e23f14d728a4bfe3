"""How tight can the threshold-ambiguity mask of the parity tests be?  (tests/test_raster_configs_gpu.py)
For one BASELINE configuration: the HIP image against the oracle image, and for a ladder of
(tol_alpha, tol_T) the number of marked pixels and the worst error on the unmarked ones.
    python tools/ambiguity_sweep.py [c2_256]      (needs a GPU)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import raster_ref as R  # noqa: E402
from tests.cases import make_workload, oracle_view_inputs, reference_cameras  # noqa: E402


def main(name):
    from pixelsplat_amd.decoder import render_cuda
    dev = torch.device("cuda:0")
    kw, vp = reference_cameras(name)
    hw, v = kw["hw"], kw["v_tgt"]
    ctx, tgt, g, _ = make_workload(kw["b"], hw, v_ctx=kw["v_ctx"], v_tgt=v, seed=kw["seed"])
    V = v
    img = render_cuda(tgt.extrinsics.reshape(V, 4, 4).to(dev), tgt.intrinsics.reshape(V, 3, 3).to(dev),
                      tgt.near.reshape(V).to(dev), tgt.far.reshape(V).to(dev), hw,
                      torch.zeros(V, 3, device=dev), g.means.to(dev), g.covariances.to(dev),
                      g.harmonics.to(dev), g.opacities.to(dev), views_per_scene=v,
                      view_params=torch.from_numpy(vp).to(dev)).cpu().numpy()
    sts = [R.forward(H=hw[0], W=hw[1], **oracle_view_inputs(g, tgt, 0, i, view_params=vp[i])) for i in range(V)]
    errs = [np.abs(img[i] - sts[i].image).max(0) for i in range(V)]
    print(f"{name}: pixels {V * hw[0] * hw[1]}, worst error anywhere {max(e.max() for e in errs):.3e}, "
          f"pixels over 1e-4: {sum(int((e > 1e-4).sum()) for e in errs)}")
    for ta, tt in [(0, 0), (1e-7, 1e-7), (3e-7, 3e-7), (1e-6, 1e-6), (3e-6, 3e-6), (1e-5, 1e-5),
                   (2e-5, 1e-4), (1e-4, 1e-3)]:
        marked, worst = 0, 0.0
        for i in range(V):
            m = R.ambiguity_mask(sts[i], tol_alpha=ta, tol_T=tt, tol_power=1e-5) != 0
            marked += int(m.sum())
            worst = max(worst, float(errs[i][~m].max()))
        print(f"  tol_alpha={ta:.0e} tol_T={tt:.0e}: marked {marked:6d}  worst unmarked error {worst:.3e}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "c2_256")
