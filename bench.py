#!/usr/bin/env python
"""bench.py -- rendered views/s (forward + backward) of the pixelSplat hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched as  python -m torch.distributed.run --nproc-per-node N ... bench.py
One "step" = one pass of the hot path over one synthetic batch already resident in HBM:
BASELINE.json configs[1] -- re10k 2-view, 256x256, batch 7 per GPU -> 7 scenes x 393 216
Gaussians, 28 target views: batched HIP rasterizer forward, MSE loss, backward to
dL/d{means, covariances, harmonics, opacities}.  Weak scaling: every rank renders its own
batch (independent scenes; the path has no data-path collective -- DESIGN.md section (e)).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--batch", type=int, default=7)
    p.add_argument("--size", type=int, default=256)
    p.add_argument("--views", type=int, default=4, help="target views per scene")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-views", type=int, default=16, help="views in the CPU-baseline sample")
    return p.parse_args()


def cpu_baseline(gaussians, tgt, vps_np, hw, n_views, gpu_images, dL):
    """The oracle (CPU port of the same algorithm) timed on the host cores over a bounded
    sample of the same workload: the first `n_views` views of the batch (scene-major),
    forward + backward.  Also the parity figure: L_inf / PSNR of the GPU render against the
    oracle render of the same views."""
    import numpy as np
    from oracle import raster_ref as R
    from tests.cases import oracle_view_inputs

    R.lib()
    R.parallel_backward(True)
    cores = os.cpu_count() or 1
    t_total = 0.0
    linf, mse = 0.0, []
    vps = tgt.near.shape[1]
    for v in range(n_views):
        inp = oracle_view_inputs(gaussians, tgt, v // vps, v % vps, view_params=vps_np[v])
        t0 = time.perf_counter()
        st = R.forward(H=hw[0], W=hw[1], **inp)
        R.backward(st, dL[v])
        t_total += time.perf_counter() - t0
        diff = np.clip(gpu_images[v], 0, 1) - np.clip(st.image, 0, 1)
        linf = max(linf, float(np.abs(gpu_images[v] - st.image).max()))
        mse.append(float((diff ** 2).mean()))
    R.parallel_backward(False)
    m = float(np.mean(mse))
    psnr = float("inf") if m == 0 else -10.0 * float(np.log10(m))
    return dict(value=n_views / t_total, unit="views/s", cores=cores, kind="port",
                sample=f"first {n_views} of {tgt.near.numel()} views of the step (scene-major), "
                       f"{hw[0]}x{hw[1]}, G={gaussians.means.shape[1]}, fwd+bwd, "
                       f"oracle/raster_ref.c with OpenMP on {cores} threads, {t_total:.1f} s"), \
        dict(linf=linf, psnr_db=psnr if psnr != float("inf") else 999.0)


def main():
    args = parse()
    from pixelsplat_amd import parallel as P

    rank, world, local = P.init_from_env()
    dev = torch.device("cuda", local)

    from pixelsplat_amd import _lib
    from pixelsplat_amd.decoder import render_cuda
    from pixelsplat_amd.raster import export_bins
    from pixelsplat_amd.synthetic import make_workload

    lib = _lib.load()  # raises if the HIP library is missing: no fallback
    hw = (args.size, args.size)
    b, v = args.batch, args.views
    ctx, tgt, g, target = make_workload(b, hw, v_ctx=2, v_tgt=v, seed=P.rank_seed(0, rank))
    G = g.means.shape[1]
    V = b * v

    means = g.means.to(dev).requires_grad_(True)
    cov = g.covariances.to(dev).requires_grad_(True)
    sh = g.harmonics.to(dev).requires_grad_(True)
    op = g.opacities.to(dev).requires_grad_(True)
    ext = tgt.extrinsics.reshape(V, 4, 4).to(dev)
    intr = tgt.intrinsics.reshape(V, 3, 3).to(dev)
    near = tgt.near.reshape(V).to(dev)
    far = tgt.far.reshape(V).to(dev)
    bg = torch.zeros((V, 3), device=dev)
    tgt_img = target.reshape(V, 3, *hw).to(dev)

    def step():
        for t in (means, cov, sh, op):
            t.grad = None
        img = render_cuda(ext, intr, near, far, hw, bg, means, cov, sh, op, views_per_scene=v)
        loss = ((img - tgt_img) ** 2).mean()
        loss.backward()
        return img

    # D = sum of tile-list lengths (reported; enters the algorithmic-bytes figure)
    img, aux = render_cuda(ext, intr, near, far, hw, bg, means, cov, sh, op, views_per_scene=v,
                           return_aux=True)
    counts, _, _ = export_bins(aux["cfg"], aux["state"], aux["layout"], aux["point_list"])
    D_total = int(counts.to(torch.int64).sum().item())
    n_visible = int((aux["radii"] > 0).sum().item())
    vps_np = aux["view_params"].cpu().numpy()
    gpu_images = img.detach().cpu().numpy()
    del aux, counts, img
    torch.cuda.empty_cache()

    for _ in range(args.warmup):
        step()
    ng = lib.ps_profile_group_count()
    tot_ms = (C.c_double * ng)()
    launches = (C.c_int64 * ng)()
    P.barrier(world)
    lib.ps_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    P.barrier(world)
    elapsed = time.perf_counter() - t0
    lib.ps_profile_enable(0)
    _lib.check(lib.ps_profile_collect(tot_ms, launches), "ps_profile_collect")
    elapsed = P.max_over_ranks(elapsed, world, dev)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = P.aggregate_throughput(V, args.steps, world, elapsed)
        groups = {lib.ps_profile_group_name(i).decode(): (tot_ms[i] / max(launches[i], 1),
                                                          int(launches[i])) for i in range(ng)}
        # algorithmic bytes per launch (DESIGN.md "kernels"): the reference-algorithm figure of
        # SURVEY.md 8(d), attributed per kernel; one launch covers all V views of the batch.
        P = hw[0] * hw[1]
        alg = {
            "preprocess_forward": 392.0 * G * V,
            "depth_sort": 48.0 * D_total,
            "tile_bins": 0.0,
            "tiles_forward": 36.0 * D_total + 20.0 * P * V,
            "tiles_backward": 76.0 * D_total + 20.0 * P * V,
            "preprocess_backward": 728.0 * G * V,
        }
        dom = max(alg, key=lambda k: groups[k][0])
        dom_ms = groups[dom][0]
        achieved = alg[dom] / (dom_ms * 1e-3) / 1e9
        out = {
            "metric": "rendered views/sec (fwd+bwd)", "value": round(value, 2), "unit": "views/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"re10k 2-view, {hw[0]}x{hw[1]}, batch_size={b} per GPU, {v} target "
                            f"views/scene (BASELINE.json configs[1]): rasterizer fwd+bwd (B)",
                "gaussians_per_scene": G, "views_per_step_per_gpu": V,
                "tile_list_entries_D": D_total, "visible_gaussian_views": n_visible,
                "D_over_GV": round(D_total / (G * V), 3), "parallelism": f"dp{world}",
            },
            "roofline": {
                "bound": "hbm", "kernel": dom, "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None, "avg_kernel_ms": round(dom_ms, 4),
                "algorithmic_bytes_per_launch": alg[dom],
            },
            "kernels_ms": {k: round(groups[k][0], 4) for k in groups},
            "whole_path": {
                "algorithmic_bytes_per_step": (1120.0 * G + 40.0 * P) * V + 160.0 * D_total,
                "hbm_frac": round(((1120.0 * G + 40.0 * P) * V + 160.0 * D_total)
                                  / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            dL = (2.0 * (torch.from_numpy(gpu_images) - target.reshape(V, 3, *hw))
                  / gpu_images.size).numpy()
            nv = min(args.cpu_views, V)
            cb, parity = cpu_baseline(g, tgt, vps_np, hw, nv, gpu_images, dL)
            out["cpu_baseline"] = cb
            out["parity_vs_oracle"] = parity
        print(json.dumps(out), flush=True)
    P.shutdown(world)


if __name__ == "__main__":
    main()
