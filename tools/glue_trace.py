#!/usr/bin/env python3
"""Which library (non-ps::) kernels does one eager (A)+(B) step launch, and from which Python line?

    python tools/glue_trace.py [--context-views 3 --batch 4] > gpurun_out/glue_trace.txt

torch.profiler with Python stacks over three steps of the same two paths bench.py times; every operator
whose device time is not spent in a ps:: kernel or a library GEMM is listed with its input shapes and the operators / autograd nodes that enclose it.
(The hot path's own kernels are timed by bench.py / rocprofv3: this is only the census of what is left
around them.)"""
from __future__ import annotations

import argparse
import collections
import os
import re
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=7)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--context-views", type=int, default=2)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()

    from pixelsplat_amd import _lib, gemm_tuning
    from pixelsplat_amd.decoder import render_cuda
    from pixelsplat_amd.epipolar import FeatureGradBatch
    from pixelsplat_amd.encoder.epipolar_transformer import (EpipolarTransformer, EpipolarTransformerCfg,
                                                             ImageSelfAttentionCfg)
    from pixelsplat_amd.loss import mse_loss
    from pixelsplat_amd.synthetic import make_workload

    _lib.load()
    gemm_tuning.enable()
    dev = torch.device("cuda", 0)
    hw, b, v, vc = (args.size, args.size), args.batch, args.views, args.context_views
    ctx, tgt, g, target = make_workload(b, hw, v_ctx=vc, v_tgt=v, seed=0)
    V = b * v
    means, cov, sh, op = (t.to(dev).requires_grad_(True)
                          for t in (g.means, g.covariances, g.harmonics, g.opacities))
    ext, intr = tgt.extrinsics.reshape(V, 4, 4).to(dev), tgt.intrinsics.reshape(V, 3, 3).to(dev)
    near, far = tgt.near.reshape(V).to(dev), tgt.far.reshape(V).to(dev)
    bg = torch.zeros((V, 3), device=dev)
    tgt_img = target.reshape(V, 3, *hw).to(dev)
    torch.manual_seed(0)
    d_feat, down = 128, 4
    et = EpipolarTransformer(EpipolarTransformerCfg(
        self_attention=ImageSelfAttentionCfg(patch_size=4, num_octaves=10, num_layers=2, num_heads=4,
                                             d_token=128, d_dot=128, d_mlp=256),
        num_octaves=10, num_layers=2, num_heads=4, num_samples=32, d_dot=128, d_mlp=256,
        downscale=down), d_feat, num_context_views=vc).to(dev)
    hA, wA = hw[0] // down, hw[1] // down
    feat = torch.randn(b, vc, hA, wA, d_feat, device=dev).requires_grad_(True)
    shuffle = torch.randperm(vc - 1, device=dev) if vc > 2 else None
    c_ext, c_intr = ctx.extrinsics.to(dev), ctx.intrinsics.to(dev)
    c_near, c_far = ctx.near.to(dev), ctx.far.to(dev)
    a_params = [p for n, p in et.named_parameters()
                if re.match(r"transformer\.layers\.\d+\.0\.|depth_encoding\.|view_embeddings\.", n)]

    def path_a():
        geo = et.epipolar_sampler.geometry(c_ext, c_intr, c_near, c_far, (hA, wA))
        x = feat.reshape(-1, 1, d_feat)
        view_emb = et.view_embeddings(shuffle) if vc > 2 else None
        folds = et.fold_layers(view_emb)
        batch = FeatureGradBatch()
        kv = batch.attach(feat)
        for (attn, _ff), folded in zip(et.transformer.layers, folds):
            x = et.fused_block(attn, x, kv, geo, view_emb=view_emb, folded=folded, batch=batch)
        return x.square().mean()

    def path_b():
        return mse_loss(render_cuda(ext, intr, near, far, hw, bg, means, cov, sh, op, views_per_scene=v),
                        tgt_img, 1.0)

    def step():
        for t in (means, cov, sh, op, feat, *a_params):
            t.grad = None
        path_a().backward()
        path_b().backward()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()

    def dev_us(e):
        for name in ("self_device_time_total", "self_cuda_time_total"):
            if hasattr(e, name):
                return float(getattr(e, name))
        return 0.0

    rows = collections.defaultdict(lambda: [0, 0.0])
    total = 0.0
    for e in prof.events():
        us = dev_us(e)
        if us <= 0 or "ps::" in e.name or "Cijk" in e.name or e.name.startswith(("void ", "hip", "Mem")):
            continue
        chain, p_ = [], e.cpu_parent        # who asked for it: the enclosing operators / autograd nodes
        while p_ is not None and len(chain) < 3:
            chain.append(p_.name.replace("autograd::engine::evaluate_function: ", "node "))
            p_ = p_.cpu_parent
        shapes = [s_ for s_ in (e.input_shapes or []) if s_]
        where = f"{shapes}  <- " + " <- ".join(chain or ["(top level)"])
        rows[(e.name, where)][0] += 1
        rows[(e.name, where)][1] += us
        total += us
    print(f"# operators with device time outside ps:: kernels, {args.steps} eager steps, "
          f"batch {b} x {vc} context views, {args.size}^2")
    print(f"# (includes the library GEMMs: aten::mm / addmm rows)   total {total / args.steps:.1f} us per step")
    print("%-34s %6s %9s  %s" % ("operator", "calls", "us/step", "input shapes <- enclosing operators / autograd nodes"))
    for (key, where), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print("%-34s %6.1f %9.1f  %s" % (key[:34], n / args.steps, us / args.steps, where))


if __name__ == "__main__":
    main()
