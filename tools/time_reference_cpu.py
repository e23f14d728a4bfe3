"""Times the REFERENCE's own path-(A) code -- `EpipolarTransformer.forward` imported unmodified through
oracle/ref_import.py -- on the host cores of THIS container, forward and forward +
backward, at the paper encoder configuration (config/model/encoder/epipolar.yaml): BASELINE configs[0] shape
(b = 1, 2 views, 64 x 64 -> 16 x 16 rays per view) and the configs[1] shape at b = 1 (256 x 256 -> 64 x 64 rays
per view).  SURVEY.md 8(d) "CPU baseline timing" / BASELINE.md 4.2.

The GPU box has no /root/reference, so this cannot be part of bench.py's run there: the result is committed
(profiles/r5_reference_cpu_container.json) and bench.py carries it in `cpu_baseline.reference_in_build_container`,
labelled with the host it was measured on.  (B) has no CPU reference at all (third-party CUDA module, absent).

    python tools/time_reference_cpu.py            (a few minutes)
"""
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import as RI  # noqa: E402
from pixelsplat_amd.synthetic import make_cameras  # noqa: E402


def run(m, hw, b=1, v=2, repeats=5, backward=True):
    cfg = m.transformer.EpipolarTransformerCfg(
        self_attention=m.self_attention.ImageSelfAttentionCfg(patch_size=4, num_octaves=10, num_layers=2,
                                                              num_heads=4, d_token=128, d_dot=128, d_mlp=256),
        num_octaves=10, num_layers=2, num_heads=4, num_samples=32, d_dot=128, d_mlp=256, downscale=4)
    torch.manual_seed(0)
    net = m.transformer.EpipolarTransformer(cfg, 128)
    gen = torch.Generator().manual_seed(0)
    ctx, _ = make_cameras(b, v, 4, hw, gen)
    feat = torch.randn((b, v, 128, *hw), generator=gen, requires_grad=True)

    def fwd():
        out, _ = net(feat, ctx.extrinsics, ctx.intrinsics, ctx.near, ctx.far)
        return out

    def fwd_bwd():
        net.zero_grad(set_to_none=True)
        feat.grad = None
        fwd().square().mean().backward()

    res = {}
    for name, fn in (("forward_no_grad", lambda: torch.no_grad()(fwd)()), ("forward_backward", fwd_bwd)):
        if name == "forward_backward" and not backward:
            continue
        fn()     # warm-up
        ts = []
        for _ in range(repeats):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        res[name + "_s"] = {"median": round(statistics.median(ts), 4), "min": round(min(ts), 4), "runs": repeats}
    return res


def main():
    m = RI.modules(2)
    cpu = "unknown"
    try:
        cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except Exception:
        pass
    out = {
        "what": "the reference's own EpipolarTransformer.forward (sampler + lstsq depth + depth encoding + 2 cross-"
                "attention layers + image self-attention + convolutions), imported unmodified, torch CPU, fp32",
        "where": "the build container (NOT the MI355X box: /root/reference does not exist there)",
        "host": {"cpu": cpu, "nproc": os.cpu_count(), "torch_threads": torch.get_num_threads(),
                 "torch": torch.__version__},
        "configs0_b1_v2_64x64": run(m, (64, 64), repeats=5),
        "configs1_shape_b1_v2_256x256": run(m, (256, 256), repeats=3),
    }
    t = out["configs1_shape_b1_v2_256x256"]["forward_backward_s"]["median"]
    out["configs1_extrapolated"] = {
        "epipolar_transformer_fwd_bwd_s_per_step_b7": round(7 * t, 2),
        "note": "7 scenes x the b = 1 time (the reference's memory footprint at b = 7 does not fit this container)"}
    path = os.path.join(ROOT, "profiles", "r5_reference_cpu_container.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
