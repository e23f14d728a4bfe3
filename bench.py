#!/usr/bin/env python
"""bench.py -- rendered views/s (forward + backward) of the pixelSplat hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched as  python -m torch.distributed.run --nproc-per-node N ... bench.py
One "step" = one pass of the hot path over one synthetic batch already resident in HBM,
BASELINE.json configs[1] -- re10k 2-view, 256x256, batch 7 per GPU:
  (A) epipolar sampler + the two epipolar cross-attention layers of the encoder on the
      [7, 2, 128, 64, 64] feature maps (geometry kernel, fused gather/attention kernels,
      the folded weight GEMMs), forward and backward to features and weights;
  (B) 7 scenes x 393 216 Gaussians -> 28 target views: batched HIP rasterizer forward, MSE
      loss, backward to dL/d{means, covariances, harmonics, opacities}.
value = 28 views / time of (A)+(B); (B) alone and (A) alone are timed after the contract's
timed region and reported next to it.  Weak scaling: every rank processes its own batch
(independent scenes; no data-path collective -- DESIGN.md section (e)).

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
# (size, batch, target views, context views) -> which BASELINE.json config the run is
BASELINE_TAG = {(256, 7, 4, 2): " (BASELINE.json configs[1])", (64, 1, 4, 2): " (BASELINE.json configs[0])",
                (256, 4, 4, 3): " (BASELINE.json configs[3])", (512, 2, 4, 2): " (BASELINE.json configs[4])"}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--batch", type=int, default=7)
    p.add_argument("--size", type=int, default=256)
    p.add_argument("--views", type=int, default=4, help="target views per scene")
    p.add_argument("--context-views", type=int, default=2,
                   help="context views per scene (3: BASELINE configs[3], acid 3-view)")
    p.add_argument("--scene", choices=["survey", "dense", "opaque", "large"], default="survey",
                   help="Gaussian distribution (pixelsplat_amd/synthetic.py SCENES): survey = SURVEY 8d's recipe "
                        "(the BASELINE workload); dense = >= 80 %% of the pairs in frame, D/G >= 2; opaque = opacity "
                        "U(0.5, 1), early termination; large = scales x 3, the > 4-tile atomic path")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-probes", action="store_true",
                   help="skip the SURVEY 8(f) side probes (adapter / depth / head chain) after the timed "
                        "region: for profiling the step alone")
    p.add_argument("--grad-payload-mb", type=float, default=0.0,
                   help="N > 1: extra synthetic fp32 gradient payload all-reduced per step, to model "
                        "the rest of the network (the reference reduces ~480 MB; SURVEY.md 5)")
    p.add_argument("--bucket-mb", type=float, default=25.0, help="gradient bucket size (torch DDP default)")
    p.add_argument("--launch", choices=["auto", "graph", "eager"], default="auto",
                   help="graph / auto: the step is replayed from two hipGraphs ((A) fwd+bwd, (B) fwd+bwd; "
                        "fixed-capacity tile lists, no host sync), the gradient all-reduce launched "
                        "between them -- for every N, so that a scaling curve compares like with like "
                        "(capture runs in thread_local error mode next to a live RCCL communicator; "
                        "if it fails the run falls back to eager launches and reports it); eager: "
                        "launch by launch.  `paths.eager_ms_per_step` carries the eager time of the "
                        "same step at every N either way")
    p.add_argument("--connected", action="store_true",
                   help="after the timed region also time ONE CONNECTED training step -- features -> "
                        "EpipolarTransformer.forward (full module) -> EncoderEpipolarHead -> DecoderSplattingCUDA -> "
                        "LossMse -> backward (pixelsplat_amd/training_step.py; model_wrapper.py:108-152) -- "
                        "eager, reported as paths.connected_ms_per_step; not part of `value`")
    p.add_argument("--scaling-sweep", type=str, default=None, metavar="1,2,4,8",
                   help="convenience for an 8-GPU node: run this command once per N (each as its own `--gpus N` job, "
                        "back to back), print each job's JSON line, then ONE summary line with "
                        "weak_scaling_efficiency[N] = value(N) / (N x value(1)).  The contract's single line per run is "
                        "what each job prints; the driver computes efficiency itself from those")
    p.add_argument("--cpu-views", type=int, default=10 ** 6,
                   help="views in the CPU-baseline sample / parity block (default: every view of the step)")
    return p.parse_args()


# profile groups whose event time is the duration of these kernels (substrings of their names)
SINGLE_KERNEL_GROUPS = {
    "tiles_backward": ["tiles_backward_kernel"], "tiles_forward": ["tiles_forward_rows_kernel"],
    "epipolar_attention_forward": ["epipolar_attn_forward_kernel"],
    "epipolar_attention_backward": ["epipolar_attn_backward_kernel"],
    "epipolar_feature_grad": ["epipolar_token_grad_kernel", "epipolar_dfmap_gather_kernel",
                              "epipolar_dfmap_kernel", "epipolar_dfmap_list_gather_kernel",
                              "epipolar_bin_count_kernel", "epipolar_bin_scan_kernel",
                              "epipolar_bin_offsets_kernel", "epipolar_bin_fill_kernel"],
    "gaussian_adapter_backward": ["adapter_backward_kernel<4, 0>"],
    "depth_sampler_forward": ["depth_sampler_forward_kernel"],
    "depth_sampler_backward": ["depth_sampler_backward_kernel"],
}
PMC_TAG = None    # "c2" | "c4" | "c5": which committed counter summaries match this run


# translation unit behind each single-kernel group: a committed counter summary is paired with a
# live kernel time only if that unit has the same hash in the summary and in the loaded library
GROUP_UNIT = {"tiles_backward": "raster_tiles", "tiles_forward": "raster_cells",
              "epipolar_attention_forward": "epipolar_attention",
              "epipolar_attention_backward": "epipolar_attention",
              "epipolar_feature_grad": "epipolar_attention",
              "gaussian_adapter_backward": "gaussian_adapter",
              "depth_sampler_forward": "depth_sampler", "depth_sampler_backward": "depth_sampler"}
_LIB_HASHES = {}


def _unit_hashes(text):
    return dict(tok.split(":", 1) for tok in (text or "").split() if ":" in tok)


def _pmc_file(kind, group):
    """Newest committed `profiles/*_<tag>_pmc_<kind>.json` whose build stamp carries the SAME hash
    for the group's translation unit as the loaded library (ps_build_info); None (-> null in the
    line) when no summary matches: stale counters are never paired with new times."""
    import glob
    if PMC_TAG is None:
        return None
    unit = GROUP_UNIT.get(group)
    want = _LIB_HASHES.get(unit)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{PMC_TAG}_pmc_{kind}.json")),
                       key=os.path.getmtime, reverse=True):
        with open(path) as f:
            stamp = json.load(f).get("build")
        if want is not None and _unit_hashes(stamp).get(unit) == want:
            return path
    return None


def pmc_traffic(group):
    """HBM-side bytes per launch of the kernel(s) behind a profile group, from the committed
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE summary of the SAME workload AND the same code
    (tools/pmc_summary.py; counters cannot be read from inside the process).  None otherwise."""
    keys, path = SINGLE_KERNEL_GROUPS.get(group), _pmc_file("traffic", group)
    if keys is None or path is None:
        return None, None
    with open(path) as f:
        doc = json.load(f)
    total = sum(float(v["bytes"]) for name, v in doc["kernels"].items() if any(k in name for k in keys))
    src = f"{os.path.basename(path)} (git {str(doc.get('git'))[:12]}, {GROUP_UNIT.get(group)}:" \
          f"{_LIB_HASHES.get(GROUP_UNIT.get(group))} = loaded library)"
    return (total, src) if total else (None, None)


def pmc_valu_busy_ms(group):
    """VALU-busy time per launch (SQ_ACTIVE_INST_VALU quad-cycles x 4 / 1024 SIMDs at 2.4 GHz) of
    the kernel(s) behind a profile group, from the committed PMC summary (tools/pmc_sq_summary.py)."""
    keys, path = SINGLE_KERNEL_GROUPS.get(group), _pmc_file("sq", group)
    if keys is None or path is None:
        return None
    with open(path) as f:
        kernels = json.load(f)["kernels"]
    tot = sum(v.get("valu_busy_ms_at_2.4GHz", 0.0) for name, v in kernels.items()
              if any(k in name for k in keys))
    return tot or None


def pmc_sq_counter(group, counter):
    """Sum of one SQ counter over the kernel(s) behind a profile group, per launch, from the committed PMC
    summary of the same workload and the same code (None otherwise)."""
    keys, path = SINGLE_KERNEL_GROUPS.get(group), _pmc_file("sq", group)
    if keys is None or path is None:
        return None
    with open(path) as f:
        kernels = json.load(f)["kernels"]
    tot = sum(float(v.get(counter, 0.0)) for name, v in kernels.items() if any(k in name for k in keys))
    return tot or None


def cpu_baseline(gaussians, tgt, vps_np, hw, n_views, gpu, dL, grad_fn):
    """The oracle (CPU port of the same algorithm) timed on the host cores over a bounded
    sample of the same workload: the first `n_views` views of the batch (scene-major; default:
    all of them), forward + backward.  Also the parity block of the line, over the SAME launch the
    timed region runs (all scenes in one call): bins, image, final_T, n_contrib per view, the
    constructive check of the threshold pixels, and the four gradient tensors per scene.

    `gpu`: dict(images, final_T, n_contrib, counts, offsets, plist, radii) of the product's
    forward as numpy arrays; `grad_fn(dL)` runs the product's backward for that dL/dimage and
    returns (d_means, d_cov, d_sh, d_opacity) per scene."""
    import numpy as np
    from oracle import raster_ref as R
    from tests.cases import oracle_view_inputs

    R.lib()
    R.parallel_backward(True)
    cores = os.cpu_count() or 1
    t_total = 0.0
    linf_all, linf_ok, linf_T, mse, n_over, n_marked, n_pix = 0.0, 0.0, 0.0, [], 0, 0, 0
    nc_mismatch = bins_mismatch = radii_mismatch = 0
    explain = dict(same=0, flipped=0, unexplained=0, exhausted=0)
    pairs_eval = pairs_contrib = quad_entries = quad_pairs = cell_pairs = cell_row_steps = 0
    vps = tgt.near.shape[1]
    G = gaussians.means.shape[1]
    n_scenes = (n_views + vps - 1) // vps
    ref = dict(means=np.zeros((n_scenes, G, 3)), cov=np.zeros((n_scenes, G, 3, 3)),
               sh=np.zeros((n_scenes, G, 3, gaussians.harmonics.shape[-1])), op=np.zeros((n_scenes, G)))
    row, col = np.triu_indices(3)
    dL = dL.copy()
    for v in range(n_views):
        inp = oracle_view_inputs(gaussians, tgt, v // vps, v % vps, view_params=vps_np[v])
        t0 = time.perf_counter()
        st = R.forward(H=hw[0], W=hw[1], **inp)
        t_total += time.perf_counter() - t0
        # ---- parity of this view (not timed) ----
        ev_, co_ = R.blend_stats(st)
        pairs_eval += ev_
        pairs_contrib += co_
        qe_, qp_ = R.quadrant_evaluations(st)      # what the tile kernels' 8x8 quadrant cull evaluates
        quad_entries += qe_
        quad_pairs += qp_
        c4_ = R.box_evaluations(st, 4)             # the forward's 4x4 cells, four 16-lane rows per wave
        cell_pairs += c4_["pairs"]
        cell_row_steps += c4_["row_steps"]
        img = gpu["images"][v]
        radii_mismatch += int((gpu["radii"][v] != st.radii).sum())
        cnt = (st.ranges[:, 1] - st.ranges[:, 0]).astype(np.int64)
        o0 = int(gpu["offsets"][v, 0])
        same_bins = (np.array_equal(gpu["counts"][v], cnt)
                     and np.array_equal(gpu["plist"][o0:o0 + st.num_rendered], st.point_list))
        bins_mismatch += 0 if same_bins else 1
        diff = np.clip(img, 0, 1) - np.clip(st.image, 0, 1)
        err = np.abs(img - st.image).max(0)
        amb = R.ambiguity_mask(st) != 0     # pixels sitting on alpha = 1/255 or T = 1e-4
        linf_all = max(linf_all, float(err.max()))
        linf_ok = max(linf_ok, float(err[~amb].max()))
        linf_T = max(linf_T, float(np.abs(gpu["final_T"][v] - st.final_T.reshape(hw))[~amb].max()))
        nc_mismatch += int((gpu["n_contrib"][v][~amb] != st.n_contrib.reshape(hw)[~amb]).sum())
        n_over += int((err[~amb] > 1e-4).sum())
        n_marked += int(amb.sum())
        n_pix += err.size
        mse.append(float((diff ** 2).mean()))
        ex = R.explain_threshold_pixels(st, img, gpu["final_T"][v], gpu["n_contrib"][v], amb, tol=1e-4)
        for k in explain:
            explain[k] += ex[k]
        dL[v][:, amb] = 0.0
        # ---- backward (timed) ----
        t0 = time.perf_counter()
        gr = R.backward(st, dL[v])
        t_total += time.perf_counter() - t0
        si, scale = v // vps, float(vps_np[v, 40])
        ref["means"][si] += gr["means3D"] * scale
        cg = np.zeros((G, 3, 3))
        cg[:, row, col] = gr["cov6"]
        ref["cov"][si] += cg * scale ** 2
        ref["sh"][si] += gr["sh"].transpose(0, 2, 1)
        ref["op"][si] += gr["opacity"]
    R.parallel_backward(False)
    # gradients of every sampled scene whose views were all sampled
    grads = {}
    full_scenes = n_views // vps
    if full_scenes:
        got = dict(zip(("means", "cov", "sh", "op"), grad_fn(dL)))
        for k, r in ref.items():
            worst = 0.0
            for si in range(full_scenes):
                a = got[k][si].astype(np.float64)
                worst = max(worst, float(np.abs(a - r[si]).max() / max(np.abs(r[si]).max(), 1e-30)))
            grads[k] = worst
    m = float(np.mean(mse))
    psnr = float("inf") if m == 0 else -10.0 * float(np.log10(m))
    return dict(value=n_views / t_total, unit="views/s", cores=cores, kind="port",
                sample=f"first {n_views} of {tgt.near.numel()} views of the step (scene-major), "
                       f"{hw[0]}x{hw[1]}, G={gaussians.means.shape[1]}, fwd+bwd, "
                       f"oracle/raster_ref.c with OpenMP on {cores} threads, {t_total:.1f} s"), \
        dict(views=n_views, pairs_evaluated_by_reference=pairs_eval,
             pairs_contributing=pairs_contrib, quadrant_entries=quad_entries, quadrant_pairs=quad_pairs,
             cell_pairs=cell_pairs, cell_row_steps=cell_row_steps), \
        dict(views_compared=n_views, launch="the benchmarked one: all scenes of the batch in one call",
             linf=linf_ok, linf_final_T=linf_T, psnr_db=psnr if psnr != float("inf") else 999.0,
             pixels_compared=n_pix, pixels_over_1e_4=n_over, pixels_on_a_threshold=n_marked,
             linf_including_threshold_pixels=linf_all,
             threshold_pixels_explained_without_flip=explain["same"],
             threshold_pixels_explained_by_flipped_flagged_decisions=explain["flipped"],
             threshold_pixels_unexplained=explain["unexplained"] + explain["exhausted"],
             n_contrib_mismatches_off_threshold=nc_mismatch,
             views_with_bin_mismatch=bins_mismatch, radii_mismatches=radii_mismatch,
             gradient_scenes_compared=full_scenes,
             gradient_err_over_max={k: float(f"{e:.3g}") for k, e in grads.items()},
             note="linf / pixels_over_1e_4 / n_contrib are over the pixels that do not sit on one of the "
                  "blend's hard thresholds (oracle.raster_ref.ambiguity_mask: alpha within 3e-6 of 1/255, "
                  "T(1-alpha) within 3e-6 of 1e-4); on those a last-bit difference of exp() flips a "
                  "branch and adds or removes one minimum-alpha contribution: every such pixel is "
                  "replayed with its flagged decisions flipped and must reproduce the product's "
                  "colour / final_T to 1e-4 and its n_contrib exactly (threshold_pixels_*); gradients: "
                  "worst |product - oracle| / max|oracle| per tensor over the scenes, dL/dimage = the "
                  "step's MSE gradient zeroed on the threshold pixels; PSNR is over all pixels")


def cpu_baseline_epipolar(et, feat_nhwc, ctx, num_samples, heads, view_shuffle=None):
    """Path (A) on the host: the oracle's unfused restatement (materialised kv, to_kv on every
    token -- what the reference computes) for ONE scene of the batch, forward + backward
    through torch autograd on the CPU.  Returns seconds per scene."""
    from oracle import epipolar_ref as E

    f = feat_nhwc[:1].detach().cpu().permute(0, 1, 4, 2, 3).contiguous().requires_grad_(True)
    ext, intr = ctx.extrinsics[:1], ctx.intrinsics[:1]
    near, far = ctx.near[:1], ctx.far[:1]
    b, v, c, h, w = f.shape
    par = {k: t.detach().cpu().requires_grad_(True) for k, t in et.state_dict().items()
           if k.startswith(("transformer.layers", "depth_encoding", "view_embeddings"))
           and "self_attention" not in k}
    t0 = time.perf_counter()
    smp = E.sample(f, ext, intr, near, far, num_samples)
    nf = (near[:, :, None, None, None], far[:, :, None, None, None])
    rd = E.relative_disparity(smp.depths.clamp(nf[0], nf[1]), nf[0], nf[1])
    enc = E.positional_encoding(rd, 10) @ par["depth_encoding.1.weight"].T \
        + par["depth_encoding.1.bias"]
    kv = smp.features + enc
    if v > 2:   # epipolar_transformer.py:126-131: one embedding per other view, shuffled
        emb = par["view_embeddings.weight"][view_shuffle.cpu()]
        kv = kv + emb[None, None, :, None, None, :]
    kv = kv.permute(0, 1, 3, 4, 2, 5).reshape(b * v * h * w, -1, c)
    x = f.permute(0, 1, 3, 4, 2).reshape(b * v * h * w, 1, c)
    for i in range(2):
        pre = f"transformer.layers.{i}.0."
        x, _ = E.attention_layer(x, kv, par[pre + "norm.weight"], par[pre + "norm.bias"],
                                 par[pre + "fn.to_q.weight"], par[pre + "fn.to_kv.weight"],
                                 par[pre + "fn.to_out.0.weight"], par[pre + "fn.to_out.0.bias"],
                                 heads)
    x.square().mean().backward()
    return time.perf_counter() - t0


def scaling_sweep(args) -> int:
    """`--scaling-sweep 1,2,4,8`: one `bench.py --gpus N` job per N with the rest of the command line unchanged
    (the CPU baseline only in the N = 1 job, as the contract says), each job's line echoed, then a summary."""
    import subprocess
    ns = [int(x) for x in args.scaling_sweep.split(",") if x.strip()]
    rest, skip = [], False
    for a in sys.argv[1:]:
        if skip:
            skip = False
            continue
        if a in ("--scaling-sweep", "--gpus"):
            skip = True
            continue
        if a.startswith("--scaling-sweep=") or a.startswith("--gpus="):
            continue
        rest.append(a)
    lines = {}
    for n in ns:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(n), *rest]
        if n != 1 and "--no-cpu-baseline" not in rest:
            cmd.append("--no-cpu-baseline")
        r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
        last = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        rec = json.loads(last[-1]) if last else {"value": 0.0, "n_gpus": n, "error": f"no line, exit code {r.returncode}"}
        lines[n] = rec
        print(json.dumps(rec), flush=True)
    base = lines.get(1, {}).get("value") or 0.0
    summary = {
        "metric": "weak scaling of rendered views/sec (fwd+bwd)", "unit": "views/s",
        "values": {str(n): lines[n].get("value") for n in ns},
        "ms_per_step": {str(n): lines[n].get("ms_per_step") for n in ns},
        "weak_scaling_efficiency": {str(n): (round(lines[n].get("value", 0.0) / (n * base), 4) if base else None)
                                    for n in ns},
        "launch": {str(n): lines[n].get("launch") for n in ns},
        "step_check_ok": {str(n): (lines[n].get("step_check") or {}).get("ok") for n in ns},
        "exposed_allreduce_ms_per_step": {str(n): (lines[n].get("comm") or {}).get("exposed_ms_per_step") for n in ns},
        "errors": {str(n): lines[n]["error"] for n in ns if "error" in lines[n]} or None}
    print(json.dumps(summary), flush=True)
    return 0 if not summary["errors"] else 1


def main():
    args = parse()
    if args.scaling_sweep:
        sys.exit(scaling_sweep(args))
    from pixelsplat_amd import parallel as P

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become N ranks of this node (the same
        # torch.distributed.run command line the driver uses), one process per GPU over RCCL
        sys.exit(P.launch_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    # stdout carries the ONE JSON line and nothing else: native libraries write to file descriptor 1
    # behind Python's back (RCCL prints a three-line version banner when a communicator is built), so
    # fd 1 is pointed at stderr for the run and the line goes to a private duplicate of the real one
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    _STATE["json_out"] = json_out
    os.dup2(2, 1)
    rank, world, local = P.init_from_env()
    if world != args.gpus and rank == 0:
        print(f"[bench] --gpus {args.gpus} but the launcher started {world} rank(s); "
              f"reporting n_gpus = {world}", file=sys.stderr)
    dev = torch.device("cuda", local)

    from pixelsplat_amd import _lib, gemm_tuning
    from pixelsplat_amd.decoder import render_cuda
    from pixelsplat_amd.loss import mse_loss
    from pixelsplat_amd.raster import export_bins
    from pixelsplat_amd.synthetic import make_workload

    lib = _lib.load()  # raises if the HIP library is missing: no fallback
    build_info = lib.ps_build_info().decode()
    _LIB_HASHES.update(_unit_hashes(build_info.split("|", 1)[-1]))
    # library GEMMs: committed TunableOp table, look-up only (pixelsplat_amd/gemm_tuning)
    tuned_gemms = (not os.environ.get("PIXELSPLAT_NO_TUNED_GEMMS")) and gemm_tuning.enable()
    hw = (args.size, args.size)
    b, v = args.batch, args.views
    vc = args.context_views
    ctx, tgt, g, target = make_workload(b, hw, v_ctx=vc, v_tgt=v, seed=P.rank_seed(0, rank),
                                        scene=args.scene)
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

    # ---- path (A): paper encoder config (config/model/encoder/epipolar.yaml) ----
    from pixelsplat_amd.epipolar import FeatureGradBatch
    from pixelsplat_amd.encoder.epipolar_transformer import (EpipolarTransformer,
                                                             EpipolarTransformerCfg,
                                                             ImageSelfAttentionCfg)
    torch.manual_seed(0)    # data parallel: identical weights on every rank, different batches
    d_feat, down, n_samp, heads = 128, 4, 32, 4
    et = EpipolarTransformer(EpipolarTransformerCfg(
        self_attention=ImageSelfAttentionCfg(patch_size=4, num_octaves=10, num_layers=2,
                                             num_heads=4, d_token=128, d_dot=128, d_mlp=256),
        num_octaves=10, num_layers=2, num_heads=heads, num_samples=n_samp, d_dot=128, d_mlp=256,
        downscale=down), d_feat, num_context_views=vc).to(dev)
    hA, wA = hw[0] // down, hw[1] // down
    torch.manual_seed(P.rank_seed(0, rank))
    feat = torch.randn(b, vc, hA, wA, d_feat, device=dev).requires_grad_(True)   # channels-last
    view_shuffle = torch.randperm(vc - 1, device=dev) if vc > 2 else None
    c_ext, c_intr = ctx.extrinsics.to(dev), ctx.intrinsics.to(dev)
    c_near, c_far = ctx.near.to(dev), ctx.far.to(dev)
    # the parameters path (A) owns: PreNorm + Attention of each epipolar layer (layers.{i}.0.*),
    # the depth encoding, the view embeddings.  NOT layers.{i}.1.* (the feed-forward block's
    # PreNorm) nor the image self-attention: this path never touches them, and an untouched
    # parameter in a bucket would hold that bucket's all-reduce back until finish()
    import re
    a_params = [p_ for n_, p_ in et.named_parameters()
                if re.match(r"transformer\.layers\.\d+\.0\.|depth_encoding\.|view_embeddings\.", n_)]

    # dL/d(output tokens) of (A): in the network the gradient arrives from the layers behind the
    # epipolar transformer (upscaler / refinement convolutions); here a fixed tensor resident in HBM
    # like every other input.  (Rounds 1-4 used a stand-in loss x.square().mean() inside the timed
    # region: 80 us of harness per step, VERDICT r4 weak #9.)
    gx = torch.randn(b * vc * hA * wA, 1, d_feat, device=dev) / (b * vc * hA * wA * d_feat)
    last = {}          # the step's outputs (static tensors of the graphs when replayed)

    def path_a():
        geo = et.epipolar_sampler.geometry(c_ext, c_intr, c_near, c_far, (hA, wA))
        x = feat.reshape(-1, 1, d_feat)
        view_emb = et.view_embeddings(view_shuffle) if vc > 2 else None  # epipolar_transformer.py:126-131
        folds = et.fold_layers(view_emb)
        grad_batch = FeatureGradBatch()
        feat_kv = grad_batch.attach(feat)     # as EpipolarTransformer.forward: deferred map gradient
        for (attn, _ff), folded in zip(et.transformer.layers, folds):
            x = et.fused_block(attn, x, feat_kv, geo, view_emb=view_emb, folded=folded, batch=grad_batch)
        last["x"] = x.detach()
        return x

    def backward_a(x):
        torch.autograd.backward(x, gx)

    list_cap = [0]     # > 0: fixed-capacity tile lists (no host sync in the forward)

    def path_b():
        img = render_cuda(ext, intr, near, far, hw, bg, means, cov, sh, op, views_per_scene=v,
                          list_capacity=list_cap[0])
        last["img"] = img.detach()
        return mse_loss(img, tgt_img, 1.0)   # LossMse (loss_mse.py:30-31), one pass

    def zero_grads():
        for t in (means, cov, sh, op, feat, *a_params):
            t.grad = None

    # the data-parallel collective of the step: bucketed asynchronous all-reduce
    # (mean) of the path's parameter gradients over RCCL; a no-op for one rank
    reducer = P.GradientReducer(a_params, world, bucket_bytes=int(args.bucket_mb * (1 << 20)),
                                extra_payload_bytes=int(args.grad_payload_mb * 1e6))

    graphs = {}

    def capture_warm_up():
        """Eager warm-up on the stream mode that will be captured.  Contains the reducer's collectives:
        a rank that fails in here cannot be waited for by its peers -- the caller aborts the
        job for N > 1 instead of falling back."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                zero_grads()
                backward_a(path_a())
                path_b().backward()
                reducer.finish()               # every rank: the hooks' buckets are reduced and re-armed
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        zero_grads()
        reducer.remove()                       # no collective may be launched from inside a capture:
                                               # replayed steps reduce with reducer.reduce_now()

    def capture_graphs():
        """The step as two hipGraphs -- (A) forward + backward, (B) forward + backward -- sharing
        one memory pool.  ~95 launches of (A) and ~30 of (B) become two graph launches: no
        per-launch host cost, no allocator traffic, no host synchronisation (fixed-capacity tile
        lists; an overflow is read back after the replay).  The gradient all-reduce is launched
        between the two replays, so it runs under (B) exactly as in the eager schedule.  No collective
        is issued in here: a failure leaves every rank with the same collective history."""
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        mode = "thread_local" if P.active(world) else "global"   # other threads (RCCL watchdog) may
        with torch.cuda.graph(ga, capture_error_mode=mode):  # call into the runtime meanwhile
            backward_a(path_a())
        with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode=mode):
            path_b().backward()
        graphs["a"], graphs["b"] = ga, gb

    def step(a=True, b_=True):
        if graphs:
            if a:
                graphs["a"].replay()
                reducer.reduce_now()
                reducer.launch_extra_payload()
            if b_:
                graphs["b"].replay()
            reducer.finish()
            return
        zero_grads()
        la = path_a() if a else None
        lb = path_b() if b_ else None
        if la is not None:
            backward_a(la)                # (A): the parameter gradients land -> buckets launch
            reducer.launch_extra_payload()
        if lb is not None:
            lb.backward()                 # (B): rasterizer backward runs over the reduction
        reducer.finish()

    # ---- Gaussian adapter, timed on its own after the contract's region:
    # the producer of the rasterizer's inputs at the same shape (b x 2 views x HxW rays x 3)
    from pixelsplat_amd.encoder import GaussianAdapter, GaussianAdapterCfg
    ga = GaussianAdapter(GaussianAdapterCfg(0.5, 15.0, 4)).to(dev)
    n_rays = hw[0] * hw[1]
    ga_in = dict(
        coordinates=torch.rand(b, vc, n_rays, 1, 1, 2, device=dev).requires_grad_(True),
        depths=(torch.rand(b, vc, n_rays, 1, 3, device=dev) * 5 + 0.5).requires_grad_(True),
        opacities=torch.rand(b, vc, n_rays, 1, 3, device=dev),
        raw=torch.randn(b, vc, n_rays, 1, 1, 82, device=dev).requires_grad_(True))
    ga_ext, ga_intr = c_ext[:, :, None, None, None], c_intr[:, :, None, None, None]

    def step_adapter():
        for t in ga_in.values():
            t.grad = None
        g_ = ga(ga_ext, ga_intr, ga_in["coordinates"], ga_in["depths"], ga_in["opacities"],
                ga_in["raw"], hw)
        (g_.means.sum() + g_.covariances.sum() + g_.harmonics.sum()).backward()

    # ---- depth predictor: ReLU + Linear + the fused sampler, same rays
    from pixelsplat_amd.encoder import DepthPredictorMonocular
    dp = DepthPredictorMonocular(d_feat, 32, 1, False).to(dev)
    dp_feat = torch.randn(b, vc, n_rays, d_feat, device=dev).requires_grad_(True)
    dp_near = torch.full((b, vc), 1.0, device=dev)
    dp_far = torch.full((b, vc), 100.0, device=dev)

    def step_depth():
        dp_feat.grad = None
        for p_ in dp.parameters():
            p_.grad = None
        dep, opa = dp.forward_mapped(dp_feat, dp_near, dp_far, False, 3, 1.0, 1.0 / 3)
        (dep.sum() + opa.sum()).backward()

    # ---- the chain either side of the boundary between (A) and (B): per-pixel features ->
    # EncoderEpipolarHead (depth sampler + head linear layers + adapter) -> rasterizer -> MSE,
    # gradients back to the features and the head weights (encoder_epipolar.py:143-214 +
    # decoder + loss_mse.py); random-init head, so its scene is not the synthetic one above
    from pixelsplat_amd.encoder import (EncoderEpipolarHead, EncoderEpipolarHeadCfg,
                                        OpacityMappingCfg)
    head = EncoderEpipolarHead(EncoderEpipolarHeadCfg(
        d_feature=d_feat, num_monocular_samples=32, num_surfaces=1, predict_opacity=False,
        gaussians_per_pixel=3, gaussian_adapter=GaussianAdapterCfg(0.5, 15.0, 4),
        opacity_mapping=OpacityMappingCfg(0.0, 0.0, 1), use_transmittance=False)).to(dev)
    head_ctx = dict(extrinsics=c_ext, intrinsics=c_intr, near=ctx.near.to(dev), far=ctx.far.to(dev))
    head_feat = torch.randn(b, vc, *hw, d_feat, device=dev).permute(0, 1, 4, 2, 3).requires_grad_(True)

    def step_chain():
        head_feat.grad = None
        for p_ in head.parameters():
            p_.grad = None
        gs = head(head_feat, head_ctx, global_step=0)
        img_ = render_cuda(ext, intr, near, far, hw, bg, gs.means, gs.covariances, gs.harmonics,
                           gs.opacities, views_per_scene=v)
        mse_loss(img_, tgt_img, 1.0).backward()

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    # D = sum of tile-list lengths (reported; enters the algorithmic-bytes figure)
    img, aux = render_cuda(ext, intr, near, far, hw, bg, means, cov, sh, op, views_per_scene=v,
                           return_aux=True)
    counts, _, _ = export_bins(aux["cfg"], aux["state"], aux["layout"], aux["point_list"])
    D_total = int(counts.to(torch.int64).sum().item())
    n_visible = int((aux["radii"] > 0).sum().item())
    # pairs whose rect covers more than 4 tiles: the tile backward accumulates their gradient with
    # float atomics instead of private slots (rects: [xmin, ymin, xmax, ymax) in tiles)
    from pixelsplat_amd.raster import state_views as _sv
    _r = _sv(aux["cfg"], aux["state"], aux["layout"])["rects"].to(torch.int32)
    _tiles = (_r[..., 2] - _r[..., 0]) * (_r[..., 3] - _r[..., 1])
    n_large = int(((aux["radii"].reshape(_tiles.shape) > 0) & (_tiles > 4)).sum().item())
    final_T_stats = _sv(aux["cfg"], aux["state"], aux["layout"])
    _nc = final_T_stats["n_contrib"].to(torch.float32).reshape(V, hw[0] // 16, 16, hw[1] // 16, 16) \
        if hw[0] % 16 == 0 and hw[1] % 16 == 0 else None
    if _nc is not None:     # where a pixel's walk ends inside its tile's list (1.0 = never stops early)
        _len = counts.to(torch.float32).reshape(V, hw[0] // 16, 1, hw[1] // 16, 1).clamp(min=1)
        walk_end_median = round(float((_nc / _len).median().item()), 3)
    else:
        walk_end_median = None
    del _r, _tiles, final_T_stats, _nc
    vps_np = aux["view_params"].cpu().numpy()
    gpu_images = img.detach().cpu().numpy()
    gpu_fwd = None
    if world == 1 and not args.no_cpu_baseline:     # the parity block compares THIS launch
        from pixelsplat_amd.raster import state_views
        sv_ = state_views(aux["cfg"], aux["state"], aux["layout"])
        c_, o_, pl_ = export_bins(aux["cfg"], aux["state"], aux["layout"], aux["point_list"])
        gpu_fwd = dict(images=gpu_images, radii=aux["radii"].cpu().numpy(),
                       final_T=sv_["final_T"].cpu().numpy().reshape(V, *hw),
                       n_contrib=sv_["n_contrib"].cpu().numpy().reshape(V, *hw),
                       counts=c_.cpu().numpy(), offsets=o_.cpu().numpy(), plist=pl_.cpu().numpy())
        del sv_, c_, o_, pl_
    del aux, counts, img
    torch.cuda.empty_cache()

    launch_mode = "eager"
    launch_fallback = None
    # auto = hipGraph replay for EVERY N (one launch mode along a scaling curve; a failed capture
    # falls back to eager launches on all ranks and says so in `launch` / `launch_fallback`)
    if args.launch in ("graph", "auto"):
        from pixelsplat_amd.raster import captured_overflow_flags
        list_cap[0] = (int(D_total * 1.25) + 4095) // 4096 * 4096
        local_ok = True
        try:
            capture_warm_up()
        except Exception as err:
            if world > 1:      # peers are inside the warm-up's collectives: no consistent way back
                raise
            local_ok, launch_fallback = False, f"warm-up: {type(err).__name__}: {err}"[:300]
        if local_ok:
            try:
                capture_graphs()
            except Exception as err:   # capture not possible in this build: the eager schedule
                print(f"[bench] hipGraph capture failed ({type(err).__name__}: {err}); "
                      f"falling back to eager launches", file=sys.stderr)
                local_ok, launch_fallback = False, f"{type(err).__name__}: {err}"[:300]
                torch.cuda.synchronize()
        # one mode for the whole job: any rank's failed capture sends every rank to eager (every rank
        # reaches this point with the same collective history: warm-up done, none in the capture)
        launch_mode, why = P.choose_launch_mode(args.launch, local_ok, world, dev)
        if why is not None and launch_fallback is None:
            launch_fallback = why
        if launch_mode == "hipgraph":
            try:
                step()
                torch.cuda.synchronize()
                captured_overflow_flags(check=True)
            except Exception as err:
                if world > 1:       # the other ranks are already replaying: no consistent way back
                    raise
                print(f"[bench] hipGraph replay failed ({type(err).__name__}: {err}); "
                      f"falling back to eager launches", file=sys.stderr)
                launch_mode, launch_fallback = "eager", f"{type(err).__name__}: {err}"[:300]
                torch.cuda.synchronize()
        if launch_mode != "hipgraph":
            from pixelsplat_amd.raster import release_captured_flags
            graphs.clear()
            release_captured_flags()
            list_cap[0] = 0
    if launch_mode == "eager" and args.launch != "eager":
        reducer.reset()
        reducer.install_hooks()
        zero_grads()
    for _ in range(args.warmup):
        step()
    ng = lib.ps_profile_group_count()
    tot_ms = (C.c_double * ng)()
    launches = (C.c_int64 * ng)()
    P.barrier(world)
    if launch_mode == "eager":
        lib.ps_profile_enable(1)       # HIP events around every library launch of the timed steps
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    P.barrier(world)
    elapsed_local = time.perf_counter() - t0
    lib.ps_profile_enable(0)
    elapsed = P.max_over_ranks(elapsed_local, world, dev)
    rank_ms = P.gather_over_ranks(elapsed_local / args.steps * 1e3, world, dev)
    exposed = reducer.exposed_ms(last=args.steps)
    launches_before_finish = reducer.stats["launches_before_finish"]
    launches_so_far = reducer.stats["launches"]
    eager_ms = elapsed / args.steps * 1e3

    def step_check():
        """What did the LAST timed step -- replayed from the graphs or launched eagerly, reduced over the
        ranks -- leave behind?  Compared with one eager step of the same inputs without the reducer, its
        parameter gradients averaged over the ranks by hand: the image against the eager launch the
        parity block checks against the oracle, the tokens of (A), every gradient tensor.  (Round 4: a
        replayed step was never compared with anything; the feature-map gradient raced -- DESIGN.md 6.)"""
        torch.cuda.synchronize()
        leaves = [("d_means", means), ("d_cov", cov), ("d_sh", sh), ("d_opacity", op), ("d_features", feat)]
        tensors = [t for _, t in leaves] + list(a_params)
        got = [(t.grad.detach().clone() if t.grad is not None else None) for t in tensors]
        got_img, got_x = last["img"].clone(), last["x"].clone()
        saved_graphs, saved_grads = dict(graphs), [(t, t.grad) for t in tensors]
        had_hooks = bool(reducer._hooks)
        graphs.clear()
        reducer.remove()
        zero_grads()
        backward_a(path_a())
        path_b().backward()
        torch.cuda.synchronize()
        want = [(t.grad.detach() if t.grad is not None else torch.zeros_like(t)) for t in tensors]
        if P.active(world) and a_params:      # the reduction by hand: mean over the ranks
            import torch.distributed as dist
            flat = torch.cat([w_.reshape(-1) for w_ in want[len(leaves):]]) / world
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            off = 0
            for i in range(len(leaves), len(want)):
                n_ = want[i].numel()
                want[i] = flat[off:off + n_].view_as(want[i])
                off += n_

        def rel(a_, b_):
            a_ = torch.zeros_like(b_) if a_ is None else a_
            return float((a_ - b_).abs().max() / b_.abs().max().clamp_min(1e-30))

        errs = {n_: rel(g_, w_) for (n_, _), g_, w_ in zip(leaves, got, want)}
        errs["d_parameters_reduced"] = max([rel(g_, w_) for g_, w_ in zip(got[len(leaves):], want[len(leaves):])]
                                          or [0.0])
        errs["tokens_out"] = rel(got_x, last["x"])
        img_now = last["img"]
        errs["image_vs_this_eager_step"] = float((got_img - img_now).abs().max())
        errs["image_vs_oracle_checked_launch"] = float(
            (got_img.cpu() - torch.from_numpy(gpu_images)).abs().max())
        # bars: (A) is deterministic (bitwise with one rank; the mean over ranks is one rounding);
        # the rasterizer's > 4-tile Gaussians sum through float atomics (<= 2e-6 of max measured)
        bars = {"d_means": 2e-5, "d_cov": 2e-5, "d_sh": 2e-5, "d_opacity": 2e-5, "d_features": 1e-6,
                "d_parameters_reduced": 1e-6, "tokens_out": 0.0, "image_vs_this_eager_step": 0.0,
                "image_vs_oracle_checked_launch": 1e-5}
        ok = all(errs[k] <= bars[k] for k in bars)
        graphs.update(saved_graphs)
        for t, g_ in saved_grads:
            t.grad = g_
        if had_hooks:
            reducer.install_hooks()
        if P.active(world):       # one verdict for the job
            ok = bool(P.max_over_ranks(0.0 if ok else 1.0, world, dev) == 0.0)
        return {"ok": ok, "launch": launch_mode, "steps_before_the_check": args.steps + args.warmup,
                "ranks": world, "max_err": {k: float(f"{e:.3g}") for k, e in errs.items()}, "bars": bars,
                "what": "the last timed step's outputs and gradients (after the all-reduce) against one eager "
                        "step without the reducer, parameter gradients averaged over the ranks by hand; "
                        "image also against the eager launch `parity_vs_oracle` is computed from"}

    check = step_check()
    if not check["ok"]:
        print(f"[bench] STEP CHECK FAILED on rank {rank}: {check['max_err']}", file=sys.stderr)
    if launch_mode == "hipgraph":
        # events cannot be read from inside a replayed graph: the same kernels are timed in an
        # eager pass of the same K steps right after the timed region (same process, same data)
        captured_overflow_flags(check=True)
        saved = dict(graphs)
        static_grads = [(t, t.grad) for t in (means, cov, sh, op, feat, *a_params)]
        graphs.clear()
        reducer.install_hooks()         # the eager schedule reduces from the hooks
        step()
        torch.cuda.synchronize()
        lib.ps_profile_enable(1)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        lib.ps_profile_enable(0)
        # the same step launched eagerly, timed like the contract's region (barrier + sync on both
        # sides, slowest rank): the like-for-like denominator / numerator of a scaling curve
        # whatever mode each N ended up in
        P.barrier(world)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        P.barrier(world)
        eager_ms = P.max_over_ranks(time.perf_counter() - t0, world, dev) / args.steps * 1e3
        reducer.remove()
        graphs.update(saved)
        for t, g_ in static_grads:      # the graphs write into these tensors
            t.grad = g_
    _lib.check(lib.ps_profile_collect(tot_ms, launches), "ps_profile_collect")
    # outside the contract's timed region: each path alone
    ms_b = timed(lambda: step(a=False), args.steps)
    ms_a = timed(lambda: step(b_=False), args.steps)
    ms_ga = ms_dp = 0.0
    ms_chain = None
    if not args.no_probes:
        step_adapter()
        step_depth()
        lib.ps_profile_enable(1)
        ms_ga = timed(step_adapter, args.steps)
        ms_dp = timed(step_depth, args.steps)
        lib.ps_profile_enable(0)
        try:
            step_chain()
            ms_chain = timed(step_chain, args.steps)
        except RuntimeError as err:   # e.g. PS_ERR_CAPACITY for a degenerate random scene
            print(f"[bench] chain probe skipped: {err}", file=sys.stderr)
            ms_chain = None
    side_ms = (C.c_double * ng)()
    side_n = (C.c_int64 * ng)()
    _lib.check(lib.ps_profile_collect(side_ms, side_n), "ps_profile_collect")
    for i in range(ng):   # the groups only these two probes launch
        if launches[i] == 0 and side_n[i]:
            tot_ms[i], launches[i] = side_ms[i], side_n[i]
    connected = None
    if args.connected:
        # the halves composed: the SAME module classes, paper encoder config, random-init weights, the
        # synthetic cameras of the step; the convolutions / image self-attention around the epipolar layers
        # are PyTorch's (MIOpen / hipBLASLt), out of the hot path's scope but inside this number
        from pixelsplat_amd.training_step import ConnectedStep
        torch.manual_seed(0)
        cs = ConnectedStep(et.cfg, d_feat, vc, head.cfg).to(dev)
        cs_feat = torch.randn(b, vc, d_feat, *hw, device=dev).requires_grad_(True)
        cs_ctx = dict(extrinsics=c_ext, intrinsics=c_intr, near=c_near, far=c_far)
        cs_tgt = dict(extrinsics=tgt.extrinsics.to(dev), intrinsics=tgt.intrinsics.to(dev),
                      near=tgt.near.to(dev), far=tgt.far.to(dev), image=tgt_img.reshape(b, v, 3, *hw))

        def step_connected():
            cs_feat.grad = None
            for p_ in cs.parameters():
                p_.grad = None
            cs(cs_feat, cs_ctx, cs_tgt, global_step=0).loss.backward()

        try:
            step_connected()
            torch.cuda.synchronize()
            n_c = max(1, min(args.steps, 5))
            lib_ms0 = (C.c_double * ng)()
            lib_n0 = (C.c_int64 * ng)()
            ms_c = timed(step_connected, n_c)
            lib.ps_profile_enable(1)
            for _ in range(n_c):
                step_connected()
            torch.cuda.synchronize()
            lib.ps_profile_enable(0)
            _lib.check(lib.ps_profile_collect(lib_ms0, lib_n0), "ps_profile_collect")
            connected = {
                "ms_per_step": round(ms_c, 3), "steps": n_c, "launch": "eager",
                "library_kernels_ms_per_step": round(sum(lib_ms0[i] for i in range(ng)) / n_c, 3),
                "views_per_s": round(V / ms_c * 1e3, 1),
                "parameters_with_gradient": sum(1 for p_ in cs.parameters() if p_.grad is not None),
                "parameters": sum(1 for _ in cs.parameters()),
                "features_grad_finite": bool(torch.isfinite(cs_feat.grad).all()),
                "what": "features [b,v,128,H,W] -> EpipolarTransformer.forward (full module incl. PyTorch's "
                        "downscale / upscale / 7x7 refinement convolutions and image self-attention) -> "
                        "EncoderEpipolarHead -> DecoderSplattingCUDA -> LossMse -> backward; random-init weights; "
                        "library_kernels_ms = HIP-event time inside this library's launches, the rest is PyTorch's"}
        except RuntimeError as err:
            print(f"[bench] connected step skipped: {err}", file=sys.stderr)
            connected = {"error": str(err)[:300]}
        del cs, cs_feat
        torch.cuda.empty_cache()

    # ---- the `dense` scene distribution in the same run: SURVEY.md 8d warns that a trained
    # model has D / G of 2-4, the contract's `survey` recipe has 1.27.  Same step, same launch mode where the
    # capture succeeds (fixed-capacity lists sized for THIS scene), the Gaussians swapped in place; one rank only.
    dense = None
    if world == 1 and not args.no_probes and args.scene == "survey":
        try:
            _, _, g_d, _ = make_workload(b, hw, v_ctx=vc, v_tgt=v, seed=P.rank_seed(0, rank), scene="dense")
            keep = [t.detach().clone() for t in (means, cov, sh, op)]
            saved_graphs_d, cap_keep = dict(graphs), list_cap[0]
            graphs.clear()
            list_cap[0] = 0
            with torch.no_grad():
                for t_, s_ in zip((means, cov, sh, op), (g_d.means, g_d.covariances, g_d.harmonics, g_d.opacities)):
                    t_.copy_(s_.to(dev))
            img_d, aux_d = render_cuda(ext, intr, near, far, hw, bg, means, cov, sh, op, views_per_scene=v,
                                       return_aux=True)
            cnt_d, _, _ = export_bins(aux_d["cfg"], aux_d["state"], aux_d["layout"], aux_d["point_list"])
            D_dense = int(cnt_d.to(torch.int64).sum().item())
            vis_dense = int((aux_d["radii"] > 0).sum().item())
            del img_d, aux_d, cnt_d
            dense_launch = "eager"
            if launch_mode == "hipgraph":
                try:
                    list_cap[0] = (int(D_dense * 1.25) + 4095) // 4096 * 4096
                    zero_grads()
                    capture_graphs()
                    step()
                    torch.cuda.synchronize()
                    captured_overflow_flags(check=True)
                    dense_launch = "hipgraph"
                except Exception as err:      # noqa: BLE001 -- the probe falls back to eager launches
                    print(f"[bench] dense probe: capture failed ({type(err).__name__}: {err}); eager", file=sys.stderr)
                    graphs.clear()
                    list_cap[0] = 0
                    torch.cuda.synchronize()
            for _ in range(2):
                step()
            n_d = max(1, min(args.steps, 10))
            ms_dense = timed(step, n_d)
            ms_dense_b = timed(lambda: step(a=False), n_d)
            dense = {"ms_per_step": round(ms_dense, 3), "views_per_s": round(V / ms_dense * 1e3, 1),
                     "raster_only_ms_per_step": round(ms_dense_b, 3), "steps": n_d, "launch": dense_launch,
                     "tile_list_entries_D": D_dense, "D_over_GV": round(D_dense / (G * V), 3),
                     "visible_frac": round(vis_dense / (G * V), 3),
                     "what": "the contract's step with the `dense` scene distribution (pixelsplat_amd/synthetic.py: "
                             "depth in the far 15 % of the disparity range) -- NOT a BASELINE config; reported beside "
                             "`value`, never as it"}
            graphs.clear()
            graphs.update(saved_graphs_d)
            list_cap[0] = cap_keep
            with torch.no_grad():
                for t_, s_ in zip((means, cov, sh, op), keep):
                    t_.copy_(s_)
            del keep, g_d
            torch.cuda.empty_cache()
        except RuntimeError as err:
            print(f"[bench] dense probe skipped: {err}", file=sys.stderr)
            dense = {"error": str(err)[:300]}

    comm_info = P.comm_info(world, dev)       # (a collective: every rank takes part)
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = P.aggregate_throughput(V, args.steps, world, elapsed)
        groups = {lib.ps_profile_group_name(i).decode(): (tot_ms[i] / max(launches[i], 1),
                                                          int(launches[i])) for i in range(ng)}
        # algorithmic bytes per launch (DESIGN.md "kernels"): the reference-algorithm figure of
        # SURVEY.md 8(d), attributed per kernel; one launch covers all V views of the batch.
        npix = hw[0] * hw[1]
        alg = {
            "preprocess_forward": 392.0 * G * V,
            "depth_sort": 48.0 * D_total,
            "tile_bins": 0.0,
            "tiles_forward": 36.0 * D_total + 20.0 * npix * V,
            "tiles_backward": 76.0 * D_total + 20.0 * npix * V,
            "preprocess_backward": 728.0 * G * V,
        }
        # (A): compulsory HBM bytes of the folded formulation per launch; these
        # kernels are VALU/latency bound, the figures are there to show how far from HBM
        RA, TA, PA = b * vc * hA * wA, n_samp * (vc - 1), 20
        fm = 4.0 * b * vc * hA * wA * d_feat
        alg.update({
            "epipolar_geometry": RA * (24.0 + 25.0 + 16.0 * TA),
            "epipolar_attention_forward": fm + RA * (12.0 * TA + 4.0 * heads * (2 * d_feat + 2 * PA + TA)),
            "epipolar_attention_backward": fm + RA * (12.0 * TA + 4.0 * heads * (2 * d_feat + 2 * PA + 2 * TA)),
            "epipolar_feature_grad": fm + RA * (8.0 * TA + 4.0 * heads * (2 * d_feat + 2 * TA)),
        })
        # the two-pass feature gradient writes and re-reads the token-gradient tensor: its floor
        alg["epipolar_feature_grad"] = fm + 2 * 4.0 * float(RA) * TA * d_feat
        dom = max((k for k in alg if k in groups), key=lambda k: groups[k][0])
        dom_ms = groups[dom][0]
        achieved = alg[dom] / (dom_ms * 1e-3) / 1e9
        # committed PMC summaries exist for the three benchmarked BASELINE configurations
        global PMC_TAG
        PMC_TAG = {(256, 256, 7, 4, 2): "c2", (256, 256, 4, 4, 3): "c4",
                   (512, 512, 2, 4, 2): "c5"}.get((hw[0], hw[1], b, v, vc))
        if args.scene != "survey":      # counters are taken on the BASELINE (survey) workload only
            PMC_TAG = None
        is_c2 = PMC_TAG is not None
        traffic, traffic_src = pmc_traffic(dom)
        # (A) against the fp32 vector peak: FOLDED useful flops = what the fused
        # kernels have to compute per launch (score + context: 2 x (c + P) MACs per (ray, token, head);
        # bilinear interpolation 4 corners x c MACs per token; token gradient 2 c MACs per (token, layer,
        # head) + the 4-corner scatter) and the REFERENCE-EQUIVALENT flops of the same stage (to_kv on
        # every token + QK^T + AV) -- the second is a folding artefact, not a utilisation
        VEC_PEAK = 157.3      # TFLOP/s fp32 vector (packed FMA), MI355X_MICROARCH.md
        n_layers_a = len(et.transformer.layers)
        tok = float(RA) * TA
        fold_attn = tok * heads * 2 * (d_feat + PA) * 2 + tok * 4 * d_feat * 2
        ref_attn = 2.0 * tok * d_feat * 2 * heads * 128 + 2 * 2.0 * tok * heads * 128
        flops_a = {
            "epipolar_attention_forward": (fold_attn, ref_attn),
            "epipolar_attention_backward": (fold_attn, 2 * ref_attn),
            # one launch group per step, both layers at once
            "epipolar_feature_grad": (tok * n_layers_a * heads * 2 * d_feat * 2 + tok * 4 * d_feat * 2, 0.0),
        }
        roofline_a = {}
        for k_, (ff, rf) in flops_a.items():
            if k_ in groups and groups[k_][0] > 0:
                # the feature-gradient group is timed per launch of its kernels: per step = ms x launches
                ms_ = groups[k_][0] * (groups[k_][1] / args.steps if k_ == "epipolar_feature_grad" else 1.0)
                roofline_a[k_] = {
                    "ms": round(ms_, 4), "folded_gflop": round(ff / 1e9, 2),
                    "folded_tflops": round(ff / (ms_ * 1e-3) / 1e12, 2),
                    "frac_of_fp32_vector_peak": round(ff / (ms_ * 1e-3) / 1e12 / VEC_PEAK, 4),
                    "reference_equivalent_gflop": round(rf / 1e9, 1) if rf else None,
                    "reference_equivalent_tflops": round(rf / (ms_ * 1e-3) / 1e12, 1) if rf else None,
                    "algorithmic_bytes": alg[k_],
                    "algorithmic_frac_of_hbm_peak": round(alg[k_] / (ms_ * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "valu_issue_frac": (round(pmc_valu_busy_ms(k_) / ms_, 3) if pmc_valu_busy_ms(k_) else None)}
        if "gemm_tn_splitk" in groups and groups["gemm_tn_splitk"][0] > 0:
            # dW = dY^T X over all rays, the two shapes of a layer: [592 x R] x [R x 128] twice per layer
            ff = 2.0 * RA * d_feat * (heads * (d_feat + PA + (vc - 1 if vc > 2 else 0)))
            ms_ = groups["gemm_tn_splitk"][0]
            roofline_a["gemm_tn_splitk"] = {
                "ms": round(ms_, 4), "gflop": round(ff / 1e9, 2), "tflops": round(ff / (ms_ * 1e-3) / 1e12, 1),
                "bound": "mfma", "peak": VEC_PEAK, "frac_of_fp32_matrix_peak": round(ff / (ms_ * 1e-3) / 1e12 / VEC_PEAK, 4)}
        valu_ms = pmc_valu_busy_ms(dom) if is_c2 else None
        out = {
            "metric": "rendered views/sec (fwd+bwd)", "value": round(value, 2), "unit": "views/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": f"{'acid' if vc > 2 else 're10k'} {vc}-view, {hw[0]}x{hw[1]}, batch_size={b} per GPU, {v} target "
                            f"views/scene{BASELINE_TAG.get((hw[0], b, v, vc), '') if args.scene == 'survey' else f' [scene distribution: {args.scene} -- NOT a BASELINE config]'}: epipolar sampler + 2 "
                            f"cross-attention layers on [{b},{vc},{d_feat},{hA},{wA}] (A) + "
                            f"rasterizer (B), fwd+bwd",
                "scene": args.scene,
                "epipolar_rays": b * vc * hA * wA, "epipolar_samples_per_ray": n_samp,
                "epipolar_kv_tokens_per_ray": n_samp * (vc - 1),
                "gaussians_per_scene": G, "views_per_step_per_gpu": V,
                "tile_list_entries_D": D_total, "visible_gaussian_views": n_visible,
                "D_over_GV": round(D_total / (G * V), 3),
                "visible_frac": round(n_visible / (G * V), 3),
                "atomic_path_pairs_frac": round(n_large / max(n_visible, 1), 4),
                "median_walk_end_over_list_length": walk_end_median,
                "parallelism": f"dp{world}",
            },
            "roofline": {
                "bound": "hbm", "kernel": dom, "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": traffic_src,
                "traffic_note": (None if traffic is not None else
                                 "no committed counter summary was taken on this build of the kernel "
                                 "(profiles/*_pmc_traffic.json `build` vs ps_build_info): null rather "
                                 "than stale; tools/profile_bench.sh re-collects it"),
                "avg_kernel_ms": round(dom_ms, 4),
                "kernel_timing": ("HIP events around the kernel's launches over the timed steps"
                                  if launch_mode == "eager" else
                                  "HIP events over an eager pass of the same K steps right after the "
                                  "timed region (the timed region replays the same kernels from hipGraphs)"),
                # the dominant kernel is VALU-issue bound, not HBM bound: share of
                # its time the SIMDs spend issuing VALU instructions, from the committed PMC run
                "valu_issue_frac": (round(valu_ms / dom_ms, 3) if valu_ms else None),
                "algorithmic_bytes_per_launch": alg[dom],
            },
            # every single-kernel group: live duration x committed PMC traffic (HBM-side bytes per
            # launch, valid for the configs[1] workload the counters were taken on)
            # NOT roofline fractions: counter traffic (useful + wasted bytes) over time
            "kernel_traffic": {
                g_: {"ms": round(groups[g_][0], 4),
                     "traffic_gb_per_s": round(pmc_traffic(g_)[0] / (groups[g_][0] * 1e-3) / 1e9, 1),
                     "traffic_frac_of_hbm_peak": round(pmc_traffic(g_)[0] / (groups[g_][0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3),
                     "algorithmic_frac_of_hbm_peak": (round(alg[g_] / (groups[g_][0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 3)
                                                      if g_ in alg else None),
                     "valu_issue_frac": (round(pmc_valu_busy_ms(g_) / groups[g_][0], 3)
                                         if pmc_valu_busy_ms(g_) else None)}
                for g_ in SINGLE_KERNEL_GROUPS
                if is_c2 and g_ in groups and groups[g_][0] > 0 and pmc_traffic(g_)[0]},
            "roofline_a": roofline_a,
            "step_check": check,
            # what the timed region contains: 2 = (A) forward + backward from a fixed upstream gradient, (B) forward
            # + LossMse + backward (round 5 on); 1 = rounds 1-4, (A) through a stand-in loss x.square().mean()
            # (+ ~0.08 ms per step): `value` of rounds 1-4 is not like-for-like
            "harness_version": 2,
            "build": build_info,
            "launch": launch_mode, "launch_requested": args.launch, "launch_fallback": launch_fallback,
            "library_gemm_table": ("pixelsplat_amd/gemm_tuning/gfx950_rocm7_torch2.10.csv"
                                   if tuned_gemms else None),
            "kernels_ms": {k: round(groups[k][0], 4) for k in groups},
            "kernel_launches_per_step": {k: groups[k][1] / args.steps for k in groups},
            "paths": {
                "eager_ms_per_step": round(eager_ms, 3),
                "raster_only_ms_per_step": round(ms_b, 3),
                "raster_only_views_per_s": round(V / ms_b * 1e3 * world, 1),
                "epipolar_only_ms_per_step": round(ms_a, 3),
                # next row of SURVEY.md 8(f): raw network outputs -> Gaussians, fwd + bwd incl.
                # the three torch .sum() reductions of this probe; not part of `value`
                "gaussian_adapter_only_ms_per_step": round(ms_ga, 3),
                # rank 3: features -> (depth, opacity): ReLU + Linear (library GEMM, split-k
                # weight gradient) + ps_depth_sampler_*, fwd + bwd; not part of `value`
                "depth_predictor_only_ms_per_step": round(ms_dp, 3),
                # features -> head -> Gaussians -> 28 rendered views -> MSE and back
                "head_decoder_loss_chain_ms_per_step": (round(ms_chain, 3) if ms_chain else None),
                "dense_scene_ms_per_step": (dense or {}).get("ms_per_step"),
                "dense_views_per_s": (dense or {}).get("views_per_s"),
                "dense_scene": dense,
                "connected_ms_per_step": (connected or {}).get("ms_per_step"),
                "connected": connected,
                "epipolar_reference_equivalent_tflops": round(
                    3.0 * 2 * (2.0 * RA * (2 * d_feat * 512 + TA * d_feat * 1024 + 2 * 4 * TA * 128))
                    / (ms_a * 1e-3) / 1e12, 1),
            },
            "comm": {
                "backend": (torch.distributed.get_backend() if P.active(world) else None),
                "world_size": world, "forced_one_rank_communicator": bool(world == 1 and P.active(world)), "collective": "all_reduce(mean) of the path's parameter gradients, "
                "bucketed, asynchronous under the rasterizer backward (parallel.GradientReducer)",
                "gradient_bytes_per_step": 4 * sum(p_.numel() for p_ in a_params),
                "extra_payload_bytes_per_step": int(args.grad_payload_mb * 1e6),
                "bucket_mb": args.bucket_mb, **{k_: v_ for k_, v_ in reducer.stats.items()},
                # collectives of the timed steps launched BEFORE finish() (from the gradient hooks /
                # reduce_now, i.e. under the rasterizer's backward) and what finish() still waited
                "launches_before_finish_total": launches_before_finish,
                "launches_total_at_end_of_timed_region": launches_so_far,
                "exposed_ms_per_step": (round(sum(exposed) / len(exposed), 4) if exposed else 0.0),
                "rank_ms_per_step": {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3),
                                     "all": [round(x, 3) for x in rank_ms]},
                **comm_info,
            },
            "whole_path": {
                # (B)'s reference-algorithm bytes over (B)'s own step time
                "raster_algorithmic_bytes_per_step": (1120.0 * G + 40.0 * npix) * V + 160.0 * D_total,
                "raster_hbm_frac": round(((1120.0 * G + 40.0 * npix) * V + 160.0 * D_total)
                                         / (ms_b * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                # BASELINE.json north_star: ">= 60 % of the MI355X HBM roofline on the raster fwd+bwd".
                # The bar stands; the line reports the distance to it
                "raster_hbm_frac_target": 0.60,
                "raster_hbm_frac_gap": round(0.60 - ((1120.0 * G + 40.0 * npix) * V + 160.0 * D_total)
                                             / (ms_b * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "raster_ms_at_target": round(((1120.0 * G + 40.0 * npix) * V + 160.0 * D_total)
                                             / (0.60 * HBM_PEAK_GBS * 1e9) * 1e3, 3),
                # the same formula with what THIS design reads / writes once per SCENE counted once per
                # scene (the 340 B/Gaussian inputs in the forward and again in the backward, the 340
                # B/Gaussian of gradients) instead of once per view: the contract's figure above charges
                # them per view ("the formula is not what those kernels do")
                "raster_bytes_inputs_once_per_scene": 3 * 340.0 * G * b + (100.0 * G + 40.0 * npix) * V
                                                      + 160.0 * D_total,
                "raster_hbm_frac_inputs_once_per_scene": round(
                    (3 * 340.0 * G * b + (100.0 * G + 40.0 * npix) * V + 160.0 * D_total)
                    / (ms_b * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            dL = (2.0 * (torch.from_numpy(gpu_images) - target.reshape(V, 3, *hw))
                  / gpu_images.size).numpy()
            nv = min(args.cpu_views, V)

            def product_gradients(dl_np):
                """The product's backward of the same one-call launch for a given dL/dimage."""
                zero_grads()
                img_ = render_cuda(ext, intr, near, far, hw, bg, means, cov, sh, op, views_per_scene=v)
                (img_ * torch.from_numpy(dl_np).to(dev)).sum().backward()
                return tuple(t.grad.cpu().numpy() for t in (means, cov, sh, op))

            cb, work, parity = cpu_baseline(g, tgt, vps_np, hw, nv, gpu_fwd, dL, product_gradients)
            # VALU roofline of the two blend kernels: USEFUL fp32 operations = the (pixel, entry)
            # pairs that pass the alpha test (counted by the oracle on the sampled views, scaled
            # to the step) x the operations of the reference algorithm per such pair (A.3: 21
            # forward, A.4: 75 backward) over the kernel's time, against the fp32
            # vector peak.  The kernels evaluate ~4x more pairs than contribute (8x8 quadrant
            # granularity of the cull), the reference 11x more (whole tiles).
            scale_v = V / work["views"]
            contrib = work["pairs_contributing"] * scale_v
            for kname, flops in (("tiles_backward", 75.0), ("tiles_forward", 21.0)):
                t_ms = groups[kname][0]
                ach = contrib * flops / (t_ms * 1e-3) / 1e12
                out.setdefault("roofline_valu", {})[kname] = {
                    "bound": "valu", "achieved": round(ach, 2), "peak": 157.3, "unit": "TFLOP/s",
                    "frac": round(ach / 157.3, 4), "useful_flops_per_contributing_pair": flops,
                    "contributing_pairs_per_step": int(contrib),
                    "pairs_evaluated_by_reference_per_step": int(work["pairs_evaluated_by_reference"] * scale_v),
                    "avg_kernel_ms": round(t_ms, 4),
                    "valu_issue_frac": (round(pmc_valu_busy_ms(kname) / t_ms, 3)
                                        if pmc_valu_busy_ms(kname) else None),
                    # what a lane does: the (entry, 8x8 quadrant) pairs the kernels'
                    # conservative cull evaluates up to every tile's last contributor (oracle walk, numpy
                    # restatement of quadrant_mask), 64 lanes each, against the pairs that contribute; and
                    # the VALU instructions the kernel really issued (committed SQ_INSTS_VALU of this build)
                    "quadrant_evaluations_per_step": int(work["quadrant_pairs"] * scale_v),
                    "list_entries_reaching_a_quadrant_per_step": int(work["quadrant_entries"] * scale_v),
                    # lanes the kernel as shipped evaluates per useful lane: the backward culls per 8x8 quadrant
                    # (64 lanes per pair); the forward (round 6, csrc/raster_cells.hip) per 4x4 cell with four
                    # 16-lane rows per wave, stepping as often as the longest of a wave's four cell queues
                    "lane_efficiency": round(contrib / max((work["cell_row_steps"] if kname == "tiles_forward"
                                                            else work["quadrant_pairs"]) * scale_v * 64.0, 1.0), 4),
                    "lane_efficiency_8x8": round(contrib / max(work["quadrant_pairs"] * scale_v * 64.0, 1.0), 4),
                    "lane_efficiency_4x4": round(contrib / max(work["cell_row_steps"] * scale_v * 64.0, 1.0), 4),
                    "lane_efficiency_4x4_rows_balanced": round(contrib / max(work["cell_pairs"] * scale_v * 16.0, 1.0), 4),
                    "cell_pairs_per_step": int(work["cell_pairs"] * scale_v),
                    "cell_row_steps_per_step": int(work["cell_row_steps"] * scale_v),
                    "valu_wave_instructions_per_launch": (int(pmc_sq_counter(kname, "SQ_INSTS_VALU"))
                                                          if pmc_sq_counter(kname, "SQ_INSTS_VALU") else None),
                    "valu_lane_instructions_per_contributing_pair": (
                        round(pmc_sq_counter(kname, "SQ_INSTS_VALU") * 64.0 / contrib, 1)
                        if pmc_sq_counter(kname, "SQ_INSTS_VALU") else None)}
            t_a = cpu_baseline_epipolar(et, feat, ctx, n_samp, heads, view_shuffle)   # one scene
            t_step = b * t_a + V / cb["value"]
            cb["raster_only_views_per_s"] = round(cb["value"], 3)
            cb["epipolar_s_per_scene"] = round(t_a, 2)
            cb["value"] = V / t_step
            cb["sample"] += (f"; (A) oracle/epipolar_ref.py (unfused, torch CPU autograd) on 1 of "
                             f"{b} scenes, fwd+bwd, {t_a:.1f} s; value = {V} views / "
                             f"({b} x scene time + {V} x view time)")
            # the REFERENCE's own path-(A) code on a CPU (SURVEY.md 8d / BASELINE.md 4.2): it cannot run on
            # this box (no /root/reference here); tools/time_reference_cpu.py measured it in the build
            # container and the committed record travels with the line, labelled with its host
            try:
                with open(os.path.join(ROOT, "profiles", "r5_reference_cpu_container.json")) as f:
                    cb["reference_in_build_container"] = json.load(f)
            except OSError:
                cb["reference_in_build_container"] = None
            out["cpu_baseline"] = cb
            out["parity_vs_oracle"] = parity
        if not check["ok"]:
            # a step whose replayed / reduced gradients differ from an eager step is not a valid measurement:
            # the line says so where a consumer of `value` alone will see it, exit code 3 below
            out["error"] = f"step_check failed: {check['max_err']}"[:600]
            out["value_unchecked"] = out["value"]
            out["value"] = 0.0
        json_out.write(json.dumps(out) + "\n")
        json_out.flush()
        _STATE["printed"] = True
    P.shutdown(world)
    if not check["ok"]:
        sys.exit(3)


_STATE = {"json_out": None, "printed": False}


def _error_line(err: BaseException) -> None:
    """A rank that fails where the process survives (a Python exception: a peer that died inside a
    collective, an allocation failure, a failed check) still leaves ONE JSON line on rank 0 -- value 0 and an
    `error` field -- so that a scaling run never yields nothing to read.  (A GPU memory fault aborts the
    process from inside the runtime: nothing can be printed then.)"""
    if _STATE["printed"] or int(os.environ.get("RANK", "0")) != 0:
        return
    out = _STATE["json_out"] or sys.stdout
    args = sys.argv[1:]
    n = int(args[args.index("--gpus") + 1]) if "--gpus" in args and args.index("--gpus") + 1 < len(args) else 1
    out.write(json.dumps({
        "metric": "rendered views/sec (fwd+bwd)", "value": 0.0, "unit": "views/s",
        "n_gpus": int(os.environ.get("WORLD_SIZE", n)), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "error": f"{type(err).__name__}: {err}"[:600], "argv": args}) + "\n")
    out.flush()


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as err:      # noqa: BLE001 -- reported, then re-raised
        _error_line(err)
        raise
