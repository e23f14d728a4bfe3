"""Generates tests/golden/decoder.npz by running the REFERENCE's decoder host code, unmodified,
on the CPU in the build container (oracle/ref_import.decoder_modules):

  DecoderSplattingCUDA.forward / .render_depth   src/model/decoder/decoder_splatting_cuda.py:35-91
  render_cuda / render_depth_cuda (4 modes) /
  render_cuda_orthographic / get_projection_matrix src/model/decoder/cuda_splatting.py:17-269
  get_fov                                          src/geometry/projection.py:233-247
  the rasterizer fixture                           src/scripts/test_splatter.py:21-101
                                                   src/visualization/camera_trajectory/spin.py:9-37

The third-party rasterizer those functions call is absent (requirements.txt:17); a RECORDING
stand-in (oracle/ref_import.RasterizerRecorder) captures every per-view call exactly as the
reference makes it -- settings (tanfov, transposed view / full-projection matrices, campos, bg,
sh_degree) and arguments (renormed means, upper-triangle covariances, [G,K,3] SH or [G,3]
colours, [G,1] opacities) -- and renders it with oracle/raster_ref.c.  What is stored:

  dec_*      a small two-scene case through DecoderSplattingCUDA (colour + the 4 depth modes +
             the orthographic path): inputs, recorded per-view settings packed as the product's
             [V,48] view block, recorded arguments of view 0, images, ambiguity masks, radii
  cam_<cfg>  the recorded settings for the target cameras of the full-size parity tests
             (BASELINE configs[0], [1], [3], [4]): the oracle AND the product are fed from these
  splat_*    scripts/test_splatter.py: 1 Gaussian, degree-4 SH, 60-frame spin, 512x512: recorded
             settings + rotated SH per frame and the oracle's per-frame channel sums

    python tests/golden/make_decoder_golden.py
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from pixelsplat_amd.synthetic import make_cameras, make_workload  # noqa: E402

# PS_VIEW_* offsets of include/pixelsplat_hip.h
VIEW, PROJ, CAMPOS, TANX, TANY, BG, SCALE, STRIDE = 0, 16, 32, 35, 36, 37, 40, 48

# full-size parity configurations: name -> (b, (h, w), v_ctx, v_tgt, seed)  (tests/cases.py)
CAMERA_CONFIGS = {
    "c1_64": (1, (64, 64), 2, 4, 0),
    "c2_256": (1, (256, 256), 2, 4, 0),
    "c2_256_s1": (1, (256, 256), 2, 4, 1),
    "c4_256_v3": (1, (256, 256), 3, 4, 0),
    "c5_512": (1, (512, 512), 2, 4, 0),
    # the BENCHMARKED launch of configs[1]: all 7 scenes x 4 views of the batch in one call
    "c2_256_b7": (7, (256, 256), 2, 4, 0),
}


def pack(calls, scale):
    """Recorded per-view settings -> [V,48] in the product's view-block layout."""
    out = np.zeros((len(calls), STRIDE), np.float32)
    for i, c in enumerate(calls):
        out[i, VIEW:VIEW + 16] = c["viewmatrix"].reshape(16)
        out[i, PROJ:PROJ + 16] = c["projmatrix"].reshape(16)
        out[i, CAMPOS:CAMPOS + 3] = c["campos"]
        out[i, TANX], out[i, TANY] = c["tanfovx"], c["tanfovy"]
        out[i, BG:BG + 3] = c["bg"]
        out[i, SCALE] = scale[i]
    return out


def rot(axis, deg):
    a = np.deg2rad(deg)
    c, s = np.cos(a), np.sin(a)
    m = np.eye(4, dtype=np.float32)
    i, j = [(1, 2), (0, 2), (0, 1)][axis]
    m[i, i], m[i, j], m[j, i], m[j, j] = c, -s, s, c
    return torch.from_numpy(m)


def small_case(out):
    rec = ref_import.RasterizerRecorder(render=True)
    m = ref_import.decoder_modules(rec)
    b, v, hw = 2, 3, (48, 64)
    ctx, _, g, _ = make_workload(b, (16, 16), v_ctx=2, v_tgt=v, seed=3)
    gen = torch.Generator().manual_seed(7)
    ext = torch.eye(4).repeat(b, v, 1, 1)
    for bi in range(b):
        for vi in range(v):
            r = rot(1, float(torch.empty(1).uniform_(-8, 8, generator=gen))) @ \
                rot(0, float(torch.empty(1).uniform_(-6, 6, generator=gen))) @ \
                rot(2, float(torch.empty(1).uniform_(-10, 10, generator=gen)))
            ext[bi, vi] = r
            ext[bi, vi, :3, 3] = torch.tensor([0.2 + 0.3 * vi, 0.05 * bi - 0.03, -0.1 * vi])
    intr = torch.eye(3).repeat(b, v, 1, 1)
    intr[..., 0, 0] = torch.tensor([0.80, 0.90, 1.05])      # fx != fy, off-centre principal
    intr[..., 1, 1] = torch.tensor([1.10, 0.95, 0.85])      # point: pins get_fov's convention
    intr[..., 0, 2] = 0.47
    intr[..., 1, 2] = 0.52
    near = torch.tensor([[0.35, 0.30, 0.40], [0.25, 0.45, 0.33]])
    far = torch.tensor([[300.0, 455.0, 200.0], [150.0, 500.0, 380.0]])
    bg = [0.1, 0.4, 0.7]
    dec = m.decoder_cuda.DecoderSplattingCUDA(
        m.decoder_cuda.DecoderSplattingCUDACfg("splatting_cuda"),
        SimpleNamespace(background_color=bg))
    gs = m.types.Gaussians(g.means, g.covariances, g.harmonics, g.opacities)

    def images(n0):
        calls = rec.calls[n0:]
        return (np.stack([c["image"] for c in calls]).reshape(b, v, 3, *hw),
                np.stack([c["ambiguous"] for c in calls]).reshape(b, v, *hw), calls)

    n0 = len(rec.calls)
    res = dec.forward(gs, ext, intr, near, far, hw)
    color, amb, calls = images(n0)
    assert np.array_equal(res.color.numpy(), color)
    scale = (1 / near).reshape(-1).numpy()
    out.update(dec_means=g.means.numpy(), dec_cov=g.covariances.numpy(), dec_sh=g.harmonics.numpy(),
               dec_op=g.opacities.numpy(), dec_ext=ext.numpy(), dec_intr=intr.numpy(),
               dec_near=near.numpy(), dec_far=far.numpy(), dec_bg=np.array(bg, np.float32),
               dec_hw=np.array(hw), dec_settings=pack(calls, scale), dec_color=color,
               dec_color_ambiguous=amb,
               dec_radii=np.stack([c["radii"] for c in calls]).reshape(b, v, -1),
               # the reference's per-view arguments for view (0, 0), as handed to the rasterizer
               # (first 256 Gaussians: enough to pin scaling, layouts and the triangle order)
               dec_args_means3D=calls[0]["means3D"][:256], dec_args_cov6=calls[0]["cov3D_precomp"][:256],
               dec_args_shs=calls[0]["shs"][:256], dec_args_opacities=calls[0]["opacities"][:256],
               dec_args_sh_degree=np.array(calls[0]["sh_degree"]),
               dec_args_campos_stride=np.array(calls[0]["campos_stride"]))
    assert calls[0]["colors_precomp"] is None and calls[0]["means2D_requires_grad"]
    for mode in ("depth", "log", "disparity", "relative_disparity"):
        n0 = len(rec.calls)
        d = dec.render_depth(gs, ext, intr, near, far, hw, mode)
        img, amb, calls = images(n0)
        assert calls[0]["shs"] is None and calls[0]["sh_degree"] == 0
        assert np.allclose(d.numpy(), img.mean(2))
        out[f"dec_depth_{mode}"] = d.numpy()
        out[f"dec_depth_{mode}_ambiguous"] = amb
        if mode == "depth":
            out["dec_depth_colors_view0"] = calls[0]["colors_precomp"][:256]
            assert np.array_equal(pack(calls, scale)[:, :37], out["dec_settings"][:, :37])
    # forward(depth_mode=...) is forward + render_depth
    res = dec.forward(gs, ext, intr, near, far, hw, depth_mode="disparity")
    assert np.array_equal(res.depth.numpy(), out["dec_depth_disparity"])

    # orthographic visualisation path (cuda_splatting.py:130-220).  The reference function only
    # runs with batch 1 (`move_back[2, 3] = -distance_to_near`, :164, needs one element -- its
    # callers, validation_in_3d.py:68 and the paper figures, all pass batch 1): 3 calls of 1 view
    width, height = torch.tensor([3.0, 2.5, 4.0]), torch.tensor([2.0, 2.5, 3.0])
    o_near, o_far = torch.tensor([0.0, 0.5, 0.2]), torch.tensor([50.0, 80.0, 60.0])
    o_img, o_amb, o_set, o_ext, o_fov = [], [], [], [], []
    for i in range(v):
        n0 = len(rec.calls)
        dump = {}
        img = m.splatting.render_cuda_orthographic(
            ext[0, i:i + 1], width[i:i + 1], height[i:i + 1], o_near[i:i + 1], o_far[i:i + 1], hw,
            torch.tensor([bg]), g.means[:1], g.covariances[:1], g.harmonics[:1], g.opacities[:1],
            fov_degrees=[0.1, 10.0, 1.0][i], dump=dump)
        (call,) = rec.calls[n0:]
        o_img.append(img[0].numpy())
        o_amb.append(call["ambiguous"])
        o_set.append(pack([call], np.ones(1, np.float32))[0])
        o_ext.append(dump["extrinsics"][0].numpy())
        o_fov.append([float(dump["fov_x"]), float(dump["fov_y"])])
    out.update(ortho_width=width.numpy(), ortho_height=height.numpy(), ortho_near=o_near.numpy(),
               ortho_far=o_far.numpy(), ortho_fov_degrees=np.array([0.1, 10.0, 1.0], np.float32),
               ortho_settings=np.stack(o_set), ortho_color=np.stack(o_img),
               ortho_ambiguous=np.stack(o_amb), ortho_dump_extrinsics=np.stack(o_ext),
               ortho_dump_fov=np.array(o_fov, np.float32))

    # scale_invariant=False, use_sh=False through render_cuda directly (flattened views)
    n0 = len(rec.calls)
    colors = torch.rand(b * v, g.means.shape[1], 3, 1, generator=gen)
    repv = lambda t: t.repeat_interleave(v, 0)
    img = m.splatting.render_cuda(
        ext.reshape(-1, 4, 4), intr.reshape(-1, 3, 3), near.reshape(-1), far.reshape(-1), hw,
        torch.tensor(bg).expand(b * v, 3), repv(g.means), repv(g.covariances), colors,
        repv(g.opacities), scale_invariant=False, use_sh=False)
    calls = rec.calls[n0:]
    out.update(raw_colors=colors.numpy(), raw_settings=pack(calls, np.ones(b * v, np.float32)),
               raw_color=img.numpy(), raw_ambiguous=np.stack([c["ambiguous"] for c in calls]))


def camera_configs(out):
    """Recorded settings of the reference's render_cuda for the synthetic target cameras of the
    full-size parity tests (one dummy Gaussian: only the host glue matters here)."""
    rec = ref_import.RasterizerRecorder(render=False)
    m = ref_import.decoder_modules(rec)
    for name, (b, hw, v_ctx, v_tgt, seed) in CAMERA_CONFIGS.items():
        _, tgt = make_cameras(b, v_ctx, v_tgt, hw, torch.Generator().manual_seed(seed))
        V = b * v_tgt
        n0 = len(rec.calls)
        m.splatting.render_cuda(
            tgt.extrinsics.reshape(V, 4, 4), tgt.intrinsics.reshape(V, 3, 3), tgt.near.reshape(V),
            tgt.far.reshape(V), hw, torch.zeros(V, 3), torch.zeros(V, 1, 3),
            torch.eye(3).expand(V, 1, 3, 3), torch.zeros(V, 1, 3, 25), torch.ones(V, 1))
        out[f"cam_{name}"] = pack(rec.calls[n0:], (1 / tgt.near).reshape(V).numpy())
        out[f"cam_{name}_def"] = np.array([b, hw[0], hw[1], v_ctx, v_tgt, seed])


def splatter_fixture(out):
    """src/scripts/test_splatter.py:21-101, line for line, on the CPU (R.random seeded)."""
    from scipy.spatial.transform import Rotation as R

    rec = ref_import.RasterizerRecorder(render=True)
    m = ref_import.decoder_modules(rec)
    NUM_FRAMES, NUM_GAUSSIANS, DEGREE, IMAGE_SHAPE = 60, 1, 4, (512, 512)
    device = torch.device("cpu")
    extrinsics = m.spin.generate_spin(60, device, 0.0, 10.0)
    intrinsics = torch.eye(3, dtype=torch.float32)
    intrinsics[:2, 2] = 0.5
    intrinsics[:2, :2] *= 0.5
    intrinsics = intrinsics.expand(NUM_FRAMES, 3, 3)
    means = torch.zeros((NUM_GAUSSIANS, 3))
    scales = torch.ones((NUM_GAUSSIANS, 3))
    rotations = torch.tensor(R.random(NUM_GAUSSIANS, random_state=0).as_matrix(), dtype=torch.float32)
    covariances = rotations @ scales.diag_embed()
    covariances = torch.einsum("bij,bkj->bik", covariances, covariances)
    sh = torch.zeros((NUM_GAUSSIANS, 3, (DEGREE + 1) ** 2))
    sh[:, 0, 4:9] = 10
    opacities = torch.ones(NUM_GAUSSIANS)
    for c2w, k in zip(extrinsics, intrinsics):
        m.splatting.render_cuda(
            c2w[None], k[None], torch.tensor([0.1]), torch.tensor([20.0]), IMAGE_SHAPE,
            torch.zeros((1, 3)), means[None], covariances[None],
            m.sh_rotation.rotate_sh(sh, c2w[:3, :3])[None], opacities[None])
    calls = rec.calls
    assert len(calls) == NUM_FRAMES
    out.update(
        splat_extrinsics=extrinsics.numpy(), splat_covariance=covariances.numpy(),
        splat_sh_unrotated=sh.numpy(),
        splat_settings=pack(calls, np.full(NUM_FRAMES, 10.0, np.float32)),
        splat_means3D=np.stack([c["means3D"] for c in calls]),       # renormed (x 1/near)
        splat_cov6=np.stack([c["cov3D_precomp"] for c in calls]),
        splat_shs=np.stack([c["shs"] for c in calls]),               # [60,1,25,3], rotated
        splat_opacities=np.stack([c["opacities"] for c in calls]),
        splat_radii=np.stack([c["radii"] for c in calls]),
        splat_image_sums=np.stack([c["image"].astype(np.float64).sum((1, 2)) for c in calls]),
        splat_image_max=np.stack([c["image"].max((1, 2)) for c in calls]),
        # frames 0 and 7 in full (quarter turn apart in SH lobes), fp16 to stay small
        splat_frames=np.stack([calls[i]["image"] for i in (0, 7)]).astype(np.float16))


def main():
    out = {}
    small_case(out)
    camera_configs(out)
    splatter_fixture(out)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "decoder.npz")
    np.savez_compressed(path, **out)
    print(path, f"{os.path.getsize(path) / 1e6:.2f} MB", len(out), "arrays")


if __name__ == "__main__":
    main()
