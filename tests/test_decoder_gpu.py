"""The product's decoder API against the REFERENCE's decoder host code (a10, a11, a12 of
SURVEY.md 8a).  tests/golden/decoder.npz holds what the reference's unmodified
`DecoderSplattingCUDA.forward / .render_depth`, `render_cuda`, `render_depth_cuda` (4 modes) and
`render_cuda_orthographic` (decoder_splatting_cuda.py:35-91, cuda_splatting.py:17-269) hand to
the rasterizer -- recorded by a stand-in for the absent third-party module -- and the images the
oracle rasterizer makes of those calls (tests/golden/make_decoder_golden.py).  Also the one
rasterizer fixture the reference ships: scripts/test_splatter.py:21-101."""
import numpy as np
import pytest
import torch

from oracle import raster_ref as R
from tests.cases import decoder_golden

pytestmark = pytest.mark.gpu

IMG_TOL = 1e-4


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _inputs(z, dev):
    from types import SimpleNamespace
    g = SimpleNamespace(means=_t(z["dec_means"], dev), covariances=_t(z["dec_cov"], dev),
                        harmonics=_t(z["dec_sh"], dev), opacities=_t(z["dec_op"], dev))
    cams = (_t(z["dec_ext"], dev), _t(z["dec_intr"], dev), _t(z["dec_near"], dev), _t(z["dec_far"], dev))
    return g, cams, tuple(int(x) for x in z["dec_hw"])


def _decoder(z, dev):
    from types import SimpleNamespace
    from pixelsplat_amd.decoder import DecoderSplattingCUDA, DecoderSplattingCUDACfg
    return DecoderSplattingCUDA(DecoderSplattingCUDACfg("splatting_cuda"),
                                SimpleNamespace(background_color=z["dec_bg"].tolist())).to(dev)


def test_camera_setup_vs_reference_settings(gpu_device):
    """ps_camera_setup == the settings the reference built (renorm, get_fov, get_projection_matrix,
    transposed view / full projection, campos: cuda_splatting.py:64-87, :110)."""
    from pixelsplat_amd.decoder import camera_setup
    z = decoder_golden()
    b, v = z["dec_near"].shape
    ext, intr = _t(z["dec_ext"], gpu_device).reshape(b * v, 4, 4), _t(z["dec_intr"], gpu_device).reshape(b * v, 3, 3)
    near, far = _t(z["dec_near"], gpu_device).reshape(-1), _t(z["dec_far"], gpu_device).reshape(-1)
    bg = _t(z["dec_bg"], gpu_device).expand(b * v, 3)
    vp = camera_setup(ext, intr, near, far, bg).cpu().numpy()
    ref = z["dec_settings"]
    np.testing.assert_allclose(vp[:, :41], ref[:, :41], rtol=3e-6, atol=3e-6)
    # scale_invariant=False leaves the pose, near and far alone
    vp = camera_setup(ext, intr, near, far, bg, scale_invariant=False).cpu().numpy()
    np.testing.assert_allclose(vp[:, :41], z["raw_settings"][:, :41], rtol=3e-6, atol=3e-6)


def test_decoder_forward_vs_reference(gpu_device):
    """DecoderSplattingCUDA.forward (colour) == reference glue + oracle rasterizer, and its
    gradients reach the Gaussians (the v-fold `repeat` of the reference is a sum over views)."""
    z = decoder_golden()
    g, (ext, intr, near, far), hw = _inputs(z, gpu_device)
    dec = _decoder(z, gpu_device)
    for t in (g.means, g.covariances, g.harmonics, g.opacities):
        t.requires_grad_(True)
    out = dec(g, ext, intr, near, far, hw)
    assert out.depth is None and out.color.shape == z["dec_color"].shape
    ok = z["dec_color_ambiguous"] == 0
    err = np.abs(out.color.detach().cpu().numpy() - z["dec_color"]).max(2)
    assert err[ok].max() <= IMG_TOL, float(err[ok].max())
    out.color.square().sum().backward()
    assert all(torch.isfinite(t.grad).all() and t.grad.abs().sum() > 0
               for t in (g.means, g.covariances, g.harmonics, g.opacities))


@pytest.mark.parametrize("mode", ["depth", "log", "disparity", "relative_disparity"])
def test_render_depth_modes_vs_reference(gpu_device, mode):
    """DecoderSplattingCUDA.render_depth / forward(depth_mode=...) == the reference's
    render_depth_cuda (cuda_splatting.py:226-269; `colors_precomp`, sh_degree 0, bg 0, channel
    mean) + oracle rasterizer.  Tolerance: 1e-4 relative to the largest rendered value."""
    z = decoder_golden()
    g, (ext, intr, near, far), hw = _inputs(z, gpu_device)
    dec = _decoder(z, gpu_device)
    ref = z[f"dec_depth_{mode}"]
    ok = z[f"dec_depth_{mode}_ambiguous"] == 0
    scale = max(1.0, float(np.abs(ref).max()))
    d = dec.render_depth(g, ext, intr, near, far, hw, mode).cpu().numpy()
    assert d.shape == ref.shape
    assert (np.abs(d - ref)[ok] / scale).max() <= IMG_TOL
    out = dec(g, ext, intr, near, far, hw, depth_mode=mode)
    assert np.array_equal(out.depth.cpu().numpy(), d)


def test_render_cuda_without_sh_and_renorm(gpu_device):
    """render_cuda(scale_invariant=False, use_sh=False), flattened views with their own Gaussians
    (the reference's calling convention: views_per_scene = 1)."""
    from pixelsplat_amd.decoder import render_cuda
    z = decoder_golden()
    g, (ext, intr, near, far), hw = _inputs(z, gpu_device)
    b, v = near.shape
    rep = lambda t: t.repeat_interleave(v, 0)
    img = render_cuda(ext.reshape(-1, 4, 4), intr.reshape(-1, 3, 3), near.reshape(-1), far.reshape(-1),
                      hw, _t(z["dec_bg"], gpu_device).expand(b * v, 3), rep(g.means),
                      rep(g.covariances), _t(z["raw_colors"], gpu_device), rep(g.opacities),
                      scale_invariant=False, use_sh=False).cpu().numpy()
    ok = z["raw_ambiguous"] == 0
    err = np.abs(img - z["raw_color"]).max(1)
    assert err[ok].max() <= IMG_TOL


def test_orthographic_vs_reference(gpu_device):
    """render_cuda_orthographic (cuda_splatting.py:130-220): the fake-orthographic camera of the
    3-D validation plots; the reference runs it with batch 1, so do we here -- one call per view
    -- plus one batched call (the product's version broadcasts)."""
    from pixelsplat_amd.decoder import render_cuda_orthographic
    z = decoder_golden()
    g, (ext, intr, near, far), hw = _inputs(z, gpu_device)
    dev = gpu_device
    bg = _t(z["dec_bg"], dev)[None]
    for i in range(3):
        dump = {}
        img = render_cuda_orthographic(
            ext[0, i:i + 1], _t(z["ortho_width"][i:i + 1], dev), _t(z["ortho_height"][i:i + 1], dev),
            _t(z["ortho_near"][i:i + 1], dev), _t(z["ortho_far"][i:i + 1], dev), hw, bg,
            g.means[:1], g.covariances[:1], g.harmonics[:1], g.opacities[:1],
            fov_degrees=float(z["ortho_fov_degrees"][i]), dump=dump)
        np.testing.assert_allclose(dump["extrinsics"][0].cpu().numpy(), z["ortho_dump_extrinsics"][i],
                                   rtol=1e-5, atol=1e-3)
        assert abs(float(dump["fov_y"]) - z["ortho_dump_fov"][i, 1]) < 1e-6
        ok = z["ortho_ambiguous"][i] == 0
        err = np.abs(img[0].cpu().numpy() - z["ortho_color"][i]).max(0)
        # the camera sits up to ~1700 units back (0.1 degree field of view): fp32 view-space
        # positions carry ~1e-4 absolute error on both sides, hence the wider image tolerance
        assert err[ok].max() <= 2e-3, (i, float(err[ok].max()))


def test_dropin_module_fed_like_the_reference(gpu_device):
    """`diff_gaussian_rasterization` (the drop-in for the third-party module) called exactly as
    the reference's loop calls it (cuda_splatting.py:91-124) for view (0, 0): renormed means,
    upper-triangle covariances, [G,K,3] SH, [G,1] opacities, non-contiguous campos.  The
    recorded arguments pin that reconstruction; the result equals the golden image and the
    batched path's image of the same view."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    z = decoder_golden()
    g, (ext, intr, near, far), hw = _inputs(z, gpu_device)
    dev = gpu_device
    s = z["dec_settings"][0]
    scale = 1 / near[0, 0]
    means3D = g.means[0] * scale
    cov = g.covariances[0] * scale ** 2
    row, col = torch.triu_indices(3, 3)
    cov6 = cov[:, row, col]
    shs = g.harmonics[0].permute(0, 2, 1).contiguous()
    op = g.opacities[0][:, None]
    n = z["dec_args_means3D"].shape[0]
    assert np.array_equal(means3D[:n].cpu().numpy(), z["dec_args_means3D"])
    assert np.array_equal(cov6[:n].cpu().numpy(), z["dec_args_cov6"])
    assert np.array_equal(shs[:n].cpu().numpy(), z["dec_args_shs"])
    assert np.array_equal(op[:n].cpu().numpy(), z["dec_args_opacities"])
    e2 = ext[0, 0].clone()
    e2[:3, 3] *= scale
    settings = GaussianRasterizationSettings(
        image_height=hw[0], image_width=hw[1], tanfovx=float(s[35]), tanfovy=float(s[36]),
        bg=_t(z["dec_bg"], dev), scale_modifier=1.0, viewmatrix=_t(s[0:16], dev).reshape(4, 4),
        projmatrix=_t(s[16:32], dev).reshape(4, 4), sh_degree=int(z["dec_args_sh_degree"]),
        campos=e2[:3, 3], prefiltered=False, debug=False)
    assert tuple(settings.campos.stride()) == tuple(z["dec_args_campos_stride"])
    m2d = torch.zeros_like(means3D, requires_grad=True)
    image, radii = GaussianRasterizer(settings)(
        means3D=means3D, means2D=m2d, shs=shs, colors_precomp=None, opacities=op,
        cov3D_precomp=cov6)
    assert np.array_equal(radii.cpu().numpy(), z["dec_radii"][0, 0])
    ok = z["dec_color_ambiguous"][0, 0] == 0
    err = np.abs(image.detach().cpu().numpy() - z["dec_color"][0, 0]).max(0)
    assert err[ok].max() <= IMG_TOL
    # the batched product path fed the same recorded settings: same kernels, same bits
    from pixelsplat_amd.decoder import render_cuda
    b, v = near.shape
    batched = render_cuda(ext.reshape(-1, 4, 4), intr.reshape(-1, 3, 3), near.reshape(-1),
                          far.reshape(-1), hw, _t(z["dec_bg"], dev).expand(b * v, 3), g.means,
                          g.covariances, g.harmonics, g.opacities, views_per_scene=v,
                          view_params=_t(z["dec_settings"], dev))
    assert torch.equal(batched[0], image.detach())


def test_splatter_fixture(gpu_device):
    """scripts/test_splatter.py:21-101 -- the only rasterizer fixture the reference holds: ONE
    Gaussian (covariance R R^T), degree-4 SH with the l = 2 band of the red channel = 10, rotated
    into each frame by rotate_sh, a 60-frame spin at radius 10 (spin.py:9-37), K = diag(.5, .5),
    near 0.1 / far 20, 512x512, black background.  The recorded per-frame calls are rendered by
    the drop-in module and by the oracle (live; its channel sums are pinned by the golden)."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    z = decoder_golden()
    dev = gpu_device
    worst = 0.0
    for f in range(60):
        s = z["splat_settings"][f]
        st = R.forward(means=z["splat_means3D"][f], cov6=z["splat_cov6"][f],
                       opacity=z["splat_opacities"][f][:, 0], view=s[0:16], proj=s[16:32],
                       campos=s[32:35], bg=s[37:40], H=512, W=512, tanfovx=float(s[35]),
                       tanfovy=float(s[36]), sh=z["splat_shs"][f], sh_degree=4)
        np.testing.assert_allclose(st.image.astype(np.float64).sum((1, 2)), z["splat_image_sums"][f],
                                   rtol=1e-9)
        settings = GaussianRasterizationSettings(
            image_height=512, image_width=512, tanfovx=float(s[35]), tanfovy=float(s[36]),
            bg=_t(s[37:40], dev), scale_modifier=1.0, viewmatrix=_t(s[0:16], dev).reshape(4, 4),
            projmatrix=_t(s[16:32], dev).reshape(4, 4), sh_degree=4, campos=_t(s[32:35], dev),
            prefiltered=False, debug=False)
        means = _t(z["splat_means3D"][f], dev)
        image, radii = GaussianRasterizer(settings)(
            means3D=means, means2D=torch.zeros_like(means), shs=_t(z["splat_shs"][f], dev),
            colors_precomp=None, opacities=_t(z["splat_opacities"][f], dev),
            cov3D_precomp=_t(z["splat_cov6"][f], dev))
        assert np.array_equal(radii.cpu().numpy(), z["splat_radii"][f])
        ok = R.ambiguity_mask(st) == 0
        err = np.abs(image.cpu().numpy() - st.image).max(0)
        # colours reach ~9 here (SH coefficients of 10): 1e-4 relative to the frame's maximum
        tol = IMG_TOL * max(1.0, float(z["splat_image_max"][f].max()))
        assert err[ok].max() <= tol, (f, float(err[ok].max()))
        worst = max(worst, float(err[ok].max()))
        if f in (0, 7):
            assert np.abs(image.cpu().numpy() - z["splat_frames"][(0, 7).index(f)].astype(np.float32)).max() < 2e-2
    print(f"\n[test_splatter] worst |image - oracle| over 60 frames: {worst:.2e}")
