"""CPU checks around tests/golden/decoder.npz (the reference's decoder host glue, recorded):
the host-side camera conventions the product keeps in PyTorch (pixelsplat_amd/geometry.py, used
by the orthographic path and by tests/cases.py) equal the reference's recorded settings, the
oracle reproduces the golden images from the recorded calls (regression pin of the oracle), and
-- in the build container, where /root/reference exists -- the golden can be regenerated
bit for bit."""
import os

import numpy as np
import pytest
import torch

from oracle import raster_ref as R
from oracle import ref_import
from tests.cases import decoder_golden, oracle_view_inputs, reference_cameras


def test_geometry_py_equals_reference_settings():
    from pixelsplat_amd.geometry import camera_matrices
    z = decoder_golden()
    ext = torch.from_numpy(z["dec_ext"]).reshape(-1, 4, 4).clone()
    intr = torch.from_numpy(z["dec_intr"]).reshape(-1, 3, 3)
    near, far = torch.from_numpy(z["dec_near"]).reshape(-1), torch.from_numpy(z["dec_far"]).reshape(-1)
    scale = 1 / near
    ext[:, :3, 3] *= scale[:, None]
    tanfov, view_t, full_t, campos = camera_matrices(ext, intr, near * scale, far * scale)
    s = z["dec_settings"]
    assert np.array_equal(view_t.reshape(-1, 16).numpy(), s[:, 0:16])
    assert np.array_equal(full_t.reshape(-1, 16).numpy(), s[:, 16:32])
    assert np.array_equal(campos.numpy(), s[:, 32:35])
    assert np.array_equal(tanfov.numpy(), s[:, 35:37])
    assert np.array_equal(scale.numpy(), s[:, 40])


@pytest.mark.parametrize("name", ["c1_64", "c2_256", "c4_256_v3", "c5_512"])
def test_synthetic_cameras_match_the_recorded_settings(name):
    """tests/cases.oracle_view_inputs' own camera math (no `view_params`) == the reference's
    recorded settings for the cameras of the full-size parity configurations."""
    from pixelsplat_amd.synthetic import make_cameras
    from types import SimpleNamespace
    kw, vp = reference_cameras(name)
    _, tgt = make_cameras(kw["b"], kw["v_ctx"], kw["v_tgt"], kw["hw"],
                          torch.Generator().manual_seed(kw["seed"]))
    g = SimpleNamespace(means=torch.zeros(1, 1, 3), covariances=torch.eye(3).reshape(1, 1, 3, 3),
                        harmonics=torch.zeros(1, 1, 3, 25), opacities=torch.ones(1, 1))
    for v in range(vp.shape[0]):
        inp = oracle_view_inputs(g, tgt, 0, v)
        assert np.array_equal(inp["view"].reshape(16), vp[v, 0:16])
        assert np.array_equal(inp["proj"].reshape(16), vp[v, 16:32])
        assert np.array_equal(inp["campos"], vp[v, 32:35])
        assert inp["tanfovx"] == float(vp[v, 35]) and inp["tanfovy"] == float(vp[v, 36])


def test_oracle_reproduces_the_golden_images():
    z = decoder_golden()
    hw = tuple(int(x) for x in z["dec_hw"])
    b, v = z["dec_near"].shape
    row, col = np.triu_indices(3)
    for bi in range(b):
        for vi in range(v):
            s = z["dec_settings"][bi * v + vi]
            scale = np.float32(s[40])
            st = R.forward(
                means=z["dec_means"][bi] * scale, cov6=(z["dec_cov"][bi] * scale ** 2)[:, row, col],
                opacity=z["dec_op"][bi], view=s[0:16], proj=s[16:32], campos=s[32:35], bg=s[37:40],
                H=hw[0], W=hw[1], tanfovx=float(s[35]), tanfovy=float(s[36]),
                sh=np.ascontiguousarray(z["dec_sh"][bi].transpose(0, 2, 1)), sh_degree=4)
            assert np.array_equal(st.image, z["dec_color"][bi, vi])
            assert np.array_equal(st.radii, z["dec_radii"][bi, vi])
            assert np.array_equal(R.ambiguity_mask(st), z["dec_color_ambiguous"][bi, vi])
    # the splatter fixture: a symmetric blob whose red channel carries the l = 2 band
    f = 0
    s = z["splat_settings"][f]
    st = R.forward(means=z["splat_means3D"][f], cov6=z["splat_cov6"][f],
                   opacity=z["splat_opacities"][f][:, 0], view=s[0:16], proj=s[16:32],
                   campos=s[32:35], bg=s[37:40], H=512, W=512, tanfovx=float(s[35]),
                   tanfovy=float(s[36]), sh=z["splat_shs"][f], sh_degree=4)
    np.testing.assert_allclose(st.image.astype(np.float64).sum((1, 2)), z["splat_image_sums"][f], rtol=1e-9)
    assert st.radii[0] == z["splat_radii"][f, 0] == 77   # 3 sigma, sigma = 256 px / 10 = 25.6 px
    assert np.abs(st.image - z["splat_frames"][0].astype(np.float32)).max() < 2e-2
    # green and blue see only the +0.5 offset: 0.5 * alpha, alpha_max = 0.99 at the centre
    assert abs(float(st.image[1].max()) - 0.495) < 1e-6


@pytest.mark.skipif(not ref_import.available(), reason="needs /root/reference (build container)")
def test_golden_regenerates_from_the_live_reference(tmp_path):
    """Re-runs the generator's small case against the live reference and compares with the
    committed file (settings, images, depth modes)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "make_decoder_golden", os.path.join(os.path.dirname(__file__), "golden", "make_decoder_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = {}
    mod.small_case(out)
    z = decoder_golden()
    for k in ("dec_settings", "dec_color", "dec_depth_depth", "dec_depth_relative_disparity",
              "ortho_settings", "ortho_color", "raw_settings", "raw_color"):
        assert np.array_equal(out[k], z[k]), k
