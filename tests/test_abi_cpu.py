"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, and
exports every symbol include/pixelsplat_hip.h declares; argument validation and the
no-fallback rule of the product path.  No compute calls (there is no GPU here)."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from pixelsplat_amd import build as hip_build
    hip_build.build()
    from pixelsplat_amd import _lib
    return _lib.load()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "pixelsplat_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(ps_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 13
    for n in sorted(names):
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    from pixelsplat_amd import _lib
    assert set(_lib.EXPORTS) == names


def test_desc_defaults_and_sizes(lib):
    from pixelsplat_amd import _lib
    d = _lib.default_desc()
    assert abs(d.near_cull - 0.2) < 1e-7 and abs(d.alpha_min - 1 / 255) < 1e-9
    assert lib.ps_raster_state_bytes(C.byref(d)) == 0  # G = 0: invalid descriptor
    d.n_scenes, d.views_per_scene, d.n_gaussians, d.height, d.width = 7, 4, 393216, 256, 256
    d.sh_degree, d.sh_coeffs = 4, 25
    state, temp = lib.ps_raster_state_bytes(C.byref(d)), lib.ps_raster_temp_bytes(C.byref(d))
    n = 28 * 393216
    assert state >= n * (48 + 8 + 4 + 8 + 1) and state < n * 96 + (1 << 24)
    assert temp >= n * 16
    bwd = lib.ps_raster_backward_temp_bytes(C.byref(d), 14_000_000)
    assert bwd >= n * 36 + n * 4 * 48 + n * 12   # accumulators, per-(Gaussian, tile) slots, dL/dRGB
    lay = _lib.PsRasterStateLayout()
    assert lib.ps_raster_state_layout(C.byref(d), C.byref(lay)) == 0
    offs = [lay.records, lay.rects, lay.sorted_idx, lay.sorted_rect, lay.n_vis, lay.final_T,
            lay.n_contrib, lay.tile_end, lay.tile_ranges, lay.num_rendered, lay.total]
    assert offs == sorted(offs) and all(o % 256 == 0 for o in offs)


def test_abi_version_and_optional_flags(lib):
    """Round 5: the library reports the layout version of its descriptor structs (PsEpipolarDesc
    grew without a guard) and the header, the loader and the library agree on it; PS_FLAG_DETERMINISTIC is
    accounted for in the backward scratch size (48 B per list entry + 4 B per (view, Gaussian))."""
    from pixelsplat_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "pixelsplat_hip.h")).read()
    ver = int(re.search(r"#define\s+PS_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert lib.ps_abi_version() == ver == _lib.PS_ABI_VERSION
    assert int(re.search(r"#define\s+PS_FLAG_DETERMINISTIC\s+(\d+)", hdr).group(1)) == _lib.PS_FLAG_DETERMINISTIC
    d = _lib.default_desc()
    d.n_scenes, d.views_per_scene, d.n_gaussians, d.height, d.width = 1, 4, 393216, 256, 256
    d.sh_degree, d.sh_coeffs = 4, 25
    cap = 2_000_000
    plain = lib.ps_raster_backward_temp_bytes(C.byref(d), cap)
    d.flags |= _lib.PS_FLAG_DETERMINISTIC
    det = lib.ps_raster_backward_temp_bytes(C.byref(d), cap)
    assert det >= plain + cap * 48 + 4 * 393216 * 4 and det < plain + cap * 48 + 4 * 393216 * 4 + 4096


def test_bad_arguments_are_status_codes_not_crashes(lib):
    from pixelsplat_amd import _lib
    d = _lib.default_desc()
    d.n_gaussians, d.height, d.width = 16, 32, 32
    rc = lib.ps_raster_forward(C.byref(d), None, None, None, None, None, None, None, None, None,
                               0, None, 0, None, 0, None)
    assert rc == -1
    assert b"bad argument" in lib.ps_status_string(rc)
    with pytest.raises(RuntimeError):
        _lib.check(rc, "ps_raster_forward")
    assert lib.ps_raster_check(C.byref(d), None, 0, None, None) == -1
    assert lib.ps_profile_group_count() >= 5


def test_product_path_refuses_cpu_tensors(lib):
    """No CPU fallback: the drop-in module raises instead of computing on the host."""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    g = 8
    s = GaussianRasterizationSettings(
        image_height=32, image_width=32, tanfovx=0.5, tanfovy=0.5, bg=torch.zeros(3),
        scale_modifier=1.0, viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0,
        campos=torch.zeros(3), prefiltered=False, debug=False)
    r = GaussianRasterizer(s)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        r(means3D=torch.zeros(g, 3), means2D=torch.zeros(g, 3), opacities=torch.ones(g, 1),
          colors_precomp=torch.ones(g, 3), cov3D_precomp=torch.ones(g, 6))
    with pytest.raises(Exception, match="excatly one"):
        r(means3D=torch.zeros(g, 3), means2D=torch.zeros(g, 3), opacities=torch.ones(g, 1),
          cov3D_precomp=torch.ones(g, 6))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from pixelsplat_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.HipExtensionMissing):
        _lib.load()


def test_graft_entry_build():
    import __graft_entry__ as ge
    ge.build()


def test_gemm_tuning_table_is_well_formed_and_inert_without_a_gpu():
    """The committed TunableOp table for the library GEMMs: validators + one row per shape;
    enable() is a no-op (False) where there is no GPU."""
    import csv

    from pixelsplat_amd import gemm_tuning

    rows = list(csv.reader(open(gemm_tuning.TABLE)))
    validators = {r[1]: r[2] for r in rows if r[0] == "Validator"}
    assert {"PT_VERSION", "HIP_VERSION", "GCN_ARCH_NAME"} <= set(validators)
    assert validators["GCN_ARCH_NAME"].startswith("gfx950")
    entries = [r for r in rows if r[0] != "Validator"]
    assert entries and all(len(r) == 4 and float(r[3]) > 0 for r in entries)
    import torch
    if not torch.cuda.is_available():
        assert gemm_tuning.enable() is False
