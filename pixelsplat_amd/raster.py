"""torch.autograd front-end of the HIP rasterizer (C ABI: include/pixelsplat_hip.h).

PyTorch is plumbing here: device buffers, the current stream, autograd bookkeeping.  All
arithmetic happens in libpixelsplat_hip.so; there is no fallback path.

`rasterize(...)` is the batched operator (S scenes x views_per_scene views in one call).
It replaces, per call, the whole loop of
/root/reference/src/model/decoder/cuda_splatting.py:91-126 (B sequential
`GaussianRasterizer` invocations with host syncs) and the v-fold `repeat` of
/root/reference/src/model/decoder/decoder_splatting_cuda.py:53-56.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch
from torch import Tensor

from . import _lib
from ._lib import (PS_COV_6, PS_COV_33, PS_SH_G3K, PS_SH_GK3, PS_VIEW_BG, PS_VIEW_CAMPOS,
                   PS_VIEW_PROJMATRIX, PS_VIEW_SCALE, PS_VIEW_STRIDE, PS_VIEW_TANFOVX,
                   PS_VIEW_TANFOVY, PS_VIEW_VIEWMATRIX)


def pack_view_params(viewmatrix: Tensor, projmatrix: Tensor, campos: Tensor, tanfov: Tensor,
                     bg: Tensor, scale: Tensor | None = None) -> Tensor:
    """[V,4,4] x2 (transposed/row-vector matrices), [V,3], [V,2], [V,3], [V] -> [V,48].
    Device-side packing: no .item(), no host sync (the reference pulls tan_fov to the host
    per view, cuda_splatting.py:102-103)."""
    v = viewmatrix.shape[0]
    out = torch.zeros((v, PS_VIEW_STRIDE), dtype=torch.float32, device=viewmatrix.device)
    out[:, PS_VIEW_VIEWMATRIX:PS_VIEW_VIEWMATRIX + 16] = viewmatrix.reshape(v, 16)
    out[:, PS_VIEW_PROJMATRIX:PS_VIEW_PROJMATRIX + 16] = projmatrix.reshape(v, 16)
    out[:, PS_VIEW_CAMPOS:PS_VIEW_CAMPOS + 3] = campos
    out[:, PS_VIEW_TANFOVX] = tanfov[:, 0]
    out[:, PS_VIEW_TANFOVY] = tanfov[:, 1]
    out[:, PS_VIEW_BG:PS_VIEW_BG + 3] = bg
    out[:, PS_VIEW_SCALE] = 1.0 if scale is None else scale
    return out


@dataclass
class RasterConfig:
    n_scenes: int
    views_per_scene: int
    n_gaussians: int
    height: int
    width: int
    sh_degree: int
    sh_coeffs: int
    sh_layout: int = PS_SH_GK3
    cov_layout: int = PS_COV_6

    def desc(self) -> _lib.PsRasterDesc:
        d = _lib.default_desc()
        d.n_scenes, d.views_per_scene, d.n_gaussians = (self.n_scenes, self.views_per_scene,
                                                        self.n_gaussians)
        d.height, d.width = self.height, self.width
        d.sh_degree, d.sh_coeffs = self.sh_degree, self.sh_coeffs
        d.sh_layout, d.cov_layout = self.sh_layout, self.cov_layout
        return d


def _p(t: Tensor | None):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _check_dev(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("pixelsplat_amd.rasterize needs GPU tensors: the HIP kernels are "
                               "the only implementation (no CPU fallback)")
        if t.dtype not in (torch.float32, torch.int32):
            raise RuntimeError(f"expected float32 tensors, got {t.dtype}")


class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg: RasterConfig, means, cov, opacity, sh, colors, view_params, means2d):
        lib = _lib.load()
        _check_dev(means, cov, opacity, sh, colors, view_params)
        means, cov, opacity = means.contiguous(), cov.contiguous(), opacity.contiguous()
        sh = None if sh is None else sh.contiguous()
        colors = None if colors is None else colors.contiguous()
        view_params = view_params.contiguous()
        d = cfg.desc()
        V = cfg.n_scenes * cfg.views_per_scene
        dev = means.device
        color = torch.empty((V, 3, cfg.height, cfg.width), dtype=torch.float32, device=dev)
        radii = torch.empty((V, cfg.n_gaussians), dtype=torch.int32, device=dev)
        state = torch.empty(lib.ps_raster_state_bytes(C.byref(d)), dtype=torch.uint8, device=dev)
        temp = torch.empty(lib.ps_raster_temp_bytes(C.byref(d)), dtype=torch.uint8, device=dev)
        _lib.check(lib.ps_raster_forward(
            C.byref(d), _p(means), _p(cov), _p(sh), _p(colors), _p(opacity), _p(view_params),
            _p(color), _p(radii), _p(state), state.numel(), _p(temp), temp.numel(), _stream()),
            "ps_raster_forward")
        ctx.cfg = cfg
        ctx.has_means2d = means2d is not None
        ctx.save_for_backward(means, cov, opacity, sh, colors, view_params, radii, state)
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, dL_dcolor, _dradii):
        lib = _lib.load()
        cfg: RasterConfig = ctx.cfg
        means, cov, opacity, sh, colors, view_params, radii, state = ctx.saved_tensors
        d = cfg.desc()
        V = cfg.n_scenes * cfg.views_per_scene
        dev = means.device
        dL_dcolor = dL_dcolor.contiguous()
        g_means = torch.empty_like(means)
        g_cov = torch.empty_like(cov)
        g_op = torch.empty_like(opacity)
        g_sh = torch.empty_like(sh) if sh is not None else None
        g_colors = torch.empty_like(colors) if colors is not None else None
        g_m2d = (torch.empty((V, cfg.n_gaussians, 3), dtype=torch.float32, device=dev)
                 if ctx.has_means2d else None)
        temp = torch.empty(lib.ps_raster_temp_bytes(C.byref(d)), dtype=torch.uint8, device=dev)
        _lib.check(lib.ps_raster_backward(
            C.byref(d), _p(means), _p(cov), _p(sh), _p(colors), _p(opacity), _p(view_params),
            _p(radii), _p(dL_dcolor), _p(state), state.numel(), _p(temp), temp.numel(),
            _p(g_means), _p(g_cov), _p(g_sh), _p(g_colors), _p(g_op), _p(g_m2d), _stream()),
            "ps_raster_backward")
        return None, g_means, g_cov, g_op, g_sh, g_colors, None, g_m2d


def rasterize(cfg: RasterConfig, means: Tensor, cov: Tensor, opacity: Tensor,
              view_params: Tensor, sh: Tensor | None = None, colors: Tensor | None = None,
              means2d: Tensor | None = None, return_state: bool = False):
    """Batched differentiable 3-D Gaussian rasterization on the HIP kernels.

    means [S,G,3]; cov [S,G,6] | [S,G,3,3]; opacity [S,G]; view_params [V,48] (see
    `pack_view_params`); sh [S,G,K,3] | [S,G,3,K] xor colors [V,G,3]; means2d [V,G,3]
    (optional, only to receive the screen-space gradient).
    Returns (color [V,3,H,W], radii [V,G] int32).
    """
    if (sh is None) == (colors is None):
        raise Exception("Please provide exactly one of either SHs or precomputed colors!")
    return _Rasterize.apply(cfg, means, cov, opacity, sh, colors, view_params, means2d)


# ---- debug / parity helpers (used by tests; thin views over the saved state) --------------
def forward_with_state(cfg: RasterConfig, means, cov, opacity, view_params, sh=None, colors=None):
    """Runs the forward kernels and returns (color, radii, state bytes tensor, layout)."""
    lib = _lib.load()
    d = cfg.desc()
    V = cfg.n_scenes * cfg.views_per_scene
    dev = means.device
    color = torch.empty((V, 3, cfg.height, cfg.width), dtype=torch.float32, device=dev)
    radii = torch.empty((V, cfg.n_gaussians), dtype=torch.int32, device=dev)
    state = torch.zeros(lib.ps_raster_state_bytes(C.byref(d)), dtype=torch.uint8, device=dev)
    temp = torch.empty(lib.ps_raster_temp_bytes(C.byref(d)), dtype=torch.uint8, device=dev)
    _lib.check(lib.ps_raster_forward(
        C.byref(d), _p(means.contiguous()), _p(cov.contiguous()),
        _p(None if sh is None else sh.contiguous()),
        _p(None if colors is None else colors.contiguous()), _p(opacity.contiguous()),
        _p(view_params.contiguous()), _p(color), _p(radii), _p(state), state.numel(), _p(temp),
        temp.numel(), _stream()), "ps_raster_forward")
    lay = _lib.PsRasterStateLayout()
    _lib.check(lib.ps_raster_state_layout(C.byref(d), C.byref(lay)), "ps_raster_state_layout")
    return color, radii, state, lay


def state_views(cfg: RasterConfig, state: Tensor, lay) -> dict:
    V = cfg.n_scenes * cfg.views_per_scene
    G, P = cfg.n_gaussians, cfg.height * cfg.width
    tiles = ((cfg.width + 15) // 16) * ((cfg.height + 15) // 16)
    N = V * G

    def view(off, nbytes, dtype, shape):
        return state[off:off + nbytes].view(dtype).reshape(shape)

    return dict(
        records=view(lay.records, N * 48, torch.float32, (V, G, 12)),
        rects=view(lay.rects, N * 8, torch.int16, (V, G, 4)),
        sorted_idx=view(lay.sorted_idx, N * 4, torch.int32, (V, G)),
        sorted_rect=view(lay.sorted_rect, N * 8, torch.int16, (V, G, 4)),
        n_vis=view(lay.n_vis, V * 4, torch.int32, (V,)),
        final_T=view(lay.final_T, V * P * 4, torch.float32, (V, P)),
        n_contrib=view(lay.n_contrib, V * P * 4, torch.int32, (V, P)),
        tile_end=view(lay.tile_end, V * tiles * 8, torch.int32, (V, tiles, 2)),
    )


def export_bins(cfg: RasterConfig, state: Tensor):
    """(tile_counts [V,T] int32, point_list int32[D]) -- the bins the tile kernels walk, in
    blend order; bit-exact counterpart of the reference's sorted point list + ranges."""
    lib = _lib.load()
    d = cfg.desc()
    V = cfg.n_scenes * cfg.views_per_scene
    tiles = ((cfg.width + 15) // 16) * ((cfg.height + 15) // 16)
    counts = torch.zeros((V, tiles), dtype=torch.int32, device=state.device)
    _lib.check(lib.ps_raster_export_bins(C.byref(d), _p(state), state.numel(), _p(counts), None,
                                         None, 0, _stream()), "ps_raster_export_bins")
    flat = counts.reshape(-1).to(torch.int64)
    offsets = (torch.cumsum(flat, 0) - flat).to(torch.int32)
    total = int(flat.sum().item())
    plist = torch.zeros(max(total, 1), dtype=torch.int32, device=state.device)
    _lib.check(lib.ps_raster_export_bins(C.byref(d), _p(state), state.numel(), _p(counts),
                                         _p(offsets), _p(plist), total, _stream()),
               "ps_raster_export_bins")
    return counts, offsets.reshape(V, tiles), plist[:total]
