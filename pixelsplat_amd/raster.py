"""torch.autograd front-end of the HIP rasterizer (C ABI: include/pixelsplat_hip.h).

PyTorch is plumbing here: device buffers, the current stream, autograd bookkeeping.  All
arithmetic happens in libpixelsplat_hip.so; there is no fallback path.

`rasterize(...)` is the batched operator (S scenes x views_per_scene views in one call).
It replaces, per call, the whole loop of
/root/reference/src/model/decoder/cuda_splatting.py:91-126 (B sequential
`GaussianRasterizer` invocations, three host syncs each) and the v-fold `repeat` of
/root/reference/src/model/decoder/decoder_splatting_cuda.py:53-56.

Host syncs per call: ONE 8-byte read-back of D (the total tile-list length, which sizes the
point list exactly) -- or ZERO when `RasterConfig.list_capacity` fixes the capacity up front;
an overflow is then detected asynchronously and raised from backward().
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
    # 0: size the tile point list exactly (one 8-byte host read-back per call);
    # > 0: fixed capacity in entries, no host sync, overflow raises from backward()
    list_capacity: int = 0
    # gradients of Gaussians over more than four tiles summed in a fixed order instead of through float
    # atomics (PS_FLAG_DETERMINISTIC: two runs are bitwise equal; more scratch, slower -- for parity and
    # reproducibility runs).  None: the PIXELSPLAT_DETERMINISTIC environment variable decides (default off)
    deterministic: bool | None = None

    def desc(self) -> _lib.PsRasterDesc:
        d = _lib.default_desc()
        d.n_scenes, d.views_per_scene, d.n_gaussians = (self.n_scenes, self.views_per_scene,
                                                        self.n_gaussians)
        d.height, d.width = self.height, self.width
        d.sh_degree, d.sh_coeffs = self.sh_degree, self.sh_coeffs
        d.sh_layout, d.cov_layout = self.sh_layout, self.cov_layout
        det = self.deterministic
        if det is None:
            import os
            det = os.environ.get("PIXELSPLAT_DETERMINISTIC", "0") not in ("", "0")
        if det:
            d.flags |= _lib.PS_FLAG_DETERMINISTIC
        return d

    @property
    def n_views(self) -> int:
        return self.n_scenes * self.views_per_scene

    @property
    def n_tiles(self) -> int:
        return ((self.width + 15) // 16) * ((self.height + 15) // 16)


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


@dataclass
class ForwardResult:
    color: Tensor
    radii: Tensor
    state: Tensor             # uint8, layout = ps_raster_state_layout
    point_list: Tensor        # int32 [capacity]
    num_rendered: int | None  # D when it was read back (exact-size mode), else None
    overflow_host: Tensor | None = None
    overflow_event: object | None = None

    def check_overflow(self, capacity: int) -> None:
        """Fixed-capacity mode: waits for the forward's 8-byte flag copy and raises when the
        tile lists did not fit (the image was then blended from truncated lists).  No-op in
        exact-sizing mode."""
        if self.overflow_host is None or self.overflow_event is None:
            return   # exact sizing, or recorded into a hipGraph (see captured_overflow_flags)
        self.overflow_event.synchronize()
        if int(self.overflow_host[1]) != 0:
            raise RuntimeError(
                f"pixelsplat_amd.rasterize: {int(self.overflow_host[0])} tile-list entries exceed "
                f"list_capacity={capacity} (PS_ERR_CAPACITY); raise it or use list_capacity=0 "
                f"(exact sizing)")


def _forward(cfg: RasterConfig, means, cov, opacity, sh, colors, view_params) -> ForwardResult:
    lib = _lib.load()
    d = cfg.desc()
    V, dev = cfg.n_views, means.device
    color = torch.empty((V, 3, cfg.height, cfg.width), dtype=torch.float32, device=dev)
    radii = torch.empty((V, cfg.n_gaussians), dtype=torch.int32, device=dev)
    state = torch.empty(lib.ps_raster_state_bytes(C.byref(d)), dtype=torch.uint8, device=dev)
    temp = torch.empty(lib.ps_raster_temp_bytes(C.byref(d)), dtype=torch.uint8, device=dev)
    if cfg.list_capacity > 0:
        plist = torch.empty(cfg.list_capacity, dtype=torch.int32, device=dev)
        _lib.check(lib.ps_raster_forward_plan(
            C.byref(d), _p(means), _p(cov), _p(sh), _p(colors), _p(opacity), _p(view_params),
            _p(radii), _p(state), state.numel(), _p(temp), temp.numel(), _stream()),
            "ps_raster_forward_plan")
        _lib.check(lib.ps_raster_forward_bins(
            C.byref(d), _p(state), state.numel(), _p(temp), temp.numel(), _p(plist), plist.numel(),
            _stream()), "ps_raster_forward_bins")
        _lib.check(lib.ps_raster_forward_tiles(
            C.byref(d), _p(view_params), _p(color), _p(state), state.numel(), _p(temp),
            temp.numel(), _p(plist), plist.numel(), _stream()), "ps_raster_forward_tiles")
        lay = _lib.PsRasterStateLayout()
        lib.ps_raster_state_layout(C.byref(d), C.byref(lay))
        flag = _pinned_flag()
        flag.copy_(state[lay.num_rendered:lay.num_rendered + 8].view(torch.int32),
                   non_blocking=True)
        ev = None
        if torch.cuda.is_current_stream_capturing():
            # hipGraph capture: no host wait may be recorded.  The copy is a node of the graph;
            # the caller checks `captured_overflow_flags()` after a replay has completed.
            _CAPTURED_FLAGS.append((flag, cfg.list_capacity))
        else:
            ev = torch.cuda.Event()
            ev.record()
        return ForwardResult(color, radii, state, plist, None, flag, ev)
    # exact sizing: D is read back once per batch.  The SH colours are deferred behind that
    # copy, so the GPU evaluates them while the host waits for D, allocates and launches.
    _defer = sh is not None
    if _defer:
        d.flags |= _lib.PS_FLAG_DEFER_SH_COLORS
    _lib.check(lib.ps_raster_forward_plan(
        C.byref(d), _p(means), _p(cov), _p(sh), _p(colors), _p(opacity), _p(view_params),
        _p(radii), _p(state), state.numel(), _p(temp), temp.numel(), _stream()),
        "ps_raster_forward_plan")
    lay = _lib.PsRasterStateLayout()
    lib.ps_raster_state_layout(C.byref(d), C.byref(lay))
    host = torch.empty(2, dtype=torch.int32, pin_memory=True)
    host.copy_(state[lay.num_rendered:lay.num_rendered + 8].view(torch.int32), non_blocking=True)
    copied = torch.cuda.Event()
    copied.record()
    if _defer:
        _lib.check(lib.ps_raster_forward_colors(
            C.byref(d), _p(means), _p(sh), _p(view_params), _p(radii), _p(state), state.numel(),
            _p(temp), temp.numel(), _stream()), "ps_raster_forward_colors")
    copied.synchronize()
    n = C.c_uint64(int(host[0]) & 0xFFFFFFFF)
    plist = torch.empty(max(int(n.value), 1), dtype=torch.int32, device=dev)
    _lib.check(lib.ps_raster_forward_bins(
        C.byref(d), _p(state), state.numel(), _p(temp), temp.numel(), _p(plist), int(n.value),
        _stream()), "ps_raster_forward_bins")
    _lib.check(lib.ps_raster_forward_tiles(
        C.byref(d), _p(view_params), _p(color), _p(state), state.numel(), _p(temp), temp.numel(),
        _p(plist), int(n.value), _stream()), "ps_raster_forward_tiles")
    return ForwardResult(color, radii, state, plist, int(n.value))


_CAPTURE_FREE: list = []      # pinned buffers reserved for forwards recorded into a hipGraph
_CAPTURED_FLAGS: list = []    # (flag, capacity) of forwards recorded into a hipGraph
_CAPTURE_RESERVE = 32


def reserve_capture_flags(n: int = _CAPTURE_RESERVE) -> None:
    """Pinned 8-byte landing buffers for fixed-capacity forwards that will be recorded into
    hipGraphs: pinned memory cannot be allocated while a stream is being captured, so they are
    set aside beforehand (the first eager fixed-capacity call reserves 32).  Every recorded
    forward gets a buffer of its OWN for the lifetime of its graph -- two forwards of one graph,
    or of two live graphs, never share a landing buffer."""
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("reserve_capture_flags() must be called outside a capture")
    while len(_CAPTURE_FREE) < n:
        _CAPTURE_FREE.append(torch.empty(2, dtype=torch.int32, pin_memory=True))


def release_captured_flags() -> None:
    """Forget the forwards recorded so far (their graphs are gone: re-capture, failed capture);
    their landing buffers return to the reserve."""
    while _CAPTURED_FLAGS:
        _CAPTURE_FREE.append(_CAPTURED_FLAGS.pop()[0])


def _pinned_flag() -> Tensor:
    """A pinned 8-byte landing buffer for (D, overflow)."""
    if torch.cuda.is_current_stream_capturing():
        if not _CAPTURE_FREE:
            raise RuntimeError(
                "pixelsplat_amd.rasterize under hipGraph capture: no landing buffer left for the "
                "overflow flag -- run one eager (warm-up) step with the same list_capacity first, "
                "or raster.reserve_capture_flags(n) for more than 32 recorded forwards, and "
                "raster.release_captured_flags() when graphs are dropped")
        return _CAPTURE_FREE.pop()
    reserve_capture_flags()
    return torch.empty(2, dtype=torch.int32, pin_memory=True)   # caching host allocator


def captured_overflow_flags(check: bool = True) -> list:
    """Fixed-capacity forwards recorded into a hipGraph cannot raise from inside the graph: after
    a replay has completed (synchronise first) this returns [(entries needed, overflowed?,
    capacity)] per recorded forward and, with `check`, raises if any list did not fit."""
    out = [(int(f[0]), bool(int(f[1])), cap) for f, cap in _CAPTURED_FLAGS]
    if check:
        for need, over, cap in out:
            if over:
                raise RuntimeError(f"pixelsplat_amd.rasterize (hipGraph replay): {need} tile-list "
                                   f"entries exceed list_capacity={cap} (PS_ERR_CAPACITY)")
    return out


class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cfg: RasterConfig, means, cov, opacity, sh, colors, view_params, means2d):
        _check_dev(means, cov, opacity, sh, colors, view_params)
        means, cov, opacity = means.contiguous(), cov.contiguous(), opacity.contiguous()
        sh = None if sh is None else sh.contiguous()
        colors = None if colors is None else colors.contiguous()
        view_params = view_params.contiguous()
        r = _forward(cfg, means, cov, opacity, sh, colors, view_params)
        if not any(ctx.needs_input_grad):
            # inference (no backward will ever read the flag): check the capacity now, one
            # host wait on an 8-byte copy that was queued right behind the forward
            r.check_overflow(cfg.list_capacity)
        ctx.cfg = cfg
        ctx.has_means2d = means2d is not None
        ctx.overflow = (r.overflow_host, r.overflow_event)
        ctx.save_for_backward(means, cov, opacity, sh, colors, view_params, r.radii, r.state,
                              r.point_list)
        ctx.mark_non_differentiable(r.radii)
        ctx.set_materialize_grads(False)    # (no zero-filled [V, G] "gradient" of the radii per backward)
        return r.color, r.radii

    @staticmethod
    def backward(ctx, dL_dcolor, _dradii):
        if dL_dcolor is None:
            return (None,) * 8
        lib = _lib.load()
        cfg: RasterConfig = ctx.cfg
        means, cov, opacity, sh, colors, view_params, radii, state, plist = ctx.saved_tensors
        d = cfg.desc()
        V, dev = cfg.n_views, means.device
        flag, ev = ctx.overflow
        if flag is not None:   # long complete: the copy was queued right after the forward
            ForwardResult(None, None, None, None, None, flag, ev).check_overflow(cfg.list_capacity)
        dL_dcolor = dL_dcolor.contiguous()
        g_means = torch.empty_like(means)
        g_cov = torch.empty_like(cov)
        g_op = torch.empty_like(opacity)
        g_sh = torch.empty_like(sh) if sh is not None else None
        g_colors = torch.empty_like(colors) if colors is not None else None
        g_m2d = (torch.empty((V, cfg.n_gaussians, 3), dtype=torch.float32, device=dev)
                 if ctx.has_means2d else None)
        # no memset: ps_raster_backward clears the few accumulator rows it adds into by itself
        # (round 3; a whole-buffer memset on a second stream under the forward cost the step 0.11 ms)
        temp = torch.empty(lib.ps_raster_backward_temp_bytes(C.byref(d), plist.numel()),
                           dtype=torch.uint8, device=dev)
        _lib.check(lib.ps_raster_backward(
            C.byref(d), _p(means), _p(cov), _p(sh), _p(colors), _p(opacity), _p(view_params),
            _p(radii), _p(dL_dcolor), _p(state), state.numel(), _p(temp), temp.numel(),
            _p(plist), plist.numel(), _p(g_means), _p(g_cov), _p(g_sh), _p(g_colors), _p(g_op),
            _p(g_m2d), _stream()), "ps_raster_backward")
        return None, g_means, g_cov, g_op, g_sh, g_colors, None, g_m2d


def rasterize(cfg: RasterConfig, means: Tensor, cov: Tensor, opacity: Tensor,
              view_params: Tensor, sh: Tensor | None = None, colors: Tensor | None = None,
              means2d: Tensor | None = None):
    """Batched differentiable 3-D Gaussian rasterization on the HIP kernels.

    means [S,G,3]; cov [S,G,6] | [S,G,3,3]; opacity [S,G]; view_params [V,48] (see
    `pack_view_params`); sh [S,G,K,3] | [S,G,3,K] xor colors [V,G,3]; means2d [V,G,3]
    (optional, only to receive the screen-space gradient).
    Returns (color [V,3,H,W], radii [V,G] int32).

    With `cfg.list_capacity > 0` (no host sync in the forward) an overflow of the tile lists is
    raised from backward(); a call that needs no gradients (inference, `torch.no_grad()`) checks
    the flag before returning instead, and `forward_with_state(...)[0].check_overflow(capacity)`
    does the same for the raw forward.
    """
    if (sh is None) == (colors is None):
        raise Exception("Please provide exactly one of either SHs or precomputed colors!")
    return _Rasterize.apply(cfg, means, cov, opacity, sh, colors, view_params, means2d)


# ---- debug / parity helpers (used by tests and bench; thin views over the saved state) ----
def forward_with_state(cfg: RasterConfig, means, cov, opacity, view_params, sh=None, colors=None):
    """Runs the forward kernels (no autograd) and returns (ForwardResult, state layout)."""
    lib = _lib.load()
    r = _forward(cfg, means.contiguous(), cov.contiguous(), opacity.contiguous(),
                 None if sh is None else sh.contiguous(),
                 None if colors is None else colors.contiguous(), view_params.contiguous())
    d = cfg.desc()
    lay = _lib.PsRasterStateLayout()
    _lib.check(lib.ps_raster_state_layout(C.byref(d), C.byref(lay)), "ps_raster_state_layout")
    return r, lay


def state_views(cfg: RasterConfig, state: Tensor, lay) -> dict:
    V, G, P, tiles = cfg.n_views, cfg.n_gaussians, cfg.height * cfg.width, cfg.n_tiles
    N = V * G

    def view(off, nbytes, dtype, shape):
        return state[off:off + nbytes].view(dtype).reshape(shape)

    # one 64-byte line per pair: the 48-byte record + its 16-byte cell window (csrc/raster_common.h: kRecFloats)
    lines = view(lay.records, N * 64, torch.float32, (V, G, 16))
    return dict(
        records=lines[..., :12],
        rects=view(lay.rects, N * 8, torch.int16, (V, G, 4)),
        sorted_idx=view(lay.sorted_idx, N * 4, torch.int32, (V, G)),
        sorted_rect=view(lay.sorted_rect, N * 8, torch.int16, (V, G, 4)),
        n_vis=view(lay.n_vis, V * 4, torch.int32, (V,)),
        final_T=view(lay.final_T, V * P * 4, torch.float32, (V, P)),
        n_contrib=view(lay.n_contrib, V * P * 4, torch.int32, (V, P)),
        tile_end=view(lay.tile_end, V * tiles * 4, torch.int32, (V, tiles)),   # the tile's last contributor
        tile_ranges=view(lay.tile_ranges, V * tiles * 8, torch.int32, (V, tiles, 2)),
        num_rendered=view(lay.num_rendered, 8, torch.int32, (2,)),
        # per pixel (quadrant, lane) of a tile whose list the backward walks as two tasks: T after the
        # first half and the colour composited behind it over that T
        checkpoint=view(lay.checkpoint, V * tiles * 256 * 16, torch.float32, (V, tiles, 4, 64, 4)),
        # which 4x4-pixel cells a visible pair can reach with alpha >= alpha_min (csrc/cell_window.h): small
        # window = (mask lo, mask hi, anchor cx | cy << 16, 0), large footprint = (cx0 | cx1 << 16, cy0 | cy1 << 16, -, 1)
        cell_windows=lines.view(torch.int32)[..., 12:16],
    )


def export_bins(cfg: RasterConfig, state: Tensor, lay, point_list: Tensor):
    """(tile_counts [V,T], tile_offsets [V,T], point_list int32[D]) -- the bins the tile
    kernels walk, in blend order; bit-exact counterpart of the reference's sorted point
    list + tile ranges."""
    sv = state_views(cfg, state, lay)
    ranges = sv["tile_ranges"]
    n = int(sv["num_rendered"][0].item())
    return ranges[..., 1].contiguous(), ranges[..., 0].contiguous(), point_list[:n]
