"""Path (A): epipolar sampler on the HIP kernels (C ABI: ps_epipolar_*).

PyTorch is plumbing (buffers, stream, tiny 3x3 / 4x4 inverses).  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import torch
from torch import Tensor

from . import _lib
from .raster import _p, _stream


@dataclass
class EpipolarGeometry:
    origins: Tensor      # [b, v, r, 3]
    directions: Tensor   # [b, v, r, 3]
    xy_min: Tensor       # [b, v, ov, r, 2]
    xy_max: Tensor       # [b, v, ov, r, 2]
    t_min: Tensor        # [b, v, ov, r]
    t_max: Tensor        # [b, v, ov, r]
    overlaps: Tensor     # [b, v, ov, r] bool
    flags: Tensor        # [b, v, ov, r] uint8 (see include/pixelsplat_hip.h)
    xy_sample: Tensor    # [b, v, ov, r, s, 2]
    depth: Tensor        # [b, v, ov, r, s]
    rel_disparity: Tensor  # [b, v, ov, r, s]


def sample_geometry(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor,
                    grid_hw: tuple[int, int], num_samples: int, w2c: Tensor | None = None,
                    k_inv: Tensor | None = None) -> EpipolarGeometry:
    """extrinsics [b,v,4,4] c2w, intrinsics [b,v,3,3] normalised, near/far [b,v]; rays are the
    pixel centres of an h x w grid.  `w2c` / `k_inv` default to torch.linalg.inv on the device
    (the reference inverts with torch too: epipolar_lines.py:167, projection.py:84)."""
    lib = _lib.load()
    if not extrinsics.is_cuda:
        raise RuntimeError("pixelsplat_amd.epipolar needs GPU tensors (no CPU fallback)")
    b, v = extrinsics.shape[:2]
    h, w = grid_hw
    s, r, ov = num_samples, h * w, v - 1
    dev = extrinsics.device
    c2w = extrinsics.contiguous().float()
    k = intrinsics.contiguous().float()
    w2c = (torch.linalg.inv(c2w) if w2c is None else w2c).contiguous().float()
    k_inv = (torch.linalg.inv(k) if k_inv is None else k_inv).contiguous().float()
    nr, fr = near.contiguous().float(), far.contiguous().float()
    f32 = dict(dtype=torch.float32, device=dev)
    origins = torch.empty((b, v, r, 3), **f32)
    directions = torch.empty((b, v, r, 3), **f32)
    seg = torch.empty((b, v, ov, r, 6), **f32)
    flags = torch.empty((b, v, ov, r), dtype=torch.uint8, device=dev)
    xy_sample = torch.empty((b, v, ov, r, s, 2), **f32)
    depth = torch.empty((b, v, ov, r, s), **f32)
    rel = torch.empty((b, v, ov, r, s), **f32)
    _lib.check(lib.ps_epipolar_geometry(
        C.c_int32(b), C.c_int32(v), C.c_int32(h), C.c_int32(w), C.c_int32(s), _p(c2w), _p(w2c),
        _p(k), _p(k_inv), _p(nr), _p(fr), _p(origins), _p(directions), _p(seg), _p(flags),
        _p(xy_sample), _p(depth), _p(rel), _stream()), "ps_epipolar_geometry")
    return EpipolarGeometry(origins, directions, seg[..., 0:2], seg[..., 2:4], seg[..., 4],
                            seg[..., 5], (flags & 1).bool(), flags, xy_sample, depth, rel)


# ---------------------------------------------------------------------------------------
# fused gather + cross-attention
# ---------------------------------------------------------------------------------------
def _desc(b, v, h, w, s, c, heads, octaves) -> _lib.PsEpipolarDesc:
    d = _lib.PsEpipolarDesc()
    d.b, d.v, d.h, d.w, d.s, d.c, d.heads, d.octaves = b, v, h, w, s, c, heads, octaves
    return d


def gather_features(fmap_nhwc: Tensor, geo: EpipolarGeometry) -> Tensor:
    """Materialised `EpipolarSampling.features` [b,v,ov,r,s,c] (visualisers / unfused path);
    fmap_nhwc [b,v,h,w,c]."""
    lib = _lib.load()
    b, v, h, w, c = fmap_nhwc.shape
    s = geo.xy_sample.shape[-2]
    out = torch.empty((b, v, v - 1, h * w, s, c), dtype=torch.float32, device=fmap_nhwc.device)
    d = _desc(b, v, h, w, s, c, 1, 1)
    _lib.check(lib.ps_epipolar_gather(C.byref(d), _p(fmap_nhwc.contiguous()),
                                      _p(geo.xy_sample), _p(geo.flags), _p(out), _stream()),
               "ps_epipolar_gather")
    return out


class _FusedEpipolarAttention(torch.autograd.Function):
    """(fmap, q~, u, e) -> (fbar, pbar, abar, attn); see csrc/epipolar_attention.hip."""

    @staticmethod
    def forward(ctx, dims, scale, fmap, xy, flags, rd, qt, u, e):
        lib = _lib.load()
        b, v, h, w, s, c, heads, octaves = dims
        d = _desc(*dims)
        R, T, P, ov = b * v * h * w, s * (v - 1), 2 * octaves, v - 1
        dev = fmap.device
        fmap, qt, u = fmap.contiguous(), qt.contiguous(), u.contiguous()
        e = None if e is None else e.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        fbar = torch.empty((R, heads, c), **f32)
        pbar = torch.empty((R, heads, P), **f32)
        abar = torch.empty((R, heads, ov), **f32)
        attn = torch.empty((R, heads, T), **f32)
        _lib.check(lib.ps_epipolar_attention_forward(
            C.byref(d), _p(fmap), _p(xy), _p(flags), _p(rd), _p(qt), _p(u), _p(e),
            C.c_float(scale), _p(fbar), _p(pbar), _p(abar), _p(attn), _stream()),
            "ps_epipolar_attention_forward")
        ctx.dims, ctx.scale, ctx.has_e = dims, scale, e is not None
        ctx.save_for_backward(fmap, xy, flags, rd, qt, attn)
        ctx.mark_non_differentiable(attn)
        return fbar, pbar, abar, attn

    @staticmethod
    def backward(ctx, dfbar, dpbar, dabar, _dattn):
        lib = _lib.load()
        fmap, xy, flags, rd, qt, attn = ctx.saved_tensors
        b, v, h, w, s, c, heads, octaves = ctx.dims
        d = _desc(*ctx.dims)
        R, T, P, ov = b * v * h * w, s * (v - 1), 2 * octaves, v - 1
        f32 = dict(dtype=torch.float32, device=fmap.device)
        dqt = torch.empty((R, heads, c), **f32)
        du = torch.empty((R, heads, P), **f32)
        de = torch.empty((R, heads, ov), **f32)
        ds = torch.empty((R, heads, T), **f32)
        dfmap = boxes = None
        if ctx.needs_input_grad[2]:
            dfmap = torch.empty_like(fmap)
            boxes = torch.empty((R * ov,), dtype=torch.int32, device=fmap.device)
        _lib.check(lib.ps_epipolar_attention_backward(
            C.byref(d), _p(fmap), _p(xy), _p(flags), _p(rd), _p(qt), _p(attn),
            _p(dfbar.contiguous()), _p(dpbar.contiguous()), _p(dabar.contiguous()),
            C.c_float(ctx.scale), _p(dqt), _p(du), _p(de), _p(ds), _p(dfmap), _p(boxes),
            _stream()),
            "ps_epipolar_attention_backward")
        return (None, None, dfmap, None, None, None, dqt, du, de if ctx.has_e else None)


def fused_cross_attention(x: Tensor, fmap_nhwc: Tensor, geo: EpipolarGeometry, *, w_q: Tensor,
                          w_kv: Tensor, w_out: Tensor, b_out: Tensor | None, heads: int,
                          depth_w: Tensor, depth_b: Tensor, octaves: int,
                          view_emb: Tensor | None = None, return_attn: bool = False):
    """Attention(x, z=kv) of the reference (attention.py:54-70) for kv = gathered features +
    Linear(PE(relative disparity)) [+ view embedding], without ever forming kv.

    x [R, 1, d] (already layer-normed), fmap_nhwc [b, v, h, w, c]; w_q [inner, d],
    w_kv [2*inner, c], w_out [d, inner]; depth_w [c, 2*octaves], depth_b [c];
    view_emb [v-1, c] (already permuted) or None.  Returns [R, 1, d] (and attn [R,H,1,T])."""
    b, v, h, w, c = fmap_nhwc.shape
    s = geo.xy_sample.shape[-2]
    inner = w_q.shape[0]
    dh = inner // heads
    R = x.shape[0]
    q = (x.reshape(R, -1) @ w_q.T).reshape(R, heads, dh)
    w_k = w_kv[:inner].reshape(heads, dh, c)
    w_v = w_kv[inner:].reshape(heads, dh, c)
    qt = torch.einsum("rhd,hdc->rhc", q, w_k)                    # q~_h = W_k,h^T q_h
    u = qt @ depth_w                                             # [R, H, P]
    e = None if view_emb is None else qt @ view_emb.T            # [R, H, ov]
    dims = (b, v, h, w, s, c, heads, octaves)
    fbar, pbar, abar, attn = _FusedEpipolarAttention.apply(
        dims, float(dh) ** -0.5, fmap_nhwc.reshape(b * v, h, w, c), geo.xy_sample, geo.flags,
        geo.rel_disparity, qt, u, e)
    ctxv = fbar + pbar @ depth_w.T + depth_b                     # sum_i a_i kv_i
    if view_emb is not None:
        ctxv = ctxv + abar @ view_emb
    out = torch.einsum("rhc,hdc->rhd", ctxv, w_v).reshape(R, 1, inner)
    out = out @ w_out.T
    if b_out is not None:
        out = out + b_out
    if return_attn:
        return out, attn.reshape(R, heads, 1, -1)
    return out
