"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's
epipolar sampler / depth encoding / cross-attention path, written as
the closed-form per-(view, other view, ray, sample) computation a fused kernel performs.

Pinned against the REAL reference (imported unmodified through oracle/ref_import.py) by
tests/test_oracle_epipolar.py in the build container and by the golden vectors it wrote to
tests/golden/ (generator: tests/golden/make_epipolar_golden.py).  Integer paths (validity,
frame selectors, bilinear corner indices) and xy_sample match the reference bit-for-bit on
CPU; this file reproduces the reference's floating-point evaluation ORDER on purpose:

  * matrix-vector products are sequential FMA chains (what MKL sgemm does for the einsums
    at src/geometry/projection.py:25-32,82-85 and epipolar_lines.py:168-171),
  * everything else is one rounding per elementwise op, as ATen evaluates it.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
from torch import Tensor

EPS = 1e-6


def _fma(a: Tensor, b: Tensor, c: Tensor) -> Tensor:
    """round(a*b + c) for float32 inputs (exact product in float64, one add, one rounding;
    the double rounding differs from a hardware FMA with probability ~2^-29)."""
    if a.dtype == torch.float64:
        return a * b + c
    return (a.double() * b.double() + c.double()).float()


def matvec(m: Tensor, x: Tensor) -> Tensor:
    """[..., n, k] @ [..., k] as MKL evaluates it: acc = m0*x0; acc = fma(mj, xj, acc)."""
    k = m.shape[-1]
    acc = m[..., :, 0] * x[..., None, 0]
    for j in range(1, k):
        acc = _fma(m[..., :, j], x[..., None, j].expand_as(acc), acc)
    return acc


def heterogeneous_index(n: int) -> Tensor:
    """index_v[v] = other views of v in increasing order (misc/heterogeneous_pairings.py:9-24)."""
    return torch.tensor([[o for o in range(n) if o != v] for v in range(n)], dtype=torch.int64)


def image_grid(h: int, w: int, dtype=torch.float32) -> Tensor:
    """[h*w, 2] pixel centres (x, y) in [0,1] (projection.py:117-137)."""
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    x = (xs.to(dtype) + 0.5) / w
    y = (ys.to(dtype) + 0.5) / h
    return torch.stack((x, y), -1).reshape(h * w, 2)


def world_rays(xy: Tensor, c2w: Tensor, k_inv: Tensor):
    """xy [..., 2], c2w [..., 4, 4], k_inv [..., 3, 3] -> origins, directions [..., 3]
    (projection.py:74-114)."""
    hom = torch.cat((xy, torch.ones_like(xy[..., :1])), -1)
    d = matvec(k_inv, hom)
    d = d * torch.ones_like(xy[..., 0])[..., None]          # "* z" with z = 1 (unproject)
    d = d / d.norm(dim=-1, keepdim=True)
    d4 = torch.cat((d, torch.zeros_like(d[..., :1])), -1)
    dw = matvec(c2w, d4)[..., :3]
    o = c2w[..., :3, 3].expand(dw.shape)
    return o, dw


@dataclass
class Segment:
    t_min: Tensor
    t_max: Tensor
    xy_min: Tensor
    xy_max: Tensor
    overlaps: Tensor          # bool
    sel_min: Tensor           # int64 0..3 frame-hit selector (x=0, x=1, y=0, y=1)
    sel_max: Tensor
    near_valid: Tensor        # bool: projection of origin + near*dir valid
    far_valid: Tensor


def _in_bounds(xy):
    return (xy >= -EPS).all(dim=-1) & (xy <= 1 + EPS).all(dim=-1)


def _frame_hit(k, o, d, dim, value):
    """epipolar_lines.py:55-104."""
    od = 1 - dim
    fs, fo = k[..., dim, dim], k[..., od, od]
    cs, co = k[..., dim, 2], k[..., od, 2]
    os_, oo = o[..., dim], o[..., od]
    ds, do = d[..., dim], d[..., od]
    oz, dz = o[..., 2], d[..., 2]
    c = (value - cs) / fs
    t = (c * oz - os_) / (ds - c * dz)
    num = fo * (oo * (c * dz - ds) + do * (os_ - c * oz))
    den = dz * os_ - ds * oz
    other = co + num / den
    same = torch.ones_like(other) * value
    xy = torch.stack((same, other) if dim == 0 else (other, same), -1)
    xyz = o + t[..., None] * d
    valid = _in_bounds(xy) & (xyz[..., 2] > -EPS) & (t > -EPS)
    return t, xy, valid


def _project_point(xyz, t, k):
    """epipolar_lines.py:134-144 with projection.py:47-56."""
    eps = torch.finfo(torch.float32).eps
    p = xyz / (xyz[..., -1:] + eps)
    p = p.nan_to_num(posinf=1e8, neginf=-1e8)
    xy = matvec(k, p)[..., :2]
    valid = _in_bounds(xy) & (xyz[..., 2] > -EPS) & (t > -EPS)
    return xy, valid


def project_rays(origins, directions, w2c, k, near, far) -> Segment:
    """origins/directions [..., 3] (world), w2c [..., 4, 4] = inv(extrinsics of the OTHER
    view), k [..., 3, 3] its intrinsics, near/far [...] of the CASTING view
    (epipolar_lines.py:157-251)."""
    o4 = torch.cat((origins, torch.ones_like(origins[..., :1])), -1)
    d4 = torch.cat((directions, torch.zeros_like(directions[..., :1])), -1)
    o = matvec(w2c, o4)[..., :3]
    d = matvec(w2c, d4)[..., :3]
    hits = [_frame_hit(k, o, d, 0, 0.0), _frame_hit(k, o, d, 0, 1.0),
            _frame_hit(k, o, d, 1, 0.0), _frame_hit(k, o, d, 1, 1.0)]
    t = torch.stack([h[0] for h in hits])
    xy = torch.stack([h[1] for h in hits])
    valid = torch.stack([h[2] for h in hits])

    def pick(reduction):
        tt = t.clone()
        tt[~valid] = math.inf if reduction == "min" else -math.inf
        red, sel = getattr(tt, reduction)(dim=0)
        pxy = xy.gather(0, sel[None, ..., None].expand(1, *sel.shape, 2))[0]
        pv = valid.gather(0, sel[None])[0]
        return red, pxy, pv, sel

    tmin_f, xymin_f, vmin_f, sel_min = pick("min")
    tmax_f, xymax_f, vmax_f, sel_max = pick("max")
    t_near = near.broadcast_to(tmin_f.shape)
    t_far = far.broadcast_to(tmin_f.shape)
    xy_near, v_near = _project_point(o + near[..., None] * d, t_near, k)
    xy_far, v_far = _project_point(o + far[..., None] * d, t_far, k)
    t_min = torch.where(v_near, t_near, tmin_f)
    t_max = torch.where(v_far, t_far, tmax_f)
    xy_min = torch.where(v_near[..., None], xy_near, xymin_f)
    xy_max = torch.where(v_far[..., None], xy_far, xymax_f)
    start_valid = torch.where(v_near, v_near, vmin_f)
    end_valid = torch.where(v_far, v_far, vmax_f)
    return Segment(t_min, t_max, xy_min, xy_max, start_valid & end_valid, sel_min, sel_max,
                   v_near, v_far)


def sample_points(seg: Segment, num_samples: int):
    """xy_sample [..., s, 2] (epipolar_sampler.py:80-88)."""
    s = num_samples
    dtype = seg.xy_min.dtype
    frac = ((torch.arange(s) + 0.5) / s).to(dtype)[:, None]
    a = seg.xy_min.nan_to_num(posinf=0, neginf=0) * seg.overlaps[..., None]
    b = seg.xy_max.nan_to_num(posinf=0, neginf=0) * seg.overlaps[..., None]
    a, b = a[..., None, :], b[..., None, :]
    return a + frac * (b - a), a, b


def bilinear_corners(xy: Tensor, h: int, w: int):
    """grid_sample(bilinear, zeros, align_corners=False) addressing for normalised xy in
    [0,1]: the reference passes grid = 2*xy - 1 (epipolar_sampler.py:98-104), ATen
    unnormalises with ((g + 1) * size - 1) / 2.
    Returns x0, y0 (int64), fractional weights wx, wy, and the 4 in-bounds masks."""
    g = 2 * xy - 1
    ix = ((g[..., 0] + 1) * w - 1) / 2
    iy = ((g[..., 1] + 1) * h - 1) / 2
    x0f, y0f = ix.floor(), iy.floor()
    x0, y0 = x0f.to(torch.int64), y0f.to(torch.int64)
    wx, wy = ix - x0f, iy - y0f
    inb = lambda xx, yy: (xx >= 0) & (xx < w) & (yy >= 0) & (yy < h)
    masks = torch.stack((inb(x0, y0), inb(x0 + 1, y0), inb(x0, y0 + 1), inb(x0 + 1, y0 + 1)), -1)
    return x0, y0, wx, wy, masks


def gather_features(fmap: Tensor, xy: Tensor) -> Tensor:
    """fmap [c, h, w], xy [n, 2] -> [n, c] bilinear, zero padding."""
    c, h, w = fmap.shape
    x0, y0, wx, wy, m = bilinear_corners(xy, h, w)
    out = torch.zeros((xy.shape[0], c), dtype=fmap.dtype)
    wts = ((1 - wx) * (1 - wy), wx * (1 - wy), (1 - wx) * wy, wx * wy)
    offs = ((0, 0), (1, 0), (0, 1), (1, 1))
    for i, ((dx, dy), wt) in enumerate(zip(offs, wts)):
        xx = (x0 + dx).clamp(0, w - 1)
        yy = (y0 + dy).clamp(0, h - 1)
        v = fmap[:, yy, xx].T
        out = out + v * (wt * m[..., i])[:, None]
    return out


def ray_depths(origins, directions, o2, d2, eps: float = 1e-5, inf: float = 1e10):
    """Depth along the casting ray of the least-squares meeting point with the second ray
    (projection.py:176-230 via epipolar_lines.py:264-292), closed form of the same 3x3
    normal equations  (sum_k n_k n_k^T - I) p = sum_k (n_k n_k^T - I) o_k."""
    parallel = (directions * d2).sum(-1) > 1 - eps
    eye = torch.eye(3, dtype=origins.dtype)
    n1 = directions[..., :, None] * directions[..., None, :] - eye
    n2 = d2[..., :, None] * d2[..., None, :] - eye
    lhs = n1 + n2
    rhs = (n1 @ origins[..., None])[..., 0] + (n2 @ o2[..., None])[..., 0]
    p = torch.linalg.solve(lhs.double(), rhs.double()).to(origins.dtype)
    p = torch.where(parallel[..., None], torch.full_like(p, inf), p)
    return (p - origins).norm(dim=-1)


def relative_disparity(depth, near, far, eps: float = 1e-10):
    """conversions.py:17-27."""
    disp_near = 1 / (near + eps)
    disp_far = 1 / (far + eps)
    disp = 1 / (depth + eps)
    return 1 - (disp - disp_far) / (disp_near - disp_far + eps)


def positional_encoding(x: Tensor, octaves: int) -> Tensor:
    """[...] -> [..., 2*octaves]: sin(x * 2 pi 2^k + {0, pi/2}) (positional_encoding.py:14-33)."""
    freq = 2 * torch.pi * 2 ** torch.arange(octaves).float()
    freq = freq[:, None].expand(octaves, 2).to(x.dtype)
    phase = torch.tensor([0, 0.5 * torch.pi], dtype=torch.float32).expand(octaves, 2).to(x.dtype)
    return torch.sin(x[..., None, None] * freq + phase).reshape(*x.shape, 2 * octaves)


@dataclass
class Sampling:
    xy_ray: Tensor        # [b, v, r, 2]
    origins: Tensor       # [b, v, r, 3]
    directions: Tensor    # [b, v, r, 3]
    segment: Segment      # fields [b, v, ov, r(, 2)]
    xy_sample: Tensor     # [b, v, ov, r, s, 2]
    features: Tensor      # [b, v, ov, r, s, c]
    depths: Tensor | None  # [b, v, ov, r, s] (unclipped)


def sample(features: Tensor, extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor,
           num_samples: int, with_depth: bool = True) -> Sampling:
    """features [b, v, c, h, w] -> everything EpipolarSampler.forward returns + depths
    (epipolar_sampler.py:51-123, epipolar_transformer.py:100-109)."""
    b, v, c, h, w = features.shape
    dtype = features.dtype
    idx = heterogeneous_index(v)
    ov = v - 1
    xy = image_grid(h, w, dtype)
    k_inv = torch.linalg.inv(intrinsics)
    w2c = torch.linalg.inv(extrinsics)
    o, d = world_rays(xy[None, None], extrinsics[:, :, None], k_inv[:, :, None])   # [b,v,r,3]
    oo = o[:, :, None].expand(b, v, ov, h * w, 3)
    dd = d[:, :, None].expand(b, v, ov, h * w, 3)
    w2c_o = w2c[:, idx][:, :, :, None]                # [b, v, ov, 1, 4, 4]
    k_o = intrinsics[:, idx][:, :, :, None]
    seg = project_rays(oo, dd, w2c_o, k_o, near[:, :, None, None].expand(b, v, ov, h * w),
                       far[:, :, None, None].expand(b, v, ov, h * w))
    xy_s, _, _ = sample_points(seg, num_samples)       # [b, v, ov, r, s, 2]
    feats = torch.zeros((b, v, ov, h * w, num_samples, c), dtype=dtype)
    for bi in range(b):
        for vi in range(v):
            for oi in range(ov):
                src = int(idx[vi, oi])
                f = gather_features(features[bi, src], xy_s[bi, vi, oi].reshape(-1, 2))
                feats[bi, vi, oi] = f.reshape(h * w, num_samples, c)
    feats = feats * seg.overlaps[..., None, None]
    depths = None
    if with_depth:
        c2w_o = extrinsics[:, idx][:, :, :, None, None]          # [b, v, ov, 1, 1, 4, 4]
        kinv_o = k_inv[:, idx][:, :, :, None, None]
        o2, d2 = world_rays(xy_s, c2w_o, kinv_o)
        depths = ray_depths(oo[..., None, :].expand_as(o2), dd[..., None, :].expand_as(d2), o2, d2)
    return Sampling(xy.expand(b, v, h * w, 2), o, d, seg, xy_s, feats, depths)


def attention_layer(x, kv, ln_w, ln_b, w_q, w_kv, w_o, b_o, heads: int):
    """One PreNorm(Attention(x, z=kv)) + residual (attention.py:54-70, pre_norm.py:34-35,
    transformer.py:69).  x [n, 1, d], kv [n, t, dk] -> (x + y, attention weights [n, h, 1, t])."""
    xh = torch.nn.functional.layer_norm(x, (x.shape[-1],), ln_w, ln_b)
    q = xh @ w_q.T
    k, vv = (kv @ w_kv.T).chunk(2, dim=-1)
    n, t, inner = k.shape
    dh = inner // heads
    sp = lambda z: z.reshape(n, -1, heads, dh).transpose(1, 2)
    q, k, vv = sp(q), sp(k), sp(vv)
    dots = (q @ k.transpose(-1, -2)) * dh ** -0.5
    attn = dots.softmax(dim=-1)
    out = (attn @ vv).transpose(1, 2).reshape(n, 1, inner)
    return x + (out @ w_o.T + b_o), attn
