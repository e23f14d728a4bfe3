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
