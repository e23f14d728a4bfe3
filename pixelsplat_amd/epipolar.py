"""Path (A): epipolar sampler on the HIP kernels (C ABI: ps_epipolar_*).

PyTorch is plumbing (buffers, stream, tiny 3x3 / 4x4 inverses).  No CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import os as _os
from contextlib import nullcontext as _nullcontext

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
    flags: Tensor        # [b, v, ov, r] uint8 (see include/pixelsplat_hip.h)
    xy_sample: Tensor    # [b, v, ov, r, s, 2]
    depth: Tensor        # [b, v, ov, r, s]
    rel_disparity: Tensor  # [b, v, ov, r, s]
    _overlaps: Tensor | None = None

    @property
    def overlaps(self) -> Tensor:
        """[b, v, ov, r] bool (`EpipolarSampling.valid`): bit 0 of `flags`, unpacked on first read --
        the kernels of the hot path read `flags` itself."""
        if self._overlaps is None:
            self._overlaps = (self.flags & 1).bool()
        return self._overlaps


def sample_geometry(extrinsics: Tensor, intrinsics: Tensor, near: Tensor, far: Tensor,
                    grid_hw: tuple[int, int], num_samples: int, w2c: Tensor | None = None,
                    k_inv: Tensor | None = None) -> EpipolarGeometry:
    """extrinsics [b,v,4,4] c2w, intrinsics [b,v,3,3] normalised, near/far [b,v]; rays are the
    pixel centres of an h x w grid.  `w2c` / `k_inv` default to ps_invert_cameras (the
    reference inverts with torch.linalg.inv: epipolar_lines.py:167, projection.py:84)."""
    lib = _lib.load()
    if not extrinsics.is_cuda:
        raise RuntimeError("pixelsplat_amd.epipolar needs GPU tensors (no CPU fallback)")
    b, v = extrinsics.shape[:2]
    h, w = grid_hw
    s, r, ov = num_samples, h * w, v - 1
    dev = extrinsics.device
    c2w = extrinsics.contiguous().float()
    k = intrinsics.contiguous().float()
    if w2c is None or k_inv is None:     # one launch, no host sync (torch.linalg.inv syncs)
        w2c_d = torch.empty_like(c2w)
        k_inv_d = torch.empty_like(k)
        _lib.check(lib.ps_invert_cameras(C.c_int32(b * v), _p(c2w), _p(k), _p(w2c_d),
                                         _p(k_inv_d), _stream()), "ps_invert_cameras")
        w2c = w2c_d if w2c is None else w2c
        k_inv = k_inv_d if k_inv is None else k_inv
    w2c, k_inv = w2c.contiguous().float(), k_inv.contiguous().float()
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
                            seg[..., 5], flags, xy_sample, depth, rel)


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


def gemm_tn(a: Tensor, b: Tensor, colsum: bool = False):
    """a [k, m], b [k, n] (fp32, row-major, last dim contiguous) -> a^T b [m, n] with the split-k
    MFMA kernel (ps_gemm_tn_colsum_f32); deterministic.  colsum=True also returns a.sum(0) [m],
    gathered from the operand stream of the same kernel."""
    lib = _lib.load()
    k, m = a.shape
    n = b.shape[1]
    if n % 4:                      # the kernel works on groups of 4 columns
        r = gemm_tn(a, torch.nn.functional.pad(b, (0, -n % 4)), colsum)
        return (r[0][:, :n], r[1]) if colsum else r[:, :n]
    if m % 4:
        r = gemm_tn(torch.nn.functional.pad(a, (0, -m % 4)), b, colsum)
        return (r[0][:m], r[1][:m]) if colsum else r[:m]
    if a.stride(1) != 1 or a.stride(0) % 4 or a.data_ptr() % 16:
        a = a.contiguous()
    if b.stride(1) != 1 or b.stride(0) % 4 or b.data_ptr() % 16:
        b = b.contiguous()
    c = torch.empty((m, n), dtype=torch.float32, device=a.device)
    cs = torch.empty((m,), dtype=torch.float32, device=a.device) if colsum else None
    ws_bytes = lib.ps_gemm_tn_workspace_bytes(m, n, k)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=a.device)
    _lib.check(lib.ps_gemm_tn_colsum_f32(m, n, k, _p(a), a.stride(0), _p(b), b.stride(0), _p(c),
                                         _p(cs), _p(ws), C.c_size_t(ws_bytes), _stream()),
               "ps_gemm_tn_colsum_f32")
    return (c, cs) if colsum else c


# PS_WGRAD_SIDE=1: weight gradients of the per-ray GEMMs on the stream the folded weights live on.
# Measured and NOT the default (profiles/r3_forward_forms_ab.txt, call 2): next to the VALU-bound
# kernels of the main stream the split-k MFMA kernel takes 0.126 instead of 0.107 ms and slows its
# neighbours by more than it hides -- (A) 3.46 vs 3.38 ms.
WGRAD_ON_SIDE_STREAM = _os.environ.get("PS_WGRAD_SIDE", "0") == "1"


class _RayLinear(torch.autograd.Function):
    """y = x W^T (+ bias) over all rays; the weight gradient dW = dy^T x is the long-k GEMM
    hipBLASLt handles badly (gemm_tn.hip).

    `wgrad_stream` (used only with PS_WGRAD_SIDE=1, an experiment that did not pay -- see
    WGRAD_ON_SIDE_STREAM): the stream the consumer of dW runs on -- for the folded attention weights
    the side stream of `EpipolarTransformer.fold_layers` (autograd replays `_FoldWeights.backward`
    on the stream of its forward).  The split-k GEMM is then launched THERE, behind the main stream's
    dy.  Ordering: same-stream FIFO with its consumer; dy and x are kept alive for the side stream
    (`record_stream`)."""

    @staticmethod
    def forward(ctx, x, w, bias, wgrad_stream=None):
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.wgrad_stream = wgrad_stream
        return x @ w.T if bias is None else torch.addmm(bias, x, w.T)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dy @ w if ctx.needs_input_grad[0] else None
        dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            if dy.shape[1] % 4 == 0 and x.shape[1] % 4 == 0:
                side = ctx.wgrad_stream if (WGRAD_ON_SIDE_STREAM and dy.is_cuda) else None
                if side is not None:
                    side.wait_stream(torch.cuda.current_stream())
                    dy.record_stream(side)
                    x.record_stream(side)
                with torch.cuda.stream(side) if side is not None else _nullcontext():
                    if want_db:      # the bias gradient falls out of the dY stream of the same kernel
                        dw, db = gemm_tn(dy, x, colsum=True)
                    else:
                        dw = gemm_tn(dy, x)
            else:
                dw = dy.T @ x
        if want_db and db is None:
            db = dy.sum(0)
        return dx, dw, db, None


class _LayerNorm(torch.autograd.Function):
    """nn.LayerNorm over the last dimension on ps_layer_norm_* (csrc/layer_norm.hip)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, fork):
        lib = _lib.load()
        dim = x.shape[-1]
        x2 = x.reshape(-1, dim).contiguous()
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        stats = torch.empty((2, rows), dtype=torch.float32, device=x.device)
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        _lib.check(lib.ps_layer_norm_forward(rows, dim, C.c_float(eps), _p(x2), _p(g), _p(b), _p(y),
                                             _p(stats[0]), _p(stats[1]), _stream()),
                   "ps_layer_norm_forward")
        ctx.save_for_backward(x2, g, stats)
        if fork:   # second output: x itself, for the residual branch (its gradient is added to
            return y.view(x.shape), x.view_as(x)   # dx inside the backward kernel)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy, d_res=None):
        lib = _lib.load()
        x2, g, stats = ctx.saved_tensors
        rows, dim = x2.shape
        if dy is None:
            dy = torch.zeros_like(x2)
        dy2 = dy.reshape(rows, dim).contiguous()
        res = None if d_res is None else d_res.reshape(rows, dim).contiguous()
        dx = torch.empty_like(x2)
        dgb = torch.empty((2, dim), dtype=torch.float32, device=x2.device)
        ws = torch.empty((lib.ps_layer_norm_workspace_floats(rows, dim),), dtype=torch.float32,
                         device=x2.device)
        _lib.check(lib.ps_layer_norm_backward(rows, dim, _p(x2), _p(g), _p(stats[0]), _p(stats[1]),
                                              _p(dy2), _p(res), _p(dx), _p(dgb[0]), _p(dgb[1]),
                                              _p(ws), _stream()), "ps_layer_norm_backward")
        return dx.view(dy.shape), dgb[0], dgb[1], None, None


def _ln_kernel_applies(x: Tensor, norm: torch.nn.LayerNorm) -> bool:
    dim = x.shape[-1]
    return (x.is_cuda and x.dtype == torch.float32 and norm.elementwise_affine
            and norm.bias is not None and tuple(norm.normalized_shape) == (dim,)
            and dim % 4 == 0 and dim <= 512 and x.numel() > 0)


def layer_norm(x: Tensor, norm: torch.nn.LayerNorm) -> Tensor:
    """`norm(x)` for an nn.LayerNorm over the last dimension; the HIP kernels when they apply
    (GPU fp32, affine, dim % 4 == 0, dim <= 512), the library op otherwise."""
    if _ln_kernel_applies(x, norm):
        return _LayerNorm.apply(x, norm.weight, norm.bias, float(norm.eps), False)
    return norm(x)


def layer_norm_fork(x: Tensor, norm: torch.nn.LayerNorm):
    """(norm(x), x) for a pre-norm residual block `x + f(norm(x))`: use the second value for the
    residual add and the gradient arriving through it is added to dx inside the LayerNorm
    backward kernel instead of a separate accumulation pass."""
    if _ln_kernel_applies(x, norm):
        return _LayerNorm.apply(x, norm.weight, norm.bias, float(norm.eps), True)
    return norm(x), x


def _pad4(n: int) -> int:
    return (n + 3) & ~3


# feature-map gradient: two passes (token gradients, then the tile gather) or the single pass
# that rebuilds every token's gradient per tile; PS_DFMAP_TWO_PASS=0 selects the latter
TWO_PASS_FEATURE_GRAD = _os.environ.get("PS_DFMAP_TWO_PASS", "1") != "0"


# The feature-map gradient (0.65 ms of HBM-bound kernels at BASELINE configs[1]) runs on a side stream,
# under the first layer's weight-gradient / input-gradient GEMMs and LayerNorm backward (matrix-pipe
# bound), and its token lists -- geometry only -- are binned there while the LAST layer's backward still
# runs.  PS_DFMAP_OVERLAP=0: everything inline on the caller's stream.
OVERLAP_FEATURE_GRAD = _os.environ.get("PS_DFMAP_OVERLAP", "1") != "0"
_FGRAD_STREAMS: dict = {}


def _fgrad_stream(device) -> "torch.cuda.Stream":
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _FGRAD_STREAMS:
        _FGRAD_STREAMS[key] = torch.cuda.Stream(device=device)
    return _FGRAD_STREAMS[key]


class _JoinFeatureGrad(torch.autograd.Function):
    """Identity on the feature map as the attention layers see it.  Its backward runs after theirs
    (it is their only consumer of that gradient) and makes the caller's stream wait for the deferred
    feature-map gradient that `FeatureGradBatch.flush` launched on the side stream."""

    @staticmethod
    def forward(ctx, fmap, batch):
        ctx.batch = batch
        return fmap.view_as(fmap)

    @staticmethod
    def backward(ctx, g):
        ctx.batch.join()
        return g, None


_PARKED: list = []     # (events, tensors ...) a FeatureGradBatch could not wait for when it was released


def _any_capture_in_progress() -> bool:
    """Is this thread's current stream capturing?  (All torch exposes.  A `global`-mode capture begun on another
    thread is not visible from here: _release therefore asks the events with query() first -- which never blocks
    and is legal during any capture -- and only waits on the host for events that are really still pending.)"""
    try:
        if not torch.cuda.is_available():
            return False
        return bool(torch.cuda.is_current_stream_capturing())
    except Exception:
        return True


def _drain_parked() -> None:
    """Drop the parked references whose events have completed (event.query() never blocks and is legal during a
    capture on another stream)."""
    keep = []
    for entry in _PARKED:
        try:
            if all(ev.query() for ev in entry[0]):
                continue
        except Exception:
            pass
        keep.append(entry)
    _PARKED[:] = keep


class FeatureGradBatch:
    """Defers the feature-map gradients of attention layers that share one geometry and one
    feature map (the layers of an EpipolarTransformer) so that they are scattered in ONE pass
    (ps_epipolar_feature_grad).  Layers register in forward order; autograd runs their
    backward in reverse, each parks its coefficients here, and the first-registered layer --
    whose backward is necessarily the last -- flushes and returns the summed gradient.  One
    batch per forward pass; the layers must be chained (each feeds the next), as in
    Transformer.forward (transformer.py:67-71).

    `attach(fmap)` (optional) returns the map the layers should be given: with it the flush runs on a
    side stream and the returned gradient is only waited for where autograd hands it on (the query
    tokens must be taken from the ORIGINAL map, so that nothing is accumulated into the gradient
    before the wait)."""

    MAX_PER_LAUNCH = 2

    def __init__(self, overlap: bool | None = None):
        self.registered = 0
        self.pending = []
        self.overlap = OVERLAP_FEATURE_GRAD if overlap is None else overlap
        self._attached = False
        self._done = None        # event: the side stream finished the flush
        self._keep = None        # tensors the side stream is using until then
        self._bins = None        # (token lists, scratch words) binned ahead of the flush
        self._bins_inputs = None # (xy, flags, event) the side stream's binning reads / ends with

    def attach(self, fmap: Tensor) -> Tensor:
        if not (self.overlap and fmap.is_cuda and fmap.requires_grad and torch.is_grad_enabled()):
            return fmap
        self._attached = True
        return _JoinFeatureGrad.apply(fmap, self)

    def join(self) -> None:
        if self._done is not None:
            torch.cuda.current_stream().wait_event(self._done)
        self._done = self._keep = None

    def _release(self) -> None:
        """The join node never ran (an exception between flush and join, a gradient nobody asked for):
        what the side stream is reading must outlive its kernels.  Outside any capture: wait for them on the
        host before the references go.  While ANY stream of the process may be capturing (this can run from
        the garbage collector, on whatever thread; a host-side event wait would invalidate a `global`-mode
        capture in progress) or when the wait itself fails: the references are parked on a module-level list
        that the next flush / release outside a capture drains.  Every attribute is read with a default: __del__
        also runs for an object whose __init__ raised."""
        done, keep = getattr(self, "_done", None), getattr(self, "_keep", None)
        bins_inputs = getattr(self, "_bins_inputs", None)
        self._done = self._keep = self._bins_inputs = None
        pending = [done] if (done is not None and keep is not None) else []
        if bins_inputs is not None:
            pending.append(bins_inputs[2])
        if not pending:
            return
        _drain_parked()
        try:
            if all(ev.query() for ev in pending):      # the side stream is done already: nothing to wait for
                return
        except Exception:
            pass
        if _any_capture_in_progress():
            _PARKED.append((pending, keep, bins_inputs))
            return
        try:
            for ev in pending:
                ev.synchronize()
        except Exception:
            _PARKED.append((pending, keep, bins_inputs))

    def register(self) -> int:
        if self.pending:
            raise RuntimeError("FeatureGradBatch reused while gradients are still parked")
        self.registered += 1
        return self.registered - 1

    def _use_side(self, t: Tensor) -> bool:
        return self.overlap and self._attached and t.is_cuda

    def park(self, qin, attn, dout, ds, desc=None, xy=None, flags=None):
        if (not self.pending and desc is not None and self._use_side(xy) and TWO_PASS_FEATURE_GRAD
                and self._bins is None and self.registered > 1):
            # first backward of the batch (the last layer's): the token lists depend on the sampling
            # geometry alone -- bin them on the side stream now, under the layers' backward
            lib = _lib.load()
            side, main = _fgrad_stream(xy.device), torch.cuda.current_stream()
            words = max(flags.numel(), lib.ps_epipolar_ray_box_words(C.byref(desc)))
            boxes = torch.empty((words,), dtype=torch.int32, device=xy.device)
            boxes.record_stream(side)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                _lib.check(lib.ps_epipolar_feature_bins(C.byref(desc), _p(xy), _p(flags), _p(boxes),
                                                        _stream()), "ps_epipolar_feature_bins")
                binned = torch.cuda.Event()
                binned.record(side)
            self._bins = boxes
            # (the side stream reads the geometry of THIS layer's node, which the engine releases when
            # this backward returns: held until the flush takes over, or until _release)
            self._bins_inputs = (xy, flags, binned)
        self.pending.append((qin, attn, dout, ds))

    def __del__(self):
        self._release()
        # the first-registered layer never ran its backward (its output was detached or unused)
        # while later layers parked their terms: those feature-map gradients were NOT applied
        if getattr(self, "pending", None):
            import warnings
            warnings.warn(f"FeatureGradBatch dropped the feature-map gradients of "
                          f"{len(self.pending)} attention layer(s): the first layer of the batch "
                          f"did not take part in the backward pass (layers must be chained)",
                          RuntimeWarning)

    def flush(self, desc, fmap, xy, flags, c):
        lib = _lib.load()
        total = None
        prebinned = self._bins is not None
        boxes = self._bins if prebinned else torch.empty(
            (max(flags.numel(), lib.ps_epipolar_ray_box_words(C.byref(desc))),),
            dtype=torch.int32, device=fmap.device)
        self._bins = None
        bins_inputs, self._bins_inputs = self._bins_inputs, None     # (alive to the end of this call)
        # two-pass scheme: token gradients once (d(kv) of the reference: b v (v-1) h w s c floats of
        # caller-owned scratch, 0.94 GB at BASELINE configs[1], 1.6 GB at configs[3] -- it sits in
        # the backward-time peak, INTEGRATION.md "memory"), then the tile gather.  One scratch for
        # all launch groups; if it cannot be allocated (or PS_DFMAP_SCRATCH_MAX_MB says it is too
        # much) the single-pass kernel -- no scratch, ~15 % slower -- takes over.
        scratch = None
        if TWO_PASS_FEATURE_GRAD:
            need = lib.ps_epipolar_token_grad_floats(C.byref(desc))
            limit_mb = float(_os.environ.get("PS_DFMAP_SCRATCH_MAX_MB", "0") or 0)
            if not (limit_mb > 0 and need * 4 > limit_mb * (1 << 20)):
                try:
                    scratch = torch.empty((need,), dtype=torch.float32, device=fmap.device)
                except torch.OutOfMemoryError:
                    scratch = None
        use_side = self._use_side(fmap)
        groups = []
        while self.pending:
            groups.append(self.pending[:self.MAX_PER_LAUNCH])
            self.pending = self.pending[self.MAX_PER_LAUNCH:]
        dfmaps = [torch.empty_like(fmap) for _ in groups]      # (allocated on the caller's stream)
        if use_side:
            side = _fgrad_stream(fmap.device)
            side.wait_stream(torch.cuda.current_stream())
        elif prebinned:     # lists were binned on the side stream: the caller's stream needs them
            torch.cuda.current_stream().wait_stream(_fgrad_stream(fmap.device))
        with torch.cuda.stream(side) if use_side else _nullcontext():
            for gi, (group, dfmap) in enumerate(zip(groups, dfmaps)):
                n = len(group)
                arr = lambda k, off=0: (C.c_void_p * n)(*[t[k].data_ptr() + 4 * off for t in group])
                if scratch is not None and prebinned:
                    _lib.check(lib.ps_epipolar_feature_grad_binned(
                        C.byref(desc), C.c_int32(n), _p(xy), _p(flags), arr(0), arr(1), arr(2), arr(3),
                        _p(dfmap), _p(boxes), _p(scratch), _stream()),
                        "ps_epipolar_feature_grad_binned")
                elif scratch is not None:
                    _lib.check(lib.ps_epipolar_feature_grad_two_pass(
                        C.byref(desc), C.c_int32(n), _p(xy), _p(flags), arr(0), arr(1), arr(2), arr(3),
                        _p(dfmap), _p(boxes), _p(scratch), _stream()),
                        "ps_epipolar_feature_grad_two_pass")
                    prebinned = True       # (the lists of this geometry are in `boxes` now)
                else:
                    _lib.check(lib.ps_epipolar_feature_grad(
                        C.byref(desc), C.c_int32(n), _p(xy), _p(flags), arr(0), arr(1), arr(2), arr(3),
                        _p(dfmap), _p(boxes), _stream()), "ps_epipolar_feature_grad")
                total = dfmap if total is None else total + dfmap
            if use_side:
                self._done = torch.cuda.Event()
                self._done.record(side)
                # alive until the caller's stream has waited (join): the side stream is still reading.
                # EVERY operand of the launches above, not only what was allocated here: xy / flags /
                # fmap are saved tensors of the calling autograd node, which the engine releases the
                # moment this backward returns -- when nothing else holds the sampling geometry (bench.py's
                # path_a) their blocks went back to the caller's stream while the side stream was still
                # reading them, the next allocation of the backward overwrote the sample coordinates under
                # the tile gather and its addresses went wild (GPU memory fault in replayed hipGraphs with a
                # second process on the device, round 4; profiles/r5_fault_root_cause.txt)
                self._keep = (groups, dfmaps, boxes, scratch, total, xy, flags, fmap, bins_inputs)
        return total


class _FusedEpipolarAttention(torch.autograd.Function):
    """(fmap, qin) -> (out, attn); see csrc/epipolar_attention.hip.

    qin and out are ONE row-major matrix each, a row being the heads' blocks
    [q~_h (c) | u_h (P) | e_h (v-1) | pad] resp. [fbar_h | pbar_h | abar_h | pad] of width
    Lh = head_width(...): exactly what a batched per-head GEMM produces / consumes, so one GEMM
    feeds the kernel and one consumes it, with no permutation or concatenation in between."""

    @staticmethod
    def head_width(c, octaves, ov, has_e):
        return _pad4(c + 2 * octaves + (ov if has_e else 0))

    @staticmethod
    def _desc(dims, has_e):
        b, v, h, w, s, c, heads, octaves = dims
        lh = _FusedEpipolarAttention.head_width(c, octaves, v - 1, has_e)
        d = _desc(*dims)
        d.ld_q = d.ld_u = d.ld_e = d.ld_f = d.ld_p = d.ld_a = heads * lh
        d.hs_in = d.hs_out = lh
        # the head stride is rounded up to a multiple of 4: the kernels zero the 0 ... 3 floats behind
        # each head's last block themselves (no clearing pass over the [R, heads * lh] matrices)
        d.tail_pad_in = d.tail_pad_out = lh - (c + 2 * octaves + ((v - 1) if has_e else 0))
        return d, lh

    @staticmethod
    def forward(ctx, dims, scale, has_e, fmap, xy, flags, rd, qin, batch=None):
        lib = _lib.load()
        b, v, h, w, s, c, heads, octaves = dims
        R, T, P = b * v * h * w, s * (v - 1), 2 * octaves
        d, lh = _FusedEpipolarAttention._desc(dims, has_e)
        fmap, qin = fmap.contiguous(), qin.contiguous()
        assert qin.shape == (R, heads * lh)
        out = torch.empty((R, heads * lh), dtype=torch.float32, device=fmap.device)
        attn = torch.empty((R, heads, T), dtype=torch.float32, device=fmap.device)
        col = lambda t, off: C.c_void_p(t.data_ptr() + 4 * off)
        _lib.check(lib.ps_epipolar_attention_forward(
            C.byref(d), _p(fmap), _p(xy), _p(flags), _p(rd), col(qin, 0), col(qin, c),
            col(qin, c + P) if has_e else None, C.c_float(scale), col(out, 0), col(out, c),
            col(out, c + P), _p(attn), _stream()), "ps_epipolar_attention_forward")
        ctx.dims, ctx.scale, ctx.has_e = dims, scale, has_e
        ctx.batch = batch
        ctx.batch_index = batch.register() if batch is not None else -1
        ctx.save_for_backward(fmap, xy, flags, rd, qin, attn, out)
        ctx.mark_non_differentiable(attn)
        ctx.set_materialize_grads(False)    # (no zero-filled [R, heads, T] "gradient" of attn per backward)
        return out, attn

    @staticmethod
    def backward(ctx, dout, _dattn):
        lib = _lib.load()
        fmap, xy, flags, rd, qin, attn, out = ctx.saved_tensors
        b, v, h, w, s, c, heads, octaves = ctx.dims
        R, T, P, ov = b * v * h * w, s * (v - 1), 2 * octaves, v - 1
        d, lh = _FusedEpipolarAttention._desc(ctx.dims, ctx.has_e)
        dout = torch.zeros_like(out) if dout is None else dout.contiguous()
        f32 = dict(dtype=torch.float32, device=fmap.device)
        dqin = torch.empty((R, heads * lh), **f32)
        ds = torch.empty((R, heads, T), **f32)
        dfmap = boxes = None
        deferred = ctx.batch is not None and ctx.needs_input_grad[3]
        if ctx.needs_input_grad[3] and not deferred:
            dfmap = torch.empty_like(fmap)
            boxes = torch.empty((R * ov,), dtype=torch.int32, device=fmap.device)
        col = lambda t, off: C.c_void_p(t.data_ptr() + 4 * off)
        de = col(dqin, c + P) if ctx.has_e else None
        _lib.check(lib.ps_epipolar_attention_backward(
            C.byref(d), _p(fmap), _p(xy), _p(flags), _p(rd), col(qin, 0), _p(attn), col(out, 0),
            col(out, c), col(out, c + P) if ctx.has_e else None,
            col(dout, 0), col(dout, c), col(dout, c + P) if ctx.has_e else None,
            C.c_float(ctx.scale), col(dqin, 0), col(dqin, c), de, _p(ds), _p(dfmap), _p(boxes),
            _stream()), "ps_epipolar_attention_backward")
        if deferred:
            # parked until the first layer of the batch (whose backward runs last) flushes
            ctx.batch.park(qin, attn, dout, ds, d, xy, flags)
            if ctx.batch_index == 0:
                dfmap = ctx.batch.flush(d, fmap, xy, flags, c)
        return (None, None, None, dfmap, None, None, None, dqin, None)


class _FoldWeights(torch.autograd.Function):
    """fold_attention_weights on the GPU: two launches forward, two backward
    (csrc/fold_weights.hip) instead of ~40 tiny library launches each way."""

    @staticmethod
    def forward(ctx, heads, w_q, w_kv, w_out, b_out, depth_w, depth_b, view_emb):
        lib = _lib.load()
        tensors = [t if t is None else t.detach().to(torch.float32).contiguous()
                   for t in (w_q, w_kv, w_out, b_out, depth_w, depth_b, view_emb)]
        w_q, w_kv, w_out, b_out, depth_w, depth_b, view_emb = tensors
        if w_q.is_cuda:
            # EpipolarTransformer.fold_layers calls this on a side stream and autograd replays the backward
            # there: an operand that was allocated on the caller's stream (the view embeddings) is
            # released by the engine the moment that backward returns -- to the CALLER's stream, with the
            # side stream's kernel still reading it.  record_stream defers the reuse (no-op for parameters)
            cur = torch.cuda.current_stream()
            for t in tensors:
                if t is not None:
                    t.record_stream(cur)
        inner, d_in = w_q.shape
        c, d_out, p2 = w_kv.shape[1], w_out.shape[0], depth_w.shape[1]
        ov = 0 if view_emb is None else view_emb.shape[0]
        desc = _lib.PsFoldDesc(heads, inner // heads, c, d_in, d_out, p2 // 2, ov)
        lh = _pad4(c + p2 + ov)
        f32 = dict(dtype=torch.float32, device=w_q.device)
        w_in = torch.empty((heads * lh, d_in), **f32)
        w_o_t = torch.empty((heads * lh, d_out), **f32)
        bias = torch.empty((d_out,), **f32)
        scratch = torch.empty((lib.ps_fold_scratch_floats(C.byref(desc)),), **f32)
        _lib.check(lib.ps_fold_attention_weights(
            C.byref(desc), _p(w_q), _p(w_kv), _p(w_out), _p(b_out), _p(depth_w), _p(depth_b),
            _p(view_emb), _p(w_in), _p(w_o_t), _p(bias), _p(scratch), _stream()),
            "ps_fold_attention_weights")
        ctx.desc = desc
        ctx.has = (b_out is not None, view_emb is not None)
        ctx.save_for_backward(*[t for t in tensors if t is not None], scratch)
        return w_in, w_o_t, bias

    @staticmethod
    def backward(ctx, d_w_in, d_w_o_t, d_bias):
        lib = _lib.load()
        saved = list(ctx.saved_tensors)
        scratch = saved.pop()
        has_b, has_e = ctx.has
        it = iter(saved)
        w_q, w_kv, w_out = next(it), next(it), next(it)
        b_out = next(it) if has_b else None
        depth_w, depth_b = next(it), next(it)
        view_emb = next(it) if has_e else None
        grads = [torch.empty_like(t) if t is not None else None
                 for t in (w_q, w_kv, w_out, b_out, depth_w, depth_b, view_emb)]
        h_lh = w_kv.shape[1] + depth_w.shape[1] + (view_emb.shape[0] if has_e else 0)
        rows = ctx.desc.heads * _pad4(h_lh)
        f32 = dict(dtype=torch.float32, device=w_q.device)
        d_w_in = (torch.zeros((rows, w_q.shape[1]), **f32) if d_w_in is None
                  else d_w_in.contiguous())
        d_w_o_t = (torch.zeros((rows, w_out.shape[0]), **f32) if d_w_o_t is None
                   else d_w_o_t.contiguous())
        d_bias = torch.zeros((w_out.shape[0],), **f32) if d_bias is None else d_bias.contiguous()
        back = torch.empty_like(scratch)
        _lib.check(lib.ps_fold_attention_weights_backward(
            C.byref(ctx.desc), _p(w_q), _p(w_kv), _p(w_out), _p(b_out), _p(depth_w), _p(depth_b),
            _p(view_emb), _p(scratch), _p(d_w_in), _p(d_w_o_t), _p(d_bias), _p(back),
            *[_p(g) for g in grads], _stream()), "ps_fold_attention_weights_backward")
        return (None, *grads)


def fold_attention_weights(*, w_q: Tensor, w_kv: Tensor, w_out: Tensor, b_out: Tensor | None,
                           heads: int, depth_w: Tensor, depth_b: Tensor,
                           view_emb: Tensor | None = None):
    """Every linear map on either side of the kernel folded into ONE weight matrix per side
    (GPU tensors: ps_fold_attention_weights; otherwise the torch statement of the same
    algebra, fold_attention_weights_torch)."""
    if w_q.is_cuda:
        return _FoldWeights.apply(heads, w_q, w_kv, w_out, b_out, depth_w, depth_b, view_emb)
    return fold_attention_weights_torch(w_q=w_q, w_kv=w_kv, w_out=w_out, b_out=b_out, heads=heads,
                                        depth_w=depth_w, depth_b=depth_b, view_emb=view_emb)


def fold_attention_weights_torch(*, w_q: Tensor, w_kv: Tensor, w_out: Tensor,
                                 b_out: Tensor | None, heads: int, depth_w: Tensor,
                                 depth_b: Tensor, view_emb: Tensor | None = None):
    """Every linear map on either side of the kernel folded into ONE weight matrix per side.
    With A = [I_c; W_d^T; E] (rows: identity, depth-encoding weights, view embeddings):
        [q~_h; u_h; e_h] = (A W_k,h^T) W_q,h x                       -> w_in   [H*Lh, d]
        y = sum_h W_o,h (W_v,h A^T) [fbar_h; pbar_h; abar_h] + bias   -> w_o_t  [H*Lh, d_out]
        bias = b_o + sum_h W_o,h W_v,h b_d                            (softmax weights sum to 1)
    Seven small launches (two matmuls per side + bias), differentiable through autograd,
    recomputed every call; rows are laid out per head, which is the kernel's layout."""
    inner, d_in = w_q.shape
    dh = inner // heads
    c = w_kv.shape[1]
    d_out = w_out.shape[0]
    P = depth_w.shape[1]
    has_e = view_emb is not None
    lh = _pad4(c + P + (view_emb.shape[0] if has_e else 0))
    rows = [torch.eye(c, device=w_q.device, dtype=w_q.dtype), depth_w.T]
    if has_e:
        rows.append(view_emb)
    n_rows = c + P + (view_emb.shape[0] if has_e else 0)
    if lh != n_rows:
        rows.append(torch.zeros((lh - n_rows, c), device=w_q.device, dtype=w_q.dtype))
    a_full = torch.cat(rows, 0)                                        # [Lh, c]
    w_k = w_kv[:inner].reshape(heads, dh, c)
    w_v = w_kv[inner:].reshape(heads, dh, c)
    k_t = torch.matmul(a_full, w_k.transpose(1, 2))                    # [H, Lh, dh] = A W_k^T
    w_in = torch.bmm(k_t, w_q.reshape(heads, dh, d_in)).reshape(heads * lh, d_in)
    v_t = torch.matmul(a_full, w_v.transpose(1, 2))                    # [H, Lh, dh] = A W_v^T
    w_o_h = w_out.reshape(d_out, heads, dh).permute(1, 2, 0)           # [H, dh, d_out] (view)
    w_o_t = torch.bmm(v_t, w_o_h).reshape(heads * lh, d_out)
    vb = torch.matmul(w_v, depth_b).reshape(inner)                     # W_v,h b_d
    bias = torch.mv(w_out, vb) if b_out is None else torch.addmv(b_out, w_out, vb)
    return w_in, w_o_t, bias


def fused_cross_attention(x: Tensor, fmap_nhwc: Tensor, geo: EpipolarGeometry, *, w_q: Tensor,
                          w_kv: Tensor, w_out: Tensor, b_out: Tensor | None, heads: int,
                          depth_w: Tensor, depth_b: Tensor, octaves: int,
                          view_emb: Tensor | None = None, return_attn: bool = False,
                          folded=None, batch: FeatureGradBatch | None = None, wgrad_stream=None):
    """Attention(x, z=kv) of the reference (attention.py:54-70) for kv = gathered features +
    Linear(PE(relative disparity)) [+ view embedding], without ever forming kv.

    x [R, 1, d] (already layer-normed), fmap_nhwc [b, v, h, w, c]; w_q [inner, d],
    w_kv [2*inner, c], w_out [d, inner]; depth_w [c, 2*octaves], depth_b [c];
    view_emb [v-1, c] (already permuted) or None; `folded` = fold_attention_weights(...) of
    the same weights when the caller computed it ahead; `batch` = a FeatureGradBatch shared by
    the chained layers of one forward pass (their feature-map gradients are then scattered
    together); `wgrad_stream` = the stream `folded` was computed on (the weight gradients of the
    two per-ray GEMMs are launched there, see _RayLinear).  Returns [R, 1, d] (and attn [R,H,1,T])."""
    b, v, h, w, c = fmap_nhwc.shape
    s = geo.xy_sample.shape[-2]
    dh = w_q.shape[0] // heads
    R, d_in = x.shape[0], x.shape[-1]
    d_out = w_out.shape[0]
    has_e = view_emb is not None
    dims = (b, v, h, w, s, c, heads, octaves)
    if folded is None:
        folded = fold_attention_weights(w_q=w_q, w_kv=w_kv, w_out=w_out, b_out=b_out, heads=heads,
                                        depth_w=depth_w, depth_b=depth_b, view_emb=view_emb)
    w_in, w_o_t, bias = folded
    qin = _RayLinear.apply(x.reshape(R, d_in), w_in, None, wgrad_stream)   # heads x [q~ | u | e]
    fused, attn = _FusedEpipolarAttention.apply(
        dims, float(dh) ** -0.5, has_e, fmap_nhwc.reshape(b * v, h, w, c), geo.xy_sample,
        geo.flags, geo.rel_disparity, qin, batch)
    out = _RayLinear.apply(fused, w_o_t.T, bias, wgrad_stream).reshape(R, 1, d_out)
    if return_attn:
        return out, attn.reshape(R, heads, 1, -1)
    return out
