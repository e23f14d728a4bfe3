"""ORACLE (test infrastructure, NOT product code) -- ctypes front-end of raster_ref.c.

CPU restatement of the third-party rasterizer pixelSplat calls at
/root/reference/src/model/decoder/cuda_splatting.py:99-124 (module
`diff_gaussian_rasterization`, unpinned, source absent).  PARITY UNPINNED: see the header
of raster_ref_impl.inc and DESIGN.md.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    path = os.path.join(_HERE, "libraster_ref.so")
    srcs = [os.path.join(_HERE, f) for f in ("raster_ref.c", "raster_ref_impl.inc")]
    stale = (not os.path.exists(path)) or any(
        os.path.getmtime(s) > os.path.getmtime(path) for s in srcs
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libraster_ref.so"],
                              stdout=subprocess.DEVNULL)
    return path


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.ps_oracle_bin.restype = C.c_int64
    return _LIB


def _params_struct(real):
    class P(C.Structure):
        _fields_ = [
            ("G", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
            ("sh_degree", C.c_int32), ("K", C.c_int32), ("use_sh", C.c_int32),
            ("tanfovx", real), ("tanfovy", real),
            ("near_cull", real), ("guard", real), ("lowpass", real), ("w_eps", real),
            ("lambda_floor", real), ("alpha_max", real), ("alpha_min", real),
            ("t_min", real), ("det2_eps", real),
        ]
    return P


_P32 = _params_struct(C.c_float)
_P64 = _params_struct(C.c_double)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@dataclass
class ForwardState:
    """Everything the oracle computed for one view (every intermediate is exposed)."""
    params: object
    dtype: object
    means: np.ndarray
    cov6: np.ndarray
    sh: np.ndarray | None
    colors: np.ndarray | None
    opacity: np.ndarray
    view: np.ndarray
    proj: np.ndarray
    campos: np.ndarray
    bg: np.ndarray
    radii: np.ndarray
    tiles_touched: np.ndarray
    rect: np.ndarray
    depth: np.ndarray
    xy: np.ndarray
    conic_opacity: np.ndarray
    rgb: np.ndarray
    clamped: np.ndarray
    point_list: np.ndarray
    keys: np.ndarray
    ranges: np.ndarray
    image: np.ndarray
    final_T: np.ndarray
    n_contrib: np.ndarray

    @property
    def num_rendered(self) -> int:
        return int(self.point_list.shape[0])


def forward(means, cov6, opacity, view, proj, campos, bg, H, W, tanfovx, tanfovy,
            sh=None, colors=None, sh_degree=0, dtype=np.float32, threads: int | None = None,
            **const_overrides) -> ForwardState:
    """One view.  `view`/`proj` are the transposed 4x4 matrices of cuda_splatting.py:84-87
    (flattened row-major of the transposed matrix, i.e. M[4*c + r])."""
    L = lib()
    f32 = dtype == np.float32
    suf = "_f32" if f32 else "_f64"
    P = (_P32 if f32 else _P64)()
    getattr(L, "ps_oracle_defaults" + suf)(C.byref(P))
    means = np.ascontiguousarray(means, dtype).reshape(-1, 3)
    G = means.shape[0]
    cov6 = np.ascontiguousarray(cov6, dtype).reshape(G, 6)
    opacity = np.ascontiguousarray(opacity, dtype).reshape(G)
    view = np.ascontiguousarray(view, dtype).reshape(16)
    proj = np.ascontiguousarray(proj, dtype).reshape(16)
    campos = np.ascontiguousarray(campos, dtype).reshape(3)
    bg = np.ascontiguousarray(bg, dtype).reshape(3)
    use_sh = sh is not None
    if use_sh:
        sh = np.ascontiguousarray(sh, dtype)
        assert sh.ndim == 3 and sh.shape[0] == G and sh.shape[2] == 3
        K = sh.shape[1]
    else:
        colors = np.ascontiguousarray(colors, dtype).reshape(G, 3)
        K = 0
    P.G, P.H, P.W, P.sh_degree, P.K, P.use_sh = G, H, W, sh_degree, K, int(use_sh)
    P.tanfovx, P.tanfovy = float(tanfovx), float(tanfovy)
    for k, v in const_overrides.items():
        setattr(P, k, v)

    radii = np.zeros(G, np.int32)
    tiles = np.zeros(G, np.uint32)
    rect = np.zeros((G, 4), np.int32)
    depth = np.zeros(G, dtype)
    xy = np.zeros((G, 2), dtype)
    co = np.zeros((G, 4), dtype)
    rgb = np.zeros((G, 3), dtype)
    clamped = np.zeros((G, 3), np.uint8)
    if threads is not None:
        os.environ["OMP_NUM_THREADS"] = str(threads)
    getattr(L, "ps_oracle_preprocess" + suf)(
        C.byref(P), _ptr(means), _ptr(cov6), _ptr(sh), _ptr(colors), _ptr(opacity), _ptr(view),
        _ptr(proj), _ptr(campos), _ptr(radii), _ptr(tiles), _ptr(rect), _ptr(depth), _ptr(xy),
        _ptr(co), _ptr(rgb), _ptr(clamped))

    gx, gy = (W + 15) // 16, (H + 15) // 16
    D = int(tiles.astype(np.int64).sum())
    plist = np.zeros(max(D, 1), np.uint32)
    keys = np.zeros(max(D, 1), np.uint64)
    ranges = np.zeros((gx * gy, 2), np.uint32)
    depth32 = np.ascontiguousarray(depth, np.float32)
    D2 = L.ps_oracle_bin(C.c_int32(G), C.c_int32(H), C.c_int32(W), _ptr(radii), _ptr(rect),
                         _ptr(depth32), _ptr(plist), _ptr(keys), _ptr(ranges))
    assert D2 == D
    plist, keys = plist[:D], keys[:D]

    image = np.zeros((3, H, W), dtype)
    final_T = np.zeros(H * W, dtype)
    n_contrib = np.zeros(H * W, np.uint32)
    pl = plist if D else np.zeros(1, np.uint32)
    getattr(L, "ps_oracle_blend_forward" + suf)(
        C.byref(P), _ptr(ranges), _ptr(pl), _ptr(xy), _ptr(co), _ptr(rgb), _ptr(bg),
        _ptr(image), _ptr(final_T), _ptr(n_contrib))
    return ForwardState(P, dtype, means, cov6, sh, colors, opacity, view, proj, campos, bg,
                        radii, tiles, rect, depth, xy, co, rgb, clamped, plist, keys, ranges,
                        image, final_T, n_contrib)


def ambiguity_mask(st: ForwardState, tol_alpha: float = 3e-6, tol_T: float = 3e-6,
                   tol_power: float = 1e-7) -> np.ndarray:
    """uint8 [H, W]: pixels whose forward walk evaluates an entry sitting on one of the blend's
    hard thresholds (bit 0: alpha ~ alpha_min, bit 1: T (1 - alpha) ~ t_min, bit 2: power ~ 0)
    within the given relative tolerances -- there two correct fp32 implementations may branch
    differently (see ps_oracle_blend_ambiguity in raster_ref_impl.inc).

    Defaults: measured on MI355X (tools/ambiguity_sweep.py, profiles/r2_ambiguity_sweep.txt):
    at BASELINE configs[1] and [4] every pixel over 1e-4 (up to 1.2e-3: one minimum-alpha
    contribution) disappears from the unmarked set at tol_alpha = tol_T = 1e-6 -- an fp32
    evaluation of `power` with the HIP kernel's log2(e)-scaled coefficients and v_exp_f32 is
    off by a few 1e-7 relative in alpha -- and the worst unmarked error is then 4.8e-7.  The
    defaults keep a 3x margin over that and mark ~0.05 % of the pixels."""
    L = lib()
    dtype = st.dtype
    suf = "_f32" if dtype == np.float32 else "_f64"
    real = C.c_float if dtype == np.float32 else C.c_double
    P = st.params
    mask = np.zeros(P.H * P.W, np.uint8)
    pl = st.point_list if st.num_rendered else np.zeros(1, np.uint32)
    getattr(L, "ps_oracle_blend_ambiguity" + suf)(
        C.byref(P), _ptr(st.ranges), _ptr(pl), _ptr(st.xy), _ptr(st.conic_opacity),
        real(tol_alpha), real(tol_T), real(tol_power), _ptr(mask))
    return mask.reshape(P.H, P.W)


def explain_threshold_pixels(st: ForwardState, image, final_T, n_contrib, select, tol: float = 1e-4,
                             tol_alpha: float = 3e-6, tol_T: float = 3e-6,
                             tol_power: float = 1e-7) -> dict:
    """Constructive check of the pixels `select` (normally the ambiguity mask) of a result under
    test (`image` [3,H,W], `final_T`, `n_contrib` [H*W]): is it the oracle's walk with nothing
    but FLAGGED decisions flipped (ps_oracle_blend_explain)?  Returns counts: selected,
    same (explained by the unflipped walk), flipped (explained with >= 1 flipped decision),
    unexplained (no such walk: a real error), exhausted (search budget), and the verdict map."""
    L = lib()
    dtype = st.dtype
    suf = "_f32" if dtype == np.float32 else "_f64"
    real = C.c_float if dtype == np.float32 else C.c_double
    P = st.params
    n = P.H * P.W
    sel = np.ascontiguousarray(np.asarray(select).reshape(n) != 0, np.uint8)
    img = np.ascontiguousarray(image, dtype).reshape(3, n)
    fT = np.ascontiguousarray(final_T, dtype).reshape(n)
    nc = np.ascontiguousarray(n_contrib, np.uint32).reshape(n)
    verdict = np.zeros(n, np.uint8)
    pl = st.point_list if st.num_rendered else np.zeros(1, np.uint32)
    getattr(L, "ps_oracle_blend_explain" + suf)(
        C.byref(P), _ptr(st.ranges), _ptr(pl), _ptr(st.xy), _ptr(st.conic_opacity), _ptr(st.rgb),
        _ptr(st.bg), real(tol_alpha), real(tol_T), real(tol_power), real(tol), _ptr(sel), _ptr(img),
        _ptr(fT), _ptr(nc), _ptr(verdict))
    return dict(selected=int(sel.sum()), same=int((verdict == 1).sum()),
                flipped=int((verdict == 2).sum()), unexplained=int((verdict == 3).sum()),
                exhausted=int((verdict == 4).sum()), verdict=verdict.reshape(P.H, P.W))


def blend_stats(st: ForwardState) -> tuple[int, int]:
    """((pixel, entry) pairs the reference walk evaluates, pairs that contribute)."""
    L = lib()
    suf = "_f32" if st.dtype == np.float32 else "_f64"
    out = np.zeros(2, np.int64)
    pl = st.point_list if st.num_rendered else np.zeros(1, np.uint32)
    getattr(L, "ps_oracle_blend_stats" + suf)(
        C.byref(st.params), _ptr(st.ranges), _ptr(pl), _ptr(st.xy), _ptr(st.conic_opacity),
        _ptr(out))
    return int(out[0]), int(out[1])


def box_evaluations(st: ForwardState, box: int = 8, alpha_min: float = 1.0 / 255.0) -> dict:
    """What a tile kernel that culls per `box` x `box` pixel block evaluates (box 8: the 8x8 quadrants of
    csrc/raster_tiles.hip's backward; box 4: the 4x4 cells of csrc/raster_cells.hip), under the product's conservative
    ellipse-vs-box bound (quadrant_mask / cell masks, restated in numpy), over the part of every tile list up to
    the tile's last contributor.  Returns dict(entries = list entries that reach at least one block, pairs =
    (entry, block) pairs, lanes = pairs x box^2, row_steps = sum over groups of four blocks (one wave64 of
    16-lane rows) of the longest of the four per-block queues -- the steps a wave of four independent rows
    takes; only meaningful for box 4).  A statistic for bench.py's `lane_efficiency*`; not a parity path."""
    n = st.num_rendered
    nb = 16 // box
    if n == 0:
        return dict(entries=0, pairs=0, lanes=0, row_steps=0)
    gx = (st.params.W + 15) // 16
    cnt = (st.ranges[:, 1] - st.ranges[:, 0]).astype(np.int64)
    tile = np.repeat(np.arange(cnt.size), cnt)
    # only the part of every list up to the tile's last contributor is walked
    nc = st.n_contrib.reshape(st.params.H, st.params.W)
    hp, wp = -(-st.params.H // 16) * 16, gx * 16
    pad = np.zeros((hp, wp), nc.dtype)
    pad[:st.params.H, :st.params.W] = nc
    last = pad.reshape(hp // 16, 16, gx, 16).max(axis=(1, 3)).reshape(-1)
    pos = np.arange(n) - np.repeat(st.ranges[:, 0].astype(np.int64), cnt)
    walked = pos < last[tile]
    ids = st.point_list.astype(np.int64)[walked]
    tile = tile[walked]
    x0 = (tile % gx * 16).astype(np.float32)
    y0 = (tile // gx * 16).astype(np.float32)
    px, py = st.xy[ids, 0].astype(np.float32), st.xy[ids, 1].astype(np.float32)
    co = st.conic_opacity[ids].astype(np.float32)
    k = np.float32(1.4426950408889634)
    A, B, Cq, op = -0.5 * k * co[:, 0], -k * co[:, 1], -0.5 * k * co[:, 2], co[:, 3]
    per_block = np.zeros((cnt.size, nb * nb), np.int64)
    with np.errstate(divide="ignore", invalid="ignore"):
        tau = np.log2(op / np.float32(alpha_min))
        pd = (A < 0) & (Cq < 0) & (4 * A * Cq - B * B > 0)
        pairs = np.zeros(ids.size, np.int64)
        e = np.float32(box - 1)
        for q in range(nb * nb):
            bx, by = x0 + box * (q % nb), y0 + box * (q // nb)
            dxlo, dxhi, dylo, dyhi = px - (bx + e), px - bx, py - (by + e), py - by
            ex = np.where(dxlo > 0, dxlo, np.where(dxhi < 0, dxhi, 0))
            ey = np.where(dylo > 0, dylo, np.where(dyhi < 0, dyhi, 0))
            qmin = np.full(ids.size, 3.0e38, np.float32)
            dy = np.minimum(dyhi, np.maximum(dylo, -B * ex / (2 * Cq)))
            qmin = np.where(ex != 0, np.minimum(qmin, -(A * ex * ex + B * ex * dy + Cq * dy * dy)), qmin)
            dx = np.minimum(dxhi, np.maximum(dxlo, -B * ey / (2 * A)))
            qmin = np.where(ey != 0, np.minimum(qmin, -(A * dx * dx + B * dx * ey + Cq * ey * ey)), qmin)
            may = ((ex == 0) & (ey == 0)) | ~(qmin > tau + 1e-4 * np.abs(tau) + 1e-3)
            hit = (tau >= 0) & (~pd | may)
            pairs += hit
            per_block[:, q] = np.bincount(tile[hit], minlength=cnt.size)
    # four blocks = the four 16-lane rows of one wave64: box 4 -> the 2x2 cells of an 8x8 quadrant
    if nb == 4:
        g = per_block.reshape(-1, 2, 2, 2, 2).transpose(0, 1, 3, 2, 4).reshape(-1, 4)
        row_steps = int(g.max(axis=1).sum())
    else:
        row_steps = int(per_block.sum())
    total = int(pairs.sum())
    return dict(entries=int((pairs > 0).sum()), pairs=total, lanes=total * box * box, row_steps=row_steps)


def quadrant_evaluations(st: ForwardState, alpha_min: float = 1.0 / 255.0) -> tuple[int, int]:
    """(list entries that reach at least one 8x8 quadrant of their tile, (entry, quadrant) pairs): box_evaluations
    with box 8 -- what the 8x8-quadrant kernels evaluate, 64 lanes per pair."""
    r = box_evaluations(st, 8, alpha_min)
    return r["entries"], r["pairs"]


def backward(st: ForwardState, dL_dimage):
    """Returns dict(means3D, means2D, cov6, sh|colors, opacity) -- the five gradients the
    reference's autograd needs (SURVEY.md section 8b 'Gradients required')."""
    L = lib()
    dtype = st.dtype
    suf = "_f32" if dtype == np.float32 else "_f64"
    P = st.params
    G = P.G
    g = np.ascontiguousarray(dL_dimage, dtype).reshape(3, P.H, P.W)
    dxy = np.zeros((G, 2), dtype)
    dconic = np.zeros((G, 3), dtype)
    dop = np.zeros(G, dtype)
    drgb = np.zeros((G, 3), dtype)
    pl = st.point_list if st.num_rendered else np.zeros(1, np.uint32)
    getattr(L, "ps_oracle_blend_backward" + suf)(
        C.byref(P), _ptr(st.ranges), _ptr(pl), _ptr(st.xy), _ptr(st.conic_opacity), _ptr(st.rgb),
        _ptr(st.bg), _ptr(st.final_T), _ptr(st.n_contrib), _ptr(g), _ptr(dxy), _ptr(dconic),
        _ptr(dop), _ptr(drgb))
    dmeans = np.zeros((G, 3), dtype)
    dcov = np.zeros((G, 6), dtype)
    dsh = np.zeros_like(st.sh) if st.sh is not None else None
    getattr(L, "ps_oracle_preprocess_backward" + suf)(
        C.byref(P), _ptr(st.means), _ptr(st.cov6), _ptr(st.sh), _ptr(st.view), _ptr(st.proj),
        _ptr(st.campos), _ptr(st.radii), _ptr(st.clamped), _ptr(dxy), _ptr(dconic), _ptr(drgb),
        _ptr(dmeans), _ptr(dcov), _ptr(dsh))
    means2d = np.zeros((G, 3), dtype)
    means2d[:, :2] = dxy
    out = dict(means3D=dmeans, means2D=means2d, cov6=dcov, opacity=dop,
               xy=dxy, conic=dconic, rgb=drgb)
    if st.sh is not None:
        out["sh"] = dsh
    else:
        out["colors"] = drgb.copy()
    return out


def parallel_backward(on: bool) -> None:
    """CPU-baseline timing only: spread the blend backward over cores (sums reorder)."""
    lib().ps_oracle_parallel_backward(C.c_int(int(on)))


def sh_basis(deg: int, dirs: np.ndarray) -> np.ndarray:
    """[N,3] unit directions -> [N,(deg+1)^2] basis values (fp64)."""
    L = lib()
    dirs = np.ascontiguousarray(dirs, np.float64).reshape(-1, 3)
    n = (deg + 1) ** 2
    out = np.zeros((dirs.shape[0], 25), np.float64)
    L.ps_oracle_sh_basis_f64.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p]
    for i, (x, y, z) in enumerate(dirs):
        L.ps_oracle_sh_basis_f64(deg, x, y, z, out[i].ctypes.data_as(C.c_void_p))
    return out[:, :n]
