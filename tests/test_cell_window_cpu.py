"""Host-side checks of csrc/cell_window.h (the header is plain arithmetic: it compiles with g++).

The cell window of a (view, Gaussian) pair is produced once by the preprocess and decoded three ways on the
device: `quad_cell_mask` (rows forward, one quadrant wave), `tile_cell_mask` (the 16 cells of a tile) and
`tile_quad_mask` (tile backward's refine, round 6).  Here: the three decoders agree on every window the
producer can emit (small and large footprints, windows far outside the tile, saturated coordinates), and the
producer is conservative against a brute-force walk over pixels (a pixel with alpha >= alpha_min always lies
in a flagged cell) -- the property that makes every cull built on it result-neutral."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pixelsplat_amd", "csrc")

SHIM = r"""
#include <cmath>
#include <cstdint>
#define __log2f log2f
#include "cell_window.h"
extern "C" {
unsigned cw_tile_quad_mask(unsigned x, unsigned y, unsigned z, unsigned w, int tx, int ty) {
  return ps::tile_quad_mask(make_uint4(x, y, z, w), tx, ty);
}
unsigned cw_tile_cell_mask(unsigned x, unsigned y, unsigned z, unsigned w, int tx, int ty) {
  return ps::tile_cell_mask(make_uint4(x, y, z, w), tx, ty);
}
unsigned cw_quad_cell_mask(unsigned x, unsigned y, unsigned z, unsigned w, int qcx, int qcy) {
  return ps::quad_cell_mask<true>(make_uint4(x, y, z, w), qcx, qcy);
}
void cw_cell_window(float px, float py, float cx, float cy, float cz, float o, float amin, unsigned* out) {
  uint4 r = ps::cell_window(px, py, cx, cy, cz, o, amin);
  out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
}
"""


@pytest.fixture(scope="module")
def cw(tmp_path_factory):
    inc = "/opt/rocm/include"
    if not os.path.exists(os.path.join(inc, "hip", "hip_runtime.h")):
        pytest.skip("no HIP headers")
    d = tmp_path_factory.mktemp("cw")
    src = d / "shim.cpp"
    src.write_text(SHIM)
    lib = d / "libcw.so"
    # -ffp-contract=off: the device build of the producer's unit (raster_preprocess.hip) has no FMA contraction
    subprocess.run(["g++", "-O1", "-ffp-contract=off", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I", inc,
                    "-I", CSRC, str(src), "-o", str(lib)], check=True, capture_output=True)
    L = C.CDLL(str(lib))
    for f in (L.cw_tile_quad_mask, L.cw_tile_cell_mask, L.cw_quad_cell_mask):
        f.restype = C.c_uint
        f.argtypes = [C.c_uint] * 4 + [C.c_int] * 2
    L.cw_cell_window.restype = None
    L.cw_cell_window.argtypes = [C.c_float] * 7 + [C.POINTER(C.c_uint)]
    return L


def _window(L, px, py, cx, cy, cz, o, amin=1.0 / 255.0):
    out = (C.c_uint * 4)()
    L.cw_cell_window(px, py, cx, cy, cz, o, amin, out)
    return tuple(int(v) for v in out)


def _random_pairs(rng, n):
    """Screen-space Gaussians of every footprint class: needles, blobs of a few pixels, footprints over 8 cells
    (the window's `big` form), centres off the image, opacities around the threshold."""
    for _ in range(n):
        px, py = rng.uniform(-40, 300, 2)
        s1 = 10 ** rng.uniform(-0.6, 1.6)
        s2 = s1 * 10 ** rng.uniform(-1.5, 0.0)
        th = rng.uniform(0, np.pi)
        c, s = np.cos(th), np.sin(th)
        cov = np.array([[c, -s], [s, c]]) @ np.diag([s1 * s1, s2 * s2]) @ np.array([[c, s], [-s, c]])
        inv = np.linalg.inv(cov)
        o = float(rng.choice([rng.uniform(0.0, 1.0), 1.0 / 255.0 * rng.uniform(0.9, 1.2), 0.99]))
        yield float(px), float(py), float(inv[0, 0]), float(inv[0, 1]), float(inv[1, 1]), o


def _quad_bits_from_cells(m16):
    q = 0
    for k, bits in enumerate((0x0033, 0x00CC, 0x3300, 0xCC00)):
        if m16 & bits:
            q |= 1 << k
    return q


def test_decoders_agree(cw):
    rng = np.random.default_rng(7)
    n_small = n_big = 0
    for px, py, cx, cy, cz, o in _random_pairs(rng, 3000):
        w = _window(cw, px, py, cx, cy, cz, o)
        n_big += w[3] != 0
        n_small += w[3] == 0 and (w[0] | w[1]) != 0
        # tiles around the footprint, tiles far away, tile 0
        tcx, tcy = int(px) // 16, int(py) // 16
        tiles = [(tcx + dx, tcy + dy) for dx in range(-3, 4) for dy in range(-3, 4)]
        tiles += [(0, 0), (15, 15), (tcx + 40, tcy), (tcx, tcy - 40), (2000, 2000)]
        for tx, ty in tiles:
            if tx < 0 or ty < 0:
                continue
            m16 = cw.cw_tile_cell_mask(*w, tx, ty)
            assert cw.cw_tile_quad_mask(*w, tx, ty) == _quad_bits_from_cells(m16), (w, tx, ty)
            for quad in range(4):
                mq = cw.cw_quad_cell_mask(*w, 4 * tx + 2 * (quad & 1), 4 * ty + 2 * (quad >> 1))
                want = 0
                for r in range(4):
                    i, j = 2 * (quad & 1) + (r & 1), 2 * (quad >> 1) + (r >> 1)
                    if m16 >> (4 * j + i) & 1:
                        want |= 1 << r
                assert mq == want, (w, tx, ty, quad)
    assert n_small > 500 and n_big > 100      # both window forms were exercised


def test_hand_made_windows(cw):
    """Windows the producer would not emit at random: anchors at the saturation bounds, negative anchors, full
    and single-bit masks, big ranges that end exactly at a quadrant boundary."""
    def pack(lo, hi):
        return (lo & 0xFFFF) | ((hi & 0xFFFF) << 16)
    cases = []
    for ax, ay in ((-30000, -30000), (-5, -3), (0, 0), (3, 2), (29990, 29990)):
        for lo, hi in ((0xFFFFFFFF, 0xFFFFFFFF), (1, 0), (0, 0x80000000), (0x00018000, 0x01000000), (0, 0)):
            cases.append((lo, hi, pack(ax, ay), 0))
    for x0, x1, y0, y1 in ((0, 1, 0, 1), (2, 2, 2, 2), (1, 2, 1, 2), (-30000, 30000, -30000, 30000), (4, 3, 0, 9),
                           (-8, -1, -8, -1), (3, 40, 5, 6)):
        cases.append((pack(x0, x1), pack(y0, y1), 0, 1))
    for w in cases:
        for tx in (0, 1, 2, 7, 7497):
            for ty in (0, 1, 3, 7497):
                m16 = cw.cw_tile_cell_mask(*w, tx, ty)
                assert cw.cw_tile_quad_mask(*w, tx, ty) == _quad_bits_from_cells(m16), (w, tx, ty)


def test_window_is_conservative(cw):
    """Every pixel whose alpha reaches alpha_min lies in a cell the window flags (fp64 walk over the pixels)."""
    rng = np.random.default_rng(11)
    amin = 1.0 / 255.0
    checked = 0
    for px, py, cx, cy, cz, o in _random_pairs(rng, 400):
        w = _window(cw, px, py, cx, cy, cz, o)
        x0, y0 = int(np.floor(px)) - 24, int(np.floor(py)) - 24
        xs, ys = np.meshgrid(np.arange(x0, x0 + 49), np.arange(y0, y0 + 49))
        dx, dy = px - xs, py - ys
        power = -0.5 * (cx * dx * dx + cz * dy * dy) - cy * dx * dy
        alpha = np.minimum(0.99, o * np.exp(power))
        hit = (power <= 0) & (alpha >= amin)
        for yy, xx in zip(*np.nonzero(hit)):
            X, Y = int(xs[yy, xx]), int(ys[yy, xx])
            if X < 0 or Y < 0:
                continue
            tx, ty = X // 16, Y // 16
            m16 = cw.cw_tile_cell_mask(*w, tx, ty)
            i, j = (X % 16) // 4, (Y % 16) // 4
            assert m16 >> (4 * j + i) & 1, (w, X, Y)
            q = ((X % 16) // 8) | (((Y % 16) // 8) << 1)
            assert cw.cw_tile_quad_mask(*w, tx, ty) >> q & 1
            checked += 1
    assert checked > 2000
