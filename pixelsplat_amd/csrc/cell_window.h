// Which 4x4-pixel cells can a projected Gaussian reach with alpha >= alpha_min?  (round 6)
//
// Producer (raster_preprocess.hip, once per visible (view, Gaussian) pair) and consumers (raster_cells.hip: the
// rows forward's per-cell cull; raster_tiles.hip: the quadrant mask of the tile backward's refine -- once per
// tile-list entry each) of the 16-byte "cell window" of a pair, the fourth 16-byte word of the pair's record line:
//   small  x, y = 64-bit mask of an 8x8 window of cells (bit 8 * wy + wx), z = anchor cell (ax | ay << 16,
//          signed 16-bit each): bit set <=> the ellipse {alpha >= alpha_min} meets the box of pixel
//          centres of cell (ax + wx, ay + wy); w = 0
//   big    the ellipse's bounding box spans more than 8 cells in x or y (or the conic is not positive
//          definite): x = cx0 | cx1 << 16, y = cy0 | cy1 << 16 (signed 16-bit cell ranges, inclusive), w = 1;
//          every cell of the range counts as reached (conservative: a large splat reaches most of them)
// The test is EXACT up to its safety margins for small windows: per row of cells (a band of four pixel rows)
// the x-extent of ellipse-and-band follows in closed form -- Q(dx, dy) = a dx^2 + b dx dy + c dy^2 <= L is
// convex, its right-most point over dy in [lo, hi] lies at the ellipse's own right-most point's dy clamped to
// the band -- so a window costs one short loop over its rows instead of one quadratic-form minimisation per
// (entry, cell).  A culled (entry, cell) pair is one every pixel of the cell would have skipped
// (alpha < 1/255): results are unchanged; the margins (L inflated by 1e-4 |tau| + 1e-3, cell ranges widened
// by 1e-3 cells) cover fp32 rounding here and in the per-pixel power.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ps {

constexpr float kCwLog2e = 1.4426950408889634f;
constexpr int kCwSat = 30000;      // cell coordinates are kept in signed 16 bits

__device__ __forceinline__ int cw_sat(float v) {      // float -> int, saturating, NaN -> -kCwSat
  if (!(v > (float)-kCwSat)) return -kCwSat;
  if (v > (float)kCwSat) return kCwSat;
  return (int)v;
}
__device__ __forceinline__ uint32_t cw_pack(int lo, int hi) {
  return ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16);
}
__device__ __forceinline__ int cw_lo(uint32_t p) { return (int)(int16_t)(p & 0xFFFFu); }
__device__ __forceinline__ int cw_hi(uint32_t p) { return (int)(int16_t)(p >> 16); }

// (px, py) pixel-space centre, (con_x, con_y, con_z) conic: power = -0.5 (con_x dx^2 + con_z dy^2) - con_y dx dy
__device__ __forceinline__ uint4 cell_window(float px, float py, float con_x, float con_y, float con_z,
                                             float opacity, float alpha_min) {
  const float tau = __log2f(opacity / alpha_min);          // need Q <= tau somewhere
  if (!(tau >= 0.f)) return make_uint4(0u, 0u, 0u, 0u);    // opacity < alpha_min (or NaN): reaches nothing
  const float a = 0.5f * kCwLog2e * con_x, b = kCwLog2e * con_y, c = 0.5f * kCwLog2e * con_z;
  const float det = a * c - 0.25f * b * b;
  const uint4 everywhere = make_uint4(cw_pack(-kCwSat, kCwSat), cw_pack(-kCwSat, kCwSat), 0u, 1u);
  if (!(a > 0.f && c > 0.f && det > 0.f)) return everywhere;      // not positive definite: keep everywhere
  const float L = tau + 1e-4f * fabsf(tau) + 1e-3f;
  const float inv_det = 1.f / det;
  const float hx = sqrtf(L * c * inv_det), hy = sqrtf(L * a * inv_det);     // half extents of the ellipse
  if (!(hx < 1e8f && hy < 1e8f)) return everywhere;
  // cell i covers pixel centres 4 i .. 4 i + 3: touched iff 4 i <= hi and 4 i + 3 >= lo
  const float eps = 1e-3f;
  const int cx0 = cw_sat(ceilf((px - hx - 3.f) * 0.25f - eps)), cx1 = cw_sat(floorf((px + hx) * 0.25f + eps));
  const int cy0 = cw_sat(ceilf((py - hy - 3.f) * 0.25f - eps)), cy1 = cw_sat(floorf((py + hy) * 0.25f + eps));
  if (cx1 < cx0 || cy1 < cy0) return make_uint4(0u, 0u, 0u, 0u);           // between two cells
  if (cx1 - cx0 >= 8 || cy1 - cy0 >= 8) return make_uint4(cw_pack(cx0, cx1), cw_pack(cy0, cy1), 0u, 1u);
  // per row of cells: the x-extent of {Q <= L} over dy in the row's band.  dx is largest at
  // dy = dys = -b hx / (2 c) (the ellipse's right-most point), smallest at -dys.
  const float inv_a = 1.f / a, kappa = det * inv_a, slope = -0.5f * b * inv_a;   // dx = slope dy +- s(dy)
  const float dys = -0.5f * b * hx / c;
  uint32_t lo = 0u, hi = 0u;
  const int rows = cy1 - cy0;
  for (int k = 0; k <= rows; ++k) {
    const float Y = 4.f * (float)(cy0 + k);
    const float d_lo = fmaxf(py - (Y + 3.f), -hy), d_hi = fminf(py - Y, hy);     // dy range of the band
    if (!(d_lo <= d_hi)) continue;
    const float dyp = fminf(d_hi, fmaxf(d_lo, dys)), dym = fminf(-d_lo, fmaxf(-d_hi, dys));
    const float dx_max = slope * dyp + sqrtf(fmaxf(L - kappa * dyp * dyp, 0.f) * inv_a);
    const float dx_min = -(slope * dym + sqrtf(fmaxf(L - kappa * dym * dym, 0.f) * inv_a));
    // pixels x in [px - dx_max, px - dx_min]
    int i0 = cw_sat(ceilf((px - dx_max - 3.f) * 0.25f - eps)) - cx0;
    int i1 = cw_sat(floorf((px - dx_min) * 0.25f + eps)) - cx0;
    i0 = i0 < 0 ? 0 : i0; i1 = i1 > 7 ? 7 : i1;
    if (i0 > i1) continue;
    const uint32_t bits = ((2u << i1) - 1u) & ~((1u << i0) - 1u);
    if (k < 4) lo |= bits << (8 * k); else hi |= bits << (8 * (k - 4));
  }
  return make_uint4(lo, hi, cw_pack(cx0, cy0), 0u);
}

// The 16-bit mask (bit 4 j + i) of the 4x4 cells of tile (tx, ty) out of a pair's cell window
__device__ __forceinline__ uint32_t tile_cell_mask(uint4 w, int tx, int ty) {
  const int tcx = 4 * tx, tcy = 4 * ty;
  uint32_t small_mask = 0u;
  {
    int ox = tcx - cw_lo(w.z), oy = tcy - cw_hi(w.z);
    ox = ox < -31 ? -31 : (ox > 31 ? 31 : ox);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int wy = oy + j;
      const uint32_t word = wy < 4 ? w.x : w.y;
      uint32_t byte = (word >> (8 * (wy & 3))) & 0xFFu;
      byte = (wy >= 0 && wy < 8) ? byte : 0u;
      const uint32_t nib = (ox >= 0 ? byte >> ox : byte << -ox) & 0xFu;
      small_mask |= nib << (4 * j);
    }
  }
  uint32_t big_mask;
  {
    int i0 = cw_lo(w.x) - tcx, i1 = cw_hi(w.x) - tcx, j0 = cw_lo(w.y) - tcy, j1 = cw_hi(w.y) - tcy;
    i0 = i0 < 0 ? 0 : i0; i1 = i1 > 3 ? 3 : i1; j0 = j0 < 0 ? 0 : j0; j1 = j1 > 3 ? 3 : j1;
    const bool any = i0 <= i1 && j0 <= j1;
    i1 = i1 < 0 ? 0 : i1; j1 = j1 < 0 ? 0 : j1; i0 = i0 > 3 ? 3 : i0; j0 = j0 > 3 ? 3 : j0;
    const uint32_t cols = ((2u << i1) - 1u) & ~((1u << i0) - 1u);                       // 4 bits
    const uint32_t rows = 0x1111u & ((2u << (4 * j1 + 3)) - 1u) & ~((1u << (4 * j0)) - 1u);
    big_mask = any ? cols * rows : 0u;
  }
  return w.w != 0u ? big_mask : small_mask;
}

// The 4-bit mask (bit 2 j + i) of the 2x2 cells whose first is cell (qcx, qcy) -- one 8x8 quadrant of a tile --
// out of a pair's cell window: tile_cell_mask for the four cells a quadrant wave owns, at a third of the price
// (`big_too` = false: the caller knows that no lane of its wave holds a large-footprint window -- a wave-uniform
// test -- and the range arithmetic of that form is not compiled in)
template <bool big_too = true>
__device__ __forceinline__ uint32_t quad_cell_mask(uint4 w, int qcx, int qcy) {
  uint32_t small_mask;
  {
    // one window row = 8 bits; ((byte << 1) >> (ox + 1)) & 3 takes cells ox, ox + 1 for ox in -1 .. 7
    int ox = qcx - cw_lo(w.z), oy = qcy - cw_hi(w.z);
    const bool x_ok = ox >= -1 && ox <= 7;
    ox = ox < -1 ? -1 : (ox > 7 ? 7 : ox);
    uint32_t two[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int wy = oy + j;
      const uint32_t word = wy < 4 ? w.x : w.y;
      const uint32_t byte = (word >> (8 * (wy & 3))) & 0xFFu;
      two[j] = (wy >= 0 && wy < 8) ? ((byte << 1) >> (ox + 1)) & 3u : 0u;
    }
    small_mask = x_ok ? (two[0] | (two[1] << 2)) : 0u;
  }
  if (!big_too) return small_mask;
  uint32_t big_mask;
  {
    const int i0 = cw_lo(w.x) - qcx, i1 = cw_hi(w.x) - qcx, j0 = cw_lo(w.y) - qcy, j1 = cw_hi(w.y) - qcy;
    const uint32_t cols = ((i0 <= 0 && i1 >= 0) ? 1u : 0u) | ((i0 <= 1 && i1 >= 1) ? 2u : 0u);
    const uint32_t rows = ((j0 <= 0 && j1 >= 0) ? 3u : 0u) | ((j0 <= 1 && j1 >= 1) ? 12u : 0u);
    big_mask = (cols | (cols << 2)) & rows;
  }
  return w.w != 0u ? big_mask : small_mask;
}

// The 4-bit mask of the 8x8 QUADRANTS of tile (tx, ty) (bit k: x half = k & 1, y half = k >> 1; a quadrant = 2x2
// cells) out of a pair's cell window -- the tile backward's cull (raster_tiles.hip: one wave walks the whole
// tile, a quadrant per evaluation).  Small window: the four window rows of the tile are one 64-bit shift, the
// two cell columns of a half are a byte mask replicated over the rows of a half.
__device__ __forceinline__ uint32_t tile_quad_mask(uint4 w, int tx, int ty) {
  const int tcx = 4 * tx, tcy = 4 * ty;
  uint32_t small_mask;
  {
    int ox = tcx - cw_lo(w.z), oy = tcy - cw_hi(w.z);
    const bool rows_ok = oy > -4 && oy < 8;
    ox = ox < -4 ? -4 : (ox > 8 ? 8 : ox);            // (outside -3 .. 7 both column masks are empty)
    oy = oy < -3 ? -3 : (oy > 7 ? 7 : oy);
    const uint64_t W = (uint64_t)w.x | ((uint64_t)w.y << 32);
    const uint32_t R = rows_ok ? (uint32_t)(oy >= 0 ? W >> (8 * oy) : W << (-8 * oy)) : 0u;   // byte j = window row oy + j
    const uint32_t cl = ((0x3u << (ox + 4)) >> 4) & 0xFFu, cr = ((0xCu << (ox + 4)) >> 4) & 0xFFu;
    const uint32_t L = R & (cl * 0x01010101u), Rr = R & (cr * 0x01010101u);
    small_mask = ((L & 0xFFFFu) ? 1u : 0u) | ((Rr & 0xFFFFu) ? 2u : 0u) | ((L >> 16) ? 4u : 0u) | ((Rr >> 16) ? 8u : 0u);
  }
  uint32_t big_mask;
  {
    const int i0 = cw_lo(w.x) - tcx, i1 = cw_hi(w.x) - tcx, j0 = cw_lo(w.y) - tcy, j1 = cw_hi(w.y) - tcy;
    const uint32_t cols = ((i0 <= 1 && i1 >= 0) ? 1u : 0u) | ((i0 <= 3 && i1 >= 2) ? 2u : 0u);
    const uint32_t rows = ((j0 <= 1 && j1 >= 0) ? 3u : 0u) | ((j0 <= 3 && j1 >= 2) ? 12u : 0u);
    big_mask = (cols | (cols << 2)) & rows;
  }
  return w.w != 0u ? big_mask : small_mask;
}

}  // namespace ps
