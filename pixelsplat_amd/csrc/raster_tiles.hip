// Tile kernels: one wave64 owns one 16x16 tile, 4 pixels per lane -- pixel k of lane l sits
// in 8x8 QUADRANT k at (l & 7, l >> 3).  Each staged entry carries a 4-bit mask of the
// quadrants its alpha >= 1/255 ellipse can reach, so the per-entry work is skipped per
// quadrant with wave-uniform (scalar) branches.  There is no duplicated (tile, Gaussian) key list and no global
// sort over it.  Per tile the wave runs a two-stage, LDS-resident pipeline over its bin:
//
//   list     the tile's bin (raster_bins.hip): Gaussian ids in (depth, id) order, exactly
//            the reference's per-tile range of its sorted point list; the
//            1-based position in it is the "contributor" index n_contrib refers to.
//   refine   64 list entries at a time, one per lane: gather the 48-byte record and test
//            the alpha >= 1/255 ellipse against the tile's pixel box (exact conservative
//            bound: the minimum of the quadratic form over the box).  Entries that cannot
//            reach 1/255 on any pixel of the tile are dropped -- every pixel would have
//            skipped them anyway (A.3), so results are unchanged -- survivors go to ring B
//            with exp2-scaled conic coefficients.
//   blend    64 ring-B entries at a time, broadcast from LDS, branch-free per-pixel update.
//
// Replaces renderCUDA fwd/bwd of the external rasterizer (call site
// /root/reference/src/model/decoder/cuda_splatting.py:117-124).
#include "raster_common.h"

#include <cstdlib>

#ifndef PS_ABLATE
#define PS_ABLATE 0   // 1..3: timing experiments that drop part of a tile kernel's work (results invalid)
#endif
// Experiment, not in the default build and not yet run on hardware (tools/build_variant.sh x
// -DPS_BWD_STAGED_SUMS=1; validate with the raster GPU tests through PIXELSPLAT_HIP_LIB): the
// backward's nine per-entry sums stop one DPP step early -- 8-lane partial sums, 24 floats per entry --
// and are staged in LDS; the lane that finalises the entry adds the halves.  Per contributing entry
// 9 DPP adds instead of 14, no flag traffic (the "entry contributed" bits stay in an SGPR), at the
// price of finalising every 32 instead of every 64 entries.  Ablation says the reduction and its
// hand-over are 26 % of the kernel.
#ifndef PS_BWD_STAGED_SUMS
#define PS_BWD_STAGED_SUMS 0
#endif

namespace ps {

constexpr int kBatch = 64;
constexpr int kQB = 128;            // ring B capacity (>= 63 + 64), power of two
constexpr int kWavesPerBlock = 4;      // forward: tiles (waves) per block
// backward: one wave per block -- a new block of four needs four free wave slots on one CU,
// i.e. it waits for four waves of that CU to retire; with single-wave blocks a slot is refilled
// the moment it frees (tiles_backward 1.868 -> 1.825 ms; the forward does not care: +0.5 %)
constexpr int kWavesPerBlockBwd = 1;
constexpr int kSlotVec = kSlotFloats / 4;   // float4 units per gradient slot
constexpr float kLog2e = 1.4426950408889634f;

struct WaveLds {
  float4 rec[kQB][3];               // {gx,gy,A,B} {C,opacity,r,g} {b,list index,quad mask,id}
};
#if PS_BWD_STAGED_SUMS
constexpr int kStage = 32;              // entries per finalisation batch
struct WaveLdsBwd : WaveLds {
  float stage[kStage][3][8];            // per entry: r1, r2, s_b as 8-lane partial sums
};
#else
struct WaveLdsBwd : WaveLds {
  float gsum[kBatch][kGradFloats + 1];  // per-entry reduced sums (+ "touched" flag)
};
#endif

__device__ __forceinline__ uint32_t wave_max_u(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const uint32_t u = __shfl_xor(v, o); v = u > v ? u : v; }
  return v;
}

// v_exp_f32: 2^x
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Conservative test: can alpha = opacity * exp(power) reach alpha_min on the box of pixel
// centres [x0,x0+15] x [y0,y0+15]?  With A,B,C the log2-scaled coefficients
// (power*log2e = A dx^2 + B dx dy + C dy^2, d = centre - pixel), Q = -(power*log2e) is a
// convex quadratic for a positive-definite conic; its minimum over the box is 0 if the
// centre is inside, else it lies on the (at most two) box edges facing the centre.
__device__ __forceinline__ bool may_contribute(float gx, float gy, float A, float B, float Cq,
                                               float opacity, float alpha_min, float x0,
                                               float y0) {
  const float tau = __log2f(opacity / alpha_min);   // need Q <= tau somewhere
  if (!(tau >= 0.f)) return false;                  // opacity < alpha_min (or NaN): never
  const float det = 4.f * A * Cq - B * B;
  if (!(A < 0.f && Cq < 0.f && det > 0.f)) return true;  // not positive definite: keep
  const float dxlo = gx - (x0 + 15.f), dxhi = gx - x0;   // range of dx over the box
  const float dylo = gy - (y0 + 15.f), dyhi = gy - y0;
  const float ex = dxlo > 0.f ? dxlo : (dxhi < 0.f ? dxhi : 0.f);  // nearest-edge offsets
  const float ey = dylo > 0.f ? dylo : (dyhi < 0.f ? dyhi : 0.f);
  if (ex == 0.f && ey == 0.f) return true;          // centre inside the box
  float qmin = 3.0e38f;
  if (ex != 0.f) {                                  // vertical edge dx = ex
    const float dy = fminf(dyhi, fmaxf(dylo, -B * ex / (2.f * Cq)));
    qmin = fminf(qmin, -(A * ex * ex + B * ex * dy + Cq * dy * dy));
  }
  if (ey != 0.f) {                                  // horizontal edge dy = ey
    const float dx = fminf(dxhi, fmaxf(dxlo, -B * ey / (2.f * A)));
    qmin = fminf(qmin, -(A * dx * dx + B * dx * ey + Cq * ey * ey));
  }
  // margin covers fp32 rounding of this bound and of the per-pixel power evaluation
  return !(qmin > tau + 1e-4f * fabsf(tau) + 1e-3f);
}

// same bound on an (8+1)x(8+1) box of pixel centres [x0,x0+7] x [y0,y0+7] given tau
__device__ __forceinline__ bool quad_may_contribute(float gx, float gy, float A, float B,
                                                    float Cq, float tau, float x0, float y0) {
  const float dxlo = gx - (x0 + 7.f), dxhi = gx - x0;
  const float dylo = gy - (y0 + 7.f), dyhi = gy - y0;
  const float ex = dxlo > 0.f ? dxlo : (dxhi < 0.f ? dxhi : 0.f);
  const float ey = dylo > 0.f ? dylo : (dyhi < 0.f ? dyhi : 0.f);
  if (ex == 0.f && ey == 0.f) return true;
  float qmin = 3.0e38f;
  if (ex != 0.f) {
    const float dy = fminf(dyhi, fmaxf(dylo, -B * ex / (2.f * Cq)));
    qmin = fminf(qmin, -(A * ex * ex + B * ex * dy + Cq * dy * dy));
  }
  if (ey != 0.f) {
    const float dx = fminf(dxhi, fmaxf(dxlo, -B * ey / (2.f * A)));
    qmin = fminf(qmin, -(A * dx * dx + B * dx * ey + Cq * ey * ey));
  }
  return !(qmin > tau + 1e-4f * fabsf(tau) + 1e-3f);
}

// 4-bit mask of the tile's 8x8 quadrants (bit k: x half = k & 1, y half = k >> 1) that the
// entry can touch with alpha >= alpha_min; 0 = drop the entry.
__device__ __forceinline__ uint32_t quadrant_mask(float gx, float gy, float A, float B, float Cq,
                                                  float opacity, float alpha_min, float x0,
                                                  float y0) {
  const float tau = __log2f(opacity / alpha_min);
  if (!(tau >= 0.f)) return 0u;
  const float det = 4.f * A * Cq - B * B;
  if (!(A < 0.f && Cq < 0.f && det > 0.f)) return 0xFu;  // not positive definite: keep all
  uint32_t m = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (quad_may_contribute(gx, gy, A, B, Cq, tau, x0 + 8.f * (k & 1), y0 + 8.f * (k >> 1)))
      m |= 1u << k;
  return m;
}

// ------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Inner loop (the result of the round-2 A/B series, profiles/r2_tiles_variants_ab.txt; the superseded
// variants are in the history before this commit):
//   * the transmittance carries the "finished" state in its sign (T > 0: live, T < 0: the pixel
//     stopped and -T is its final value), so there is no separate flag to test, mask and update;
//   * power and the (T (1 - a), T a) pair are written on 2-vectors -> v_pk_add / v_pk_mul;
//   * two entries per trip, each one's record read from LDS while the other is blended (the latency
//     is hidden without moving a prefetched record between registers);
//   * "is every pixel finished?" is asked once per 8 entries, not per entry;
//   * the refine's two dependent global latencies (list -> record gather) are off the wave's critical
//     path: the records of batch i + 1 and the list indices of batch i + 2 are in flight while batch i
//     is refined and blended.
__global__ void __launch_bounds__(kWavesPerBlock* kWave)
tiles_forward_kernel(PsRasterDesc d, const float* __restrict__ records,
                     const uint32_t* __restrict__ tile_order,
                     const uint32_t* __restrict__ tile_ranges,
                     const uint32_t* __restrict__ point_list, uint32_t capacity,
                     const float* __restrict__ view_params, float* __restrict__ out_color,
                     float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                     uint32_t* __restrict__ tile_end) {
  __shared__ WaveLds lds_all[kWavesPerBlock];
  const int G = d.n_gaussians, H = d.height, W = d.width;
  const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
  const int tiles = gx * gy;
  const int V = d.n_scenes * d.views_per_scene;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slot_global = blockIdx.x * kWavesPerBlock + w;
  if (slot_global >= V * tiles) return;
  const int tile_global = (int)tile_order[slot_global];   // longest lists are launched first
  WaveLds& lds = lds_all[w];
  const int v = tile_global / tiles, t = tile_global % tiles;
  const uint32_t tx = t % gx, ty = t / gx;
  const size_t vo = (size_t)v * G;
  const float* recs = records + vo * kRecFloats;
  uint32_t l_start = tile_ranges[2 * (size_t)tile_global];
  uint32_t l_count = tile_ranges[2 * (size_t)tile_global + 1];
  if (l_start > capacity) l_start = capacity;                       // overflowed step: stay in
  if (l_count > capacity - l_start) l_count = capacity - l_start;   // bounds (flag is raised)
  const uint32_t* list = point_list + l_start;

  const float x0 = (float)(tx * kTile), y0 = (float)(ty * kTile);
  int px[4], py[4]; float pxf[4], pyf[4]; bool live[4];
  float T[4], C0[4], C1[4], C2[4]; uint32_t last[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    px[k] = tx * kTile + 8 * (k & 1) + (lane & 7);
    py[k] = ty * kTile + 8 * (k >> 1) + (lane >> 3);
    pxf[k] = (float)px[k]; pyf[k] = (float)py[k];
    live[k] = px[k] < W && py[k] < H;
    T[k] = 1.f; C0[k] = C1[k] = C2[k] = 0.f; last[k] = 0;
  }
  uint32_t b_head = 0, b_tail = 0;  // wave-uniform ring cursors
  const uint64_t lt = lanemask_lt();
  const float alpha_max = d.alpha_max, alpha_min = d.alpha_min, t_min = d.t_min;
  bool all_done = !__any(live[0] | live[1] | live[2] | live[3]);

  // Ts[k] = T while the pixel is live, -T once it has stopped (pixels outside the image start
  // stopped)
  float Ts[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) Ts[k] = live[k] ? 1.f : -1.f;
  // one ring entry against the (up to four) quadrants it can reach
  auto process_entry = [&](const float4 q0, const float4 q1, const float4 q2) {
    const uint32_t hidx = __float_as_uint(q2.y);
#if PS_ABLATE == 3   // timing experiment (tools/build_variant.sh): refine + ring only, no blend math
    const uint32_t qm = __builtin_amdgcn_readfirstlane(__float_as_uint(q2.z)) & 0u;
#else
    const uint32_t qm = __builtin_amdgcn_readfirstlane(__float_as_uint(q2.z));
#endif
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (qm & (1u << k)) {   // wave-uniform: the entry cannot reach the other quadrants
        const f32x2 dd = f32x2{q0.x, q0.y} - f32x2{pxf[k], pyf[k]};      // (dx, dy)
        const f32x2 bc = f32x2{q0.w, q1.x} * f32x2{dd.y, dd.y};          // (B dy, C dy)
        const float pw = fmaf(dd.x, fmaf(q0.z, dd.x, bc.x), dd.y * bc.y);  // power * log2(e)
        const float alpha = fminf(alpha_max, q1.y * fast_exp2(pw));
        const bool ok = (pw <= 0.f) & (alpha >= alpha_min);
        const float ale = ok ? alpha : 0.f;          // 0 => every update below is a no-op
        float Tp;                                    // max(T, 0): 0 for a stopped pixel (one
        asm("v_max_f32 %0, 0, %1" : "=v"(Tp) : "v"(Ts[k]));   // instruction; fmaxf adds a canonicalize)
        const f32x2 tw = f32x2{Tp, Tp} * f32x2{1.f - ale, ale};          // (T (1 - a), T a)
        const bool stop = tw.x < t_min;              // a live pixel can only stop when ale > 0;
        const float wgt = stop ? 0.f : tw.y;         // a stopped one always "stops" again
        Ts[k] = stop ? -fabsf(Ts[k]) : tw.x;
        C0[k] = fmaf(q1.z, wgt, C0[k]);
        C1[k] = fmaf(q1.w, wgt, C1[k]);
        C2[k] = fmaf(q2.x, wgt, C2[k]);
        last[k] = (ok & !stop) ? hidx : last[k];
      }
    }
  };
  auto every_pixel_stopped = [&]() {
    return !__any((Ts[0] > 0.f) | (Ts[1] > 0.f) | (Ts[2] > 0.f) | (Ts[3] > 0.f));
  };
  auto blend1 = [&](uint32_t m) {
    uint32_t slot = b_head & (kQB - 1);
    float4 a0 = lds.rec[slot][0], a1 = lds.rec[slot][1], a2 = lds.rec[slot][2];
    for (uint32_t j = 0; j < m; j += 2) {
      slot = (b_head + j + 1) & (kQB - 1);         // (stale beyond m: never processed)
      const float4 b0 = lds.rec[slot][0], b1 = lds.rec[slot][1], b2 = lds.rec[slot][2];
      process_entry(a0, a1, a2);
      if (j + 1 >= m) break;
      slot = (b_head + j + 2) & (kQB - 1);
      a0 = lds.rec[slot][0]; a1 = lds.rec[slot][1]; a2 = lds.rec[slot][2];
      process_entry(b0, b1, b2);
      if ((j & 7u) == 6u && every_pixel_stopped()) { all_done = true; break; }
    }
    b_head += m;
    wave_lds_sync();
  };

  {
    auto gather = [&](uint32_t id, float4& r0, float4& r1, float4& r2) {
      const float4* r = reinterpret_cast<const float4*>(recs + (size_t)id * kRecFloats);
      r0 = r[0]; r1 = r[1]; r2 = r[2];
    };
    auto idx_of = [&](uint32_t first) {
      return first + (uint32_t)lane < l_count ? list[first + lane] : list[0];
    };
    float4 n0, n1, n2;
    uint32_t id2 = 0;
    if (l_count > 0) {
      gather(idx_of(0), n0, n1, n2);
      id2 = idx_of(kBatch);
    }
    for (uint32_t first = 0; first < l_count && !all_done; first += kBatch) {
      const uint32_t m = l_count - first < (uint32_t)kBatch ? l_count - first : (uint32_t)kBatch;
      const float4 r0 = n0, r1 = n1, r2 = n2;
      if (first + kBatch < l_count) {
        gather(id2, n0, n1, n2);
        id2 = idx_of(first + 2 * kBatch);
      }
      bool keep = false;
      float4 q0, q1, q2;
      if ((uint32_t)lane < m) {
        const float A = -0.5f * kLog2e * r0.z, B = -kLog2e * r0.w, Cq = -0.5f * kLog2e * r1.x;
        const uint32_t qm = quadrant_mask(r0.x, r0.y, A, B, Cq, r1.y, alpha_min, x0, y0);
        keep = qm != 0u;
        q0 = make_float4(r0.x, r0.y, A, B);
        q1 = make_float4(Cq, r1.y, r2.x, r2.y);
        q2 = make_float4(r2.z, __uint_as_float(first + lane + 1u), __uint_as_float(qm), 0.f);
      }
      const uint64_t mask = __ballot(keep);
      if (keep) {
        const uint32_t slot = (b_tail + (uint32_t)__popcll(mask & lt)) & (kQB - 1);
        lds.rec[slot][0] = q0; lds.rec[slot][1] = q1; lds.rec[slot][2] = q2;
      }
      b_tail += (uint32_t)__popcll(mask);
      wave_lds_sync();
      while (!all_done && b_tail - b_head >= (uint32_t)kBatch) blend1(kBatch);
    }
  }
  if (!all_done && b_tail != b_head) blend1(b_tail - b_head);
#pragma unroll
  for (int k = 0; k < 4; ++k) T[k] = fabsf(Ts[k]);

  // epilogue
  const float* bg = view_params + (size_t)v * PS_VIEW_STRIDE + PS_VIEW_BG;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const size_t P = (size_t)H * W;
  uint32_t max_c = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (px[k] < W && py[k] < H) {
      const size_t pix = (size_t)py[k] * W + px[k];
      float* oc = out_color + (size_t)v * 3 * P;
      oc[pix] = C0[k] + T[k] * bg0;
      oc[P + pix] = C1[k] + T[k] * bg1;
      oc[2 * P + pix] = C2[k] + T[k] * bg2;
      final_T[(size_t)v * P + pix] = T[k];
      n_contrib[(size_t)v * P + pix] = last[k];
      max_c = last[k] > max_c ? last[k] : max_c;
    }
  }
  max_c = wave_max_u(max_c);
  if (lane == 0) tile_end[tile_global] = max_c;
}

void launch_tiles_forward(const PsRasterDesc& d, const float* records,
                          const uint32_t* tile_order, const uint32_t* tile_ranges,
                          const uint32_t* point_list,
                          uint32_t capacity, const float* view_params, float* out_color,
                          float* final_T, uint32_t* n_contrib, uint32_t* tile_end,
                          hipStream_t st) {
  const Dims m = make_dims(d);
  const int total = m.V * m.tiles;
  dim3 grid((total + kWavesPerBlock - 1) / kWavesPerBlock), block(kWavesPerBlock * kWave);
  hipLaunchKernelGGL(tiles_forward_kernel, grid, block, 0, st, d, records, tile_order, tile_ranges,
                     point_list, capacity, view_params, out_color, final_T, n_contrib, tile_end);
}

// ------------------------------------------------------------------------------------
// backward: walk the tile's bin back to front from the last contributor
// ------------------------------------------------------------------------------------
// Nine wave64 sums with the gfx950 lane-swap instructions.  v_permlane32_swap exchanges the
// upper half of one register with the lower half of another, so ONE swap + ONE add folds two
// values from 64 to 32 lanes each (a butterfly step that halves the number of live
// registers); v_permlane16_swap does the same between odd and even rows of 16.  Eight values
// end up as the four 16-lane rows of two registers, reduced by four DPP row_shr adds each:
//   r1 rows = [a, c, b, d], r2 rows = [e, g, f, h]  (row total in lane 15 of the row),
// and the ninth value takes the plain DPP chain (total in lane 63).  26 instructions instead
// of 54, and the totals are stored from four lanes at once.
__device__ __forceinline__ float fold32(float x, float y) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);       // [x_lo + x_hi | y_lo + y_hi]
}
__device__ __forceinline__ float fold16(float x, float y) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);       // rows [x0+x1, y0+y1, x2+x3, y2+y3]
}
__device__ __forceinline__ void wave_sum9_rows(float a, float b, float c, float d, float e,
                                               float f, float g, float h, float& i, float& r1,
                                               float& r2) {
  r1 = fold16(fold32(a, b), fold32(c, d));
  r2 = fold16(fold32(e, f), fold32(g, h));
  asm volatile(
      "s_nop 1\n"
      "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_f32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_f32_dpp %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_f32_dpp %2, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "s_nop 1\n"
      "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
      "s_nop 1\n"
      "v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
      "s_nop 1\n"
      : "+v"(r1), "+v"(r2), "+v"(i));
}

#if PS_BWD_STAGED_SUMS
// the nine sums down to 8-lane partials: r1 rows = [Mx, Mxx, My, Mxy], r2 rows = [Myy, s_r, s_op, s_g]
// (as in wave_sum9_rows), i = s_b; afterwards lanes 7 and 15 of every 16-lane row hold the sums of
// lanes 0-7 and 8-15 of that row
__device__ __forceinline__ void wave_sum9_partials(float a, float b, float c, float d, float e,
                                                   float f, float g, float h, float& i, float& r1,
                                                   float& r2) {
  r1 = fold16(fold32(a, b), fold32(c, d));
  r2 = fold16(fold32(e, f), fold32(g, h));
  asm volatile(
      "s_nop 1\n"
      "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_f32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_add_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "s_nop 1\n"
      : "+v"(r1), "+v"(r2), "+v"(i));
}
#endif

// The loop is the last of the round-2 A/B series (profiles/r2_tiles_variants_ab.txt: 2.14 -> 1.82 ms at
// BASELINE configs[1]; the superseded variants are in the history before this commit): the record is read
// where it is used, (dx, dy) / (B dy, C dy) and the per-pixel colour state and sums are 2-vectors (the
// packed instructions take their operands in place: no v_mov to build pairs; 126 VGPRs, no spill),
// min(alpha_max, .) is one instruction, two entries per trip with each record read from LDS while the
// other entry is processed.  Held to 4 waves per SIMD (128 VGPRs): 3 waves with 168 registers measured
// slower (2.17 ms).
__global__ void __launch_bounds__(kWavesPerBlockBwd* kWave, 4)
tiles_backward_kernel(PsRasterDesc d, const float* __restrict__ records,
                      const uint32_t* __restrict__ tile_order,
                      const uint32_t* __restrict__ tile_ranges,
                      const uint32_t* __restrict__ point_list, uint32_t capacity,
                      const float* __restrict__ view_params, const float* __restrict__ final_T,
                      const uint32_t* __restrict__ n_contrib,
                      const uint32_t* __restrict__ tile_end, const float* __restrict__ dL_dcolor,
                      float* __restrict__ grad2d, float* __restrict__ tile_grads) {
  __shared__ WaveLdsBwd lds_all[kWavesPerBlockBwd];
  const int G = d.n_gaussians, H = d.height, W = d.width;
  const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
  const int tiles = gx * gy;
  const int V = d.n_scenes * d.views_per_scene;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slot_global = blockIdx.x * kWavesPerBlockBwd + w;
  if (slot_global >= V * tiles) return;
  const int tile_global = (int)tile_order[slot_global];   // longest lists are launched first
  WaveLdsBwd& lds = lds_all[w];
  const int v = tile_global / tiles, t = tile_global % tiles;
  const uint32_t tx = t % gx, ty = t / gx;
  const size_t vo = (size_t)v * G;
  const float* recs = records + vo * kRecFloats;
  float* gacc = grad2d + vo * kGradFloats;
  uint32_t l_start = tile_ranges[2 * (size_t)tile_global];
  if (l_start > capacity) l_start = capacity;
  const uint32_t* list = point_list + l_start;

  // entries behind the tile's last contributor are never blended (1-based index c_max); the
  // geometry backward still sums the private slot of every (Gaussian, tile) pair of a small
  // Gaussian, so the slots of those entries are cleared here (nothing is memset)
  const uint32_t c_max = tile_end[tile_global];
  float4* const slots = reinterpret_cast<float4*>(tile_grads) + vo * (kInvSlots * kSlotVec);
  {
    uint32_t l_count = tile_ranges[2 * (size_t)tile_global + 1];
    if (l_count > capacity - l_start) l_count = capacity - l_start;
    for (uint32_t e = c_max + (uint32_t)lane; e < l_count; e += kWave) {
      const uint32_t id = list[e];
      const uint32_t pk = __float_as_uint(recs[(size_t)id * kRecFloats + 7]);
      if (pk & kSmallFlag) {
        const uint32_t kk = (ty - ((pk >> 15) & 0x3FFFu)) * (((pk >> 29) & 3u) + 1u) + (tx - (pk & 0x7FFFu));
        float4* tg = slots + ((size_t)id * kInvSlots + kk) * kSlotVec;
        tg[0] = tg[1] = tg[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kSlotVec == 4) tg[3] = make_float4(0.f, 0.f, 0.f, 0.f);   // 64-byte slots: whole line
      }
    }
  }
  if (c_max == 0) return;

  const float* bg = view_params + (size_t)v * PS_VIEW_STRIDE + PS_VIEW_BG;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const size_t P = (size_t)H * W;
  const float x0 = (float)(tx * kTile), y0 = (float)(ty * kTile);
  float pxf[4], pyf[4], T[4], Tfb[4], g0[4], g1[4], g2[4], acc0[4], acc1[4], acc2[4];
  uint32_t nc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int px = tx * kTile + 8 * (k & 1) + (lane & 7);
    const int py = ty * kTile + 8 * (k >> 1) + (lane >> 3);
    pxf[k] = (float)px; pyf[k] = (float)py;
    const bool inside = px < W && py < H;
    const size_t pix = inside ? (size_t)py * W + px : 0;
    nc[k] = inside ? n_contrib[(size_t)v * P + pix] : 0u;
    T[k] = inside ? final_T[(size_t)v * P + pix] : 0.f;
    const float* gp = dL_dcolor + (size_t)v * 3 * P;
    g0[k] = inside ? gp[pix] : 0.f;
    g1[k] = inside ? gp[P + pix] : 0.f;
    g2[k] = inside ? gp[2 * P + pix] : 0.f;
    Tfb[k] = -T[k] * (bg0 * g0[k] + bg1 * g1[k] + bg2 * g2[k]);  // -T_final * (bg . dL/dC)
    acc0[k] = acc1[k] = acc2[k] = 0.f;   // colour composited BEHIND the current entry
  }
  const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
  const float alpha_max = d.alpha_max, alpha_min = d.alpha_min;
  const uint64_t lt = lanemask_lt();
  uint32_t b_head = 0, b_tail = 0;
  // the per-pixel colour state as 2-vectors (channels 0,1 | channel 2) so that the packed
  // instructions take their operands in place (the auto-vectoriser packs the scalar form too, but
  // assembles the register pairs with ~5 v_mov per pixel and entry)
  f32x2 acc01[4], g01[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { acc01[k] = f32x2{0.f, 0.f}; g01[k] = f32x2{g0[k], g1[k]}; }

  // refine list entries with 1-based indices top, top-1, ..., top-m+1 (lane i takes top - i)
  // list index of this lane's entry in the batch whose first (highest) 1-based index is `top`;
  // loaded one batch ahead, which takes the first of the refine's two dependent global latencies
  // (list -> record gather) off the critical path for one register
  auto idx_of = [&](uint32_t top) {
    return (uint32_t)lane < top ? list[top - 1u - lane] : list[0];
  };
  uint32_t id_ahead = c_max > 0 ? idx_of(c_max) : 0u;
  auto refine = [&](uint32_t top, uint32_t m) {
    bool keep = false;
    float4 q0, q1, q2;
    const uint32_t id_now = id_ahead;
    if (top > m) id_ahead = idx_of(top - m);
    if ((uint32_t)lane < m) {
      const uint32_t id = id_now;
      const float4* r = reinterpret_cast<const float4*>(recs + (size_t)id * kRecFloats);
      const float4 r0 = r[0], r1 = r[1], r2 = r[2];   // {px,py,cx,cy} {cz,o,depth,radius} {r,g,b,-}
      const float A = -0.5f * kLog2e * r0.z, B = -kLog2e * r0.w, Cq = -0.5f * kLog2e * r1.x;
      const uint32_t qm = quadrant_mask(r0.x, r0.y, A, B, Cq, r1.y, alpha_min, x0, y0);
      keep = qm != 0u;
      q0 = make_float4(r0.x, r0.y, A, B);
      q1 = make_float4(Cq, r1.y, r2.x, r2.y);
      // bit 4: the Gaussian touches <= 4 tiles -> its partial gradient goes to its private
      // slot of this tile (index id * 4 + position of the tile inside its rect), else id
      const uint32_t pk = __float_as_uint(r1.w);
      const uint32_t small = (pk & kSmallFlag) ? 16u : 0u;
      uint32_t target = id;
      if (small) {
        const uint32_t kk = (ty - ((pk >> 15) & 0x3FFFu)) * (((pk >> 29) & 3u) + 1u) + (tx - (pk & 0x7FFFu));
        target = id * (uint32_t)kInvSlots + kk;
        if (!keep) {   // cannot reach any quadrant: never blended, its slot is still summed
          float4* tg = slots + (size_t)target * kSlotVec;
          tg[0] = tg[1] = tg[2] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (kSlotVec == 4) tg[3] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      q2 = make_float4(r2.z, __uint_as_float(top - lane), __uint_as_float(qm | small),
                       __uint_as_float(target));
    }
    const uint64_t mask = __ballot(keep);
    if (keep) {
      const uint32_t slot = (b_tail + (uint32_t)__popcll(mask & lt)) & (kQB - 1);
      lds.rec[slot][0] = q0; lds.rec[slot][1] = q1; lds.rec[slot][2] = q2;
    }
    b_tail += (uint32_t)__popcll(mask);
    wave_lds_sync();
  };

#if PS_BWD_STAGED_SUMS
  uint32_t hitbits = 0;   // bit j: entry j of the finalisation batch contributed
#endif
  // one ring entry: the pixels' updates, the nine wave sums, their hand-over through LDS
  auto entry = [&](uint32_t j, const float4 q0, const float4 q1, const float4 q2) {
      const float o = q1.y, c0 = q1.z, c1 = q1.w, c2 = q2.x;
      const uint32_t hidx = __float_as_uint(q2.y);
#if PS_ABLATE == 2   // timing experiment: no per-pixel math (and so no reduction)
      const uint32_t qm = __builtin_amdgcn_readfirstlane(__float_as_uint(q2.z)) & 0u;
#else
      const uint32_t qm = __builtin_amdgcn_readfirstlane(__float_as_uint(q2.z));
#endif
      float Mx = 0.f, My = 0.f, Mxx = 0.f, Mxy = 0.f, Myy = 0.f;
      float s_op = 0.f, s_r = 0.f, s_g = 0.f, s_b = 0.f;
      bool any = false;
      {
        f32x2 M1 = {0.f, 0.f}, M2 = {0.f, 0.f}, s_rg = {0.f, 0.f};   // (Mx, My) (Mxx, Mxy) (s_r, s_g)
        const f32x2 c01 = f32x2{c0, c1};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (qm & (1u << k)) {   // wave-uniform quadrant skip
            const f32x2 dd = f32x2{q0.x, q0.y} - f32x2{pxf[k], pyf[k]};
            const f32x2 bc = f32x2{q0.w, q1.x} * f32x2{dd.y, dd.y};        // (B dy, C dy)
            const float pw = fmaf(dd.x, fmaf(q0.z, dd.x, bc.x), dd.y * bc.y);
            const float Gv = fast_exp2(pw);
            float alpha;                                   // min(alpha_max, o G) in one instruction
            asm("v_min_f32 %0, %1, %2" : "=v"(alpha) : "s"(alpha_max), "v"(o * Gv));
            const bool ok = (hidx <= nc[k]) & (pw <= 0.f) & (alpha >= alpha_min);
            const float ale = ok ? alpha : 0.f;            // 0 => all updates are no-ops
            const float rcp = __builtin_amdgcn_rcpf(1.f - ale);   // 1 ulp; exact 1 when ale == 0
            const float Tn = T[k] * rcp;                    // T in front of this entry
            const f32x2 d01 = c01 - acc01[k];
            const float d2 = c2 - acc2[k];
            const f32x2 t01 = d01 * g01[k];
            float dL_dalpha = fmaf(d2, g2[k], t01.x + t01.y) * Tn;
            dL_dalpha = fmaf(Tfb[k], rcp, dL_dalpha);       // -T_final/(1-alpha) * bg.dL/dC
            const float dch = ale * Tn;
            s_rg = f32x2{dch, dch} * g01[k] + s_rg;
            s_b = fmaf(dch, g2[k], s_b);
            const float gda = ok ? Gv * dL_dalpha : 0.f;    // G * dL/dalpha
            s_op += gda;
            const float q = o * gda;                        // G * dL/dG
            const f32x2 qxy = f32x2{q, q} * dd;             // (q dx, q dy)
            M1 += qxy;
            M2 = f32x2{qxy.x, qxy.x} * dd + M2;             // (Mxx, Mxy) += q dx (dx, dy)
            Myy = fmaf(qxy.y, dd.y, Myy);
            T[k] = Tn;
            acc01[k] = f32x2{ale, ale} * d01 + acc01[k];    // alpha c + (1 - alpha) acc
            acc2[k] = fmaf(ale, d2, acc2[k]);
            any |= ok;
          }
        }
        Mx = M1.x; My = M1.y; Mxx = M2.x; Mxy = M2.y; s_r = s_rg.x; s_g = s_rg.y;
      }
#if PS_ABLATE == 1   // timing experiment: per-pixel math kept, the nine wave sums and their hand-over dropped
      asm volatile("" :: "v"(Mx), "v"(My), "v"(Mxx), "v"(Mxy), "v"(Myy), "v"(s_op), "v"(s_r), "v"(s_g), "v"(s_b));
      if (false) {
#else
      if (__any(any)) {
#endif
#if PS_BWD_STAGED_SUMS
        float r1, r2;
        wave_sum9_partials(Mx, My, Mxx, Mxy, Myy, s_op, s_r, s_g, s_b, r1, r2);
        if ((lane & 7) == 7) {
          float* st = &lds.stage[j][0][lane >> 3];
          st[0] = r1; st[8] = r2; st[16] = s_b;
        }
        hitbits |= 1u << j;                 // wave-uniform: stays in an SGPR
      }
#else
        // moments about the Gaussian centre: Mx = sum q dx, My = sum q dy, ...
        float r1, r2;
        wave_sum9_rows(Mx, My, Mxx, Mxy, Myy, s_op, s_r, s_g, s_b, r1, r2);
        if ((lane & 15) == 15) {
          // row r of r1 holds value {0, 2, 1, 3}[r], row r of r2 value 4 + {0, 2, 1, 3}[r]
          const int row = lane >> 4, idx = ((row & 1) << 1) | (row >> 1);
          float* gs = lds.gsum[j];
          gs[idx] = r1; gs[4 + idx] = r2;
          if (lane == 63) { gs[8] = s_b; gs[9] = 1.f; }
        }
      } else if (lane == 63) {
        lds.gsum[j][9] = 0.f;
      }
#endif
  };

#if PS_BWD_STAGED_SUMS
  auto blend = [&](uint32_t m) {
    for (uint32_t base = 0; base < m; base += kStage) {
      const uint32_t cnt = m - base < (uint32_t)kStage ? m - base : (uint32_t)kStage;
      hitbits = 0;
      {
        uint32_t slot = (b_head + base) & (kQB - 1);
        float4 a0 = lds.rec[slot][0], a1 = lds.rec[slot][1], a2 = lds.rec[slot][2];
        for (uint32_t j = 0; j < cnt; j += 2) {
          slot = (b_head + base + j + 1) & (kQB - 1);          // (stale beyond cnt: never processed)
          const float4 b0 = lds.rec[slot][0], b1 = lds.rec[slot][1], b2 = lds.rec[slot][2];
          entry(j, a0, a1, a2);
          if (j + 1 >= cnt) break;
          slot = (b_head + base + j + 2) & (kQB - 1);
          a0 = lds.rec[slot][0]; a1 = lds.rec[slot][1]; a2 = lds.rec[slot][2];
          entry(j + 1, b0, b1, b2);
        }
      }
      wave_lds_sync();
      if ((uint32_t)lane < cnt) {
        const uint32_t slot = (b_head + base + lane) & (kQB - 1);
        const float4 q0 = lds.rec[slot][0], q1 = lds.rec[slot][1], q2 = lds.rec[slot][2];
        const float4* sp = reinterpret_cast<const float4*>(&lds.stage[lane][0][0]);
        const float4 u0 = sp[0], u1 = sp[1], v0 = sp[2], v1 = sp[3], w0 = sp[4], w1 = sp[5];
        const bool hit = (hitbits >> lane) & 1u;
        // r1 rows = [Mx, Mxx, My, Mxy], r2 rows = [Myy, s_r, s_op, s_g]; a row = two 8-lane halves
        const float Mx = u0.x + u0.y, Mxx = u0.z + u0.w, My = u1.x + u1.y, Mxy = u1.z + u1.w;
        const float Myy = v0.x + v0.y, s_r = v0.z + v0.w, s_op = v1.x + v1.y, s_g = v1.z + v1.w;
        const float s_b = ((w0.x + w0.y) + (w0.z + w0.w)) + ((w1.x + w1.y) + (w1.z + w1.w));
        const float cx = q0.z * (-2.f / kLog2e), cy = q0.w * (-1.f / kLog2e),
                    cz = q1.x * (-2.f / kLog2e);
        const float o0 = (-cx * Mx - cy * My) * ddelx_dx, o1 = (-cz * My - cy * Mx) * ddely_dy;
        const float o2 = -0.5f * Mxx, o3 = -0.5f * Mxy, o4 = -0.5f * Myy;
        if (__float_as_uint(q2.z) & 16u) {
          float4* tg = slots + (size_t)__float_as_uint(q2.w) * kSlotVec;
          tg[0] = hit ? make_float4(o0, o1, o2, o3) : make_float4(0.f, 0.f, 0.f, 0.f);
          tg[1] = hit ? make_float4(o4, s_op, s_r, s_g) : make_float4(0.f, 0.f, 0.f, 0.f);
          tg[2] = make_float4(hit ? s_b : 0.f, 0.f, 0.f, 0.f);
          if (kSlotVec == 4) tg[3] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else if (hit) {
          float* ga = gacc + (size_t)__float_as_uint(q2.w) * kGradFloats;
          atomicAdd(ga + 0, o0); atomicAdd(ga + 1, o1); atomicAdd(ga + 2, o2);
          atomicAdd(ga + 3, o3); atomicAdd(ga + 4, o4); atomicAdd(ga + 5, s_op);
          atomicAdd(ga + 6, s_r); atomicAdd(ga + 7, s_g); atomicAdd(ga + 8, s_b);
        }
      }
      wave_lds_sync();     // the staging rows are rewritten by the next batch
    }
    b_head += m;
  };
#else
  auto blend = [&](uint32_t m) {
    {
      // two entries per trip, each one's record read from LDS while the other is processed
      uint32_t slot = b_head & (kQB - 1);
      float4 a0 = lds.rec[slot][0], a1 = lds.rec[slot][1], a2 = lds.rec[slot][2];
      for (uint32_t j = 0; j < m; j += 2) {
        slot = (b_head + j + 1) & (kQB - 1);                   // (stale beyond m: never processed)
        const float4 b0 = lds.rec[slot][0], b1 = lds.rec[slot][1], b2 = lds.rec[slot][2];
        entry(j, a0, a1, a2);
        if (j + 1 >= m) break;
        slot = (b_head + j + 2) & (kQB - 1);
        a0 = lds.rec[slot][0]; a1 = lds.rec[slot][1]; a2 = lds.rec[slot][2];
        entry(j + 1, b0, b1, b2);
      }
    }
    wave_lds_sync();
    // lane j finalises entry j: its private slot (always written: values or zeros), or one set
    // of 9 atomics per (tile, Gaussian) for the large ones
    if ((uint32_t)lane < m) {
      const uint32_t slot = (b_head + lane) & (kQB - 1);
      const float4 q0 = lds.rec[slot][0], q1 = lds.rec[slot][1], q2 = lds.rec[slot][2];
      const float* gs = lds.gsum[lane];
      const bool hit = gs[9] != 0.f;
      const float cx = q0.z * (-2.f / kLog2e), cy = q0.w * (-1.f / kLog2e),
                  cz = q1.x * (-2.f / kLog2e);
      const float Mx = gs[0], My = gs[1];
      const float o0 = (-cx * Mx - cy * My) * ddelx_dx, o1 = (-cz * My - cy * Mx) * ddely_dy;
      const float o2 = -0.5f * gs[2], o3 = -0.5f * gs[3], o4 = -0.5f * gs[4];
      if (__float_as_uint(q2.z) & 16u) {
        // private slot of this (Gaussian, tile): plain 16-byte stores, summed later in a fixed
        // order by the geometry backward (no atomics, deterministic)
        float4* tg = slots + (size_t)__float_as_uint(q2.w) * kSlotVec;
        tg[0] = hit ? make_float4(o0, o1, o2, o3) : make_float4(0.f, 0.f, 0.f, 0.f);
        tg[1] = hit ? make_float4(o4, gs[5], gs[6], gs[7]) : make_float4(0.f, 0.f, 0.f, 0.f);
        tg[2] = make_float4(hit ? gs[8] : 0.f, 0.f, 0.f, 0.f);
        if (kSlotVec == 4) tg[3] = make_float4(0.f, 0.f, 0.f, 0.f);
      } else if (hit) {
        float* ga = gacc + (size_t)__float_as_uint(q2.w) * kGradFloats;
        atomicAdd(ga + 0, o0); atomicAdd(ga + 1, o1); atomicAdd(ga + 2, o2);
        atomicAdd(ga + 3, o3); atomicAdd(ga + 4, o4); atomicAdd(ga + 5, gs[5]);
        atomicAdd(ga + 6, gs[6]); atomicAdd(ga + 7, gs[7]); atomicAdd(ga + 8, gs[8]);
      }
    }
    b_head += m;
    wave_lds_sync();
  };
#endif

  for (uint32_t top = c_max; top > 0;) {
    const uint32_t m = top < (uint32_t)kBatch ? top : (uint32_t)kBatch;
    refine(top, m);
    while (b_tail - b_head >= (uint32_t)kBatch) blend(kBatch);
    top -= m;
  }
  if (b_tail != b_head) blend(b_tail - b_head);
}

void launch_tiles_backward(const PsRasterDesc& d, const float* records,
                           const uint32_t* tile_order, const uint32_t* tile_ranges,
                           const uint32_t* point_list,
                           uint32_t capacity, const float* view_params, const float* final_T,
                           const uint32_t* n_contrib, const uint32_t* tile_end,
                           const float* dL_dcolor, float* grad2d, float* tile_grads,
                           hipStream_t st) {
  const Dims m = make_dims(d);
  const int total = m.V * m.tiles;
  dim3 grid((total + kWavesPerBlockBwd - 1) / kWavesPerBlockBwd), block(kWavesPerBlockBwd * kWave);
  hipLaunchKernelGGL(tiles_backward_kernel, grid, block, 0, st, d, records, tile_order, tile_ranges,
                     point_list, capacity, view_params, final_T, n_contrib, tile_end, dL_dcolor,
                     grad2d, tile_grads);
}

}  // namespace ps
