// Tile kernels on 8x8 QUADRANTS.  A 16x16 tile is four quadrants; one pixel per lane and quadrant -- pixel k of
// lane l sits in quadrant k at (l & 7, l >> 3).  Two kernels live here:
//
//   tiles_backward_kernel   the shipped backward: one wave walks a tile's bin back to front (two tasks per tile),
//                           all four quadrants per wave.  Its refine reads an entry's 4-bit quadrant mask off the
//                           pair's cell window (cell_window.h: written once per pair by the preprocess), its per-pixel
//                           state is T, dL/dC and the scalar (colour behind, background included) . dL/dC, its nine
//                           per-entry sums are finished through LDS by eight lanes per entry (round 6).
//   tiles_forward_kernel    the round-2..5 forward (two waves per tile, kFwdQW quadrants each), kept as the A/B
//                           reference of the shipped 4x4-cell forward (raster_cells.hip; PS_FORWARD_QUADRANTS=1 selects
//                           this one, tools/ab_cells.sh): its refine still minimises the quadratic form over each
//                           quadrant's box per list entry (exact conservative bound).
//
// Common to both: every staged entry carries the mask of the quadrants its alpha >= 1/255 ellipse can reach, so the
// per-entry work is skipped per quadrant with wave-uniform (scalar) branches; an entry that reaches no quadrant is
// dropped -- every pixel would have skipped it anyway, so results are unchanged.  There is no
// duplicated (tile, Gaussian) key list and no global sort over it.  Per tile a wave runs a two-stage, LDS-resident
// pipeline over the tile's bin:
//
//   list     the tile's bin (raster_bins.hip): Gaussian ids in (depth, id) order, exactly the reference's per-tile
//            range of its sorted point list; the 1-based position in it is the "contributor" index
//            n_contrib refers to.
//   refine   64 list entries at a time, one per lane: gather the pair's 64-byte line (record + cell window), decide
//            the quadrant mask, compact the survivors into a ring in LDS with exp2-scaled conic coefficients.
//   blend    ring entries broadcast from LDS, branch-free per-pixel update.
//
// What was tried on these two kernels and what an instruction costs here: DESIGN.md 4 / 4a / 15,
// profiles/r2_tiles_variants_ab.txt, profiles/r3_{issue_model,forward_forms_ab,forward_split_ab}.txt,
// profiles/r6_ab_backward_lds_reduce.txt.
//
// Replaces renderCUDA fwd/bwd of the external rasterizer (call site
// /root/reference/src/model/decoder/cuda_splatting.py:117-124).
#include "raster_common.h"
#include "cell_window.h"

#include <cstdlib>
#include <type_traits>

// Quadrants per forward wave.  2 = two waves per tile (top / bottom half, 2 pixels per lane): fewer
// quadrants per wave = more, smaller tasks (7168 one-wave tile tasks on 4096 wave slots leave a launch
// tail that costs ~14 % at BASELINE configs[1]) and fewer registers (more waves per SIMD), paid for with
// a refine pass per wave over the tile's whole list.  Results are identical by construction (a pixel's
// walk does not depend on which wave owns it).  Measured (profiles/r3_forward_split_ab.txt, configs[1]):
// one wave per tile 0.93 ms, two 0.87 ms, four 0.86-0.87 ms with four times the record gathers.

namespace ps {

constexpr int kFwdQW = 2;
constexpr int kFwdParts = 4 / kFwdQW;
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
// The backward's nine per-entry sums leave the VALU after the two lane-swap folds (round 6): eight of them are
// then the four 16-lane rows of two registers (16 partial sums per value), the ninth is still one value per lane,
// and all 64 lanes stage the three registers in LDS (768 bytes per entry).  The LDS pipe does the rest of the
// transposition: eight lanes finalise an entry, each reads the 16 partials of ONE value (+ 8 of the ninth's 64)
// as 16-byte words and adds them -- 64 busy lanes per batch of 8 entries, where rounds 3 - 5 spent 9 DPP adds per
// contributing entry to get down to 8-lane partials and finalised with one lane per entry.  The stage is paid
// for with the ring: 64 records, refilled when it has run dry and blended down to the last entry after every
// refine (9216 bytes per wave as before = 16 waves per CU).
constexpr int kBwdBatch = 64;           // list entries per refine
constexpr int kBwdQB = 64;              // ring capacity = one refine
constexpr int kStage = 8;               // entries per finalisation batch (8 lanes each)
struct WaveLdsBwd {
  float4 rec[kBwdQB][3];                // {gx,gy,A,B} {C,opacity,r,g} {b,list index,quad mask,id}
  float stage[kStage][3][kWave];        // per entry: r1, r2 (rows of 16 partials), the ninth sum (64 partials)
};

__device__ __forceinline__ uint32_t wave_max_u(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const uint32_t u = __shfl_xor(v, o); v = u > v ? u : v; }
  return v;
}

// v_exp_f32: 2^x
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// Conservative test of the round-2..5 forward: can alpha = opacity * exp(power) reach alpha_min on the box of pixel
// centres [x0,x0+7] x [y0,y0+7], given tau = log2(opacity / alpha_min)?  With A,B,C the log2-scaled coefficients
// (power*log2e = A dx^2 + B dx dy + C dy^2, d = centre - pixel), Q = -(power*log2e) is a convex quadratic for a
// positive-definite conic; its minimum over the box is 0 if the centre is inside, else it lies on the (at most two)
// box edges facing the centre.  The margin covers fp32 rounding of this bound and of the per-pixel power evaluation.
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

// Is an entry "plain"?  Its conic is positive definite with a condition number far from fp32 round-off
// (so `power > 0` cannot happen for any pixel: the true power is <= -lambda_min |d|^2 and its fp32
// evaluation is off by < 4e-7 lambda_max |d|^2) and its opacity is below the alpha_max clamp.  Such an
// entry needs neither the per-pixel sign test of the power nor the min() -- two of the half-rate
// compare / select class instructions the blend loops are made of (tools/issue_model.hip: v_cmp /
// v_cndmask / v_min / DPP issue at ~4.4 cycles per wave64 instruction, v_fma / v_mul / v_add at ~2.9)
// -- and skipping them changes no result.  Used by the backward's short form (a finalisation batch
// whose ring holds plain entries only); the forward's short form did not survive measurement
// (DESIGN.md 4a).
__device__ __forceinline__ bool entry_is_plain(float gx, float gy, float A, float B, float Cq,
                                               float opacity, float alpha_max) {
  const float det = 4.f * A * Cq - B * B, tr = A + Cq;
  const bool finite = fabsf(gx) < 1e30f && fabsf(gy) < 1e30f && fabsf(tr) < 1e30f && fabsf(B) < 1e30f;
  return finite && A < 0.f && Cq < 0.f && det > 1e-4f * tr * tr && opacity <= 0.98f * alpha_max &&
         opacity >= 0.f;
}

// the mask of the QW quadrants first .. first + QW - 1 of a tile an entry can reach (bit k: quadrant first + k);
// 0 = drop the entry
template <int QW>
__device__ __forceinline__ uint32_t quadrant_mask_part(float gx, float gy, float A, float B,
                                                       float Cq, float opacity, float alpha_min,
                                                       float x0, float y0, int first) {
  const float tau = __log2f(opacity / alpha_min);
  if (!(tau >= 0.f)) return 0u;
  const float det = 4.f * A * Cq - B * B;
  if (!(A < 0.f && Cq < 0.f && det > 0.f)) return (1u << QW) - 1u;
  uint32_t m = 0;
#pragma unroll
  for (int k = 0; k < QW; ++k) {
    const int q = first + k;
    if (quad_may_contribute(gx, gy, A, B, Cq, tau, x0 + 8.f * (q & 1), y0 + 8.f * (q >> 1)))
      m |= 1u << k;
  }
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
constexpr int kFwdMinWaves = 6;      // waves per SIMD the forward's register allocation aims at (80 VGPRs, no spill)
__global__ void __launch_bounds__(kWavesPerBlock* kWave, kFwdMinWaves)
tiles_forward_kernel(PsRasterDesc d, const float* __restrict__ records,
                     const uint32_t* __restrict__ tile_order,
                     const uint32_t* __restrict__ tile_ranges,
                     const uint32_t* __restrict__ point_list, uint32_t capacity,
                     const float* __restrict__ view_params, float* __restrict__ out_color,
                     float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                     float4* __restrict__ checkpoint, uint32_t* __restrict__ tile_end) {
  constexpr int QW = kFwdQW;          // quadrants (= pixels per lane) of this wave
  __shared__ WaveLds lds_all[kWavesPerBlock];
  const int G = d.n_gaussians, H = d.height, W = d.width;
  const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
  const int tiles = gx * gy;
  const int V = d.n_scenes * d.views_per_scene;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slot_global = blockIdx.x * kWavesPerBlock + w;
  if (slot_global >= V * tiles * kFwdParts) return;
  // longest lists are launched first; the parts of a tile are neighbours in the launch order
  const int tile_global = (int)tile_order[slot_global / kFwdParts];
  const int q_first = (slot_global % kFwdParts) * QW;     // first quadrant of this wave
  WaveLds& lds = lds_all[w];
  const int v = tile_global / tiles, t = tile_global % tiles;
  const uint32_t tx = t % gx, ty = t / gx;
  const size_t vo = (size_t)v * G;
  const float* recs = records + vo * kRecFloats;
  uint32_t l_start = tile_ranges[2 * (size_t)tile_global];
  uint32_t l_count = tile_ranges[2 * (size_t)tile_global + 1];
  if (l_start > capacity) l_start = capacity;                       // overflowed step: stay in
  if (l_count > capacity - l_start) l_count = capacity - l_start;   // bounds (flag is raised)
  // (loaded with vector loads: tell the compiler they are wave-uniform, or every loop bounded by
  // them is compiled as divergent control flow)
  l_start = __builtin_amdgcn_readfirstlane(l_start);
  l_count = __builtin_amdgcn_readfirstlane(l_count);
  const uint32_t* list = point_list + l_start;
  // the backward walks a long list as two tasks (raster_common.h: split_point): this wave leaves its
  // pixels' state at the split point -- T and the accumulated colour, rewritten in the epilogue as the
  // colour composited BEHIND the split point over that T, which is what the front task starts from
  const uint32_t ck_at = split_point(l_count);
  // [tile][quadrant][lane]; the address is rebuilt where it is used (two registers less across the walk)
  auto ck_ptr = [&]() { return checkpoint + ((size_t)tile_global * 4 + q_first) * kWave + lane; };
  bool ck_written = false;

  const float x0 = (float)(tx * kTile), y0 = (float)(ty * kTile);
  int px[QW], py[QW]; float pxf[QW], pyf[QW]; bool live[QW];
  float T[QW], C0[QW], C1[QW], C2[QW]; uint32_t last[QW];
  bool any_live = false;
#pragma unroll
  for (int k = 0; k < QW; ++k) {
    const int q = q_first + k;
    px[k] = tx * kTile + 8 * (q & 1) + (lane & 7);
    py[k] = ty * kTile + 8 * (q >> 1) + (lane >> 3);
    pxf[k] = (float)px[k]; pyf[k] = (float)py[k];
    live[k] = px[k] < W && py[k] < H;
    any_live |= live[k];
    T[k] = 1.f; C0[k] = C1[k] = C2[k] = 0.f; last[k] = 0;
  }
  uint32_t b_head = 0, b_tail = 0;  // wave-uniform ring cursors
  const uint64_t lt = lanemask_lt();
  const float alpha_max = d.alpha_max, alpha_min = d.alpha_min, t_min = d.t_min;
  bool all_done = !__any(any_live);

  // Ts[k] = T while the pixel is live, -T once it has stopped (pixels outside the image start
  // stopped)
  float Ts[QW];
#pragma unroll
  for (int k = 0; k < QW; ++k) Ts[k] = live[k] ? 1.f : -1.f;
  // one ring entry against the (up to QW) quadrants of this wave it can reach
  auto process_entry = [&](const float4 q0, const float4 q1, const float4 q2) {
    const uint32_t hidx = __float_as_uint(q2.y);
    const uint32_t qm = __builtin_amdgcn_readfirstlane(__float_as_uint(q2.z));
#pragma unroll
    for (int k = 0; k < QW; ++k) {
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
    bool any = false;
#pragma unroll
    for (int k = 0; k < QW; ++k) any |= Ts[k] > 0.f;
    return !__any(any);
  };
  // m ring entries, two per trip, each one's record read from LDS while the other is blended; "is every
  // pixel finished?" is asked once per 8 entries.  The ring cursors are wave-uniform by construction
  // (sums of ballot popcounts) and the compiler is TOLD so (readfirstlane): left to its own analysis it
  // keeps them in vector registers and compiles every loop bounded by them as divergent control flow
  // (s_and_saveexec chains): tiles_forward 0.86 -> 0.82 ms at BASELINE configs[1]
  // (profiles/r4_forward_forms_ab.txt).
  auto blend1 = [&](uint32_t m) {
    m = __builtin_amdgcn_readfirstlane(m);
    const uint32_t bh = __builtin_amdgcn_readfirstlane(b_head);
    uint32_t slot = bh & (kQB - 1);
    float4 a0 = lds.rec[slot][0], a1 = lds.rec[slot][1], a2 = lds.rec[slot][2];
    for (uint32_t j = 0; j < m; j += 2) {
      slot = (bh + j + 1) & (kQB - 1);         // (stale beyond m: never processed)
      const float4 b0 = lds.rec[slot][0], b1 = lds.rec[slot][1], b2 = lds.rec[slot][2];
      process_entry(a0, a1, a2);
      if (j + 1 >= m) break;
      slot = (bh + j + 2) & (kQB - 1);
      a0 = lds.rec[slot][0]; a1 = lds.rec[slot][1]; a2 = lds.rec[slot][2];
      process_entry(b0, b1, b2);
      if ((j & 7u) == 6u && every_pixel_stopped()) { all_done = true; break; }
    }
    b_head = __builtin_amdgcn_readfirstlane(bh + m);
    wave_lds_sync();
  };

  auto gather = [&](uint32_t id, float4& r0, float4& r1, float4& r2) {
    const float4* r = reinterpret_cast<const float4*>(recs + (size_t)id * kRecFloats);
    r0 = r[0]; r1 = r[1]; r2 = r[2];
  };
  auto idx_of = [&](uint32_t at) {
    return at + (uint32_t)lane < l_count ? list[at + lane] : list[0];
  };
  float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0, n2 = n0;
  uint32_t id2 = 0;
  if (l_count > 0) {
    gather(idx_of(0), n0, n1, n2);
    id2 = idx_of(kBatch);
  }
  // Loop order: blend what the ring allows, THEN refine the next batch (the ring is tested first).  The
  // same work as "refine, then blend" -- measured 5 % faster (0.82 vs 0.86 ms, profiles/r4_forward_forms_ab.txt).
  uint32_t first = 0;                  // next list entry to refine
  auto refine = [&]() {
    // refine the batch of list entries [first, first + kBatch) into the ring; the records of batch i + 1
    // and the list indices of batch i + 2 are in flight meanwhile
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
      const uint32_t qm = quadrant_mask_part<QW>(r0.x, r0.y, A, B, Cq, r1.y, alpha_min, x0, y0, q_first);
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
    b_tail = __builtin_amdgcn_readfirstlane(b_tail + (uint32_t)__popcll(mask));
    wave_lds_sync();
    first += kBatch;
  };
  for (;;) {
    const bool refined_all = first >= l_count;
    // full batches while the list lasts, then whatever is left -- also at the split point, where the
    // ring is drained so that the pixel state is exactly "after entry ck_at"
    const bool at_split = ck_at != 0u && first == ck_at && !ck_written;
    const bool drain = refined_all || at_split;
    for (;;) {
      const uint32_t have = __builtin_amdgcn_readfirstlane(b_tail - b_head);
      if (all_done || !(have >= (uint32_t)kBatch || (drain && have != 0u))) break;
      blend1(have < (uint32_t)kBatch ? have : (uint32_t)kBatch);
    }
    if (all_done) break;
    if (at_split) {
      float4* const ck = ck_ptr();
#pragma unroll
      for (int k = 0; k < QW; ++k) ck[k * kWave] = make_float4(Ts[k], C0[k], C1[k], C2[k]);
      ck_written = true;
    }
    if (refined_all) break;
    refine();
  }
#pragma unroll
  for (int k = 0; k < QW; ++k) T[k] = fabsf(Ts[k]);

  // epilogue
  const float* bg = view_params + (size_t)v * PS_VIEW_STRIDE + PS_VIEW_BG;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const size_t P = (size_t)H * W;
  uint32_t max_c = 0;
#pragma unroll
  for (int k = 0; k < QW; ++k) {
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
  if (ck_written) {     // (wave-uniform) checkpoint -> (T at the split, colour behind the split / that T)
    float4* const ck = ck_ptr();
#pragma unroll
    for (int k = 0; k < QW; ++k) {
      const float4 c = ck[k * kWave];
      const float inv = c.x > 0.f ? 1.f / c.x : 0.f;       // stopped before the split: never read
      ck[k * kWave] = make_float4(c.x, (C0[k] - c.y) * inv, (C1[k] - c.z) * inv, (C2[k] - c.w) * inv);
    }
  }
  // the tile's last contributor (both half-tile waves max into it; cleared by the binning's scan kernel):
  // the backward orders its tasks by it and starts its walks there
  max_c = wave_max_u(max_c);
  if (lane == 0 && max_c != 0u) atomicMax(&tile_end[tile_global], max_c);
}

void launch_tiles_forward(const PsRasterDesc& d, const float* records,
                          const uint32_t* tile_order, const uint32_t* tile_ranges,
                          const uint32_t* point_list,
                          uint32_t capacity, const float* view_params, float* out_color,
                          float* final_T, uint32_t* n_contrib, float4* checkpoint,
                          uint32_t* tile_end, hipStream_t st) {
  const Dims m = make_dims(d);
  const int total = m.V * m.tiles * kFwdParts;
  dim3 grid((total + kWavesPerBlock - 1) / kWavesPerBlock), block(kWavesPerBlock * kWave);
  hipLaunchKernelGGL(tiles_forward_kernel, grid, block, 0, st, d, records, tile_order, tile_ranges,
                     point_list, capacity, view_params, out_color, final_T, n_contrib, checkpoint, tile_end);
}

// ------------------------------------------------------------------------------------
// backward: walk the tile's bin back to front from the last contributor
// ------------------------------------------------------------------------------------
// Eight of the nine wave64 sums with the gfx950 lane-swap instructions.  v_permlane32_swap exchanges the
// upper half of one register with the lower half of another, so ONE swap + ONE add folds two values from
// 64 to 32 lanes each (a butterfly step that halves the number of live registers); v_permlane16_swap does
// the same between odd and even rows of 16.  The values come as the four register PAIRS the blend loop
// accumulates them in -- (a, e) (c, g) (b, f) (d, h) -- and the swaps are taken pair against pair, so that
// both adds of a level are ONE packed add on registers that are pairs already:
//   level 1:  swap32(a, c) swap32(e, g) -> (a, e) + (c, g) = U = ([a | c], [e | g])     (32-lane halves)
//             swap32(b, d) swap32(f, h) -> (b, f) + (d, h) = W = ([b | d], [f | h])
//   level 2:  swap16(U.x, W.x) swap16(U.y, W.y) -> U + W = (rows [a, b, c, d], rows [e, f, g, h])
// 6 swaps + 3 packed adds; every 16-lane row of the result holds 16 partial sums of one value.
__device__ __forceinline__ f32x2 wave_fold8_rows(f32x2 ae, f32x2 cg, f32x2 bf, f32x2 dh) {
  auto swap32 = [](float x, float y) {
    return __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  };
  auto swap16 = [](float x, float y) {
    return __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  };
  const auto s0 = swap32(ae.x, cg.x), s1 = swap32(ae.y, cg.y);
  const f32x2 U = f32x2{__uint_as_float(s0[0]), __uint_as_float(s1[0])} +
                  f32x2{__uint_as_float(s0[1]), __uint_as_float(s1[1])};
  const auto s2 = swap32(bf.x, dh.x), s3 = swap32(bf.y, dh.y);
  const f32x2 W = f32x2{__uint_as_float(s2[0]), __uint_as_float(s3[0])} +
                  f32x2{__uint_as_float(s2[1]), __uint_as_float(s3[1])};
  const auto t0 = swap16(U.x, W.x), t1 = swap16(U.y, W.y);
  return f32x2{__uint_as_float(t0[0]), __uint_as_float(t1[0])} +
         f32x2{__uint_as_float(t0[1]), __uint_as_float(t1[1])};
}
// DPP moves inside the 8 lanes that finalise one entry (quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror)
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(x), CTRL, 0xf, 0xf, true));
}

// The loop is the last of the round-2 A/B series (profiles/r2_tiles_variants_ab.txt: 2.14 -> 1.82 ms at
// BASELINE configs[1]; the superseded variants are in the history before this commit): the record is read
// where it is used, (dx, dy) / (B dy, C dy) and the per-pixel colour state and sums are 2-vectors (the
// packed instructions take their operands in place: no v_mov to build pairs; 126 VGPRs, no spill),
// min(alpha_max, .) is one instruction, two entries per trip with each record read from LDS while the
// other entry is processed.  Held to 4 waves per SIMD (128 VGPRs): 3 waves with 168 registers measured
// slower (2.17 ms).
// DET (PS_FLAG_DETERMINISTIC): the partial gradient of a Gaussian over more than kInvSlots tiles goes to the
// slot of its list entry (det_slots[list position], cleared by the caller) instead of nine float atomics.
template <bool DET>
__global__ void __launch_bounds__(kWavesPerBlockBwd* kWave, 4)
tiles_backward_kernel(PsRasterDesc d, const float* __restrict__ records,
                      const uint4* __restrict__ cell_windows,
                      const uint32_t* __restrict__ task_order,
                      const uint32_t* __restrict__ tile_ranges,
                      const uint32_t* __restrict__ point_list, uint32_t capacity,
                      const float* __restrict__ view_params, const float* __restrict__ final_T,
                      const uint32_t* __restrict__ n_contrib,
                      const float4* __restrict__ checkpoint,
                      const uint32_t* __restrict__ tile_end, const float* __restrict__ dL_dcolor,
                      float* __restrict__ grad2d, float* __restrict__ tile_grads,
                      float4* __restrict__ det_slots) {
  __shared__ WaveLdsBwd lds_all[kWavesPerBlockBwd];
  const int G = d.n_gaussians, H = d.height, W = d.width;
  const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
  const int tiles = gx * gy;
  const int V = d.n_scenes * d.views_per_scene;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slot_global = blockIdx.x * kWavesPerBlockBwd + w;
  // two tasks per tile: seg 0 walks the list
  // entries above the tile's split point, seg 1 those up to it, from the state the forward left in
  // `checkpoint` (a list shorter than kSplitMin is not split: seg 1 has nothing to do).  A list entry
  // -- and with it its gradient slot -- belongs to exactly one of the two.  Half-length tasks, twice as
  // many: the launch tail of 7168 one-wave tile tasks on 4096 wave slots shrinks.
  // Launch order = task_order: the tasks sorted by the length of their WALK, longest first
  // (backward_task_order_kernel) -- the walk starts at the tile's last contributor, which the forward
  // left in tile_end, not at the end of the list: where pixels stop early the list length says little
  // about a task's duration, and tasks with little or nothing to do must not sit between working ones
  // (interleaved, nearly-empty one-wave workgroups halved this kernel's throughput:
  // profiles/r4_backward_split_ab.txt); sorted, they trail the grid.
  if (slot_global >= 2 * V * tiles) return;
  const uint32_t task = task_order[slot_global];
  const int tile_global = (int)(task >> 1);
  const uint32_t seg = task & 1u;
  WaveLdsBwd& lds = lds_all[w];
  const int v = tile_global / tiles, t = tile_global % tiles;
  const uint32_t tx = t % gx, ty = t / gx;
  const size_t vo = (size_t)v * G;
  const float* recs = records + vo * kRecFloats;
  const uint4* wins = cell_windows + vo * (kRecFloats / 4);     // (one per record line)
  float* gacc = grad2d + vo * kGradFloats;
  uint32_t l_start = tile_ranges[2 * (size_t)tile_global];
  if (l_start > capacity) l_start = capacity;
  l_start = __builtin_amdgcn_readfirstlane(l_start);       // (wave-uniform: say so)
  const uint32_t* list = point_list + l_start;

  // entries behind the tile's last contributor are never blended (1-based index c_max, left by the
  // forward in tile_end); the geometry backward still sums the private slot of every (Gaussian, tile)
  // pair of a small Gaussian, so the slots of those entries are cleared here (nothing is memset)
  const uint32_t c_max = __builtin_amdgcn_readfirstlane(tile_end[tile_global]);
  float4* const slots = reinterpret_cast<float4*>(tile_grads) + vo * (kInvSlots * kSlotVec);
  uint32_t lo, hi;        // this task walks the entries with 1-based index in (lo, hi], back to front
  {
    uint32_t l_count = tile_ranges[2 * (size_t)tile_global + 1];
    if (l_count > capacity - l_start) l_count = capacity - l_start;
    l_count = __builtin_amdgcn_readfirstlane(l_count);
    const uint32_t split = split_point(l_count);
    if (seg == 1u && split == 0u) return;
    lo = seg == 0u ? split : 0u;
    hi = seg == 0u ? c_max : (c_max < split ? c_max : split);
    if (hi < lo) hi = lo;
    // (the entries of this task's part of the list that lie behind the last contributor)
    const uint32_t clear_end = seg == 0u ? l_count : split;
    for (uint32_t e = hi + (uint32_t)lane; e < clear_end; e += kWave) {
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
  if (hi == lo) return;

  const float* bg = view_params + (size_t)v * PS_VIEW_STRIDE + PS_VIEW_BG;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const size_t P = (size_t)H * W;
  const float x0 = (float)(tx * kTile), y0 = (float)(ty * kTile);
  float T[4], g0[4], g1[4], g2[4], hb[4];
  uint32_t nc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int px = tx * kTile + 8 * (k & 1) + (lane & 7);
    const int py = ty * kTile + 8 * (k >> 1) + (lane >> 3);
    const bool inside = px < W && py < H;
    const size_t pix = inside ? (size_t)py * W + px : 0;
    nc[k] = inside ? n_contrib[(size_t)v * P + pix] : 0u;
    T[k] = inside ? final_T[(size_t)v * P + pix] : 0.f;
    const float* gp = dL_dcolor + (size_t)v * 3 * P;
    g0[k] = inside ? gp[pix] : 0.f;
    g1[k] = inside ? gp[P + pix] : 0.f;
    g2[k] = inside ? gp[2 * P + pix] : 0.f;
    // acc = the colour composited BEHIND the current entry, over the transmittance in front of it -- the
    // BACKGROUND included: behind a pixel's last contributor that is bg itself (T_final bg / T_final), and the
    // recurrence acc <- alpha c + (1 - alpha) acc carries it forward.  dC/dalpha_i = T_i (c_i - acc_i) then holds
    // the reference's second term, -T_final / (1 - alpha_i) (bg . dL/dC), already: no per-pixel register and no
    // per-evaluation FMA for it (rounds 1 - 6a kept -T_final (bg . dL/dC) per pixel).
    // And acc is only ever used in (c_i - acc_i) . dL/dC: the state kept per pixel is the SCALAR
    // hb = acc . dL/dC (dL/dC is fixed per pixel), with the same recurrence hb <- hb + alpha (c . dL/dC - hb):
    // one register instead of three per pixel, and per evaluation c . g - hb instead of a 3-vector difference.
    float a0 = bg0, a1 = bg1, a2 = bg2;
    if (seg == 1u && nc[k] > hi) {       // the pixel's walk continues behind the split point: start from
      const float4 c = checkpoint[((size_t)tile_global * 4 + k) * kWave + lane];   // the forward's state
      const float tb = T[k] / c.x;       // T_final over T at the split (> 0: the pixel passed the split)
      T[k] = c.x; a0 = fmaf(bg0, tb, c.y); a1 = fmaf(bg1, tb, c.z); a2 = fmaf(bg2, tb, c.w);
    }
    hb[k] = fmaf(a2, g2[k], fmaf(a1, g1[k], a0 * g0[k]));
  }
  const f32x2 pxy0 = f32x2{(float)(tx * kTile + (lane & 7)), (float)(ty * kTile + (lane >> 3))};   // pixel of quadrant 0
  const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
  const float alpha_max = d.alpha_max, alpha_min = d.alpha_min;
  const uint64_t lt = lanemask_lt();
  uint32_t n_ring = 0;      // records in the ring: one refine's survivors, from index 0 (wave-uniform)
  // finalisation (stage_batch): entry and value of this lane, float offset of the value's 16 partials in the stage
  // (stage rows 0..7 = Mx, s_r, Mxx, s_b, My, s_g, Mxy, Myy  ->  row of fp = 0, 4, 2, 6, 7, 3, 1, 5)
  const uint32_t fe = (uint32_t)lane >> 3, fp = (uint32_t)lane & 7u;
  const uint32_t f_voff = ((0x51376240u >> (4u * fp)) & 15u) * 16u;
  // the per-pixel colour state as 2-vectors (channels 0,1 | channel 2) so that the packed
  // instructions take their operands in place (the auto-vectoriser packs the scalar form too, but
  // assembles the register pairs with ~5 v_mov per pixel and entry)
  f32x2 g01[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) g01[k] = f32x2{g0[k], g1[k]};

  // refine list entries with 1-based indices top, top-1, ..., top-m+1 (lane i takes top - i)
  // list index of this lane's entry in the batch whose first (highest) 1-based index is `top`;
  // loaded one batch ahead, which takes the first of the refine's two dependent global latencies
  // (list -> record gather) off the critical path for one register
  auto idx_of = [&](uint32_t top) {
    return (uint32_t)lane < top ? list[top - 1u - lane] : list[0];
  };
  uint32_t id_ahead = idx_of(hi);
  uint32_t np_end = 0;     // ring index one past the last entry that is NOT plain
  auto refine = [&](uint32_t top, uint32_t m) {
    bool keep = false, not_plain = false;
    float4 q0, q1, q2;
    const uint32_t id_now = id_ahead;
    if (top - m > lo) id_ahead = idx_of(top - m);
    if ((uint32_t)lane < m) {
      const uint32_t id = id_now;
      const float4* r = reinterpret_cast<const float4*>(recs + (size_t)id * kRecFloats);
      const float4 r0 = r[0], r1 = r[1], r2 = r[2];   // {px,py,cx,cy} {cz,o,depth,radius} {r,g,b,-}
      const uint4 win = wins[(size_t)id * (kRecFloats / 4)];
      const float A = -0.5f * kLog2e * r0.z, B = -kLog2e * r0.w, Cq = -0.5f * kLog2e * r1.x;
      // which quadrants the entry can reach with alpha >= alpha_min: read off the pair's cell window (the
      // preprocess computed it once per pair, cell_window.h; rounds 2 - 5 minimised the quadratic form over
      // each quadrant's box here, per list entry: nine divisions, 2/3 of the refine's instructions)
      const uint32_t qm = tile_quad_mask(win, (int)tx, (int)ty);
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
      not_plain = keep && !entry_is_plain(r0.x, r0.y, A, B, Cq, r1.y, alpha_max);
    }
    const uint64_t mask = __ballot(keep);
    {
      const uint64_t npm = __ballot(not_plain);
      np_end = 0u;
      if (npm) {
        const int top_lane = 63 - __builtin_clzll(npm);
        np_end = __builtin_amdgcn_readfirstlane((uint32_t)__popcll(mask & ((2ull << top_lane) - 1ull)));
      }
    }
    if (keep) {
      const uint32_t slot = (uint32_t)__popcll(mask & lt);
      lds.rec[slot][0] = q0; lds.rec[slot][1] = q1; lds.rec[slot][2] = q2;
    }
    n_ring = __builtin_amdgcn_readfirstlane((uint32_t)__popcll(mask));   // (wave-uniform: say so)
    wave_lds_sync();
  };

  uint32_t hitbits = 0;   // bit j: entry j of the finalisation batch contributed (wave-uniform)
  // one ring entry: the pixels' updates, the nine wave sums down to 16-lane partials, staged in LDS.
  // FAST (decided per blend call, see blend()): every entry of the call is plain (entry_is_plain:
  // no sign test of the power, no alpha_max clamp) and lies at or before EVERY pixel's last
  // contributor (no `hidx <= n_contrib` test), and alpha = opacity * G lets q = opacity * G * dL/dalpha
  // be formed as alpha * dL/dalpha without the select on G dL/dalpha: five of the half-rate compare /
  // select class instructions and a multiply less per quadrant evaluation (tools/issue_model.hip).
  // The opacity sum then holds sum(q) = opacity * sum(G dL/dalpha); the finalising lane divides.
  auto entry = [&](auto fast_tag, uint32_t j, const float4 q0, const float4 q1, const float4 q2) {
      constexpr bool FAST = decltype(fast_tag)::value;
      const float o = q1.y, c0 = q1.z, c1 = q1.w, c2 = q2.x;
      const uint32_t hidx = __float_as_uint(q2.y);
      const uint32_t qm = __builtin_amdgcn_readfirstlane(__float_as_uint(q2.z));
      // the nine sums as the blend loop's register pairs: (Mx, My) (Mxx, Mxy) (s_r, s_g) (s_b, Myy) + s_op
      f32x2 M1 = {0.f, 0.f}, M2 = {0.f, 0.f}, s_rg = {0.f, 0.f}, sbM = {0.f, 0.f};
      float s_op = 0.f;
      uint64_t any = 0ull;      // lanes with a contributing pixel, as the compares' own masks (scalar ORs)
      {
        const f32x2 c01 = f32x2{c0, c1};
        const f32x2 d0 = f32x2{q0.x, q0.y} - pxy0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (qm & (1u << k)) {   // wave-uniform quadrant skip
            // (dx, dy) to this lane's pixel of quadrant k: the entry's offset to the pixel of quadrant 0, shifted
            const f32x2 dd = k == 0 ? d0 : d0 - f32x2{8.f * (float)(k & 1), 8.f * (float)(k >> 1)};
            const f32x2 bc = f32x2{q0.w, q1.x} * f32x2{dd.y, dd.y};        // (B dy, C dy)
            const float pw = fmaf(dd.x, fmaf(q0.z, dd.x, bc.x), dd.y * bc.y);
            const float Gv = fast_exp2(pw);
            float alpha;
            bool ok;
            if (FAST) {
              alpha = o * Gv;
              ok = alpha >= alpha_min;
            } else {                                         // min(alpha_max, o G) in one instruction
              asm("v_min_f32 %0, %1, %2" : "=v"(alpha) : "s"(alpha_max), "v"(o * Gv));
              ok = (hidx <= nc[k]) & (pw <= 0.f) & (alpha >= alpha_min);
            }
            const float ale = ok ? alpha : 0.f;            // 0 => all updates are no-ops
            const float rcp = __builtin_amdgcn_rcpf(1.f - ale);   // 1 ulp; exact 1 when ale == 0
            const float Tn = T[k] * rcp;                    // T in front of this entry
            const float e = fmaf(c2, g2[k], fmaf(c01.y, g01[k].y, fmaf(c01.x, g01[k].x, -hb[k])));   // (c - acc) . dL/dC
            const float dch = ale * Tn;
            s_rg = f32x2{dch, dch} * g01[k] + s_rg;
            float q;                                        // opacity * G * dL/dalpha = G * dL/dG
            if (FAST) {
              q = dch * e;                                  // alpha dL/dalpha = (alpha T) e
              s_op += q;
            } else {
              const float gda = ok ? Gv * (e * Tn) : 0.f;   // G * dL/dalpha
              s_op += gda;
              q = o * gda;
            }
            const f32x2 qxy = f32x2{q, q} * dd;             // (q dx, q dy)
            M1 += qxy;
            M2 = f32x2{qxy.x, qxy.x} * dd + M2;             // (Mxx, Mxy) += q dx (dx, dy)
            // (s_b, Myy) += (dch g2, q dy dy) as TWO scalar FMAs on the halves of a pair: packed, the two factor
            // pairs would have to be assembled with a v_mov each, per evaluation
            asm("v_fmac_f32 %0, %1, %2" : "+v"(sbM.x) : "v"(dch), "v"(g2[k]));
            asm("v_fmac_f32 %0, %1, %2" : "+v"(sbM.y) : "v"(qxy.y), "v"(dd.y));
            T[k] = Tn;
            hb[k] = fmaf(ale, e, hb[k]);                    // (alpha c + (1 - alpha) acc) . dL/dC
            if (FAST) any |= __builtin_amdgcn_ballot_w64(ok);
            else any |= __builtin_amdgcn_ballot_w64(hidx <= nc[k]) & __builtin_amdgcn_ballot_w64(pw <= 0.f) &
                        __builtin_amdgcn_ballot_w64(alpha >= alpha_min);
          }
        }
      }
      if (any != 0ull) {
        // moments about the Gaussian centre: Mx = sum q dx, My = sum q dy, ...
        // r.x rows = [Mx, s_r, Mxx, s_b], r.y rows = [My, s_g, Mxy, Myy]; all 64 lanes stage their partials
        const f32x2 r = wave_fold8_rows(M1, M2, s_rg, sbM);
        float* st = &lds.stage[j][0][lane];
        st[0] = r.x; st[kWave] = r.y; st[2 * kWave] = s_op;
        hitbits |= 1u << j;                 // wave-uniform: stays in an SGPR
      }
  };

  // kStage ring entries from entry `base` of the current blend call, two per trip (each one's
  // record read from LDS while the other is processed), then lane j finalises entry j: its
  // private slot (always written: values or zeros), or one set of 9 atomics per (tile, Gaussian)
  // for the large ones
  auto stage_batch = [&](auto fast_tag, uint32_t base, uint32_t cnt) {
    constexpr bool FAST = decltype(fast_tag)::value;
    hitbits = 0;
    {
      // (the records are read through ONE running pointer: entry j + 1 / j + 2 at fixed offsets from it; a read
      // beyond cnt -- at most two records past the ring, still inside this wave's LDS -- is never used)
      const float4* rp = &lds.rec[base][0];
      float4 a0 = rp[0], a1 = rp[1], a2 = rp[2];
      for (uint32_t j = 0; j < cnt; j += 2) {
        const float4 b0 = rp[3], b1 = rp[4], b2 = rp[5];
        entry(fast_tag, j, a0, a1, a2);
        if (j + 1 >= cnt) break;
        a0 = rp[6]; a1 = rp[7]; a2 = rp[8];
        entry(fast_tag, j + 1, b0, b1, b2);
        rp += 6;
      }
    }
    wave_lds_sync();
    // eight lanes finalise entry fe: lane fp adds the 16 partials of ONE of the eight folded values and 8 of the
    // ninth's 64, the ninth is then summed over the eight lanes; lane fp produces float fp of the 9-float result
    //   fp:     0    1    2    3    4    5    6    7
    //   value:  Mx   My   Mxx  Mxy  Myy  s_b  s_r  s_g      (stage rows: r1 = [Mx, s_r, Mxx, s_b], r2 = [My, s_g, Mxy, Myy])
    //   result: o0   o1   o2   o3   o4   s_op s_r  s_g  + float 8 = s_b (from lane 5)
    // private slot (always written: values or zeros), per-entry slot (deterministic mode), or atomics
    if (fe < cnt) {
      const float4* fr = &lds.rec[base + fe][0];
      const float4 q0 = fr[0], q1 = fr[1], q2 = fr[2];
      const float* sp = &lds.stage[fe][0][0];
      // (the 16-byte words are visited in a lane-rotated order: the eight lanes of an entry -- and with them all
      // 64 -- read eight different words of the 128-byte bank window at a time)
      const float4* vp = reinterpret_cast<const float4*>(sp + f_voff);
      const float4 u0 = vp[fp & 3u], u1 = vp[(fp + 1u) & 3u], u2 = vp[(fp + 2u) & 3u], u3 = vp[(fp + 3u) & 3u];
      const float4* wp = reinterpret_cast<const float4*>(sp + 2 * kWave + fp * 8u);
      const float4 w0 = wp[(fp >> 2) & 1u], w1 = wp[((fp >> 2) & 1u) ^ 1u];
      const bool hit = (hitbits >> fe) & 1u;
      const float4 us = (u0 + u1) + (u2 + u3);
      const float own = (us.x + us.y) + (us.z + us.w);
      const float4 ws = w0 + w1;
      float s_op = (ws.x + ws.y) + (ws.z + ws.w);
      s_op += dpp_mov<0xB1>(s_op);          // quad_perm [1,0,3,2]
      s_op += dpp_mov<0x4E>(s_op);          // quad_perm [2,3,0,1]
      s_op += dpp_mov<0x141>(s_op);         // row_half_mirror: the other quad of the eight
      if (FAST) s_op *= __builtin_amdgcn_rcpf(q1.y);   // the fast form summed opacity * G dL/dalpha (rcp: 1 ulp)
      const float other = dpp_mov<0xB1>(own);   // lanes 0 / 1: My / Mx
      const float cx = q0.z * (-2.f / kLog2e), cy = q0.w * (-1.f / kLog2e),
                  cz = q1.x * (-2.f / kLog2e);
      float out = fp == 0u ? (-cx * own - cy * other) * ddelx_dx : (-cz * own - cy * other) * ddely_dy;
      out = fp >= 2u ? -0.5f * own : out;
      out = fp >= 6u ? own : out;
      out = fp == 5u ? s_op : out;
      const uint32_t i2 = 8u + ((fp - 5u) & 7u);      // float 8 (s_b) from lane 5, the slot's padding from its neighbours
      // (32-bit byte offsets from wave-uniform bases: a view's slots and rows are far below 4 GB)
      if (__float_as_uint(q2.z) & 16u) {
        // private slot of this (Gaussian, tile): plain stores, summed later in a fixed order by the geometry
        // backward (no atomics, deterministic)
        char* tg = reinterpret_cast<char*>(slots) + __float_as_uint(q2.w) * (uint32_t)(kSlotFloats * 4);
        *reinterpret_cast<float*>(tg + 4u * fp) = hit ? out : 0.f;
        if (i2 < (uint32_t)kSlotFloats) *reinterpret_cast<float*>(tg + 4u * i2) = (hit && fp == 5u) ? own : 0.f;
      } else if (DET) {
        if (hit) {      // (entries that never get here keep the zeros the caller cleared the slots to)
          float* ds = reinterpret_cast<float*>(det_slots + ((size_t)l_start + (__float_as_uint(q2.y) - 1u)) * kSlotVec);
          ds[fp] = out;
          if (i2 < (uint32_t)kSlotFloats) ds[i2] = fp == 5u ? own : 0.f;
        }
      } else if (hit) {
        float* ga = reinterpret_cast<float*>(reinterpret_cast<char*>(gacc) +
                                             __float_as_uint(q2.w) * (uint32_t)(kGradFloats * 4) + 4u * fp);
        atomicAdd(ga, out);
        if (fp == 5u) atomicAdd(ga + 3, own);      // float 8 = s_b
      }
    }
    wave_lds_sync();     // the staging rows are rewritten by the next batch
  };
  // the first pixel to finish: entries at or before it are inside EVERY pixel's walk (pixels outside
  // the image count as finished at 0)
  uint32_t nc_min;
  {
    uint32_t n = nc[0] < nc[1] ? nc[0] : nc[1];
    n = nc[2] < n ? nc[2] : n;
    n = nc[3] < n ? nc[3] : n;
    nc_min = ~wave_max_u(~n);
  }
  auto blend = [&](uint32_t m) {
    for (uint32_t base = 0; base < m; base += kStage) {
      const uint32_t cnt = m - base < (uint32_t)kStage ? m - base : (uint32_t)kStage;
      // the batch's first entry has its highest list index (the walk runs back to front)
      const uint32_t top_h =
          __builtin_amdgcn_readfirstlane(__float_as_uint(lds.rec[base][2].y));
      // short form: every entry from here to the ring's tail is plain and inside every pixel's walk
      if (base >= np_end && top_h <= nc_min) stage_batch(std::true_type{}, base, cnt);
      else stage_batch(std::false_type{}, base, cnt);
    }
  };

  for (uint32_t top = hi; top > lo;) {
    const uint32_t m = top - lo < (uint32_t)kBwdBatch ? top - lo : (uint32_t)kBwdBatch;
    refine(top, m);           // (the ring holds exactly one refine)
    if (n_ring != 0u) blend(n_ring);
    top -= m;
  }
}

// Launch order of the tile backward: its 2 x V x tiles tasks sorted by walk length, longest first
// (counting sort into 1024 buckets, one block: pure latency, ~15 us).  Task id = tile << 1 | seg.
__global__ void __launch_bounds__(1024)
backward_task_order_kernel(int n_tiles, const uint32_t* __restrict__ tile_ranges,
                           const uint32_t* __restrict__ tile_end, uint32_t capacity,
                           uint32_t* __restrict__ task_order) {
  __shared__ uint32_t hist[1024];
  __shared__ uint32_t part[1024];
  __shared__ uint32_t s_max;
  const int n_tasks = 2 * n_tiles;
  const int per = (n_tasks + 1023) / 1024;
  const int lo = threadIdx.x * per, hi = lo + per < n_tasks ? lo + per : n_tasks;
  auto walk = [&](int task) -> uint32_t {       // entries the task walks (tiles_backward_kernel: lo / hi)
    const int tile = task >> 1;
    uint32_t l_start = tile_ranges[2 * (size_t)tile], l_count = tile_ranges[2 * (size_t)tile + 1];
    if (l_start > capacity) l_start = capacity;
    if (l_count > capacity - l_start) l_count = capacity - l_start;
    const uint32_t split = split_point(l_count), c = tile_end[tile];
    if (task & 1) return c < split ? c : split;          // seg 1: (0, min(c_max, split)]
    return c > split ? c - split : 0u;                    // seg 0: (split, c_max]
  };
  // a single block: pure latency.  The thread's (up to kPer) walk lengths are computed ONCE, every load
  // issued before the first use, and kept in registers for all three uses (maximum, histogram, scatter)
  constexpr int kPer = 16;
  const bool fits = per <= kPer;                                  // uniform
  uint32_t wl[kPer];
#pragma unroll
  for (int u = 0; u < kPer; ++u) wl[u] = walk(lo + u < n_tasks ? lo + u : n_tasks - 1);   // (unused when !fits)
  if (threadIdx.x == 0) s_max = 1u;
  hist[threadIdx.x] = 0u;
  __syncthreads();
  uint32_t mx = 0;
  if (fits) {
#pragma unroll
    for (int u = 0; u < kPer; ++u) if (lo + u < hi) mx = wl[u] > mx ? wl[u] : mx;
  } else {
    for (int i = lo; i < hi; ++i) { const uint32_t c = walk(i); mx = c > mx ? c : mx; }
  }
  atomicMax(&s_max, mx);
  __syncthreads();
  const uint32_t maxc = s_max;
  // bucket 0 = longest.  Any monotone map of the walk length will do (the order inside a bucket is arbitrary but
  // fixed): one float multiply -- the 64-bit division this replaced was ~40 instructions per task, twice
  const float scale = 1023.f / (float)maxc;
  auto bucket = [&](uint32_t c) -> uint32_t {
    const uint32_t b = (uint32_t)((float)c * scale);
    return 1023u - (b > 1023u ? 1023u : b);
  };
  if (fits) {
#pragma unroll
    for (int u = 0; u < kPer; ++u) { wl[u] = bucket(wl[u]); if (lo + u < hi) atomicAdd(&hist[wl[u]], 1u); }   // (wl = bucket from here on)
  } else {
    for (int i = lo; i < hi; ++i) atomicAdd(&hist[bucket(walk(i))], 1u);
  }
  __syncthreads();
  // exclusive scan of the 1024 bucket counts: inside each wave by DPP-free shuffles (no barrier), the 16 wave totals
  // by the first wave, two barriers in all (the Hillis-Steele scan over LDS this replaced took twenty)
  const uint32_t hsum = hist[threadIdx.x];
  uint32_t hx = hsum;
  const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t y = __shfl_up(hx, off);
    if (ln >= off) hx += y;
  }
  if (ln == 63) part[wv] = hx;
  __syncthreads();
  if (threadIdx.x < 64) {
    const uint32_t t = threadIdx.x < 16 ? part[threadIdx.x] : 0u;
    uint32_t tx = t;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const uint32_t y = __shfl_up(tx, off);
      if ((int)threadIdx.x >= off) tx += y;
    }
    if (threadIdx.x < 16) part[threadIdx.x] = tx - t;       // exclusive start of the wave's buckets
  }
  __syncthreads();
  hist[threadIdx.x] = part[wv] + hx - hsum;   // exclusive start of the bucket
  __syncthreads();
  if (fits) {
#pragma unroll
    for (int u = 0; u < kPer; ++u)
      if (lo + u < hi) task_order[atomicAdd(&hist[wl[u]], 1u)] = (uint32_t)(lo + u);
  } else {
    for (int i = lo; i < hi; ++i) task_order[atomicAdd(&hist[bucket(walk(i))], 1u)] = (uint32_t)i;
  }
}

void launch_backward_task_order(const PsRasterDesc& d, const uint32_t* tile_ranges,
                                const uint32_t* tile_end, uint32_t capacity, uint32_t* task_order,
                                hipStream_t st) {
  const Dims m = make_dims(d);
  hipLaunchKernelGGL(backward_task_order_kernel, dim3(1), dim3(1024), 0, st, m.V * m.tiles, tile_ranges,
                     tile_end, capacity, task_order);
}

void launch_tiles_backward(const PsRasterDesc& d, const float* records, const uint4* cell_windows,
                           const uint32_t* task_order, const uint32_t* tile_ranges,
                           const uint32_t* point_list,
                           uint32_t capacity, const float* view_params, const float* final_T,
                           const uint32_t* n_contrib, const float4* checkpoint,
                           const uint32_t* tile_end, const float* dL_dcolor, float* grad2d,
                           float* tile_grads, float* det_slots, hipStream_t st) {
  const Dims m = make_dims(d);
  const int total = 2 * m.V * m.tiles;
  dim3 grid((total + kWavesPerBlockBwd - 1) / kWavesPerBlockBwd), block(kWavesPerBlockBwd * kWave);
  if (det_slots != nullptr)
    hipLaunchKernelGGL(tiles_backward_kernel<true>, grid, block, 0, st, d, records, cell_windows, task_order, tile_ranges,
                       point_list, capacity, view_params, final_T, n_contrib, checkpoint, tile_end,
                       dL_dcolor, grad2d, tile_grads, reinterpret_cast<float4*>(det_slots));
  else
    hipLaunchKernelGGL(tiles_backward_kernel<false>, grid, block, 0, st, d, records, cell_windows, task_order, tile_ranges,
                       point_list, capacity, view_params, final_T, n_contrib, checkpoint, tile_end,
                       dL_dcolor, grad2d, tile_grads, (float4*)nullptr);
}

}  // namespace ps
