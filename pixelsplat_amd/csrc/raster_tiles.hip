// Tile kernels: one wave64 owns one 16x16 tile (4 pixels per lane: x = lane & 15,
// y = (lane >> 4) + 4k).  There is no duplicated (tile, Gaussian) key list and no global
// sort over it: the wave streams its view's depth-sorted rect array (L2 resident, 8 B per
// Gaussian), keeps the entries whose rect covers the tile with ballot + popcount
// compaction, stages the surviving records in LDS 64 at a time and alpha-composites them
// front to back.  The sequence of survivors IS the reference's per-tile bin
// (SURVEY.md A.2/A.3), in the same order.
//
// Replaces renderCUDA fwd/bwd of the external rasterizer (call site
// /root/reference/src/model/decoder/cuda_splatting.py:117-124).
#include "raster_common.h"

namespace ps {

constexpr int kBatch = 64;         // entries blended per LDS stage
constexpr int kQueue = 2 * kBatch; // compaction queue capacity (positions)
constexpr int kWavesPerBlock = 4;

struct WaveLds {
  uint32_t queue[kQueue];
  float4 rec[kBatch][3];
};

__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ uint32_t wave_max_u(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const uint32_t u = __shfl_xor(v, o); v = u > v ? u : v; }
  return v;
}

// ------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kWavesPerBlock* kWave)
tiles_forward_kernel(PsRasterDesc d, const float* __restrict__ records,
                     const uint32_t* __restrict__ sorted_idx,
                     const uint2* __restrict__ sorted_rect, const uint32_t* __restrict__ n_vis,
                     const float* __restrict__ view_params, float* __restrict__ out_color,
                     float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                     uint32_t* __restrict__ tile_end) {
  __shared__ WaveLds lds_all[kWavesPerBlock];
  const int G = d.n_gaussians, H = d.height, W = d.width;
  const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
  const int tiles = gx * gy;
  const int V = d.n_scenes * d.views_per_scene;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile_global = blockIdx.x * kWavesPerBlock + w;
  if (tile_global >= V * tiles) return;
  WaveLds& lds = lds_all[w];
  const int v = tile_global / tiles, t = tile_global % tiles;
  const uint32_t tx = t % gx, ty = t / gx;
  const size_t vo = (size_t)v * G;
  const uint32_t n = n_vis[v];
  const uint2* srect = sorted_rect + vo;
  const uint32_t* sidx = sorted_idx + vo;
  const float* recs = records + vo * kRecFloats;

  const int px = tx * kTile + (lane & 15);
  const float pxf = (float)px;
  int py[4]; float pyf[4]; bool done[4];
  float T[4], C0[4], C1[4], C2[4]; uint32_t last[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    py[k] = ty * kTile + (lane >> 4) + 4 * k;
    pyf[k] = (float)py[k];
    done[k] = !(px < W && py[k] < H);
    T[k] = 1.f; C0[k] = C1[k] = C2[k] = 0.f; last[k] = 0;
  }
  uint32_t last_pos = 0;       // sorted position + 1 of my latest contributor
  uint32_t contributor = 0;    // wave-uniform count of list entries walked so far
  uint32_t qn = 0;             // wave-uniform queue fill
  const uint64_t lt = lanemask_lt();
  const float alpha_max = d.alpha_max, alpha_min = d.alpha_min, t_min = d.t_min;
  bool all_done = __all(done[0] & done[1] & done[2] & done[3]);

  auto process = [&](uint32_t m) {
    // stage m records (lane j fetches entry j)
    if ((uint32_t)lane < m) {
      const uint32_t p = lds.queue[lane];
      const uint32_t id = sidx[p];
      const float4* r = reinterpret_cast<const float4*>(recs + (size_t)id * kRecFloats);
      float4 r0 = r[0], r1 = r[1], r2 = r[2];
      r2.y = __uint_as_float(p);  // carry the sorted position in the (unused here) depth slot
      lds.rec[lane][0] = r0; lds.rec[lane][1] = r1; lds.rec[lane][2] = r2;
    }
    wave_lds_sync();
    for (uint32_t j = 0; j < m; ++j) {
      const float4 r0 = lds.rec[j][0], r1 = lds.rec[j][1];
      const float4 r2 = lds.rec[j][2];
      contributor++;
      const float gxp = r0.x, gyp = r0.y, cx = r0.z, cy = r0.w, cz = r1.x, o = r1.y;
      const float dx = gxp - pxf;
      bool any_contrib = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dy = gyp - pyf[k];
        const float power = -0.5f * (cx * dx * dx + cz * dy * dy) - cy * dx * dy;
        const float alpha = fminf(alpha_max, o * __expf(power));
        const bool ok = !done[k] && power <= 0.f && alpha >= alpha_min;
        if (ok) {
          const float test_T = T[k] * (1.f - alpha);
          if (test_T < t_min) {
            done[k] = true;
          } else {
            const float wgt = alpha * T[k];
            C0[k] += r1.z * wgt; C1[k] += r1.w * wgt; C2[k] += r2.x * wgt;
            T[k] = test_T;
            last[k] = contributor;
            any_contrib = true;
          }
        }
      }
      if (any_contrib) last_pos = __float_as_uint(r2.y) + 1u;
      if (__all(done[0] & done[1] & done[2] & done[3])) { all_done = true; break; }
    }
    wave_lds_sync();
  };

  for (uint32_t basep = 0; basep < n && !all_done; basep += kWave) {
    const uint32_t p = basep + lane;
    bool hit = false;
    if (p < n) hit = rect_covers(srect[p], tx, ty);
    const uint64_t mask = __ballot(hit);
    if (mask != 0ull) {
      if (hit) lds.queue[qn + (uint32_t)__popcll(mask & lt)] = p;
      qn += (uint32_t)__popcll(mask);
      wave_lds_sync();
      if (qn >= (uint32_t)kBatch) {
        process(kBatch);
        // shift the remainder down
        const uint32_t rem = qn - kBatch;
        uint32_t tmp = 0;
        if ((uint32_t)lane < rem) tmp = lds.queue[kBatch + lane];
        wave_lds_sync();
        if ((uint32_t)lane < rem) lds.queue[lane] = tmp;
        wave_lds_sync();
        qn = rem;
      }
    }
  }
  if (!all_done && qn > 0) process(qn);

  // epilogue
  const float* bg = view_params + (size_t)v * PS_VIEW_STRIDE + PS_VIEW_BG;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const size_t P = (size_t)H * W;
  uint32_t max_c = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (px < W && py[k] < H) {
      const size_t pix = (size_t)py[k] * W + px;
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
  const uint32_t max_p = wave_max_u(last_pos);
  if (lane == 0) {
    tile_end[2 * (size_t)tile_global] = max_c;
    tile_end[2 * (size_t)tile_global + 1] = max_p;
  }
}

void launch_tiles_forward(const PsRasterDesc& d, const float* records, const uint32_t* sorted_idx,
                          const uint2* sorted_rect, const uint32_t* n_vis,
                          const float* view_params, float* out_color, float* final_T,
                          uint32_t* n_contrib, uint32_t* tile_end, hipStream_t st) {
  const Dims m = make_dims(d);
  const int total = m.V * m.tiles;
  dim3 grid((total + kWavesPerBlock - 1) / kWavesPerBlock), block(kWavesPerBlock * kWave);
  hipLaunchKernelGGL(tiles_forward_kernel, grid, block, 0, st, d, records, sorted_idx,
                     sorted_rect, n_vis, view_params, out_color, final_T, n_contrib, tile_end);
}

// ------------------------------------------------------------------------------------
// parity export of the bins (tests only; not on the training path)
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kWavesPerBlock* kWave)
export_bins_kernel(PsRasterDesc d, const uint32_t* __restrict__ sorted_idx,
                   const uint2* __restrict__ sorted_rect, const uint32_t* __restrict__ n_vis,
                   uint32_t* __restrict__ tile_counts, const uint32_t* __restrict__ tile_offsets,
                   uint32_t* __restrict__ point_list, size_t capacity) {
  const int G = d.n_gaussians;
  const int gx = (d.width + kTile - 1) / kTile, gy = (d.height + kTile - 1) / kTile;
  const int tiles = gx * gy;
  const int V = d.n_scenes * d.views_per_scene;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile_global = blockIdx.x * kWavesPerBlock + w;
  if (tile_global >= V * tiles) return;
  const int v = tile_global / tiles, t = tile_global % tiles;
  const uint32_t tx = t % gx, ty = t / gx;
  const size_t vo = (size_t)v * G;
  const uint32_t n = n_vis[v];
  const uint64_t lt = lanemask_lt();
  uint32_t count = 0;
  const size_t off = (point_list && tile_offsets) ? tile_offsets[tile_global] : 0;
  for (uint32_t basep = 0; basep < n; basep += kWave) {
    const uint32_t p = basep + lane;
    bool hit = false;
    if (p < n) hit = rect_covers(sorted_rect[vo + p], tx, ty);
    const uint64_t mask = __ballot(hit);
    if (hit && point_list) {
      const size_t q = off + count + (uint32_t)__popcll(mask & lt);
      if (q < capacity) point_list[q] = sorted_idx[vo + p];
    }
    count += (uint32_t)__popcll(mask);
  }
  if (lane == 0 && tile_counts) tile_counts[tile_global] = count;
}

void launch_export_bins(const PsRasterDesc& d, const uint32_t* sorted_idx,
                        const uint2* sorted_rect, const uint32_t* n_vis, uint32_t* tile_counts,
                        const uint32_t* tile_offsets, uint32_t* point_list, size_t capacity,
                        hipStream_t st) {
  const Dims m = make_dims(d);
  const int total = m.V * m.tiles;
  dim3 grid((total + kWavesPerBlock - 1) / kWavesPerBlock), block(kWavesPerBlock * kWave);
  hipLaunchKernelGGL(export_bins_kernel, grid, block, 0, st, d, sorted_idx, sorted_rect, n_vis,
                     tile_counts, tile_offsets, point_list, capacity);
}

// ------------------------------------------------------------------------------------
// backward: walk the tile's bin back to front from the last contributor
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__global__ void __launch_bounds__(kWavesPerBlock* kWave)
tiles_backward_kernel(PsRasterDesc d, const float* __restrict__ records,
                      const uint32_t* __restrict__ sorted_idx,
                      const uint2* __restrict__ sorted_rect,
                      const float* __restrict__ view_params, const float* __restrict__ final_T,
                      const uint32_t* __restrict__ n_contrib,
                      const uint32_t* __restrict__ tile_end, const float* __restrict__ dL_dcolor,
                      float* __restrict__ grad2d) {
  __shared__ WaveLds lds_all[kWavesPerBlock];
  const int G = d.n_gaussians, H = d.height, W = d.width;
  const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
  const int tiles = gx * gy;
  const int V = d.n_scenes * d.views_per_scene;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile_global = blockIdx.x * kWavesPerBlock + w;
  if (tile_global >= V * tiles) return;
  WaveLds& lds = lds_all[w];
  const int v = tile_global / tiles, t = tile_global % tiles;
  const uint32_t tx = t % gx, ty = t / gx;
  const size_t vo = (size_t)v * G;
  const uint2* srect = sorted_rect + vo;
  const uint32_t* sidx = sorted_idx + vo;
  const float* recs = records + vo * kRecFloats;
  float* gacc = grad2d + vo * kGradFloats;

  const uint32_t c_max = tile_end[2 * (size_t)tile_global];
  const uint32_t p_end = tile_end[2 * (size_t)tile_global + 1];  // position + 1
  if (c_max == 0) return;

  const float* bg = view_params + (size_t)v * PS_VIEW_STRIDE + PS_VIEW_BG;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const size_t P = (size_t)H * W;
  const int px = tx * kTile + (lane & 15);
  const float pxf = (float)px;
  float pyf[4], T[4], Tfin[4], g0[4], g1[4], g2[4], bgdot[4];
  float acc0[4], acc1[4], acc2[4], last_alpha[4], lc0[4], lc1[4], lc2[4];
  uint32_t nc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int py = ty * kTile + (lane >> 4) + 4 * k;
    pyf[k] = (float)py;
    const bool inside = px < W && py < H;
    const size_t pix = inside ? (size_t)py * W + px : 0;
    nc[k] = inside ? n_contrib[(size_t)v * P + pix] : 0u;
    Tfin[k] = inside ? final_T[(size_t)v * P + pix] : 0.f;
    T[k] = Tfin[k];
    const float* gp = dL_dcolor + (size_t)v * 3 * P;
    g0[k] = inside ? gp[pix] : 0.f;
    g1[k] = inside ? gp[P + pix] : 0.f;
    g2[k] = inside ? gp[2 * P + pix] : 0.f;
    bgdot[k] = bg0 * g0[k] + bg1 * g1[k] + bg2 * g2[k];
    acc0[k] = acc1[k] = acc2[k] = 0.f; last_alpha[k] = 0.f; lc0[k] = lc1[k] = lc2[k] = 0.f;
  }
  const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
  const float alpha_max = d.alpha_max, alpha_min = d.alpha_min;
  const uint64_t lt = lanemask_lt();
  uint32_t c_run = c_max;  // 1-based list index of the next entry to process
  uint32_t qn = 0;

  auto process = [&](uint32_t m) {
    uint32_t my_id = 0;
    if ((uint32_t)lane < m) {
      const uint32_t p = lds.queue[lane];
      my_id = sidx[p];
      const float4* r = reinterpret_cast<const float4*>(recs + (size_t)my_id * kRecFloats);
      lds.rec[lane][0] = r[0]; lds.rec[lane][1] = r[1]; lds.rec[lane][2] = r[2];
    }
    wave_lds_sync();
    for (uint32_t j = 0; j < m; ++j) {
      const uint32_t cidx = c_run - j;  // >= 1
      const float4 r0 = lds.rec[j][0], r1 = lds.rec[j][1];
      const float4 r2 = lds.rec[j][2];
      const float gxp = r0.x, gyp = r0.y, cx = r0.z, cy = r0.w, cz = r1.x, o = r1.y;
      const float c0 = r1.z, c1 = r1.w, c2 = r2.x;
      const float dx = gxp - pxf;
      float s_dx = 0.f, s_dy = 0.f, s_ca = 0.f, s_cb = 0.f, s_cc = 0.f, s_op = 0.f;
      float s_r = 0.f, s_g = 0.f, s_b = 0.f;
      bool any = false;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dy = gyp - pyf[k];
        const float power = -0.5f * (cx * dx * dx + cz * dy * dy) - cy * dx * dy;
        const float Gv = __expf(power);
        const float alpha = fminf(alpha_max, o * Gv);
        const bool ok = (cidx <= nc[k]) && power <= 0.f && alpha >= alpha_min;
        if (ok) {
          any = true;
          T[k] = T[k] / (1.f - alpha);
          const float dch = alpha * T[k];
          acc0[k] = last_alpha[k] * lc0[k] + (1.f - last_alpha[k]) * acc0[k]; lc0[k] = c0;
          acc1[k] = last_alpha[k] * lc1[k] + (1.f - last_alpha[k]) * acc1[k]; lc1[k] = c1;
          acc2[k] = last_alpha[k] * lc2[k] + (1.f - last_alpha[k]) * acc2[k]; lc2[k] = c2;
          float dL_dalpha = (c0 - acc0[k]) * g0[k] + (c1 - acc1[k]) * g1[k] +
                            (c2 - acc2[k]) * g2[k];
          s_r += dch * g0[k]; s_g += dch * g1[k]; s_b += dch * g2[k];
          dL_dalpha *= T[k];
          last_alpha[k] = alpha;
          dL_dalpha += (-Tfin[k] / (1.f - alpha)) * bgdot[k];
          const float dL_dG = o * dL_dalpha;
          const float gdx = Gv * dx, gdy = Gv * dy;
          s_dx += dL_dG * (-gdx * cx - gdy * cy);
          s_dy += dL_dG * (-gdy * cz - gdx * cy);
          s_ca += -0.5f * gdx * dx * dL_dG;
          s_cb += -0.5f * gdx * dy * dL_dG;
          s_cc += -0.5f * gdy * dy * dL_dG;
          s_op += Gv * dL_dalpha;
        }
      }
      if (__any(any)) {
        s_dx = wave_sum(s_dx) * ddelx_dx; s_dy = wave_sum(s_dy) * ddely_dy;
        s_ca = wave_sum(s_ca); s_cb = wave_sum(s_cb); s_cc = wave_sum(s_cc);
        s_op = wave_sum(s_op);
        s_r = wave_sum(s_r); s_g = wave_sum(s_g); s_b = wave_sum(s_b);
        // lane q (< 9) owns component q of entry j
        float mine = s_dx;
        mine = lane == 1 ? s_dy : mine; mine = lane == 2 ? s_ca : mine;
        mine = lane == 3 ? s_cb : mine; mine = lane == 4 ? s_cc : mine;
        mine = lane == 5 ? s_op : mine; mine = lane == 6 ? s_r : mine;
        mine = lane == 7 ? s_g : mine;  mine = lane == 8 ? s_b : mine;
        const uint32_t id = __shfl(my_id, (int)j);
        if (lane < kGradFloats) atomicAdd(gacc + (size_t)id * kGradFloats + lane, mine);
      }
    }
    c_run -= m;
    wave_lds_sync();
  };

  // stream positions p_end-1 ... 0 in descending order; lane 0 takes the highest
  for (uint32_t top = p_end; top > 0 && c_run > 0;) {
    const bool in = (uint32_t)lane < top;
    const uint32_t p = in ? top - 1u - lane : 0u;
    bool hit = false;
    if (in) hit = rect_covers(srect[p], tx, ty);
    const uint64_t mask = __ballot(hit);
    if (mask != 0ull) {
      if (hit) lds.queue[qn + (uint32_t)__popcll(mask & lt)] = p;
      qn += (uint32_t)__popcll(mask);
      wave_lds_sync();
      if (qn >= (uint32_t)kBatch) {
        process(kBatch);
        const uint32_t rem = qn - kBatch;
        uint32_t tmp = 0;
        if ((uint32_t)lane < rem) tmp = lds.queue[kBatch + lane];
        wave_lds_sync();
        if ((uint32_t)lane < rem) lds.queue[lane] = tmp;
        wave_lds_sync();
        qn = rem;
      }
    }
    top = top > (uint32_t)kWave ? top - kWave : 0u;
  }
  if (qn > 0 && c_run > 0) process(qn < c_run ? qn : c_run);
}

void launch_tiles_backward(const PsRasterDesc& d, const float* records,
                           const uint32_t* sorted_idx, const uint2* sorted_rect,
                           const float* view_params, const float* final_T,
                           const uint32_t* n_contrib, const uint32_t* tile_end,
                           const float* dL_dcolor, float* grad2d, hipStream_t st) {
  const Dims m = make_dims(d);
  const int total = m.V * m.tiles;
  dim3 grid((total + kWavesPerBlock - 1) / kWavesPerBlock), block(kWavesPerBlock * kWave);
  hipLaunchKernelGGL(tiles_backward_kernel, grid, block, 0, st, d, records, sorted_idx,
                     sorted_rect, view_params, final_T, n_contrib, tile_end, dL_dcolor, grad2d);
}

}  // namespace ps
