// Preprocess backward: screen-space gradients (dxy in NDC units, dconic, dopacity, drgb
// accumulated per (view, Gaussian) by the tile backward) -> dL/d{means, cov, SH|colors,
// opacity}.  Two kernels, both one thread per scene Gaussian looping over the scene's
// views and summing in registers: each output is written once, without atomics, and the
// v-fold `repeat` backward of decoder_splatting_cuda.py:53-56 never materialises.
//
//   geometry  EWA covariance + projection backward -> dL/dmeans (part 1), dL/dcov,
//             dL/dopacity, dL/dmeans2D, dL/dcolors.  Small register footprint.
//   colour    SH backward, one wave per 64 Gaussians: coefficients staged in LDS with
//             coalesced 16-byte loads, dL/dSH (75 accumulators per lane) leaves through the
//             same slab with coalesced stores; adds the view-direction term to dL/dmeans.
//
// Semantics: SURVEY.md Appendix A.4 (upstream quirks kept: the 2-D mean gradient is in NDC
// units, the frustum-guard clamp zeroes d/dt.x, d/dt.y, the off-diagonal conic gradient is
// stored once).  Replaces computeCov2D/preprocess backward of the external rasterizer
// reached from /root/reference/src/model/decoder/cuda_splatting.py:117-124 via autograd.
#include "raster_common.h"
#include "sh_math.h"

namespace ps {

__global__ void __launch_bounds__(256)
geometry_backward_kernel(PsRasterDesc d, const float* __restrict__ means,
                         const float* __restrict__ cov, const float* __restrict__ view_params,
                         const int32_t* __restrict__ radii, const uint2* __restrict__ rects,
                         const float* __restrict__ tile_grads,
                         const uint8_t* __restrict__ clamp_bits, float* __restrict__ color_grads,
                         const float* __restrict__ grad2d,
                         float* __restrict__ dL_dmeans, float* __restrict__ dL_dcov,
                         float* __restrict__ dL_dcolors, float* __restrict__ dL_dopacity,
                         float* __restrict__ dL_dmeans2D) {
  const int G = d.n_gaussians, vps = d.views_per_scene, H = d.height, W = d.width;
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int s = blockIdx.y;
  if (g >= G) return;
  const size_t sg = (size_t)s * G + g;

  float m0[3], c6[6];
  {
    const float* mp = means + sg * 3;
    m0[0] = mp[0]; m0[1] = mp[1]; m0[2] = mp[2];
    if (d.cov_layout == PS_COV_6) {
      const float* cp = cov + sg * 6;
#pragma unroll
      for (int k = 0; k < 6; ++k) c6[k] = cp[k];
    } else {
      const float* cp = cov + sg * 9;
      c6[0] = cp[0]; c6[1] = cp[1]; c6[2] = cp[2]; c6[3] = cp[4]; c6[4] = cp[5]; c6[5] = cp[8];
    }
  }
  float gm[3] = {0.f, 0.f, 0.f}, gc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, gop = 0.f;

  // radius, rect and clamp bits of four views at a time, branch free (clamped view index): read
  // one after the other inside the view loop (radius -> rect -> slots -> clamp bits) they were
  // four dependent memory round trips per view, sixteen per thread at four views
  const uint8_t* cbp = color_grads ? clamp_bits : reinterpret_cast<const uint8_t*>(radii);
  for (int j0 = 0; j0 < vps; j0 += 4) {
   int32_t rad4[4];
   uint2 rect4[4];
   uint32_t cb4[4];
#pragma unroll
   for (int u = 0; u < 4; ++u) {
     const size_t vgc = (size_t)(s * vps + min(j0 + u, vps - 1)) * G + g;
     rad4[u] = radii[vgc];
     rect4[u] = rects[vgc];
     cb4[u] = cbp[vgc];
   }
#pragma unroll
   for (int u = 0; u < 4; ++u) {
    const int j = j0 + u;
    if (j >= vps) break;
    const int v = s * vps + j;
    const size_t vg = (size_t)v * G + g;
    const bool vis = rad4[u] > 0;
    float gr[kGradFloats];
#pragma unroll
    for (int c = 0; c < kGradFloats; ++c) gr[c] = 0.f;
    if (vis) {
      // Gaussians touching <= 4 tiles: sum their private (Gaussian, tile) slots in tile order
      // (deterministic); larger ones were accumulated with atomics into grad2d
      const uint2 r = rect4[u];
      const uint32_t area = ((r.y & 0xFFFFu) - (r.x & 0xFFFFu)) * ((r.y >> 16) - (r.x >> 16));
      if (area <= (uint32_t)kInvSlots && (H + kTile - 1) / kTile <= 16383) {
        // its private slots (one per tile of the rect, row-major; the tile backward wrote every
        // one of them): all loads are issued before the first add, the sum keeps the tile order
        float4 a0[kInvSlots], a1[kInvSlots];
        float a2[kInvSlots];
        const float* sp0 = tile_grads + vg * (size_t)(kInvSlots * kSlotFloats);
#pragma unroll
        for (int k = 0; k < kInvSlots; ++k) {
          // unconditional loads from a clamped slot, masked when summed: `on ? tg[0] : 0` is a
          // load the compiler may not speculate, so it became a branch per slot with a wait at
          // every join -- the "all loads first" of the comment above did not survive
          const float* sp = sp0 + ((uint32_t)k < area ? k : 0) * kSlotFloats;
          const float4* tg = reinterpret_cast<const float4*>(sp);
          a0[k] = tg[0];
          a1[k] = tg[1];
          a2[k] = sp[8];
        }
#pragma unroll
        for (int k = 0; k < kInvSlots; ++k) {
          if ((uint32_t)k < area) {
            gr[0] += a0[k].x; gr[1] += a0[k].y; gr[2] += a0[k].z; gr[3] += a0[k].w;
            gr[4] += a1[k].x; gr[5] += a1[k].y; gr[6] += a1[k].z; gr[7] += a1[k].w;
            gr[8] += a2[k];
          }
        }
      } else {
        const float* gi = grad2d + vg * kGradFloats;
#pragma unroll
        for (int c = 0; c < kGradFloats; ++c) gr[c] = gi[c];
      }
    }
    if (dL_dmeans2D) {
      float* o = dL_dmeans2D + vg * 3;
      o[0] = vis ? gr[0] : 0.f;
      o[1] = vis ? gr[1] : 0.f;
      o[2] = 0.f;
    }
    if (dL_dcolors) {
      float* o = dL_dcolors + vg * 3;
      o[0] = vis ? gr[6] : 0.f;
      o[1] = vis ? gr[7] : 0.f;
      o[2] = vis ? gr[8] : 0.f;
    }
    if (!vis) continue;
    if (color_grads) {   // SH path: dL/dRGB with the forward's clamp applied, 12-byte rows
      const uint32_t cb = cb4[u];
      float* o = color_grads + vg * 3;
      o[0] = (cb & 1u) ? 0.f : gr[6];
      o[1] = (cb & 2u) ? 0.f : gr[7];
      o[2] = (cb & 4u) ? 0.f : gr[8];
    }
    const float* vp = view_params + (size_t)v * PS_VIEW_STRIDE;
    const float* V = vp + PS_VIEW_VIEWMATRIX;
    const float* PV = vp + PS_VIEW_PROJMATRIX;
    const float tanfovx = vp[PS_VIEW_TANFOVX], tanfovy = vp[PS_VIEW_TANFOVY];
    const float scale = vp[PS_VIEW_SCALE], scale2 = scale * scale;
    const float gx2 = gr[0], gy2 = gr[1], gcx = gr[2], gcy = gr[3], gcz = gr[4];
    const float g_op = gr[5];

    const float mx = m0[0] * scale, my = m0[1] * scale, mz = m0[2] * scale;
    const float tvx = V[0] * mx + V[4] * my + V[8] * mz + V[12];
    const float tvy = V[1] * mx + V[5] * my + V[9] * mz + V[13];
    const float tvz = V[2] * mx + V[6] * my + V[10] * mz + V[14];
    const float fx = (float)W / (2.0f * tanfovx), fy = (float)H / (2.0f * tanfovy);
    const float limx = d.guard * tanfovx, limy = d.guard * tanfovy;
    const float txtz = tvx / tvz, tytz = tvy / tvz;
    const float tx = fminf(limx, fmaxf(-limx, txtz)) * tvz;
    const float ty = fminf(limy, fmaxf(-limy, tytz)) * tvz;
    const float xgm = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    const float ygm = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    const float J00 = fx / tvz, J02 = -(fx * tx) / (tvz * tvz);
    const float J11 = fy / tvz, J12 = -(fy * ty) / (tvz * tvz);
    const float R00 = V[0], R01 = V[4], R02 = V[8], R10 = V[1], R11 = V[5], R12 = V[9],
                R20 = V[2], R21 = V[6], R22 = V[10];
    const float M00 = J00 * R00 + J02 * R20, M01 = J00 * R01 + J02 * R21,
                M02 = J00 * R02 + J02 * R22;
    const float M10 = J11 * R10 + J12 * R20, M11 = J11 * R11 + J12 * R21,
                M12 = J11 * R12 + J12 * R22;
    const float S00 = c6[0] * scale2, S01 = c6[1] * scale2, S02 = c6[2] * scale2,
                S11 = c6[3] * scale2, S12 = c6[4] * scale2, S22 = c6[5] * scale2;
    const float a0 = S00 * M00 + S01 * M01 + S02 * M02;
    const float a1 = S01 * M00 + S11 * M01 + S12 * M02;
    const float a2 = S02 * M00 + S12 * M01 + S22 * M02;
    const float b0 = S00 * M10 + S01 * M11 + S02 * M12;
    const float b1 = S01 * M10 + S11 * M11 + S12 * M12;
    const float b2 = S02 * M10 + S12 * M11 + S22 * M12;
    const float a = (M00 * a0 + M01 * a1 + M02 * a2) + d.lowpass;
    const float b = M10 * a0 + M11 * a1 + M12 * a2;
    const float c = (M10 * b0 + M11 * b1 + M12 * b2) + d.lowpass;
    const float denom = a * c - b * b;
    const float denom2inv = 1.0f / (denom * denom + d.det2_eps);
    float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
    if (denom2inv != 0.f) {
      dL_da = denom2inv * (-c * c * gcx + 2.f * b * c * gcy + (denom - a * c) * gcz);
      dL_dc = denom2inv * (-a * a * gcz + 2.f * a * b * gcy + (denom - a * c) * gcx);
      dL_db = denom2inv * 2.f * (b * c * gcx - (denom + 2.f * b * b) * gcy + a * b * gcz);
      gc[0] += scale2 * (M00 * M00 * dL_da + M00 * M10 * dL_db + M10 * M10 * dL_dc);
      gc[3] += scale2 * (M01 * M01 * dL_da + M01 * M11 * dL_db + M11 * M11 * dL_dc);
      gc[5] += scale2 * (M02 * M02 * dL_da + M02 * M12 * dL_db + M12 * M12 * dL_dc);
      gc[1] += scale2 * (2.f * M00 * M01 * dL_da + (M00 * M11 + M01 * M10) * dL_db +
                         2.f * M10 * M11 * dL_dc);
      gc[2] += scale2 * (2.f * M00 * M02 * dL_da + (M00 * M12 + M02 * M10) * dL_db +
                         2.f * M10 * M12 * dL_dc);
      gc[4] += scale2 * (2.f * M01 * M02 * dL_da + (M01 * M12 + M02 * M11) * dL_db +
                         2.f * M11 * M12 * dL_dc);
    }
    const float dM00 = 2.f * a0 * dL_da + b0 * dL_db, dM01 = 2.f * a1 * dL_da + b1 * dL_db,
                dM02 = 2.f * a2 * dL_da + b2 * dL_db;
    const float dM10 = 2.f * b0 * dL_dc + a0 * dL_db, dM11 = 2.f * b1 * dL_dc + a1 * dL_db,
                dM12 = 2.f * b2 * dL_dc + a2 * dL_db;
    const float dJ00 = R00 * dM00 + R01 * dM01 + R02 * dM02;
    const float dJ02 = R20 * dM00 + R21 * dM01 + R22 * dM02;
    const float dJ11 = R10 * dM10 + R11 * dM11 + R12 * dM12;
    const float dJ12 = R20 * dM10 + R21 * dM11 + R22 * dM12;
    const float tz = 1.0f / tvz, tz2 = tz * tz, tz3 = tz2 * tz;
    const float dtx = xgm * -fx * tz2 * dJ02;
    const float dty = ygm * -fy * tz2 * dJ12;
    const float dtz = -fx * tz2 * dJ00 - fy * tz2 * dJ11 + (2.f * fx * tx) * tz3 * dJ02 +
                      (2.f * fy * ty) * tz3 * dJ12;
    float gmx = R00 * dtx + R10 * dty + R20 * dtz;
    float gmy = R01 * dtx + R11 * dty + R21 * dtz;
    float gmz = R02 * dtx + R12 * dty + R22 * dtz;
    {
      const float hx = PV[0] * mx + PV[4] * my + PV[8] * mz + PV[12];
      const float hy = PV[1] * mx + PV[5] * my + PV[9] * mz + PV[13];
      const float hw = PV[3] * mx + PV[7] * my + PV[11] * mz + PV[15];
      const float m_w = 1.0f / (hw + d.w_eps);
      const float mul1 = hx * m_w * m_w, mul2 = hy * m_w * m_w;
      gmx += (PV[0] * m_w - PV[3] * mul1) * gx2 + (PV[1] * m_w - PV[3] * mul2) * gy2;
      gmy += (PV[4] * m_w - PV[7] * mul1) * gx2 + (PV[5] * m_w - PV[7] * mul2) * gy2;
      gmz += (PV[8] * m_w - PV[11] * mul1) * gx2 + (PV[9] * m_w - PV[11] * mul2) * gy2;
    }
    gm[0] += scale * gmx; gm[1] += scale * gmy; gm[2] += scale * gmz;
    gop += g_op;
   }
  }

  float* om = dL_dmeans + sg * 3;
  om[0] = gm[0]; om[1] = gm[1]; om[2] = gm[2];
  if (d.cov_layout == PS_COV_6) {
    float* oc = dL_dcov + sg * 6;
#pragma unroll
    for (int k = 0; k < 6; ++k) oc[k] = gc[k];
  } else {
    float* oc = dL_dcov + sg * 9;
    oc[0] = gc[0]; oc[1] = gc[1]; oc[2] = gc[2]; oc[3] = 0.f; oc[4] = gc[3]; oc[5] = gc[4];
    oc[6] = 0.f; oc[7] = 0.f; oc[8] = gc[5];
  }
  dL_dopacity[sg] = gop;
}

// SH backward (runs after geometry_backward_kernel: it ADDS its dL/dmeans term).
// 75 gradient accumulators + 25 basis values + 25 weights per lane: the kernel sits at the
// 256-VGPR boundary (250 before the compact colour gradients, 258 after -- one wave per SIMD
// instead of two, 479 -> 600 us); asking for three waves per SIMD makes the compiler settle at 242 VGPRs, no scratch (two waves)
template <int DEG, bool LDS_SH>
__global__ void __launch_bounds__(kWave) __attribute__((amdgpu_waves_per_eu(3)))
color_backward_kernel(PsRasterDesc d, const float* __restrict__ means,
                      const float* __restrict__ sh, const float* __restrict__ view_params,
                      const int32_t* __restrict__ radii, const float* __restrict__ color_grads,
                      float* __restrict__ dL_dmeans, float* __restrict__ dL_dsh) {
  constexpr int NB = (DEG + 1) * (DEG + 1);
  const int G = d.n_gaussians, vps = d.views_per_scene, K = d.sh_coeffs;
  const int lane = threadIdx.x;
  const int g = blockIdx.x * kWave + lane;
  const int s = blockIdx.y;
  const bool active = g < G;
  const size_t sg = (size_t)s * G + (active ? g : 0);
  const int S3 = K * 3;
  const bool gk3 = d.sh_layout == PS_SH_GK3;

  __shared__ __attribute__((aligned(16))) float slab[LDS_SH ? kWave * 75 : 4];
  const size_t g0 = (size_t)s * G + (size_t)blockIdx.x * kWave;
  const int rem = G - (int)(blockIdx.x * kWave);
  const int nflt = (rem < kWave ? rem : kWave) * S3;
  float* my_slab = slab + (LDS_SH ? lane * S3 : 0);
  const float* my_sh;
  if (LDS_SH) {
    stage_slab<(kWave * 75 + 255) / 256>(sh + g0 * (size_t)S3, slab, nflt, lane);
    __syncthreads();
    my_sh = my_slab;
  } else {
    my_sh = sh + sg * (size_t)S3;
  }

  const float* mp = means + sg * 3;
  const float m0x = mp[0], m0y = mp[1], m0z = mp[2];
  float dsh[NB * 3];
#pragma unroll
  for (int i = 0; i < NB * 3; ++i) dsh[i] = 0.f;
  float gmx = 0.f, gmy = 0.f, gmz = 0.f;

  for (int j = 0; j < vps; ++j) {
    const int v = s * vps + j;
    const size_t vg = (size_t)v * G + (active ? g : 0);
    if (!(active && radii[vg] > 0)) continue;
    const float* vp = view_params + (size_t)v * PS_VIEW_STRIDE;
    const float scale = vp[PS_VIEW_SCALE];
    const float* cam = vp + PS_VIEW_CAMPOS;
    const float* gr = color_grads + vg * 3;   // clamp already applied (geometry backward)
    const float gch[3] = {gr[0], gr[1], gr[2]};
    const float ox = m0x * scale - cam[0], oy = m0y * scale - cam[1], oz = m0z * scale - cam[2];
    const float inv = 1.0f / sqrtf(ox * ox + oy * oy + oz * oz);
    const float x = ox * inv, y = oy * inv, z = oz * inv;
    float bas[25], w[25];
    sh_basis(DEG, x, y, z, bas);
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      float wk = 0.f;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        dsh[k * 3 + ch] = fmaf(bas[k], gch[ch], dsh[k * 3 + ch]);
        wk = fmaf(my_sh[gk3 ? k * 3 + ch : ch * K + k], gch[ch], wk);
      }
      w[k] = wk;
    }
    float ddx, ddy, ddz;
    sh_grad_dot(DEG, x, y, z, w, ddx, ddy, ddz);
    const float dot = x * ddx + y * ddy + z * ddz;
    gmx += scale * ((ddx - x * dot) * inv);
    gmy += scale * ((ddy - y * dot) * inv);
    gmz += scale * ((ddz - z * dot) * inv);
  }
  if (active) {
    float* om = dL_dmeans + sg * 3;
    om[0] += gmx; om[1] += gmy; om[2] += gmz;
  }
  if (LDS_SH) {
    // dL/dSH leaves through the slab: every lane drops its 3K values, then the wave streams
    // the 64 x 3K block out with coalesced 16-byte stores
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NB; ++k)
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) my_slab[gk3 ? k * 3 + ch : ch * K + k] = dsh[k * 3 + ch];
    for (int k = NB; k < K; ++k)
      for (int ch = 0; ch < 3; ++ch) my_slab[gk3 ? k * 3 + ch : ch * K + k] = 0.f;
    __syncthreads();
    float* dst = dL_dsh + g0 * (size_t)S3;
    if ((reinterpret_cast<size_t>(dst) & 15) == 0) {
      float4* dst4 = reinterpret_cast<float4*>(dst);
      for (int i = lane; i * 4 < nflt; i += kWave) {
        if (i * 4 + 3 < nflt)
          dst4[i] = make_float4(slab[i * 4], slab[i * 4 + 1], slab[i * 4 + 2], slab[i * 4 + 3]);
        else
          for (int e = i * 4; e < nflt; ++e) dst[e] = slab[e];
      }
    } else {
      for (int i = lane; i < nflt; i += kWave) dst[i] = slab[i];
    }
  } else if (active) {
    float* os = dL_dsh + sg * (size_t)S3;
#pragma unroll
    for (int k = 0; k < NB; ++k)
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) os[gk3 ? k * 3 + ch : ch * K + k] = dsh[k * 3 + ch];
    for (int k = NB; k < K; ++k)
      for (int ch = 0; ch < 3; ++ch) os[gk3 ? k * 3 + ch : ch * K + k] = 0.f;
  }
}

// Rows of grad2d the tile backward accumulates into with atomics: visible pairs whose rect covers
// more than kInvSlots tiles -- the predicate of packed_small_rect (raster_preprocess.hip) and of the
// slot branch of geometry_backward_kernel above, which reads exactly these rows.  Clearing only
// them replaces a memset of the whole [V, G, 9] array: 396 MB at configs[1], a 0.16 - 0.19 ms fill
// kernel that took 0.11 ms out of the step even on a second stream under the forward.
// Round 6: driven by the view's DEPTH-SORTED arrays (sorted_rect / sorted_idx hold the visible pairs only, densely:
// 12 bytes per visible pair, coalesced) instead of radii + rects over all (view, Gaussian) pairs (4 bytes for every
// pair and a dependent, scattered 8-byte rect for the 40 % that are visible): blocks beyond a view's n_vis leave at once.
__global__ void __launch_bounds__(256)
clear_atomic_rows_kernel(int G, int tiles_y, const uint2* __restrict__ sorted_rect,
                         const uint32_t* __restrict__ sorted_idx, const uint32_t* __restrict__ n_vis,
                         float* __restrict__ grad2d) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const size_t vo = (size_t)blockIdx.y * G;
  if (i >= n_vis[blockIdx.y]) return;
  const uint2 r = sorted_rect[vo + i];
  const uint32_t id = sorted_idx[vo + i];
  const uint32_t area = ((r.y & 0xFFFFu) - (r.x & 0xFFFFu)) * ((r.y >> 16) - (r.x >> 16));
  if (area <= (uint32_t)kInvSlots && tiles_y <= 16383) return;
  float* g = grad2d + (vo + id) * kGradFloats;
#pragma unroll
  for (int c = 0; c < kGradFloats; ++c) g[c] = 0.f;
}

void launch_clear_atomic_rows(const PsRasterDesc& d, const uint2* sorted_rect, const uint32_t* sorted_idx,
                              const uint32_t* n_vis, float* grad2d, hipStream_t st) {
  const int V = d.n_scenes * d.views_per_scene;
  if (V == 0 || d.n_gaussians == 0) return;
  hipLaunchKernelGGL(clear_atomic_rows_kernel, dim3((unsigned)((d.n_gaussians + 255) / 256), (unsigned)V), dim3(256),
                     0, st, d.n_gaussians, (d.height + kTile - 1) / kTile, sorted_rect, sorted_idx, n_vis, grad2d);
}

// ---- PS_FLAG_DETERMINISTIC -------------------------------------------------------------------------
// rank_of[v][id] = position of the Gaussian in its view's (depth, id) order: the order of every tile list
__global__ void __launch_bounds__(256)
det_rank_kernel(int G, const uint32_t* __restrict__ sorted_idx, const uint32_t* __restrict__ n_vis,
                uint32_t* __restrict__ rank_of) {
  const uint32_t r = blockIdx.x * 256u + threadIdx.x;
  const size_t vo = (size_t)blockIdx.y * G;
  if (r < n_vis[blockIdx.y]) rank_of[vo + sorted_idx[vo + r]] = r;
}

// One thread per visible (view, Gaussian) pair over more than kInvSlots tiles: the tiles of its rect in
// row-major order; in each, its entry is found by bisection on the ranks (a tile list is sorted by rank)
// and the slot the tile backward left there is added.  Fixed order, no atomics: bitwise reproducible.
__global__ void __launch_bounds__(256)
det_reduce_kernel(size_t n, int G, int gx, int gy, const int32_t* __restrict__ radii,
                  const uint2* __restrict__ rects, const uint32_t* __restrict__ rank_of,
                  const uint32_t* __restrict__ tile_ranges, const uint32_t* __restrict__ point_list,
                  uint32_t capacity, const float4* __restrict__ det_slots, float* __restrict__ grad2d) {
  const size_t vg = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (vg >= n) return;
  if (radii[vg] <= 0) return;
  const uint2 r = rects[vg];
  const uint32_t xmin = r.x & 0xFFFFu, ymin = r.x >> 16, xmax = r.y & 0xFFFFu, ymax = r.y >> 16;
  if ((xmax - xmin) * (ymax - ymin) <= (uint32_t)kInvSlots && gy <= 16383) return;   // private slots
  const size_t v = vg / (size_t)G, vo = v * (size_t)G;
  const uint32_t id = (uint32_t)(vg - vo), my = rank_of[vg];
  float acc[kGradFloats];
#pragma unroll
  for (int c = 0; c < kGradFloats; ++c) acc[c] = 0.f;
  for (uint32_t ty = ymin; ty < ymax; ++ty)
    for (uint32_t tx = xmin; tx < xmax; ++tx) {
      const size_t tile = v * (size_t)(gx * gy) + (size_t)ty * gx + tx;
      uint32_t start = tile_ranges[2 * tile], cnt = tile_ranges[2 * tile + 1];
      if (start > capacity) start = capacity;
      if (cnt > capacity - start) cnt = capacity - start;
      const uint32_t* list = point_list + start;
      uint32_t lo = 0, hi = cnt;
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (rank_of[vo + list[mid]] < my) lo = mid + 1; else hi = mid;
      }
      if (lo < cnt && list[lo] == id) {
        const float4* s = det_slots + ((size_t)start + lo) * (kSlotFloats / 4);
        const float4 a = s[0], b = s[1], c = s[2];
        acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
        acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w; acc[8] += c.x;
      }
    }
  float* g = grad2d + vg * kGradFloats;
#pragma unroll
  for (int c = 0; c < kGradFloats; ++c) g[c] = acc[c];
}

// the per-entry slots start from zero (a kernel, not hipMemsetAsync: the one memset node this library would
// put into a captured step gave wrong gradients when the step was replayed from a hipGraph -- bench.py's
// step_check caught it, round 5 -- while every kernel launch replays correctly)
__global__ void __launch_bounds__(256)
det_clear_kernel(size_t n4, float4* __restrict__ p) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
    p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
void launch_deterministic_clear(float* det_slots, size_t entries, hipStream_t st) {
  const size_t n4 = entries * (kSlotFloats / 4);
  if (n4 == 0) return;
  const size_t blocks = (n4 + 255) / 256;
  hipLaunchKernelGGL(det_clear_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st, n4,
                     reinterpret_cast<float4*>(det_slots));
}

void launch_deterministic_reduce(const PsRasterDesc& d, const int32_t* radii, const uint2* rects,
                                 const uint32_t* sorted_idx, const uint32_t* n_vis,
                                 const uint32_t* tile_ranges, const uint32_t* point_list,
                                 uint32_t capacity, const float* det_slots, uint32_t* rank_of,
                                 float* grad2d, hipStream_t st) {
  const Dims m = make_dims(d);
  if (m.N == 0) return;
  hipLaunchKernelGGL(det_rank_kernel, dim3((unsigned)((m.G + 255) / 256), (unsigned)m.V), dim3(256), 0, st,
                     m.G, sorted_idx, n_vis, rank_of);
  hipLaunchKernelGGL(det_reduce_kernel, dim3((unsigned)((m.N + 255) / 256)), dim3(256), 0, st, m.N, m.G,
                     m.gx, m.gy, radii, rects, rank_of, tile_ranges, point_list, capacity,
                     reinterpret_cast<const float4*>(det_slots), grad2d);
}

void launch_preprocess_backward(const PsRasterDesc& d, const float* means, const float* cov,
                                const float* sh, const float* view_params, const float* records,
                                const int32_t* radii, const uint2* rects,
                                const float* tile_grads, const uint8_t* clamp_bits,
                                float* color_grads,
                                float* grad2d, float* dL_dmeans,
                                float* dL_dcov, float* dL_dsh, float* dL_dcolors,
                                float* dL_dopacity, float* dL_dmeans2D, hipStream_t st) {
  {
    dim3 grid((d.n_gaussians + 255) / 256, d.n_scenes), block(256);
    hipLaunchKernelGGL(geometry_backward_kernel, grid, block, 0, st, d, means, cov, view_params,
                       radii, rects, tile_grads, clamp_bits,
                       sh ? color_grads : (float*)nullptr, grad2d, dL_dmeans, dL_dcov,
                       sh ? (float*)nullptr : dL_dcolors, dL_dopacity, dL_dmeans2D);
  }
  if (!sh) return;
  const int deg = d.sh_degree;
  const bool lds = ((d.sh_coeffs * 3) & 1) && d.sh_coeffs * 3 <= 75;
  dim3 grid((d.n_gaussians + kWave - 1) / kWave, d.n_scenes), block(kWave);
#define PS_LAUNCH(DEG)                                                                         \
  do {                                                                                         \
    if (lds)                                                                                   \
      hipLaunchKernelGGL((color_backward_kernel<DEG, true>), grid, block, 0, st, d, means, sh, \
                         view_params, radii, color_grads, dL_dmeans, dL_dsh);                  \
    else                                                                                       \
      hipLaunchKernelGGL((color_backward_kernel<DEG, false>), grid, block, 0, st, d, means,    \
                         sh, view_params, radii, color_grads, dL_dmeans, dL_dsh);              \
  } while (0)
  switch (deg) {
    case 0: PS_LAUNCH(0); break;
    case 1: PS_LAUNCH(1); break;
    case 2: PS_LAUNCH(2); break;
    case 3: PS_LAUNCH(3); break;
    default: PS_LAUNCH(4); break;
  }
#undef PS_LAUNCH
}

}  // namespace ps
