// Per-Gaussian preprocess (forward):
//
//   fused     (default, <= 4 views per scene) one lane per (view, Gaussian) pair, a wave per 16
//             Gaussians: geometry, then -- if any lane of the wave survived the cull -- the SH slab
//             through LDS and the colour; the 48-byte record is written once.  See the comment at
//             preprocess_fused_kernel.
//   geometry  one thread per scene Gaussian, looping over the scene's views: frustum cull,
//             projection, EWA 2-D covariance, conic, 3-sigma radius, tile rect, depth key.
//             {mean, cov, opacity} are read once per scene, not once per view (the reference
//             materialises v copies: decoder_splatting_cuda.py:53-56).
//   colour    one wave per 32 Gaussians: the wave stages their SH coefficients (32 x 3K
//             contiguous floats) in LDS with coalesced 16-byte loads -- 300 of the 340 input
//             bytes per Gaussian -- then evaluates the degree<=4 colour for every view in
//             which the Gaussian survived the cull.  Lane stride 3K is odd => conflict-free.
//             (geometry + colour: the path for colors_precomp, deferred colours, > 4 views.)
//
// Semantics: SURVEY.md Appendix A.1 -- what `GaussianRasterizer.forward` computes per
// Gaussian for the call at /root/reference/src/model/decoder/cuda_splatting.py:117-124.
//
// THIS TRANSLATION UNIT IS BUILT WITH -ffp-contract=off: radius, tile rect and the depth
// sort key are integer-valued functions of this arithmetic and must be bit-exact.
//
// Record (12 floats per (view, Gaussian), written only when visible):
//   [0..3] px, py, conic.x, conic.y   [4..7] conic.z, opacity, depth, packed small-rect origin
//          (kSmallFlag | width-1 | ymin | xmin for rects of <= 4 tiles, else 0)
//   [8..11] r, g, b, clamp bits
#include "raster_common.h"
#include "cell_window.h"
#include "sh_math.h"

#include <cstdlib>

namespace ps {

__device__ __forceinline__ int sat_int(float v) {  // trunc, saturating, NaN -> -2^30
  if (!(v > -1073741824.0f)) return -1073741824;
  if (v > 1073741824.0f) return 1073741824;
  return (int)v;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) {
  return v < lo ? lo : (v > hi ? hi : v);
}

// Geometry of ONE (view, Gaussian) pair: frustum cull, projection, EWA 2-D covariance, conic,
// 3-sigma radius, tile rect.  m0 / c6 are the scene-space mean and covariance (upper triangle), vp the
// view's parameter block (global or LDS).  Shared by the two kernels below so that the arithmetic --
// and with it radius, rect and depth-key bits -- is the same whichever of them ran.
struct GeoOut {
  bool vis;
  int radius, xmin, ymin, xmax, ymax;
  float px, py, con_x, con_y, con_z, tvz;
};
__device__ __forceinline__ GeoOut project_gaussian(const PsRasterDesc& d, const float* m0,
                                                   const float* c6, const float* vp, int gx,
                                                   int gy) {
  const int H = d.height, W = d.width;
  const float* V = vp + PS_VIEW_VIEWMATRIX;
  const float* PV = vp + PS_VIEW_PROJMATRIX;
  const float tanfovx = vp[PS_VIEW_TANFOVX], tanfovy = vp[PS_VIEW_TANFOVY];
  const float scale = vp[PS_VIEW_SCALE];
  const float scale2 = scale * scale;
  GeoOut o;
  const float mx = m0[0] * scale, my = m0[1] * scale, mz = m0[2] * scale;
  const float tvx = V[0] * mx + V[4] * my + V[8] * mz + V[12];
  const float tvy = V[1] * mx + V[5] * my + V[9] * mz + V[13];
  const float tvz = V[2] * mx + V[6] * my + V[10] * mz + V[14];
  // Branch free: every lane runs the whole chain and the verdict is formed at the end (the values
  // of a culled pair are never stored).  With the cull as a branch the compiler sinks the
  // covariance loads into it -- a second dependent memory round trip per wave.
  const float hx = PV[0] * mx + PV[4] * my + PV[8] * mz + PV[12];
  const float hy = PV[1] * mx + PV[5] * my + PV[9] * mz + PV[13];
  const float hw = PV[3] * mx + PV[7] * my + PV[11] * mz + PV[15];
  const float pw = 1.0f / (hw + d.w_eps);
  const float ppx = hx * pw, ppy = hy * pw;

  const float fx = (float)W / (2.0f * tanfovx), fy = (float)H / (2.0f * tanfovy);
  const float limx = d.guard * tanfovx, limy = d.guard * tanfovy;
  const float txtz = tvx / tvz, tytz = tvy / tvz;
  const float tx = fminf(limx, fmaxf(-limx, txtz)) * tvz;
  const float ty = fminf(limy, fmaxf(-limy, tytz)) * tvz;
  const float J00 = fx / tvz, J02 = -(fx * tx) / (tvz * tvz);
  const float J11 = fy / tvz, J12 = -(fy * ty) / (tvz * tvz);
  const float M00 = J00 * V[0] + J02 * V[2], M01 = J00 * V[4] + J02 * V[6],
              M02 = J00 * V[8] + J02 * V[10];
  const float M10 = J11 * V[1] + J12 * V[2], M11 = J11 * V[5] + J12 * V[6],
              M12 = J11 * V[9] + J12 * V[10];
  const float S00 = c6[0] * scale2, S01 = c6[1] * scale2, S02 = c6[2] * scale2,
              S11 = c6[3] * scale2, S12 = c6[4] * scale2, S22 = c6[5] * scale2;
  const float a0 = S00 * M00 + S01 * M01 + S02 * M02;
  const float a1 = S01 * M00 + S11 * M01 + S12 * M02;
  const float a2 = S02 * M00 + S12 * M01 + S22 * M02;
  const float b0 = S00 * M10 + S01 * M11 + S02 * M12;
  const float b1 = S01 * M10 + S11 * M11 + S12 * M12;
  const float b2 = S02 * M10 + S12 * M11 + S22 * M12;
  const float c00 = (M00 * a0 + M01 * a1 + M02 * a2) + d.lowpass;
  const float c01 = M10 * a0 + M11 * a1 + M12 * a2;
  const float c11 = (M10 * b0 + M11 * b1 + M12 * b2) + d.lowpass;
  const float det = c00 * c11 - c01 * c01;
  const float det_inv = 1.0f / det;
  const float con_x = c11 * det_inv, con_y = -c01 * det_inv, con_z = c00 * det_inv;
  const float mid = 0.5f * (c00 + c11);
  const float sq = sqrtf(fmaxf(d.lambda_floor, mid * mid - det));
  const float l1 = mid + sq, l2 = mid - sq;
  const int radius = sat_int(ceilf(3.0f * sqrtf(fmaxf(l1, l2))));
  const float px = ((ppx + 1.0f) * (float)W - 1.0f) * 0.5f;
  const float py = ((ppy + 1.0f) * (float)H - 1.0f) * 0.5f;
  const float rf = (float)radius;
  const int xmin = clampi(sat_int((px - rf) / 16.0f), 0, gx);
  const int ymin = clampi(sat_int((py - rf) / 16.0f), 0, gy);
  const int xmax = clampi(sat_int((px + rf + 15.0f) / 16.0f), 0, gx);
  const int ymax = clampi(sat_int((py + rf + 15.0f) / 16.0f), 0, gy);
  const bool vis = (tvz > d.near_cull) & !(det == 0.0f || det != det) &
                   ((xmax - xmin) * (ymax - ymin) > 0);
  o.vis = vis; o.radius = radius; o.xmin = xmin; o.ymin = ymin; o.xmax = xmax; o.ymax = ymax;
  o.px = px; o.py = py; o.con_x = con_x; o.con_y = con_y; o.con_z = con_z; o.tvz = tvz;
  return o;
}

// records[vg][0..7] of a visible pair (the colour kernel / path fills [8..11])
__device__ __forceinline__ uint32_t packed_small_rect(const GeoOut& o, int gy) {
  const int rw = o.xmax - o.xmin, area = rw * (o.ymax - o.ymin);
  return (area <= kInvSlots && gy <= 16383)
      ? (kSmallFlag | ((uint32_t)(rw - 1) << 29) | ((uint32_t)o.ymin << 15) | (uint32_t)o.xmin) : 0u;
}

__device__ __forceinline__ void load_gaussian(const PsRasterDesc& d, const float* means,
                                              const float* cov, const float* opacity, size_t sg,
                                              float* m0, float* c6, float& opac) {
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
  opac = opacity[sg];
}

__global__ void __launch_bounds__(256)
geometry_forward_kernel(PsRasterDesc d, const float* __restrict__ means,
                        const float* __restrict__ cov, const float* __restrict__ colors,
                        const float* __restrict__ opacity, const float* __restrict__ view_params,
                        float* __restrict__ records, uint32_t* __restrict__ keys,
                        uint2* __restrict__ rects, int32_t* __restrict__ radii,
                        uint4* __restrict__ cell_windows) {
  const int G = d.n_gaussians, vps = d.views_per_scene, H = d.height, W = d.width;
  const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int s = blockIdx.y;
  if (g >= G) return;
  const size_t sg = (size_t)s * G + g;

  float m0[3], c6[6], opac;
  load_gaussian(d, means, cov, opacity, sg, m0, c6, opac);

  for (int j = 0; j < vps; ++j) {
    const int v = s * vps + j;
    const size_t vg = (size_t)v * G + g;
    const GeoOut o = project_gaussian(d, m0, c6, view_params + (size_t)v * PS_VIEW_STRIDE, gx, gy);
    radii[vg] = o.vis ? o.radius : 0;
    keys[vg] = o.vis ? __float_as_uint(o.tvz) : kCulledKey;
    if (o.vis) {
      rects[vg] = make_uint2((uint32_t)o.xmin | ((uint32_t)o.ymin << 16),
                             (uint32_t)o.xmax | ((uint32_t)o.ymax << 16));
      float4* r = reinterpret_cast<float4*>(records + vg * kRecFloats);
      r[0] = make_float4(o.px, o.py, o.con_x, o.con_y);
      r[1] = make_float4(o.con_z, opac, o.tvz, __uint_as_float(packed_small_rect(o, gy)));
      cell_windows[vg * (kRecFloats / 4)] = cell_window(o.px, o.py, o.con_x, o.con_y, o.con_z, opac, d.alpha_min);
      if (colors != nullptr) {   // colors_precomp: verbatim, no clamp
        const float* cp = colors + vg * 3;
        r[2] = make_float4(cp[0], cp[1], cp[2], __uint_as_float(0u));
      }
    }
  }
}

// SH colour of one (view, Gaussian) pair from coefficients at my_sh (global or LDS); returns the
// clamp bits (channels clamped at 0)
template <int DEG>
__device__ __forceinline__ uint32_t sh_color(const float* my_sh, bool gk3, int K, float m0x,
                                             float m0y, float m0z, const float* vp,
                                             float* rgb) {
  constexpr int NB = (DEG + 1) * (DEG + 1);
  const float scale = vp[PS_VIEW_SCALE];
  const float* cam = vp + PS_VIEW_CAMPOS;
  const float mx = m0x * scale, my = m0y * scale, mz = m0z * scale;
  float dx = mx - cam[0], dy = my - cam[1], dz = mz - cam[2];
  const float len = sqrtf(dx * dx + dy * dy + dz * dz);
  dx = dx / len; dy = dy / len; dz = dz / len;
  float b[25];
  sh_basis(DEG, dx, dy, dz, b);
  uint32_t clamp_bits = 0;
#pragma unroll 1
  for (int c = 0; c < 3; ++c) {      // not unrolled: 25 LDS operands live at a time, not 75
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < NB; ++k) acc = acc + b[k] * my_sh[gk3 ? k * 3 + c : c * K + k];
    acc = acc + 0.5f;
    if (acc < 0.f) clamp_bits |= (1u << c);
    rgb[c] = fmaxf(acc, 0.f);
  }
  return clamp_bits;
}

// SH -> RGB for the (view, Gaussian) pairs that survived the cull.
// LDS_SH: slab staged through LDS (3K odd and <= 75); otherwise direct per-lane loads.
// This kernel is a stream over the SH array (300 B per Gaussian, 0.83 GB at the paper
// config) and bandwidth follows occupancy: measured with tools/stream_microbench.hip, MI355X
// reads 6.1 TB/s at full occupancy but 3.8-4.5 TB/s when LDS limits a CU to 8 waves.  So a
// wave takes only 32 Gaussians (9.6 KB slab, 16 waves/CU) and uses its two half-waves for the
// even / odd views of the same Gaussians.
template <int DEG, bool LDS_SH>
__global__ void __launch_bounds__(kWave, LDS_SH ? 4 : 1)
color_forward_kernel(PsRasterDesc d, const float* __restrict__ means,
                     const float* __restrict__ sh, const float* __restrict__ view_params,
                     const int32_t* __restrict__ radii, float* __restrict__ records,
                     uint8_t* __restrict__ clamp_out) {
  constexpr int GPW = LDS_SH ? 32 : kWave;          // Gaussians per wave
  const int G = d.n_gaussians, vps = d.views_per_scene, K = d.sh_coeffs;
  const int lane = threadIdx.x;
  const int gl = lane % GPW, part = lane / GPW, parts = kWave / GPW;
  const int g = blockIdx.x * GPW + gl;
  const int s = blockIdx.y;
  const bool active = g < G;
  const size_t sg = (size_t)s * G + (active ? g : 0);
  const int S3 = K * 3;

  // which views need a colour at all?  (skip the slab load if nobody in the wave does)
  uint32_t vis_bits = 0;
  for (int j = 0; j < vps && j < 32; ++j)
    if (active && radii[(size_t)(s * vps + j) * G + g] > 0) vis_bits |= 1u << j;
  // views beyond the 32 tracked bits are re-tested in the loop below; only the wave-level
  // early-out must not fire on their behalf
  const bool any_vis = vps > 32 ? active : vis_bits != 0u;
  if (__ballot(any_vis) == 0ull) return;

  __shared__ __attribute__((aligned(16))) float slab[LDS_SH ? GPW * 75 + 4 : 4];
  const float* my_sh;
  if (LDS_SH) {
    const size_t g0 = (size_t)s * G + (size_t)blockIdx.x * GPW;
    const int rem = G - (int)(blockIdx.x * GPW);
    const int nflt = (rem < GPW ? rem : GPW) * S3;
    stage_slab<(GPW * 75 + 255) / 256>(sh + g0 * (size_t)S3, slab, nflt, lane);
    __syncthreads();
    my_sh = slab + gl * S3;
  } else {
    my_sh = sh + sg * (size_t)S3;
  }
  const bool gk3 = d.sh_layout == PS_SH_GK3;
  const float* mp = means + sg * 3;
  const float m0x = mp[0], m0y = mp[1], m0z = mp[2];

  for (int j = part; j < vps; j += parts) {
    const bool vis = j < 32 ? ((vis_bits >> j) & 1u) != 0u
                            : (active && radii[(size_t)(s * vps + j) * G + g] > 0);
    if (!vis) continue;
    const int v = s * vps + j;
    float rgb[3];
    const uint32_t clamp_bits = sh_color<DEG>(my_sh, gk3, K, m0x, m0y, m0z,
                                              view_params + (size_t)v * PS_VIEW_STRIDE, rgb);
    float4* r = reinterpret_cast<float4*>(records + ((size_t)v * G + g) * kRecFloats);
    r[2] = make_float4(rgb[0], rgb[1], rgb[2], __uint_as_float(clamp_bits));
    clamp_out[(size_t)v * G + g] = (uint8_t)clamp_bits;   // compact copy for the backward
  }
}

// Geometry AND SH colour in one kernel (the default when a scene has <= 4 views, which is every
// configuration of the reference: 4 target views, or 1 for the per-view drop-in call).  A wave
// takes 16 Gaussians; lane (j, i) = view j of Gaussian i, so the 4.8 KB SH slab serves the whole
// wave in ONE round and a lane holds one pair's geometry in registers until its colour is known:
//   * the 48-byte record of a visible pair is written once, in full (the two-kernel path writes
//     32 + 16 bytes from two launches: each a partial-line write the memory side has to merge --
//     measured 0.10 of the colour kernel's 0.33 ms at BASELINE configs[1]);
//   * radii / means are not re-read, the geometry arithmetic hides under the slab's latency;
//   * the scene's view blocks are staged in LDS once per wave (lanes of different views would
//     otherwise need ~40 vector loads each where the per-view kernel uses scalar loads).
// Same arithmetic as the two kernels above (project_gaussian / sh_color): identical bits.
constexpr int kFusedGpw = 16;
constexpr int kFusedViews = kWave / kFusedGpw;
template <int DEG>
__global__ void __launch_bounds__(kWave, 8)
preprocess_fused_kernel(PsRasterDesc d, const float* __restrict__ means,
                        const float* __restrict__ cov, const float* __restrict__ sh,
                        const float* __restrict__ opacity, const float* __restrict__ view_params,
                        float* __restrict__ records, uint32_t* __restrict__ keys,
                        uint2* __restrict__ rects, int32_t* __restrict__ radii,
                        uint8_t* __restrict__ clamp_out, uint4* __restrict__ cell_windows) {
  constexpr int GPW = kFusedGpw;
  __shared__ __attribute__((aligned(16))) float slab[GPW * 75 + 4];
  __shared__ __attribute__((aligned(16))) float vslab[kFusedViews * PS_VIEW_STRIDE];
  const int G = d.n_gaussians, vps = d.views_per_scene, K = d.sh_coeffs;
  const int gx = (d.width + kTile - 1) / kTile, gy = (d.height + kTile - 1) / kTile;
  const int lane = threadIdx.x;
  const int gl = lane % GPW, j = lane / GPW;
  const int g = blockIdx.x * GPW + gl;
  const int s = blockIdx.y;
  const bool active = g < G && j < vps;
  const size_t sg = (size_t)s * G + (g < G ? g : G - 1);    // clamped: every load unconditional
  const int S3 = K * 3;

  // one batch of loads: this lane's Gaussian and the scene's view blocks (-> LDS)
  constexpr int kViewFloats = kFusedViews * PS_VIEW_STRIDE;
  static_assert(kViewFloats % kWave == 0, "view staging assumes whole wave-rows");
  float m0[3], c6[6], opac;
  load_gaussian(d, means, cov, opacity, sg, m0, c6, opac);
  const float* vsrc = view_params + (size_t)s * vps * PS_VIEW_STRIDE;
  const int vlast = vps * PS_VIEW_STRIDE - 1;
  float vstage[kViewFloats / kWave];
#pragma unroll
  for (int u = 0; u < kViewFloats / kWave; ++u) {
    const int i = lane + u * kWave;
    vstage[u] = vsrc[i < vlast ? i : vlast];
  }
#pragma unroll
  for (int u = 0; u < kViewFloats / kWave; ++u) vslab[lane + u * kWave] = vstage[u];
  __syncthreads();
  const float* vp = vslab + (j < vps ? j : 0) * PS_VIEW_STRIDE;

  const GeoOut o = project_gaussian(d, m0, c6, vp, gx, gy);
  const bool vis = active && o.vis;
  const int v = s * vps + j;
  const size_t vg = (size_t)v * G + g;
  if (active) {
    radii[vg] = vis ? o.radius : 0;
    keys[vg] = vis ? __float_as_uint(o.tvz) : kCulledKey;
  }
  if (__ballot(vis) == 0ull) return;     // nobody in the wave needs a colour: no slab load

  {
    const size_t g0 = (size_t)s * G + (size_t)blockIdx.x * GPW;
    const int rem = G - (int)(blockIdx.x * GPW);
    const int nflt = (rem < GPW ? rem : GPW) * S3;
    stage_slab<(GPW * 75 + 255) / 256>(sh + g0 * (size_t)S3, slab, nflt, lane);
    __syncthreads();
  }
  if (!vis) return;
  float rgb[3];
  const uint32_t clamp_bits = sh_color<DEG>(slab + gl * S3, d.sh_layout == PS_SH_GK3, K, m0[0],
                                            m0[1], m0[2], vp, rgb);
  rects[vg] = make_uint2((uint32_t)o.xmin | ((uint32_t)o.ymin << 16),
                         (uint32_t)o.xmax | ((uint32_t)o.ymax << 16));
  float4* r = reinterpret_cast<float4*>(records + vg * kRecFloats);
  r[0] = make_float4(o.px, o.py, o.con_x, o.con_y);
  r[1] = make_float4(o.con_z, opac, o.tvz, __uint_as_float(packed_small_rect(o, gy)));
  r[2] = make_float4(rgb[0], rgb[1], rgb[2], __uint_as_float(clamp_bits));
  clamp_out[vg] = (uint8_t)clamp_bits;   // compact copy for the backward
  // which 4x4 cells the pair can reach (the tile forward's cull, cell_window.h)
  cell_windows[vg * (kRecFloats / 4)] = cell_window(o.px, o.py, o.con_x, o.con_y, o.con_z, opac, d.alpha_min);
}

void launch_preprocess_forward(const PsRasterDesc& d, const float* means, const float* cov,
                               const float* sh, const float* colors, const float* opacity,
                               const float* view_params, float* records, uint32_t* keys,
                               uint2* rects, int32_t* radii, uint8_t* clamp_bits, uint4* cell_windows,
                               bool geometry, bool sh_colors, hipStream_t st) {
  // LDS staging needs an odd float stride per Gaussian (K = 1, 9, 25), at most 75
  const bool lds = ((d.sh_coeffs * 3) & 1) && d.sh_coeffs * 3 <= 75;
  static const bool fused_ok = [] { const char* e = getenv("PS_PREPROCESS_FUSED"); return !e || atoi(e) != 0; }();
  if (fused_ok && geometry && sh && sh_colors && !colors && lds &&
      d.views_per_scene <= kFusedViews) {
    dim3 grid((d.n_gaussians + kFusedGpw - 1) / kFusedGpw, d.n_scenes), block(kWave);
#define PS_FUSED(DEG)                                                                          \
  hipLaunchKernelGGL((preprocess_fused_kernel<DEG>), grid, block, 0, st, d, means, cov, sh,     \
                     opacity, view_params, records, keys, rects, radii, clamp_bits, cell_windows)
    switch (d.sh_degree) {
      case 0: PS_FUSED(0); break;
      case 1: PS_FUSED(1); break;
      case 2: PS_FUSED(2); break;
      case 3: PS_FUSED(3); break;
      default: PS_FUSED(4); break;
    }
#undef PS_FUSED
    return;
  }
  if (geometry) {
    dim3 grid((d.n_gaussians + 255) / 256, d.n_scenes), block(256);
    hipLaunchKernelGGL(geometry_forward_kernel, grid, block, 0, st, d, means, cov, colors,
                       opacity, view_params, records, keys, rects, radii, cell_windows);
  }
  if (!sh || !sh_colors) return;
  const int deg = d.sh_degree;
  dim3 grid((d.n_gaussians + kWave - 1) / kWave, d.n_scenes), block(kWave);
  dim3 grid32((d.n_gaussians + 31) / 32, d.n_scenes);
#define PS_LAUNCH(DEG)                                                                        \
  do {                                                                                        \
    if (lds)                                                                                  \
      hipLaunchKernelGGL((color_forward_kernel<DEG, true>), grid32, block, 0, st, d, means,   \
                         sh, view_params, radii, records, clamp_bits);                        \
    else                                                                                      \
      hipLaunchKernelGGL((color_forward_kernel<DEG, false>), grid, block, 0, st, d, means,    \
                         sh, view_params, radii, records, clamp_bits);                        \
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
