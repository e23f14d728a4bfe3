// Depth predictor sampling: everything DepthPredictorMonocular.forward
// does after its linear projection (src/model/encoder/epipolar/depth_predictor_monocular.py:52-81),
// with the discrete sampler (src/misc/discrete_probability_distribution.py:7-33), the
// disparity conversion (epipolar/conversions.py:5-14) and the encoder's opacity mapping
// (encoder_epipolar.py:97-110, :170), as ONE forward and ONE backward kernel.
//
// Both are HBM streams over the projection ([rays][2 * buckets * surfaces] floats, 256 B per
// ray in the shipped config).  A wave reads a run of whole rays with coalesced 8-byte loads
// (all issued before the first LDS store) and keeps only the LOGITS in its private LDS slab;
// then ONE LANE PER (ray, surface) ROW walks its row serially: max, exp + sum, and the CDF
// search / top-k are three short loops of ds_read + 1-8 VALU per bucket.  The few offsets a
// row needs (one per drawn sample) are gathered from global memory, lines the staging has just
// pulled through the cache.  Row stride = an odd number of floats, so any 64 rows are
// bank-conflict-free.  The backward overwrites the logits with their gradients in place,
// keeps the (sparse) offset gradients in a second slab and streams both out as coalesced
// pairs.  (A first version gave each row a 32-lane group and did the reductions with
// cross-lane permutes: ~800 instructions per row pair, 0.35 ms forward for 279 MB.)
#include "raster_common.h"

namespace ps {

namespace {

constexpr float kF32Eps = 1.1920928955078125e-07f;
constexpr int kSampleChunk = 4;       // CDF searches sharing one pass over the buckets
constexpr int kStageBatch = 8;        // float2 loads in flight per lane while staging
constexpr unsigned kSlabBytes = 9216; // LDS budget per wave
constexpr int kChunk = 8;             // bucket values loaded together (see for_buckets)

struct RowCtx {   // per-view constants of conversions.py:11-13
  float disp_near, disp_far;
};
__device__ inline RowCtx row_ctx(const float* near, const float* far, unsigned view) {
  return {1.f / (near[view] + 1e-10f), 1.f / (far[view] + 1e-10f)};
}
__device__ inline float depth_of(float rd, const RowCtx& cx) {   // conversions.py:14
  return 1.f / ((1.f - rd) * (cx.disp_near - cx.disp_far) + cx.disp_far + 1e-10f);
}

// encoder_epipolar.py:109-110; exponent 1 (the shipped config) is the identity, 0 = unmapped
__device__ inline float map_opacity(float p, float e) {
  if (e == 1.f || e == 0.f) return p;
  return 0.5f * (1.f - powf(1.f - p, e) + powf(p, 1.f / e));
}
__device__ inline float map_opacity_grad(float p, float e) {
  if (e == 1.f || e == 0.f) return 1.f;
  return 0.5f * (e * powf(1.f - p, e - 1.f) + (1.f / e) * powf(p, 1.f / e - 1.f));
}
// accurate exp: the depth is ill-conditioned in the offset at far depths
__device__ inline float sigmoid(float y) { return 1.f / (1.f + expf(-y)); }

// Walks `n_pairs` consecutive (logit, offset) pairs of the projection with position
// (ray, pos) in a slab of `stride` floats per ray: logits -> slab.
__device__ inline void stage_logits(float* slab, const float* __restrict__ src, unsigned n_pairs,
                                    unsigned ppr, unsigned stride, unsigned lane) {
  const float2* s2 = reinterpret_cast<const float2*>(src);
  unsigned ray = lane / ppr, pos = lane - ray * ppr;
  const unsigned step_ray = 64u / ppr, step_pos = 64u - step_ray * ppr;
  for (unsigned base = 0; base < n_pairs; base += 64u * kStageBatch) {
    float2 v[kStageBatch];
#pragma unroll
    for (int i = 0; i < kStageBatch; ++i) {
      unsigned e = base + i * 64u + lane;
      v[i] = e < n_pairs ? s2[e] : make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < kStageBatch; ++i) {
      unsigned e = base + i * 64u + lane;
      if (e < n_pairs) slab[ray * stride + pos] = v[i].x;
      pos += step_pos; ray += step_ray;
      if (pos >= ppr) { pos -= ppr; ++ray; }
    }
  }
}
// (gx, gy) slabs -> interleaved gradient pairs
__device__ inline void unstage_pairs(const float* gx, const float* gy, float* __restrict__ dst,
                                     unsigned n_pairs, unsigned ppr, unsigned stride,
                                     unsigned lane) {
  float2* d2 = reinterpret_cast<float2*>(dst);
  unsigned ray = lane / ppr, pos = lane - ray * ppr;
  const unsigned step_ray = 64u / ppr, step_pos = 64u - step_ray * ppr;
  for (unsigned e = lane; e < n_pairs; e += 64u) {
    d2[e] = make_float2(gx[ray * stride + pos], gy[ray * stride + pos]);
    pos += step_pos; ray += step_ray;
    if (pos >= ppr) { pos -= ppr; ++ray; }
  }
}

// Bucket loops run in chunks of kChunk values loaded together: a load-use-per-iteration loop
// pays one LDS round trip per bucket.  STEP = 1 for one surface (constant ds_read offsets),
// 0 = run-time stride.
template <int STEP, typename F>
__device__ __forceinline__ void for_buckets(const float* p, int S, int step_rt, F&& f) {
  const int step = STEP ? STEP : step_rt;
  int k0 = 0;
  for (; k0 + kChunk <= S; k0 += kChunk) {
    float v[kChunk];
#pragma unroll
    for (int i = 0; i < kChunk; ++i) v[i] = p[(k0 + i) * step];
#pragma unroll
    for (int i = 0; i < kChunk; ++i) f(k0 + i, v[i]);
  }
  for (; k0 < S; ++k0) f(k0, p[k0 * step]);
}

// softmax numerators in place (row := exp(x - max)); returns their sum and the max
template <int STEP>
__device__ inline float row_exponentials(float* p, int S, int step_rt, float* max_out) {
  const int step = STEP ? STEP : step_rt;
  float m = -INFINITY;
  for_buckets<STEP>(p, S, step_rt, [&](int, float v) { m = fmaxf(m, v); });
  float sum = 0.f;
  for_buckets<STEP>(p, S, step_rt, [&](int k, float v) {
    float e = __expf(v - m);
    p[k * step] = e;
    sum += e;
  });
  *max_out = m;
  return sum;
}

struct RowScale {   // pdf_k = e_k * inv (:56); normalized_k = pdf_k * it
  float inv, tot, it;   // (discrete_probability_distribution.py:17, :31)
};
__device__ inline RowScale row_scale(float sum) {
  RowScale r;
  r.inv = 1.f / sum;
  r.tot = sum * r.inv;                 // sum of the pdf, 1 up to rounding
  r.it = 1.f / (kF32Eps + r.tot);
  return r;
}

}  // namespace

template <int STEP>
__global__ __launch_bounds__(256) void depth_sampler_forward_kernel(
    PsDepthSamplerDesc d, unsigned rays_per_wave, unsigned stride,
    const float* __restrict__ projected, const float* __restrict__ near,
    const float* __restrict__ far, const float* __restrict__ uniforms, float* __restrict__ depth,
    float* __restrict__ opacity, int32_t* __restrict__ index) {
  extern __shared__ float lds[];
  const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  const unsigned total_rays = (unsigned)d.n_views * d.rays_per_view;
  const unsigned ray0 = (blockIdx.x * (blockDim.x >> 6) + w) * rays_per_wave;
  if (ray0 >= total_rays) return;   // wave-uniform, no block barriers below
  const unsigned n_rays = min(rays_per_wave, total_rays - ray0);
  const int S = d.buckets, srf = d.surfaces, spp = d.spp, step = STEP ? STEP : srf;
  const unsigned ppr = (unsigned)S * srf;
  float* slab = lds + w * rays_per_wave * stride;
  stage_logits(slab, projected + (size_t)ray0 * (2u * ppr), n_rays * ppr, ppr, stride, lane);
  __builtin_amdgcn_wave_barrier();

  for (unsigned row = lane; row < n_rays * srf; row += 64u) {
    const unsigned r = STEP == 1 ? row : row / (unsigned)srf, sf = row - r * srf;
    float* p = slab + r * stride + sf;
    const float* offsets = projected + (size_t)(ray0 + r) * (2u * ppr) + 2u * sf + 1u;
    float row_max;
    const RowScale sc = row_scale(row_exponentials<STEP>(p, S, srf, &row_max));
    const float scale_n = sc.inv * sc.it;
    const RowCtx cx = row_ctx(near, far, (ray0 + r) / (unsigned)d.rays_per_view);
    const size_t out0 = ((size_t)(ray0 + r) * srf + sf) * spp;
    float prev_p = INFINITY;
    int prev_k = -1;
    for (int t0 = 0; t0 < spp; t0 += kSampleChunk) {
      int idx[kSampleChunk];
      if (d.deterministic) {
        // gather_discrete_topk: buckets in the order (probability descending, index ascending)
#pragma unroll
        for (int c = 0; c < kSampleChunk; ++c) {
          float best_p = -1.f;
          int best_k = 0;
          if (t0 + c < spp) {
            for_buckets<STEP>(p, S, srf, [&](int k, float v) {
              bool after = v < prev_p || (v == prev_p && k > prev_k);
              if (after && v > best_p) { best_p = v; best_k = k; }
            });
            prev_p = best_p; prev_k = best_k;
          }
          idx[c] = best_k;
        }
      } else {
        // sample_discrete_distribution: searchsorted(cumsum(normalized), u, right=True)
        float u[kSampleChunk];
#pragma unroll
        for (int c = 0; c < kSampleChunk; ++c) {
          u[c] = t0 + c < spp ? uniforms[out0 + t0 + c] : -1.f;
          idx[c] = 0;
        }
        float cdf = 0.f;
        for_buckets<STEP>(p, S, srf, [&](int, float v) {
          cdf += v * scale_n;
#pragma unroll
          for (int c = 0; c < kSampleChunk; ++c) idx[c] += cdf <= u[c] ? 1 : 0;
        });
#pragma unroll
        for (int c = 0; c < kSampleChunk; ++c) idx[c] = min(idx[c], S - 1);
      }
#pragma unroll
      for (int c = 0; c < kSampleChunk; ++c) {
        if (t0 + c >= spp) break;
        const int k = idx[c];
        const float e = p[k * step], oraw = offsets[2 * k * srf];
        float prob = e * scale_n;
        if (d.use_transmittance) {   // depth_predictor_monocular.py:72-78
          float part = 0.f;
          for (int i = 0; i < k; ++i) part += p[i * step] * sc.inv;
          prob = e * sc.inv / (1.f - part + 1e-10f);
        }
        const float rd = ((float)k + sigmoid(oraw)) / (float)S;   // :64
        depth[out0 + t0 + c] = depth_of(rd, cx);
        opacity[out0 + t0 + c] = map_opacity(prob, d.opacity_exponent) * d.opacity_scale;
        index[out0 + t0 + c] = k;
      }
    }
  }
}

// LDS per wave: logits -> gradients in place | offset gradients | one fix-up value per drawn
// sample | (transmittance only) point and range terms
template <int STEP>
__global__ __launch_bounds__(256) void depth_sampler_backward_kernel(
    PsDepthSamplerDesc d, unsigned rays_per_wave, unsigned stride, unsigned wave_floats,
    const float* __restrict__ projected, const float* __restrict__ near,
    const float* __restrict__ far, const int32_t* __restrict__ index,
    const float* __restrict__ d_depth, const float* __restrict__ d_opacity,
    float* __restrict__ d_projected) {
  extern __shared__ float lds[];
  const unsigned lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  const unsigned total_rays = (unsigned)d.n_views * d.rays_per_view;
  const unsigned ray0 = (blockIdx.x * (blockDim.x >> 6) + w) * rays_per_wave;
  if (ray0 >= total_rays) return;
  const unsigned n_rays = min(rays_per_wave, total_rays - ray0);
  const int S = d.buckets, srf = d.surfaces, spp = d.spp, step = STEP ? STEP : srf;
  const unsigned ppr = (unsigned)S * srf;
  float* slab = lds + w * wave_floats;
  float* grad_y = slab + rays_per_wave * stride;
  float* fixes = grad_y + rays_per_wave * stride;
  float* points = fixes + rays_per_wave * srf * spp;   // transmittance only
  float* ranges = points + rays_per_wave * stride;
  stage_logits(slab, projected + (size_t)ray0 * (2u * ppr), n_rays * ppr, ppr, stride, lane);
  __builtin_amdgcn_wave_barrier();

  for (unsigned row = lane; row < n_rays * srf; row += 64u) {
    const unsigned r = STEP == 1 ? row : row / (unsigned)srf, sf = row - r * srf;
    float* p = slab + r * stride + sf;
    float* gy = grad_y + r * stride + sf;
    float* fix = fixes + row * spp;
    const float* offsets = projected + (size_t)(ray0 + r) * (2u * ppr) + 2u * sf + 1u;
    float row_max;
    const RowScale sc = row_scale(row_exponentials<STEP>(p, S, srf, &row_max));
    const float scale_n = sc.inv * sc.it;
    for (int k = 0; k < S; ++k) gy[k * step] = 0.f;
    const RowCtx cx = row_ctx(near, far, (ray0 + r) / (unsigned)d.rays_per_view);
    const size_t out0 = ((size_t)(ray0 + r) * srf + sf) * spp;
    // the chosen bucket's offset gets the depth gradient; the index is not differentiable
    float dot_n = 0.f;
    for (int t = 0; t < spp; ++t) {
      const int k = index[out0 + t];
      const float sig = sigmoid(offsets[2 * k * srf]);
      const float dep = depth_of(((float)k + sig) / (float)S, cx);
      const float g_rd = d_depth[out0 + t] * dep * dep * (cx.disp_near - cx.disp_far);
      gy[k * step] += g_rd / (float)S * sig * (1.f - sig);
      if (!d.use_transmittance) {
        const float e = p[k * step];
        const float gn = d_opacity[out0 + t] * d.opacity_scale *
                         map_opacity_grad(e * scale_n, d.opacity_exponent);
        fix[t] = e * sc.inv * gn * sc.it;   // the sample's own term of d/d logit_k, added below
        dot_n += gn * e * sc.inv;
      }
    }
    if (!d.use_transmittance) {
      // normalized = pdf / (eps + sum pdf), then the softmax: every bucket gets
      // pdf_k (-shift - dot_p), with sum_k pdf_k g_pdf_k = dot_p in closed form
      const float shift = dot_n * sc.it * sc.it;
      const float dot_p = sc.it * dot_n - shift * sc.tot;
      const float common = (-shift - dot_p) * sc.inv;
      for_buckets<STEP>(p, S, srf, [&](int k, float e) { p[k * step] = e * common; });
      for (int t = 0; t < spp; ++t) p[index[out0 + t] * step] += fix[t];
    } else {
      // opacity_k = pdf_k / (1 - sum_{i<k} pdf_i + 1e-10): a point term at k and the same
      // amount on every bucket below k (the range term stored at k applies to all i < k)
      float* pt = points + r * stride + sf;
      float* rg = ranges + r * stride + sf;
      for (int k = 0; k < S; ++k) { pt[k * step] = 0.f; rg[k * step] = 0.f; }
      for (int t = 0; t < spp; ++t) {
        const int k = index[out0 + t];
        float part = 0.f;
        for (int i = 0; i < k; ++i) part += p[i * step] * sc.inv;
        const float pk = p[k * step] * sc.inv, den = 1.f - part + 1e-10f;
        const float go = d_opacity[out0 + t] * d.opacity_scale *
                         map_opacity_grad(pk / den, d.opacity_exponent);
        pt[k * step] += go / den;
        rg[k * step] += go * pk / (den * den);
      }
      float above = 0.f, dot_p = 0.f;
      for (int k = S - 1; k >= 0; --k) {
        const float gp = pt[k * step] + above;
        above += rg[k * step];
        pt[k * step] = gp;
        dot_p += gp * p[k * step] * sc.inv;
      }
      for (int k = 0; k < S; ++k) p[k * step] = p[k * step] * sc.inv * (pt[k * step] - dot_p);
    }
  }
  __builtin_amdgcn_wave_barrier();
  unstage_pairs(slab, grad_y, d_projected + (size_t)ray0 * (2u * ppr), n_rays * ppr, ppr, stride,
                lane);
}

namespace {
struct SamplerPlan {
  unsigned rays_per_wave, stride, wave_floats, blocks;
  size_t lds_bytes;
};
// odd row stride; rays per wave = the power of two (<= 64 rows) that fits the LDS budget
bool plan(const PsDepthSamplerDesc& d, bool backward, SamplerPlan* out) {
  const unsigned ppr = (unsigned)d.buckets * d.surfaces;
  const unsigned stride = ppr | 1u;
  const unsigned slabs = backward ? (d.use_transmittance ? 4u : 2u) : 1u;
  const unsigned per_ray = slabs * stride + (backward ? (unsigned)d.surfaces * d.spp : 0u);
  const unsigned fit = kSlabBytes / (per_ray * 4u);
  if (fit == 0) return false;
  unsigned rpw = 1;
  while (rpw * 2 <= fit && rpw * 2 * d.surfaces <= 64u) rpw *= 2;
  const unsigned total_rays = (unsigned)d.n_views * d.rays_per_view;
  const unsigned waves = (total_rays + rpw - 1) / rpw;
  *out = {rpw, stride, rpw * per_ray, (waves + 3) / 4, (size_t)4 * rpw * per_ray * 4};
  return true;
}
}  // namespace

int launch_depth_sampler_forward(const PsDepthSamplerDesc& d, const float* projected,
                                 const float* near, const float* far, const float* uniforms,
                                 float* depth, float* opacity, int32_t* index, hipStream_t st) {
  SamplerPlan pl;
  if (!plan(d, false, &pl)) return PS_ERR_UNSUPPORTED;
  auto kernel = d.surfaces == 1 ? depth_sampler_forward_kernel<1> : depth_sampler_forward_kernel<0>;
  kernel<<<pl.blocks, 256, pl.lds_bytes, st>>>(d, pl.rays_per_wave, pl.stride, projected, near,
                                               far, uniforms, depth, opacity, index);
  return PS_OK;
}

int launch_depth_sampler_backward(const PsDepthSamplerDesc& d, const float* projected,
                                  const float* near, const float* far, const int32_t* index,
                                  const float* d_depth, const float* d_opacity,
                                  float* d_projected, hipStream_t st) {
  SamplerPlan pl;
  if (!plan(d, true, &pl)) return PS_ERR_UNSUPPORTED;
  auto kernel = d.surfaces == 1 ? depth_sampler_backward_kernel<1> : depth_sampler_backward_kernel<0>;
  kernel<<<pl.blocks, 256, pl.lds_bytes, st>>>(d, pl.rays_per_wave, pl.stride, pl.wave_floats,
                                               projected, near, far, index, d_depth, d_opacity,
                                               d_projected);
  return PS_OK;
}

}  // namespace ps
