// Image-side losses, each as one pass over the rendered images:
//   * LossMse (src/loss/loss_mse.py:22-31) + compute_psnr (src/evaluation/metrics.py:12-19):
//     one kernel reads prediction and target once and leaves the per-image squared-error sums
//     (raw for the loss, clipped to [0, 1] for the PSNR) and dL/d prediction, the tensor the
//     rasterizer backward consumes -- instead of sub, pow, mean and their three backward kernels;
//   * LossDepth (src/loss/loss_depth.py:26-60): depth normalisation, first / second finite
//     differences, optional bilateral weights from the target image, |.| and both means in one
//     forward kernel; the backward gathers every pixel's <= 6 stencil terms (no atomics).
// Sums are deterministic: per-block partials in fixed order, folded in double by one block.
#include "raster_common.h"

namespace ps {

namespace {

constexpr int kMseVec = 4;   // float4 per thread and pass

__device__ inline float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// block (256 threads) sum of two values -> thread 0
__device__ inline float2 block_sum2(float a, float b, float* scratch /* [8] */) {
  a = wave_sum(a); b = wave_sum(b);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { scratch[w] = a; scratch[4 + w] = b; }
  __syncthreads();
  float2 r = make_float2(0.f, 0.f);
  if (threadIdx.x == 0) {
    r.x = (scratch[0] + scratch[1]) + (scratch[2] + scratch[3]);
    r.y = (scratch[4] + scratch[5]) + (scratch[6] + scratch[7]);
  }
  return r;
}

__device__ inline float clip01(float v) { return fminf(fmaxf(v, 0.f), 1.f); }

}  // namespace

__global__ __launch_bounds__(256) void image_mse_kernel(
    unsigned elems, unsigned blocks_per_image, const float* __restrict__ pred,
    const float* __restrict__ target, float grad_scale, float* __restrict__ grad,
    float* __restrict__ partials) {
  __shared__ float scratch[8];
  const unsigned img = blockIdx.y;
  const size_t base = (size_t)img * elems;
  float sse = 0.f, ssec = 0.f;
  const unsigned per_block = 256u * kMseVec * 4u;
  const unsigned start = blockIdx.x * per_block;
  if ((elems & 3u) == 0 && ((reinterpret_cast<size_t>(pred) | reinterpret_cast<size_t>(target) |
                             reinterpret_cast<size_t>(grad)) & 15u) == 0) {
    float4 p[kMseVec], t[kMseVec];
#pragma unroll
    for (int i = 0; i < kMseVec; ++i) {
      unsigned e = start + (i * 256u + threadIdx.x) * 4u;
      bool ok = e < elems;
      p[i] = ok ? *reinterpret_cast<const float4*>(pred + base + e) : make_float4(0, 0, 0, 0);
      t[i] = ok ? *reinterpret_cast<const float4*>(target + base + e) : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < kMseVec; ++i) {
      unsigned e = start + (i * 256u + threadIdx.x) * 4u;
      const float pv[4] = {p[i].x, p[i].y, p[i].z, p[i].w}, tv[4] = {t[i].x, t[i].y, t[i].z, t[i].w};
      float gv[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float dlt = pv[c] - tv[c], dc = clip01(tv[c]) - clip01(pv[c]);
        sse += dlt * dlt; ssec += dc * dc;
        gv[c] = grad_scale * dlt;
      }
      if (grad && e < elems)
        *reinterpret_cast<float4*>(grad + base + e) = make_float4(gv[0], gv[1], gv[2], gv[3]);
    }
  } else {
    for (unsigned e = start + threadIdx.x; e < min(start + per_block, elems); e += 256u) {
      float dlt = pred[base + e] - target[base + e];
      float dc = clip01(target[base + e]) - clip01(pred[base + e]);
      sse += dlt * dlt; ssec += dc * dc;
      if (grad) grad[base + e] = grad_scale * dlt;
    }
  }
  float2 s = block_sum2(sse, ssec, scratch);
  if (threadIdx.x == 0) {
    partials[((size_t)img * blocks_per_image + blockIdx.x) * 2] = s.x;
    partials[((size_t)img * blocks_per_image + blockIdx.x) * 2 + 1] = s.y;
  }
}

// one wave per image: partial pairs in a fixed order, in double
__global__ __launch_bounds__(64) void fold_pairs_kernel(unsigned n_partials,
                                                        const float* __restrict__ partials,
                                                        float* __restrict__ out_a,
                                                        float* __restrict__ out_b) {
  const unsigned img = blockIdx.x;
  double a = 0.0, b = 0.0;
  for (unsigned i = threadIdx.x; i < n_partials; i += 64u) {
    a += (double)partials[((size_t)img * n_partials + i) * 2];
    b += (double)partials[((size_t)img * n_partials + i) * 2 + 1];
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
  if (threadIdx.x == 0) { out_a[img] = (float)a; if (out_b) out_b[img] = (float)b; }
}

// ---- depth smoothness -----------------------------------------------------------------------
namespace {

struct DepthView {
  const float* depth;    // [h][w] of this image
  const float* image;    // [c][h][w] of this image or nullptr
  float ln, inv_range;   // log(near), 1 / (log(far) - log(near))
  float lf;
  int h, w, channels, second;
  float sigma;
  bool use_sigma;

  // loss_depth.py:35-39
  __device__ float nd(int y, int x) const {
    float d = depth[y * w + x];
    d = fmaxf(fminf(d, lf), ln);
    return (d - ln) / (lf - ln);
  }
  // max over channels of the signed colour difference (:52-53)
  __device__ float cdx(int y, int i) const {
    float m = -INFINITY;
    for (int c = 0; c < channels; ++c) {
      const float* r = image + ((size_t)c * h + y) * w;
      m = fmaxf(m, r[i + 1] - r[i]);
    }
    return m;
  }
  __device__ float cdy(int i, int x) const {
    float m = -INFINITY;
    for (int c = 0; c < channels; ++c) {
      const float* r = image + (size_t)c * h * w;
      m = fmaxf(m, r[(i + 1) * w + x] - r[i * w + x]);
    }
    return m;
  }
  // x-term i of row y: finite difference (:42-48) and its bilateral weight (:51-59)
  __device__ float tx(int y, int i) const {
    if (!second) return nd(y, i + 1) - nd(y, i);
    float a = nd(y, i), b = nd(y, i + 1), c = nd(y, i + 2);
    return (c - b) - (b - a);
  }
  __device__ float wx(int y, int i) const {
    if (!use_sigma) return 1.f;
    float c = second ? fmaxf(cdx(y, i + 1), cdx(y, i)) : cdx(y, i);
    return expf(-c * sigma);
  }
  __device__ float ty(int i, int x) const {
    if (!second) return nd(i + 1, x) - nd(i, x);
    float a = nd(i, x), b = nd(i + 1, x), c = nd(i + 2, x);
    return (c - b) - (b - a);
  }
  __device__ float wy(int i, int x) const {
    if (!use_sigma) return 1.f;
    float c = second ? fmaxf(cdy(i + 1, x), cdy(i, x)) : cdy(i, x);
    return expf(-c * sigma);
  }
  // d |t w| / d t
  __device__ float qx(int y, int i) const {
    float t = tx(y, i), wt = wx(y, i);
    return t * wt > 0.f ? wt : (t * wt < 0.f ? -wt : 0.f);
  }
  __device__ float qy(int i, int x) const {
    float t = ty(i, x), wt = wy(i, x);
    return t * wt > 0.f ? wt : (t * wt < 0.f ? -wt : 0.f);
  }
};

__device__ inline DepthView make_view(const PsDepthLossDesc& d, unsigned img, const float* depth,
                                      const float* near, const float* far, const float* image) {
  DepthView v;
  v.h = d.height; v.w = d.width; v.channels = d.channels; v.second = d.use_second_derivative;
  v.sigma = d.sigma_image; v.use_sigma = d.use_sigma != 0;
  v.depth = depth + (size_t)img * d.height * d.width;
  v.image = v.use_sigma ? image + (size_t)img * d.channels * d.height * d.width : nullptr;
  v.ln = logf(near[img]); v.lf = logf(far[img]);
  v.inv_range = 1.f / (v.lf - v.ln);
  return v;
}

}  // namespace

__global__ __launch_bounds__(256) void depth_smoothness_forward_kernel(
    PsDepthLossDesc d, unsigned blocks_per_image, const float* __restrict__ depth,
    const float* __restrict__ near, const float* __restrict__ far,
    const float* __restrict__ image, float* __restrict__ partials) {
  __shared__ float scratch[8];
  const unsigned img = blockIdx.y;
  const DepthView v = make_view(d, img, depth, near, far, image);
  const int o = d.use_second_derivative ? 2 : 1;
  const unsigned pix = blockIdx.x * 256u + threadIdx.x;
  float sx = 0.f, sy = 0.f;
  if (pix < (unsigned)(d.height * d.width)) {
    const int y = pix / d.width, x = pix - y * d.width;
    if (x < d.width - o) sx = fabsf(v.tx(y, x) * v.wx(y, x));
    if (y < d.height - o) sy = fabsf(v.ty(y, x) * v.wy(y, x));
  }
  float2 s = block_sum2(sx, sy, scratch);
  if (threadIdx.x == 0) {
    partials[((size_t)img * blocks_per_image + blockIdx.x) * 2] = s.x;
    partials[((size_t)img * blocks_per_image + blockIdx.x) * 2 + 1] = s.y;
  }
}

// loss = weight (sum_x / n_x + sum_y / n_y) over all images (:60)
__global__ __launch_bounds__(64) void depth_smoothness_finish_kernel(
    unsigned n_images, const float* __restrict__ sums_x, const float* __restrict__ sums_y,
    double n_x, double n_y, float weight, float* __restrict__ loss) {
  double a = 0.0, b = 0.0;
  for (unsigned i = threadIdx.x; i < n_images; i += 64u) { a += sums_x[i]; b += sums_y[i]; }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { a += __shfl_xor(a, m); b += __shfl_xor(b, m); }
  if (threadIdx.x == 0) loss[0] = (float)(weight * (a / n_x + b / n_y));
}

__global__ __launch_bounds__(256) void depth_smoothness_backward_kernel(
    PsDepthLossDesc d, const float* __restrict__ depth, const float* __restrict__ near,
    const float* __restrict__ far, const float* __restrict__ image,
    const float* __restrict__ d_loss, float scale_x, float scale_y, float* __restrict__ d_depth) {
  const unsigned img = blockIdx.y;
  const DepthView v = make_view(d, img, depth, near, far, image);
  const unsigned pix = blockIdx.x * 256u + threadIdx.x;
  if (pix >= (unsigned)(d.height * d.width)) return;
  const int y = pix / d.width, x = pix - y * d.width;
  float gx = 0.f, gy = 0.f;
  if (!d.use_second_derivative) {          // term i = d[i+1] - d[i], i in [0, n-2]
    if (x >= 1) gx += v.qx(y, x - 1);
    if (x <= d.width - 2) gx -= v.qx(y, x);
    if (y >= 1) gy += v.qy(y - 1, x);
    if (y <= d.height - 2) gy -= v.qy(y, x);
  } else {                                 // term i = d[i+2] - 2 d[i+1] + d[i], i in [0, n-3]
    if (x >= 2) gx += v.qx(y, x - 2);
    if (x >= 1 && x <= d.width - 2) gx -= 2.f * v.qx(y, x - 1);
    if (x <= d.width - 3) gx += v.qx(y, x);
    if (y >= 2) gy += v.qy(y - 2, x);
    if (y >= 1 && y <= d.height - 2) gy -= 2.f * v.qy(y - 1, x);
    if (y <= d.height - 3) gy += v.qy(y, x);
  }
  const float raw = v.depth[y * d.width + x];
  const float inside = (raw > v.ln && raw < v.lf) ? 1.f / (v.lf - v.ln) : 0.f;
  d_depth[(size_t)img * d.height * d.width + pix] =
      d_loss[0] * (gx * scale_x + gy * scale_y) * inside;
}

// ---- launchers --------------------------------------------------------------------------------
namespace {
unsigned mse_blocks(unsigned elems) { return (elems + 256u * kMseVec * 4u - 1) / (256u * kMseVec * 4u); }
unsigned pixel_blocks(const PsDepthLossDesc& d) { return ((unsigned)d.height * d.width + 255u) / 256u; }
}  // namespace

size_t image_mse_workspace_bytes(int n_images, int elems) {
  return (size_t)n_images * mse_blocks((unsigned)elems) * 2 * sizeof(float);
}

int launch_image_mse(int n_images, int elems, const float* pred, const float* target,
                     float grad_scale, float* grad, float* sse, float* sse_clipped,
                     void* workspace, hipStream_t st) {
  const unsigned bpi = mse_blocks((unsigned)elems);
  float* partials = static_cast<float*>(workspace);
  image_mse_kernel<<<dim3(bpi, n_images), 256, 0, st>>>((unsigned)elems, bpi, pred, target,
                                                        grad_scale, grad, partials);
  fold_pairs_kernel<<<n_images, 64, 0, st>>>(bpi, partials, sse, sse_clipped);
  return PS_OK;
}

size_t depth_smoothness_workspace_bytes(const PsDepthLossDesc& d) {
  return ((size_t)d.n_images * pixel_blocks(d) * 2 + 2 * (size_t)d.n_images) * sizeof(float);
}

int launch_depth_smoothness_forward(const PsDepthLossDesc& d, const float* depth,
                                    const float* near, const float* far, const float* image,
                                    float* loss, void* workspace, hipStream_t st) {
  const unsigned bpi = pixel_blocks(d);
  float* partials = static_cast<float*>(workspace);
  float* sums_x = partials + (size_t)d.n_images * bpi * 2;
  float* sums_y = sums_x + d.n_images;
  depth_smoothness_forward_kernel<<<dim3(bpi, d.n_images), 256, 0, st>>>(d, bpi, depth, near, far,
                                                                         image, partials);
  fold_pairs_kernel<<<d.n_images, 64, 0, st>>>(bpi, partials, sums_x, sums_y);
  const int o = d.use_second_derivative ? 2 : 1;
  const double n_x = (double)d.n_images * d.height * (d.width - o);
  const double n_y = (double)d.n_images * (d.height - o) * d.width;
  depth_smoothness_finish_kernel<<<1, 64, 0, st>>>((unsigned)d.n_images, sums_x, sums_y, n_x, n_y,
                                                   d.weight, loss);
  return PS_OK;
}

int launch_depth_smoothness_backward(const PsDepthLossDesc& d, const float* depth,
                                     const float* near, const float* far, const float* image,
                                     const float* d_loss, float* d_depth, hipStream_t st) {
  const int o = d.use_second_derivative ? 2 : 1;
  const double n_x = (double)d.n_images * d.height * (d.width - o);
  const double n_y = (double)d.n_images * (d.height - o) * d.width;
  depth_smoothness_backward_kernel<<<dim3(pixel_blocks(d), d.n_images), 256, 0, st>>>(
      d, depth, near, far, image, d_loss, (float)(d.weight / n_x), (float)(d.weight / n_y), d_depth);
  return PS_OK;
}

}  // namespace ps
