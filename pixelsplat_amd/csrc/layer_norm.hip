// LayerNorm over the last dimension for the per-ray token matrix of path (A) (PreNorm of the
// cross-attention layers: src/model/transformer/pre_norm.py:34-35 -> nn.LayerNorm), forward
// and backward.  [rows][dim] fp32, dim % 4 == 0, dim <= 512: 16 lanes own a row (float4 chunks
// in registers), 4 rows per wave, so a wave instruction reads whole 512-byte rows and the two
// row reductions are 4 DPP-width shuffles.  The backward produces dx and per-block partial
// column sums for d_gamma / d_beta, folded in a fixed order by a second tiny kernel
// (deterministic).  The library kernels this replaces took 57 us forward and 41 + 61..89 us
// backward at [57 344][128]; these are HBM streams of 59 / 88 MB.
#include "raster_common.h"

namespace ps {

namespace {
constexpr int kLnMaxChunks = 8;    // float4 chunks per lane -> dim <= 512
constexpr int kLnRowsPerBlock = 128;

__device__ __forceinline__ float row_sum16(float v) {
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) v += __shfl_xor(v, m, 16);
  return v;
}
}  // namespace

template <int CH>
__global__ __launch_bounds__(256) void layer_norm_forward_kernel(
    int rows, int dim, float eps, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ mean_out,
    float* __restrict__ rstd_out) {
  const int sub = threadIdx.x & 15;
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool ok = row < rows;
  const int n4 = dim >> 2;
  const float4* xr = reinterpret_cast<const float4*>(x + (size_t)(ok ? row : 0) * dim);
  float4 v[CH];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int i = sub + c * 16;
    v[c] = (ok && i < n4) ? xr[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[c].x + v[c].y) + (v[c].z + v[c].w);
  }
  const float mean = row_sum16(s) / (float)dim;
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int i = sub + c * 16;
    if (i < n4) {
      const float a = v[c].x - mean, b = v[c].y - mean, cc = v[c].z - mean, d = v[c].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = 1.f / sqrtf(row_sum16(q) / (float)dim + eps);
  if (!ok) return;
  float4* yr = reinterpret_cast<float4*>(y + (size_t)row * dim);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int i = sub + c * 16;
    if (i < n4) {
      const float4 g = reinterpret_cast<const float4*>(gamma)[i];
      const float4 b = reinterpret_cast<const float4*>(beta)[i];
      yr[i] = make_float4(fmaf((v[c].x - mean) * rstd, g.x, b.x), fmaf((v[c].y - mean) * rstd, g.y, b.y),
                          fmaf((v[c].z - mean) * rstd, g.z, b.z), fmaf((v[c].w - mean) * rstd, g.w, b.w));
    }
  }
  if (sub == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
}

// partials: [blocks][2][dim] (d_gamma | d_beta of the block's rows)
template <int CH>
__global__ __launch_bounds__(256) void layer_norm_backward_kernel(
    int rows, int dim, const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
    const float* __restrict__ dy, const float* __restrict__ d_residual, float* __restrict__ dx,
    float* __restrict__ partials) {
  extern __shared__ float lds[];   // [16 row groups][2 dim]
  const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const int n4 = dim >> 2;
  float4 gam[CH], dg[CH], db[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int i = sub + c * 16;
    gam[c] = i < n4 ? reinterpret_cast<const float4*>(gamma)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    dg[c] = db[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int row0 = blockIdx.x * kLnRowsPerBlock;
  for (int it = 0; it < kLnRowsPerBlock / 16; ++it) {
    const int row = row0 + it * 16 + grp;
    const bool ok = row < rows;
    const size_t base = (size_t)(ok ? row : 0) * dim;
    const float mean = ok ? mean_in[row] : 0.f, rstd = ok ? rstd_in[row] : 0.f;
    float4 xh[CH], g[CH];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const int i = sub + c * 16;
      const bool live = ok && i < n4;
      const float4 xv = live ? reinterpret_cast<const float4*>(x + base)[i] : make_float4(0, 0, 0, 0);
      const float4 dv = live ? reinterpret_cast<const float4*>(dy + base)[i] : make_float4(0, 0, 0, 0);
      xh[c] = live ? make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd,
                                 (xv.w - mean) * rstd) : make_float4(0, 0, 0, 0);
      g[c] = make_float4(dv.x * gam[c].x, dv.y * gam[c].y, dv.z * gam[c].z, dv.w * gam[c].w);
      s1 += (g[c].x + g[c].y) + (g[c].z + g[c].w);
      s2 += (g[c].x * xh[c].x + g[c].y * xh[c].y) + (g[c].z * xh[c].z + g[c].w * xh[c].w);
      dg[c].x = fmaf(dv.x, xh[c].x, dg[c].x); dg[c].y = fmaf(dv.y, xh[c].y, dg[c].y);
      dg[c].z = fmaf(dv.z, xh[c].z, dg[c].z); dg[c].w = fmaf(dv.w, xh[c].w, dg[c].w);
      db[c].x += dv.x; db[c].y += dv.y; db[c].z += dv.z; db[c].w += dv.w;
    }
    const float m1 = row_sum16(s1) / (float)dim, m2 = row_sum16(s2) / (float)dim;
    if (ok) {
      float4* dxr = reinterpret_cast<float4*>(dx + base);
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        const int i = sub + c * 16;
        if (i < n4) {
          // gradient arriving through the residual branch around the normalised sub-layer
          const float4 r = d_residual ? reinterpret_cast<const float4*>(d_residual + base)[i]
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
          dxr[i] = make_float4(fmaf(rstd, g[c].x - m1 - xh[c].x * m2, r.x),
                               fmaf(rstd, g[c].y - m1 - xh[c].y * m2, r.y),
                               fmaf(rstd, g[c].z - m1 - xh[c].z * m2, r.z),
                               fmaf(rstd, g[c].w - m1 - xh[c].w * m2, r.w));
        }
      }
    }
  }
  // fold the 16 row groups of the block in a fixed order
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int i = sub + c * 16;
    if (i < n4) {
      reinterpret_cast<float4*>(lds + (size_t)grp * 2 * dim)[i] = dg[c];
      reinterpret_cast<float4*>(lds + (size_t)grp * 2 * dim + dim)[i] = db[c];
    }
  }
  __syncthreads();
  for (int col = threadIdx.x; col < 2 * dim; col += 256) {
    float s = 0.f;
#pragma unroll
    for (int gi = 0; gi < 16; ++gi) s += lds[(size_t)gi * 2 * dim + col];
    partials[(size_t)blockIdx.x * 2 * dim + col] = s;
  }
}

__global__ __launch_bounds__(256) void layer_norm_fold_kernel(int blocks, int dim,
                                                              const float* __restrict__ partials,
                                                              float* __restrict__ d_gamma,
                                                              float* __restrict__ d_beta) {
  // one wave per column: lanes take the block partials in a fixed interleaved order
  const int lane = threadIdx.x & 63, col = blockIdx.x * 4 + (threadIdx.x >> 6);
  float s = 0.f;
  if (col < 2 * dim)
    for (int b = lane; b < blocks; b += 64) s += partials[(size_t)b * 2 * dim + col];
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  if (col < 2 * dim && lane == 0) (col < dim ? d_gamma[col] : d_beta[col - dim]) = s;
}

namespace {
int chunks_for(int dim) {
  if (dim <= 0 || (dim & 3) || dim > 16 * 4 * kLnMaxChunks) return 0;
  const int ch = (dim / 4 + 15) / 16;
  return ch <= 2 ? 2 : ch <= 4 ? 4 : 8;
}
}  // namespace

size_t layer_norm_workspace_floats(int rows, int dim) {
  const size_t blocks = ((size_t)rows + kLnRowsPerBlock - 1) / kLnRowsPerBlock;
  return blocks * 2 * dim;
}

int launch_layer_norm_forward(int rows, int dim, float eps, const float* x, const float* gamma,
                              const float* beta, float* y, float* mean, float* rstd,
                              hipStream_t st) {
  const unsigned blocks = (unsigned)((rows + 15) / 16);
  switch (chunks_for(dim)) {
    case 2: layer_norm_forward_kernel<2><<<blocks, 256, 0, st>>>(rows, dim, eps, x, gamma, beta, y, mean, rstd); break;
    case 4: layer_norm_forward_kernel<4><<<blocks, 256, 0, st>>>(rows, dim, eps, x, gamma, beta, y, mean, rstd); break;
    case 8: layer_norm_forward_kernel<8><<<blocks, 256, 0, st>>>(rows, dim, eps, x, gamma, beta, y, mean, rstd); break;
    default: return PS_ERR_UNSUPPORTED;
  }
  return PS_OK;
}

int launch_layer_norm_backward(int rows, int dim, const float* x, const float* gamma,
                               const float* mean, const float* rstd, const float* dy,
                               const float* d_residual, float* dx, float* d_gamma, float* d_beta,
                               float* workspace, hipStream_t st) {
  const unsigned blocks = (unsigned)((rows + kLnRowsPerBlock - 1) / kLnRowsPerBlock);
  const size_t lds = (size_t)16 * 2 * dim * sizeof(float);
  switch (chunks_for(dim)) {
    case 2: layer_norm_backward_kernel<2><<<blocks, 256, lds, st>>>(rows, dim, x, gamma, mean, rstd, dy, d_residual, dx, workspace); break;
    case 4: layer_norm_backward_kernel<4><<<blocks, 256, lds, st>>>(rows, dim, x, gamma, mean, rstd, dy, d_residual, dx, workspace); break;
    case 8: layer_norm_backward_kernel<8><<<blocks, 256, lds, st>>>(rows, dim, x, gamma, mean, rstd, dy, d_residual, dx, workspace); break;
    default: return PS_ERR_UNSUPPORTED;
  }
  layer_norm_fold_kernel<<<(unsigned)((2 * dim + 3) / 4), 256, 0, st>>>((int)blocks, dim, workspace,
                                                                               d_gamma, d_beta);
  return PS_OK;
}

}  // namespace ps
