// Camera set-up for a batch of views in ONE launch: what the reference does on the host side
// of the rasterizer call with ~60 tiny PyTorch kernels and two .item() syncs per view
// (/root/reference/src/model/decoder/cuda_splatting.py:64-71 scale-invariant renorm,
// :80-82 field of view via src/geometry/projection.py:233-247 get_fov, :17-44 projection
// matrix, :84-87 transposed view / full-projection matrices, :110 camera position).
// One thread per view; output is the packed per-view parameter block of
// include/pixelsplat_hip.h (PS_VIEW_*).
#include "raster_common.h"

namespace ps {

template <typename T>
__device__ inline bool invert4(const T* m, T* inv) {  // general 4x4, cofactors
  T a[16];
  a[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] +
         m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
  a[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] -
         m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
  a[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] +
         m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
  a[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] -
          m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
  a[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] -
         m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
  a[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] +
         m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
  a[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] -
         m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
  a[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] +
          m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
  a[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] +
         m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
  a[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] -
         m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
  a[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] +
          m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
  a[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] -
          m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
  a[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] -
         m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
  a[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] +
         m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
  a[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] -
          m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
  a[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] +
          m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
  const T det = m[0] * a[0] + m[1] * a[4] + m[2] * a[8] + m[3] * a[12];
  const T id = T(1) / det;
#pragma unroll
  for (int i = 0; i < 16; ++i) inv[i] = a[i] * id;
  return det != T(0);
}

template <typename T>
__device__ inline void invert3(const T* k, T* inv) {
  const T c00 = k[4] * k[8] - k[5] * k[7], c01 = k[5] * k[6] - k[3] * k[8],
          c02 = k[3] * k[7] - k[4] * k[6];
  const T det = k[0] * c00 + k[1] * c01 + k[2] * c02;
  const T id = T(1) / det;
  inv[0] = c00 * id; inv[1] = (k[2] * k[7] - k[1] * k[8]) * id; inv[2] = (k[1] * k[5] - k[2] * k[4]) * id;
  inv[3] = c01 * id; inv[4] = (k[0] * k[8] - k[2] * k[6]) * id; inv[5] = (k[2] * k[3] - k[0] * k[5]) * id;
  inv[6] = c02 * id; inv[7] = (k[1] * k[6] - k[0] * k[7]) * id; inv[8] = (k[0] * k[4] - k[1] * k[3]) * id;
}

__device__ inline float ray_cos(const float* ki, float ax, float ay, float bx, float by) {
  // normalise(K^-1 [ax,ay,1]) . normalise(K^-1 [bx,by,1])
  float a[3], b[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    a[i] = ki[3 * i] * ax + ki[3 * i + 1] * ay + ki[3 * i + 2];
    b[i] = ki[3 * i] * bx + ki[3 * i + 1] * by + ki[3 * i + 2];
  }
  const float na = sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
  const float nb = sqrtf(b[0] * b[0] + b[1] * b[1] + b[2] * b[2]);
  return (a[0] / na) * (b[0] / nb) + (a[1] / na) * (b[1] / nb) + (a[2] / na) * (b[2] / nb);
}

__global__ void camera_setup_kernel(int n_views, const float* __restrict__ extrinsics,
                                    const float* __restrict__ intrinsics,
                                    const float* __restrict__ near, const float* __restrict__ far,
                                    const float* __restrict__ bg, int scale_invariant,
                                    float* __restrict__ view_params) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n_views) return;
  float e[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) e[i] = extrinsics[16 * v + i];
  float nr = near[v], fr = far[v];
  const float scale = scale_invariant ? 1.0f / nr : 1.0f;
  e[3] *= scale; e[7] *= scale; e[11] *= scale;   // translation column
  nr *= scale; fr *= scale;

  float ki[9];
  invert3(intrinsics + 9 * v, ki);
  const float fov_x = acosf(ray_cos(ki, 0.f, 0.5f, 1.f, 0.5f));
  const float fov_y = acosf(ray_cos(ki, 0.5f, 0.f, 0.5f, 1.f));
  const float tan_x = tanf(0.5f * fov_x), tan_y = tanf(0.5f * fov_y);

  // projection (column-vector convention): x,y -> (-1,1), z -> (0,1), w = z
  const float top = tan_y * nr, right = tan_x * nr;
  float P[16] = {0};
  P[0] = 2.f * nr / (2.f * right);
  P[5] = 2.f * nr / (2.f * top);
  P[14] = 1.f;                      // row 3, col 2
  P[10] = fr / (fr - nr);           // row 2, col 2
  P[11] = -(fr * nr) / (fr - nr);   // row 2, col 3

  float w2c[16];
  invert4(e, w2c);
  float* out = view_params + (size_t)v * PS_VIEW_STRIDE;
  // transposed (row-vector) matrices: out[4*c + r] = M[r][c]
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      out[PS_VIEW_VIEWMATRIX + 4 * c + r] = w2c[4 * r + c];
      float acc = 0.f;   // full = P @ w2c
#pragma unroll
      for (int k = 0; k < 4; ++k) acc += P[4 * r + k] * w2c[4 * k + c];
      out[PS_VIEW_PROJMATRIX + 4 * c + r] = acc;
    }
  out[PS_VIEW_CAMPOS + 0] = e[3]; out[PS_VIEW_CAMPOS + 1] = e[7]; out[PS_VIEW_CAMPOS + 2] = e[11];
  out[PS_VIEW_TANFOVX] = tan_x; out[PS_VIEW_TANFOVY] = tan_y;
  out[PS_VIEW_BG + 0] = bg[3 * v]; out[PS_VIEW_BG + 1] = bg[3 * v + 1]; out[PS_VIEW_BG + 2] = bg[3 * v + 2];
  out[PS_VIEW_SCALE] = scale;
#pragma unroll
  for (int i = PS_VIEW_SCALE + 1; i < PS_VIEW_STRIDE; ++i) out[i] = 0.f;
}

void launch_camera_setup(int n_views, const float* extrinsics, const float* intrinsics,
                         const float* near, const float* far, const float* bg,
                         int scale_invariant, float* view_params, hipStream_t st) {
  hipLaunchKernelGGL(camera_setup_kernel, dim3((n_views + 63) / 64), dim3(64), 0, st, n_views,
                     extrinsics, intrinsics, near, far, bg, scale_invariant, view_params);
}

// w2c = c2w^-1 and K^-1 for n cameras in one launch (the sampler's torch.linalg.inv calls:
// /root/reference/src/geometry/epipolar_lines.py:167, src/geometry/projection.py:84; torch's
// inv also checks for singularity on the host, i.e. synchronises).  Cofactors in double,
// rounded once.
__global__ void camera_inverse_kernel(int n, const float* __restrict__ c2w,
                                      const float* __restrict__ k, float* __restrict__ w2c,
                                      float* __restrict__ k_inv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double m[16], mi[16], kk[9], ki[9];
#pragma unroll
  for (int j = 0; j < 16; ++j) m[j] = (double)c2w[16 * i + j];
#pragma unroll
  for (int j = 0; j < 9; ++j) kk[j] = (double)k[9 * i + j];
  invert4<double>(m, mi);
  invert3<double>(kk, ki);
#pragma unroll
  for (int j = 0; j < 16; ++j) w2c[16 * i + j] = (float)mi[j];
#pragma unroll
  for (int j = 0; j < 9; ++j) k_inv[9 * i + j] = (float)ki[j];
}

void launch_camera_inverse(int n, const float* c2w, const float* k, float* w2c, float* k_inv,
                           hipStream_t st) {
  hipLaunchKernelGGL(camera_inverse_kernel, dim3((n + 63) / 64), dim3(64), 0, st, n, c2w, k, w2c,
                     k_inv);
}

}  // namespace ps
