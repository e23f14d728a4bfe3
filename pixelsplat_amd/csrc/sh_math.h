// Real spherical-harmonics basis up to degree 4 in the polynomial form the rasterizer
// boundary uses (degree 4 is required by the reference:
// /root/reference/src/model/decoder/cuda_splatting.py:73-75, config epipolar.yaml sh_degree 4).
// colour_c = sum_k basis[k] * sh[k][c] + 0.5, clamped at 0.
#pragma once
#include <hip/hip_runtime.h>

namespace ps {

#define PS_C0 0.28209479177387814f
#define PS_C1 0.4886025119029199f
#define PS_C2_0 1.0925484305920792f
#define PS_C2_1 -1.0925484305920792f
#define PS_C2_2 0.31539156525252005f
#define PS_C2_3 -1.0925484305920792f
#define PS_C2_4 0.5462742152960396f
#define PS_C3_0 -0.5900435899266435f
#define PS_C3_1 2.890611442640554f
#define PS_C3_2 -0.4570457994644658f
#define PS_C3_3 0.3731763325901154f
#define PS_C3_4 -0.4570457994644658f
#define PS_C3_5 1.445305721320277f
#define PS_C3_6 -0.5900435899266435f
#define PS_C4_0 2.5033429417967046f
#define PS_C4_1 -1.7701307697799304f
#define PS_C4_2 0.9461746957575601f
#define PS_C4_3 -0.6690465435572892f
#define PS_C4_4 0.10578554691520431f
#define PS_C4_5 -0.6690465435572892f
#define PS_C4_6 0.47308734787878004f
#define PS_C4_7 -1.7701307697799304f
#define PS_C4_8 0.6258357354491761f

// DEG is a compile-time bound (loops fully unroll); `deg` the runtime active degree.
__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float* b) {
  b[0] = PS_C0;
  if (deg < 1) return;
  b[1] = -PS_C1 * y;
  b[2] = PS_C1 * z;
  b[3] = -PS_C1 * x;
  if (deg < 2) return;
  const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
  b[4] = PS_C2_0 * xy;
  b[5] = PS_C2_1 * yz;
  b[6] = PS_C2_2 * (2.0f * zz - xx - yy);
  b[7] = PS_C2_3 * xz;
  b[8] = PS_C2_4 * (xx - yy);
  if (deg < 3) return;
  b[9] = PS_C3_0 * y * (3.0f * xx - yy);
  b[10] = PS_C3_1 * xy * z;
  b[11] = PS_C3_2 * y * (4.0f * zz - xx - yy);
  b[12] = PS_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
  b[13] = PS_C3_4 * x * (4.0f * zz - xx - yy);
  b[14] = PS_C3_5 * z * (xx - yy);
  b[15] = PS_C3_6 * x * (xx - 3.0f * yy);
  if (deg < 4) return;
  b[16] = PS_C4_0 * xy * (xx - yy);
  b[17] = PS_C4_1 * yz * (3.0f * xx - yy);
  b[18] = PS_C4_2 * xy * (7.0f * zz - 1.0f);
  b[19] = PS_C4_3 * yz * (7.0f * zz - 3.0f);
  b[20] = PS_C4_4 * (zz * (35.0f * zz - 30.0f) + 3.0f);
  b[21] = PS_C4_5 * xz * (7.0f * zz - 3.0f);
  b[22] = PS_C4_6 * (xx - yy) * (7.0f * zz - 1.0f);
  b[23] = PS_C4_7 * xz * (xx - 3.0f * yy);
  b[24] = PS_C4_8 * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy));
}

// partial derivatives of the polynomials above w.r.t. x, y, z (independent variables)
__device__ __forceinline__ void sh_basis_grad(int deg, float x, float y, float z, float* bx,
                                              float* by, float* bz) {
#pragma unroll
  for (int i = 0; i < 25; ++i) { bx[i] = 0.f; by[i] = 0.f; bz[i] = 0.f; }
  if (deg < 1) return;
  by[1] = -PS_C1; bz[2] = PS_C1; bx[3] = -PS_C1;
  if (deg < 2) return;
  const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
  bx[4] = PS_C2_0 * y; by[4] = PS_C2_0 * x;
  by[5] = PS_C2_1 * z; bz[5] = PS_C2_1 * y;
  bx[6] = PS_C2_2 * (-2.0f * x); by[6] = PS_C2_2 * (-2.0f * y); bz[6] = PS_C2_2 * (4.0f * z);
  bx[7] = PS_C2_3 * z; bz[7] = PS_C2_3 * x;
  bx[8] = PS_C2_4 * (2.0f * x); by[8] = PS_C2_4 * (-2.0f * y);
  if (deg < 3) return;
  bx[9] = PS_C3_0 * (6.0f * xy); by[9] = PS_C3_0 * (3.0f * xx - 3.0f * yy);
  bx[10] = PS_C3_1 * yz; by[10] = PS_C3_1 * xz; bz[10] = PS_C3_1 * xy;
  bx[11] = PS_C3_2 * (-2.0f * xy); by[11] = PS_C3_2 * (4.0f * zz - xx - 3.0f * yy);
  bz[11] = PS_C3_2 * (8.0f * yz);
  bx[12] = PS_C3_3 * (-6.0f * xz); by[12] = PS_C3_3 * (-6.0f * yz);
  bz[12] = PS_C3_3 * (6.0f * zz - 3.0f * xx - 3.0f * yy);
  bx[13] = PS_C3_4 * (4.0f * zz - 3.0f * xx - yy); by[13] = PS_C3_4 * (-2.0f * xy);
  bz[13] = PS_C3_4 * (8.0f * xz);
  bx[14] = PS_C3_5 * (2.0f * xz); by[14] = PS_C3_5 * (-2.0f * yz); bz[14] = PS_C3_5 * (xx - yy);
  bx[15] = PS_C3_6 * (3.0f * xx - 3.0f * yy); by[15] = PS_C3_6 * (-6.0f * xy);
  if (deg < 4) return;
  bx[16] = PS_C4_0 * (3.0f * xx * y - yy * y); by[16] = PS_C4_0 * (xx * x - 3.0f * x * yy);
  bx[17] = PS_C4_1 * (6.0f * xy * z); by[17] = PS_C4_1 * (z * (3.0f * xx - 3.0f * yy));
  bz[17] = PS_C4_1 * (y * (3.0f * xx - yy));
  bx[18] = PS_C4_2 * (y * (7.0f * zz - 1.0f)); by[18] = PS_C4_2 * (x * (7.0f * zz - 1.0f));
  bz[18] = PS_C4_2 * (14.0f * xy * z);
  by[19] = PS_C4_3 * (z * (7.0f * zz - 3.0f)); bz[19] = PS_C4_3 * (y * (21.0f * zz - 3.0f));
  bz[20] = PS_C4_4 * (140.0f * zz * z - 60.0f * z);
  bx[21] = PS_C4_5 * (z * (7.0f * zz - 3.0f)); bz[21] = PS_C4_5 * (x * (21.0f * zz - 3.0f));
  bx[22] = PS_C4_6 * (2.0f * x * (7.0f * zz - 1.0f));
  by[22] = PS_C4_6 * (-2.0f * y * (7.0f * zz - 1.0f));
  bz[22] = PS_C4_6 * (14.0f * z * (xx - yy));
  bx[23] = PS_C4_7 * (z * (3.0f * xx - 3.0f * yy)); by[23] = PS_C4_7 * (-6.0f * xy * z);
  bz[23] = PS_C4_7 * (x * (xx - 3.0f * yy));
  bx[24] = PS_C4_8 * (4.0f * xx * x - 12.0f * x * yy);
  by[24] = PS_C4_8 * (4.0f * yy * y - 12.0f * xx * y);
}

}  // namespace ps

namespace ps {

// (ddx, ddy, ddz) = sum_k w[k] * d basis_k / d(x, y, z) without materialising the 75
// partial derivatives (keeps the SH backward kernel's live registers low).
__device__ __forceinline__ void sh_grad_dot(int deg, float x, float y, float z, const float* w,
                                            float& ddx, float& ddy, float& ddz) {
  ddx = 0.f; ddy = 0.f; ddz = 0.f;
#define PS_ACC(k, gx_, gy_, gz_) \
  do { ddx = fmaf((gx_), w[k], ddx); ddy = fmaf((gy_), w[k], ddy); ddz = fmaf((gz_), w[k], ddz); } while (0)
  if (deg < 1) return;
  PS_ACC(1, 0.f, -PS_C1, 0.f);
  PS_ACC(2, 0.f, 0.f, PS_C1);
  PS_ACC(3, -PS_C1, 0.f, 0.f);
  if (deg < 2) return;
  const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
  PS_ACC(4, PS_C2_0 * y, PS_C2_0 * x, 0.f);
  PS_ACC(5, 0.f, PS_C2_1 * z, PS_C2_1 * y);
  PS_ACC(6, PS_C2_2 * (-2.0f * x), PS_C2_2 * (-2.0f * y), PS_C2_2 * (4.0f * z));
  PS_ACC(7, PS_C2_3 * z, 0.f, PS_C2_3 * x);
  PS_ACC(8, PS_C2_4 * (2.0f * x), PS_C2_4 * (-2.0f * y), 0.f);
  if (deg < 3) return;
  PS_ACC(9, PS_C3_0 * (6.0f * xy), PS_C3_0 * (3.0f * xx - 3.0f * yy), 0.f);
  PS_ACC(10, PS_C3_1 * yz, PS_C3_1 * xz, PS_C3_1 * xy);
  PS_ACC(11, PS_C3_2 * (-2.0f * xy), PS_C3_2 * (4.0f * zz - xx - 3.0f * yy), PS_C3_2 * (8.0f * yz));
  PS_ACC(12, PS_C3_3 * (-6.0f * xz), PS_C3_3 * (-6.0f * yz),
         PS_C3_3 * (6.0f * zz - 3.0f * xx - 3.0f * yy));
  PS_ACC(13, PS_C3_4 * (4.0f * zz - 3.0f * xx - yy), PS_C3_4 * (-2.0f * xy), PS_C3_4 * (8.0f * xz));
  PS_ACC(14, PS_C3_5 * (2.0f * xz), PS_C3_5 * (-2.0f * yz), PS_C3_5 * (xx - yy));
  PS_ACC(15, PS_C3_6 * (3.0f * xx - 3.0f * yy), PS_C3_6 * (-6.0f * xy), 0.f);
  if (deg < 4) return;
  PS_ACC(16, PS_C4_0 * (3.0f * xx * y - yy * y), PS_C4_0 * (xx * x - 3.0f * x * yy), 0.f);
  PS_ACC(17, PS_C4_1 * (6.0f * xy * z), PS_C4_1 * (z * (3.0f * xx - 3.0f * yy)),
         PS_C4_1 * (y * (3.0f * xx - yy)));
  PS_ACC(18, PS_C4_2 * (y * (7.0f * zz - 1.0f)), PS_C4_2 * (x * (7.0f * zz - 1.0f)),
         PS_C4_2 * (14.0f * xy * z));
  PS_ACC(19, 0.f, PS_C4_3 * (z * (7.0f * zz - 3.0f)), PS_C4_3 * (y * (21.0f * zz - 3.0f)));
  PS_ACC(20, 0.f, 0.f, PS_C4_4 * (140.0f * zz * z - 60.0f * z));
  PS_ACC(21, PS_C4_5 * (z * (7.0f * zz - 3.0f)), 0.f, PS_C4_5 * (x * (21.0f * zz - 3.0f)));
  PS_ACC(22, PS_C4_6 * (2.0f * x * (7.0f * zz - 1.0f)), PS_C4_6 * (-2.0f * y * (7.0f * zz - 1.0f)),
         PS_C4_6 * (14.0f * z * (xx - yy)));
  PS_ACC(23, PS_C4_7 * (z * (3.0f * xx - 3.0f * yy)), PS_C4_7 * (-6.0f * xy * z),
         PS_C4_7 * (x * (xx - 3.0f * yy)));
  PS_ACC(24, PS_C4_8 * (4.0f * xx * x - 12.0f * x * yy), PS_C4_8 * (4.0f * yy * y - 12.0f * xx * y),
         0.f);
#undef PS_ACC
}

}  // namespace ps
