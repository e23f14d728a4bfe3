// Gaussian adapter: raw network outputs -> the rasterizer's inputs, and its backward
// (SURVEY.md 8(f) rank 2).  One fused pass replaces, per step, a dozen PyTorch passes over the
// 300-byte-per-Gaussian SH tensor:
//   /root/reference/src/model/encoder/common/gaussian_adapter.py:48-95   scale map, quaternion,
//                                                                         covariance, means, mask
//   /root/reference/src/model/encoder/common/gaussians.py:8-41           quaternion -> R S S^T R^T
//   /root/reference/src/geometry/projection.py:65-108                    unproject / world rays
//   /root/reference/src/misc/sh_rotation.py:10-31                        e3nn Wigner-D rotation
// Outputs are written directly in the layouts ps_raster_forward consumes (PS_SH_G3K,
// PS_COV_33), Gaussian index g = ((view * r + pixel) * srf + surface) * spp + sample.
//
// A wave owns 16 consecutive (pixel, surface) entries of one view (7.5 KB of LDS: 20 waves per
// CU; with 64 entries the kernels ran at 5 waves per CU: forward 0.42 -> 0.31 ms, backward
// 0.72 -> 0.46 ms).  Their raw vectors (7 + 3K floats each, one contiguous 5.2 KB block at
// K = 25) are staged in LDS with coalesced
// 16-byte loads; each lane rotates its SH coefficients in place in that slab; the harmonics of
// the spp samples of a pixel are identical, so the slab is streamed out spp times into the
// (again contiguous) output block.  Per-view constants -- c2w rotation, origin, K^-1, the
// scale multiplier and the block-diagonal Wigner-D -- come from a tiny per-view kernel.
#include "raster_common.h"

namespace ps {

constexpr int kViewStride = 192;   // floats per view block
constexpr int kEntries = 16;       // (pixel, surface) entries per wave: LDS sets the occupancy
constexpr int kViewRot = 0, kViewOrigin = 9, kViewKinv = 12, kViewMult = 21, kViewD = 24;
__host__ __device__ constexpr int wigner_block(int l) {   // start of the (2l+1)^2 block
  return l == 0 ? 0 : l == 1 ? 1 : l == 2 ? 10 : l == 3 ? 35 : 84;
}

// ---------------------------------------------------------------------------------------
// per-view constants: one block per view, wave w builds the Wigner-D of degree w + 1 in LDS
// (double precision; Z(t) has two entries per row, so only P Z(b) P^T costs a real product)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
adapter_view_kernel(int n_views, int sh_degree, int img_h, int img_w,
                    const float* __restrict__ extrinsics, const float* __restrict__ intrinsics,
                    const double* __restrict__ conj /*P_1..P_4*/, float* __restrict__ views) {
  __shared__ double ang[3];
  __shared__ double buf[4][2][81];
  const int v = blockIdx.x;
  const float* e = extrinsics + 16 * v;
  const float* k = intrinsics + 9 * v;
  float* out = views + (size_t)v * kViewStride;
  if (threadIdx.x == 0) {
    double R[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) { R[3 * i + j] = e[4 * i + j]; out[kViewRot + 3 * i + j] = e[4 * i + j]; }
    for (int i = 0; i < 3; ++i) out[kViewOrigin + i] = e[4 * i + 3];
    {  // K^-1 (cofactors) and the scale multiplier 0.1 * sum(K[:2,:2]^-1 (1/w, 1/h))
      double m[9], inv[9];
      for (int i = 0; i < 9; ++i) m[i] = k[i];
      const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8],
                   c02 = m[3] * m[7] - m[4] * m[6];
      const double id = 1.0 / (m[0] * c00 + m[1] * c01 + m[2] * c02);
      inv[0] = c00 * id; inv[1] = (m[2] * m[7] - m[1] * m[8]) * id; inv[2] = (m[1] * m[5] - m[2] * m[4]) * id;
      inv[3] = c01 * id; inv[4] = (m[0] * m[8] - m[2] * m[6]) * id; inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
      inv[6] = c02 * id; inv[7] = (m[1] * m[6] - m[0] * m[7]) * id; inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
      for (int i = 0; i < 9; ++i) out[kViewKinv + i] = (float)inv[i];
      const double d2 = 1.0 / (m[0] * m[4] - m[1] * m[3]);
      const double px = 1.0 / img_w, py = 1.0 / img_h;
      const double mx = (m[4] * px - m[1] * py) * d2, my = (-m[3] * px + m[0] * py) * d2;
      out[kViewMult] = (float)(0.1 * (mx + my));
    }
    // e3nn.o3.matrix_to_angles: R = Y(alpha) X(beta) Y(gamma)
    double x0 = R[1], x1 = R[4], x2 = R[7];               // R (0, 1, 0)
    const double nx = sqrt(x0 * x0 + x1 * x1 + x2 * x2);
    x0 /= nx; x1 /= nx; x2 /= nx;
    x1 = fmin(1.0, fmax(-1.0, x1));
    const double beta = acos(x1), alpha = atan2(x0, x2);
    // first row of (Y(alpha) X(beta))^T R: column 0 of Y(a) X(b) is (cos a, 0, -sin a)
    const double ca = cos(alpha), sa = sin(alpha);
    const double gamma = atan2(ca * R[2] - sa * R[8], ca * R[0] - sa * R[6]);
    ang[0] = alpha; ang[1] = beta; ang[2] = gamma;
    out[kViewD] = 1.0f;
  }
  __syncthreads();
  const int l = (threadIdx.x >> 6) + 1, lane = threadIdx.x & 63;
  const bool on = l <= sh_degree && l <= 4;
  const int n = 2 * l + 1, nn = n * n;
  int off = 0;
  for (int j = 1; j < l; ++j) off += (2 * j + 1) * (2 * j + 1);
  const double* P = conj + off;
  double* A = buf[l - 1][0];
  double* B = buf[l - 1][1];
  // entry (i, j) of X Z(t) for a matrix X: column j mixes columns j and 2l - j
  auto times_z = [&](const double* X, int i, int j, double t) {
    const int m = j - l;                                 // Z[j][j] = cos |m| t
    if (m == 0) return X[i * n + j];
    const double c = cos(m * t), s_ = sin(m * t);        // Z[2l-j][j] = sin(m t) (sign of m)
    return X[i * n + j] * c + X[i * n + (2 * l - j)] * s_;
  };
  if (on) for (int idx = lane; idx < nn; idx += 64) A[idx] = times_z(P, idx / n, idx % n, ang[1]);
  __syncthreads();
  if (on)
    for (int idx = lane; idx < nn; idx += 64) {          // B = (P Z(b)) P^T
      const int i = idx / n, j = idx % n;
      double acc = 0.0;
      for (int kk = 0; kk < n; ++kk) acc += A[i * n + kk] * P[j * n + kk];
      B[idx] = acc;
    }
  __syncthreads();
  if (on) for (int idx = lane; idx < nn; idx += 64) A[idx] = times_z(B, idx / n, idx % n, ang[2]);
  __syncthreads();
  if (on)
    for (int idx = lane; idx < nn; idx += 64) {          // rows: Z(a) A
      const int i = idx / n, j = idx % n, m = l - i;     // Z[i][i] = cos |m| a, Z[i][2l-i] = sin(m a)
      double val = A[idx];
      if (m != 0) val = A[idx] * cos(m * ang[0]) + A[(2 * l - i) * n + j] * sin(m * ang[0]);
      out[kViewD + wigner_block(l) + idx] = (float)val;
    }
}

// ---------------------------------------------------------------------------------------
// shared per-lane math
// ---------------------------------------------------------------------------------------
struct AdapterDims {
  int n_views, rp /* pixels * surfaces per view */, spp, k /* SH per channel */;
  int srf, grid_w, grid_h;   // head layout only: surfaces per pixel and the pixel grid
};

// Head layout (SKIP = 2): an entry row is the whole output row of the encoder's `to_gaussians`
// linear layer for one (pixel, surface) -- [xy offset (2) | scale (3) | quaternion (4) | SH] --
// and the ray coordinate is the pixel centre moved by (sigmoid(offset) - 0.5) pixels
// (encoder_epipolar.py:155-164): no slice copy, no coordinate tensor.
__device__ __forceinline__ float2 head_coordinates(const AdapterDims& dm, int entry, float ox,
                                                   float oy, float* sig_out) {
  const int ray = entry / dm.srf, y = ray / dm.grid_w, x = ray - y * dm.grid_w;
  const float sx = 1.f / (1.f + expf(-ox)), sy = 1.f / (1.f + expf(-oy));
  sig_out[0] = sx; sig_out[1] = sy;
  const float pw = 1.f / (float)dm.grid_w, ph = 1.f / (float)dm.grid_h;
  return make_float2(((float)x + 0.5f) / (float)dm.grid_w + (sx - 0.5f) * pw,
                     ((float)y + 0.5f) / (float)dm.grid_h + (sy - 0.5f) * ph);
}

struct LaneGeom {
  float base[3];      // scale before depth * multiplier
  float sig[3];       // sigmoid(raw scale)
  float q[4], qn[4];  // raw and normalised quaternion (xyzw)
  float qnorm;
  float t;            // two_s = 2 / (|qn|^2 + 1e-8)
  float M[9];         // c2w rotation * quaternion rotation
  float dir[3], dirw[3], dnorm;
};

__device__ __forceinline__ void lane_geometry(const float* __restrict__ vw, const float* raw7,
                                              float cx, float cy, float smin, float smax,
                                              float eps, LaneGeom& g) {
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    g.sig[i] = 1.0f / (1.0f + __expf(-raw7[i]));
    g.base[i] = smin + (smax - smin) * g.sig[i];
  }
  float n2 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) { g.q[i] = raw7[3 + i]; n2 = fmaf(g.q[i], g.q[i], n2); }
  g.qnorm = sqrtf(n2);
  const float inv = 1.0f / (g.qnorm + eps);
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) { g.qn[i] = g.q[i] * inv; s2 = fmaf(g.qn[i], g.qn[i], s2); }
  g.t = 2.0f / (s2 + 1e-8f);
  const float qi = g.qn[0], qj = g.qn[1], qk = g.qn[2], qr = g.qn[3], t = g.t;
  const float Rq[9] = {1.f - t * (qj * qj + qk * qk), t * (qi * qj - qk * qr), t * (qi * qk + qj * qr),
                       t * (qi * qj + qk * qr), 1.f - t * (qi * qi + qk * qk), t * (qj * qk - qi * qr),
                       t * (qi * qk - qj * qr), t * (qj * qk + qi * qr), 1.f - t * (qi * qi + qj * qj)};
  const float* C = vw + kViewRot;
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b)
      g.M[3 * a + b] = C[3 * a] * Rq[b] + C[3 * a + 1] * Rq[3 + b] + C[3 * a + 2] * Rq[6 + b];
  const float* Ki = vw + kViewKinv;
  float dc[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) dc[a] = Ki[3 * a] * cx + Ki[3 * a + 1] * cy + Ki[3 * a + 2];
  g.dnorm = sqrtf(dc[0] * dc[0] + dc[1] * dc[1] + dc[2] * dc[2]);
#pragma unroll
  for (int a = 0; a < 3; ++a) g.dir[a] = dc[a] / g.dnorm;
#pragma unroll
  for (int a = 0; a < 3; ++a)
    g.dirw[a] = C[3 * a] * g.dir[0] + C[3 * a + 1] * g.dir[1] + C[3 * a + 2] * g.dir[2];
}

__device__ __forceinline__ float sh_mask_of(int l) {   // gaussian_adapter.py:41-46
  return l == 0 ? 1.0f : l == 1 ? 0.025f : l == 2 ? 0.00625f : l == 3 ? 0.0015625f : 0.000390625f;
}

// rows of one contiguous [rows][K3] block <-> lanes: full 64-float pieces first, then the
// (K3 - 64)-float tails packed 64 to an instruction (K3 = 3K = 75 at degree 4)
template <typename F>
__device__ __forceinline__ void for_each_row_element(int rows, int k3, int lane, F f) {
  if (k3 <= kWave) {
    const int per = kWave / k3;                    // rows per instruction
    for (int r0 = 0; r0 < rows; r0 += per) {
      const int r = r0 + lane / k3, j = lane % k3;
      if (lane < per * k3 && r < rows) f(r, j);
    }
    return;
  }
  for (int r = 0; r < rows; ++r) f(r, lane);
  const int tail = k3 - kWave;
  for (int e0 = 0; e0 < rows * tail; e0 += kWave) {
    const int e = e0 + lane;
    if (e < rows * tail) f(e / tail, kWave + e % tail);
  }
}

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
template <int DEG, int SKIP>
__global__ void __launch_bounds__(kWave)
adapter_forward_kernel(AdapterDims dm, float smin, float smax, float eps,
                       const float* __restrict__ views, const float* __restrict__ coords,
                       const float* __restrict__ depths, const float* __restrict__ raw,
                       float* __restrict__ means, float* __restrict__ cov,
                       float* __restrict__ harmonics) {
  constexpr int K = (DEG + 1) * (DEG + 1), K3 = 3 * K, CIN = SKIP + 7 + K3;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* rawS = lds;                                   // [kEntries][CIN]
  float* geoS = lds + kEntries * CIN;                  // [kEntries * spp][12]: cov 9 + mean 3
  const int lane = threadIdx.x;
  const int view = blockIdx.y;
  const int p0 = blockIdx.x * kEntries;
  const int rows = min(kEntries, dm.rp - p0);             // entries of this wave
  const size_t e0 = (size_t)view * dm.rp + p0;         // first (view, pixel, surface) entry
  const float* vw = views + (size_t)view * kViewStride;

  stage_slab<(kEntries * CIN + 255) / 256>(raw + e0 * CIN, rawS, rows * CIN, lane);
  wave_lds_sync();

  if (lane < rows) {
    float* mine = rawS + lane * CIN + SKIP;
    float raw7[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) raw7[i] = mine[i];
    float2 cxy;
    if (SKIP) {
      float sig_xy[2];
      cxy = head_coordinates(dm, p0 + lane, mine[-2], mine[-1], sig_xy);
    } else {
      cxy = *reinterpret_cast<const float2*>(coords + 2 * (e0 + lane));
    }
    LaneGeom g;
    lane_geometry(vw, raw7, cxy.x, cxy.y, smin, smax, eps, g);
    const float mult = vw[kViewMult];
    for (int k = 0; k < dm.spp; ++k) {
      const float depth = depths[(e0 + lane) * dm.spp + k];
      float s2[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) { const float s = g.base[i] * depth * mult; s2[i] = s * s; }
      float* o = geoS + (lane * dm.spp + k) * 12;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
          o[3 * a + b] = g.M[3 * a] * g.M[3 * b] * s2[0] + g.M[3 * a + 1] * g.M[3 * b + 1] * s2[1] +
                         g.M[3 * a + 2] * g.M[3 * b + 2] * s2[2];
#pragma unroll
      for (int a = 0; a < 3; ++a) o[9 + a] = fmaf(g.dirw[a], depth, vw[kViewOrigin + a]);
    }
  }
  // SH: mask, then the block-diagonal Wigner-D of this view, in place in the slab.  Lane =
  // (entry, colour channel): 3 x kEntries lanes work here instead of kEntries (the per-entry
  // geometry above runs on 16 lanes; these 165 FMAs per channel were 55 % of the kernel's
  // VALU time when one lane did all three channels)
  if (lane < 3 * kEntries && (lane % kEntries) < rows) {
    const int c = lane / kEntries;
    float* sh = rawS + (lane % kEntries) * CIN + SKIP + 7 + c * K;
    const float* D = vw + kViewD;
    float in[K], out[K];
#pragma unroll
    for (int j = 0; j < K; ++j) in[j] = sh[j];
    out[0] = in[0];
#pragma unroll
    for (int l = 1; l <= DEG; ++l) {
      const int n = 2 * l + 1, b0 = l * l;
      const float* Dl = D + wigner_block(l);
      const float mk = sh_mask_of(l);
#pragma unroll
      for (int i = 0; i < n; ++i) {
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < n; ++j) acc = fmaf(Dl[i * n + j], in[b0 + j], acc);
        out[b0 + i] = acc * mk;
      }
    }
#pragma unroll
    for (int j = 0; j < K; ++j) sh[j] = out[j];
  }
  wave_lds_sync();

  // stream out: the wave's output blocks are contiguous
  const size_t g0 = e0 * dm.spp;                       // first Gaussian of the wave
  const int n_g = rows * dm.spp;
  for (int i = lane; i < n_g * 9; i += kWave) cov[g0 * 9 + i] = geoS[(i / 9) * 12 + i % 9];
  for (int i = lane; i < n_g * 3; i += kWave) means[g0 * 3 + i] = geoS[(i / 3) * 12 + 9 + i % 3];
  float* hout = harmonics + g0 * K3;
  for_each_row_element(n_g, K3, lane, [&](int r, int j) {
    hout[(size_t)r * K3 + j] = rawS[(r / dm.spp) * CIN + SKIP + 7 + j];
  });
}

// ---------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------
template <int DEG, int SKIP>
__global__ void __launch_bounds__(kWave)
adapter_backward_kernel(AdapterDims dm, float smin, float smax, float eps,
                        const float* __restrict__ views, const float* __restrict__ coords,
                        const float* __restrict__ depths, const float* __restrict__ raw,
                        const float* __restrict__ d_means, const float* __restrict__ d_cov,
                        const float* __restrict__ d_harmonics, float* __restrict__ d_raw,
                        float* __restrict__ d_depths, float* __restrict__ d_coords) {
  constexpr int K = (DEG + 1) * (DEG + 1), K3 = 3 * K, CIN = SKIP + 7 + K3;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* outS = lds;                                   // [kEntries][CIN]: d_raw staging
  float* geoS = lds + kEntries * CIN;                  // [kEntries * spp][12]: d_cov 9 + d_mean 3
  const int lane = threadIdx.x;
  const int view = blockIdx.y;
  const int p0 = blockIdx.x * kEntries;
  const int rows = min(kEntries, dm.rp - p0);
  const size_t e0 = (size_t)view * dm.rp + p0;
  const float* vw = views + (size_t)view * kViewStride;
  const size_t g0 = e0 * dm.spp;
  const int n_g = rows * dm.spp;

  // incoming gradients: coalesced into LDS, eight loads in flight per lane (a load -> LDS store
  // per iteration is a chain of memory latencies: 100 us per wave in the first version); the
  // harmonics of the spp samples are summed on the way in (they share one raw vector)
  {
    constexpr int U = 8;
    for (int i0 = lane; i0 < n_g * 9; i0 += kWave * U) {
      float r[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { const int i = i0 + u * kWave; r[u] = i < n_g * 9 ? d_cov[g0 * 9 + i] : 0.f; }
#pragma unroll
      for (int u = 0; u < U; ++u) { const int i = i0 + u * kWave; if (i < n_g * 9) geoS[(i / 9) * 12 + i % 9] = r[u]; }
    }
    for (int i0 = lane; i0 < n_g * 3; i0 += kWave * U) {
      float r[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { const int i = i0 + u * kWave; r[u] = i < n_g * 3 ? d_means[g0 * 3 + i] : 0.f; }
#pragma unroll
      for (int u = 0; u < U; ++u) { const int i = i0 + u * kWave; if (i < n_g * 3) geoS[(i / 3) * 12 + 9 + i % 3] = r[u]; }
    }
    // all rows x samples of the wave are one contiguous [n_g][K3] block: walk it linearly,
    // element o belongs to entry o / (spp K3), coefficient o % K3
    const float* hin = d_harmonics + g0 * K3;
    const int total = n_g * K3, per_entry = dm.spp * K3;
    for (int i = lane; i < rows * CIN; i += kWave) outS[i] = 0.f;
    wave_lds_sync();
    for (int o0 = lane; o0 < total; o0 += kWave * U) {
      float r[U];
#pragma unroll
      for (int u = 0; u < U; ++u) { const int o = o0 + u * kWave; r[u] = o < total ? hin[o] : 0.f; }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int o = o0 + u * kWave;
        if (o < total) {
          const int ent = o / per_entry, j = (o - ent * per_entry) % K3;
          // lanes of one instruction hit distinct (entry, j) unless spp rows of an entry fall
          // into the same 64-element window (K3 >= 64: they never do; else serialise)
          if (K3 >= kWave) outS[ent * CIN + SKIP + 7 + j] += r[u];
          else atomicAdd(&outS[ent * CIN + SKIP + 7 + j], r[u]);
        }
      }
    }
  }
  wave_lds_sync();

  if (lane < rows) {
    float* mine = outS + lane * CIN + SKIP;
    float raw7[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) raw7[i] = raw[(e0 + lane) * CIN + SKIP + i];
    float2 cxy;
    float sig_xy[2] = {0.f, 0.f};
    if (SKIP) {
      cxy = head_coordinates(dm, p0 + lane, raw[(e0 + lane) * CIN], raw[(e0 + lane) * CIN + 1],
                             sig_xy);
    } else {
      cxy = *reinterpret_cast<const float2*>(coords + 2 * (e0 + lane));
    }
    LaneGeom g;
    lane_geometry(vw, raw7, cxy.x, cxy.y, smin, smax, eps, g);
    const float mult = vw[kViewMult];
    float dM[9], dbase[3] = {0.f, 0.f, 0.f}, ddw[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 9; ++i) dM[i] = 0.f;
    for (int k = 0; k < dm.spp; ++k) {
      const float depth = depths[(e0 + lane) * dm.spp + k];
      const float* gi = geoS + (lane * dm.spp + k) * 12;
      float Gs[9];                                       // Gc + Gc^T
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) Gs[3 * a + b] = gi[3 * a + b] + gi[3 * b + a];
      float GM[9];                                       // (Gc + Gc^T) M
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int i = 0; i < 3; ++i)
          GM[3 * a + i] = Gs[3 * a] * g.M[i] + Gs[3 * a + 1] * g.M[3 + i] + Gs[3 * a + 2] * g.M[6 + i];
      float dd = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float s = g.base[i] * depth * mult, s2 = s * s;
        // dS2_i = (M^T Gc M)_ii = 0.5 * sum_a M_ai (Gs M)_ai
        const float ds2 = 0.5f * (g.M[i] * GM[i] + g.M[3 + i] * GM[3 + i] + g.M[6 + i] * GM[6 + i]);
        const float dsc = 2.f * s * ds2;
        dbase[i] = fmaf(dsc, depth * mult, dbase[i]);
        dd = fmaf(dsc, g.base[i] * mult, dd);
#pragma unroll
        for (int a = 0; a < 3; ++a) dM[3 * a + i] = fmaf(GM[3 * a + i], s2, dM[3 * a + i]);
      }
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        dd = fmaf(g.dirw[a], gi[9 + a], dd);
        ddw[a] = fmaf(depth, gi[9 + a], ddw[a]);
      }
      d_depths[(e0 + lane) * dm.spp + k] = dd;
    }
    const float* C = vw + kViewRot;
    // scale features
#pragma unroll
    for (int i = 0; i < 3; ++i) mine[i] = dbase[i] * (smax - smin) * g.sig[i] * (1.f - g.sig[i]);
    // quaternion: G = C^T dM, then through the matrix formula and the normalisation
    float G[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b)
        G[3 * a + b] = C[a] * dM[b] + C[3 + a] * dM[3 + b] + C[6 + a] * dM[6 + b];
    const float qi = g.qn[0], qj = g.qn[1], qk = g.qn[2], qr = g.qn[3], t = g.t;
    const float dt = -G[0] * (qj * qj + qk * qk) + G[1] * (qi * qj - qk * qr) + G[2] * (qi * qk + qj * qr) +
                     G[3] * (qi * qj + qk * qr) - G[4] * (qi * qi + qk * qk) + G[5] * (qj * qk - qi * qr) +
                     G[6] * (qi * qk - qj * qr) + G[7] * (qj * qk + qi * qr) - G[8] * (qi * qi + qj * qj);
    float dqn[4];
    dqn[0] = t * (G[1] * qj + G[2] * qk + G[3] * qj - 2.f * G[4] * qi - G[5] * qr + G[6] * qk + G[7] * qr - 2.f * G[8] * qi);
    dqn[1] = t * (-2.f * G[0] * qj + G[1] * qi + G[2] * qr + G[3] * qi + G[5] * qk - G[6] * qr + G[7] * qk - 2.f * G[8] * qj);
    dqn[2] = t * (-2.f * G[0] * qk - G[1] * qr + G[2] * qi + G[3] * qr - 2.f * G[4] * qk + G[5] * qj + G[6] * qi + G[7] * qj);
    dqn[3] = t * (-G[1] * qk + G[2] * qj + G[3] * qk - G[5] * qi - G[6] * qj + G[7] * qi);
    float dot = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) { dqn[a] = fmaf(-dt * t * t, g.qn[a], dqn[a]); dot = fmaf(g.q[a], dqn[a], dot); }
    const float den = g.qnorm + eps;
    const float cfac = g.qnorm > 0.f ? dot / (g.qnorm * den * den) : 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) mine[3 + a] = dqn[a] / den - g.q[a] * cfac;
    // coordinates: through C, the normalisation of the camera ray and K^-1
    float dd3[3], dotd = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      dd3[a] = C[a] * ddw[0] + C[3 + a] * ddw[1] + C[6 + a] * ddw[2];
      dotd = fmaf(g.dir[a], dd3[a], dotd);
    }
    const float* Ki = vw + kViewKinv;
    float dcx = 0.f, dcy = 0.f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float ddc = (dd3[a] - g.dir[a] * dotd) / g.dnorm;
      dcx = fmaf(Ki[3 * a], ddc, dcx);
      dcy = fmaf(Ki[3 * a + 1], ddc, dcy);
    }
    if (SKIP) {   // through (sigmoid(offset) - 0.5) * pixel size
      mine[-2] = dcx * sig_xy[0] * (1.f - sig_xy[0]) / (float)dm.grid_w;
      mine[-1] = dcy * sig_xy[1] * (1.f - sig_xy[1]) / (float)dm.grid_h;
    } else {
      *reinterpret_cast<float2*>(d_coords + 2 * (e0 + lane)) = make_float2(dcx, dcy);
    }
  }
  // SH: D^T and the mask, in place; lane = (entry, colour channel) as in the forward
  if (lane < 3 * kEntries && (lane % kEntries) < rows) {
    const int c = lane / kEntries;
    float* sh = outS + (lane % kEntries) * CIN + SKIP + 7 + c * K;
    const float* D = vw + kViewD;
    float in[K], out[K];
#pragma unroll
    for (int j = 0; j < K; ++j) in[j] = sh[j];
    out[0] = in[0];
#pragma unroll
    for (int l = 1; l <= DEG; ++l) {
      const int n = 2 * l + 1, b0 = l * l;
      const float* Dl = D + wigner_block(l);
      const float mk = sh_mask_of(l);
#pragma unroll
      for (int j = 0; j < n; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < n; ++i) acc = fmaf(Dl[i * n + j], in[b0 + i], acc);
        out[b0 + j] = acc * mk;
      }
    }
#pragma unroll
    for (int j = 0; j < K; ++j) sh[j] = out[j];
  }
  wave_lds_sync();
  float* dst = d_raw + e0 * CIN;
  for (int i = lane; i < rows * CIN; i += kWave) dst[i] = outS[i];
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
int launch_adapter_views(int n_views, int sh_degree, int img_h, int img_w, const float* extrinsics,
                         const float* intrinsics, const double* conj, float* views,
                         hipStream_t st) {
  hipLaunchKernelGGL(adapter_view_kernel, dim3(n_views), dim3(256), 0, st, n_views,
                     sh_degree, img_h, img_w, extrinsics, intrinsics, conj, views);
  return PS_OK;
}

static size_t adapter_lds(int deg, int spp, int skip) {
  const int cin = skip + 7 + 3 * (deg + 1) * (deg + 1);
  return (size_t)kEntries * (cin + 12 * spp) * sizeof(float);
}

// head = {surfaces, grid_w, grid_h} selects the head layout (coords unused), nullptr the plain one
int launch_adapter_forward(int n_views, int rp, int spp, int sh_degree, float smin, float smax,
                           float eps, const float* views, const float* coords,
                           const float* depths, const float* raw, float* means, float* cov,
                           float* harmonics, const int* head, hipStream_t st) {
  const int skip = head ? 2 : 0;
  if (sh_degree < 0 || sh_degree > 4 || spp < 1 || adapter_lds(sh_degree, spp, skip) > 160 * 1024)
    return PS_ERR_UNSUPPORTED;
  const AdapterDims dm{n_views, rp, spp, (sh_degree + 1) * (sh_degree + 1),
                       head ? head[0] : 1, head ? head[1] : 1, head ? head[2] : 1};
  dim3 grid((rp + kEntries - 1) / kEntries, n_views), block(kWave);
  const size_t sm = adapter_lds(sh_degree, spp, skip);
#define PS_GO2(D, SK)                                                                         \
  do {                                                                                        \
    static const bool lds_ok_ = (hipFuncSetAttribute((const void*)adapter_forward_kernel<D, SK>,\
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true); (void)lds_ok_;           \
    hipLaunchKernelGGL((adapter_forward_kernel<D, SK>), grid, block, sm, st, dm, smin, smax,  \
                       eps, views, coords, depths, raw, means, cov, harmonics);               \
  } while (0)
#define PS_GO(D) do { if (skip) PS_GO2(D, 2); else PS_GO2(D, 0); } while (0)
  switch (sh_degree) {
    case 0: PS_GO(0); break; case 1: PS_GO(1); break; case 2: PS_GO(2); break;
    case 3: PS_GO(3); break; default: PS_GO(4); break;
  }
#undef PS_GO
#undef PS_GO2
  return PS_OK;
}

int launch_adapter_backward(int n_views, int rp, int spp, int sh_degree, float smin, float smax,
                            float eps, const float* views, const float* coords,
                            const float* depths, const float* raw, const float* d_means,
                            const float* d_cov, const float* d_harmonics, float* d_raw,
                            float* d_depths, float* d_coords, const int* head, hipStream_t st) {
  const int skip = head ? 2 : 0;
  if (sh_degree < 0 || sh_degree > 4 || spp < 1 || adapter_lds(sh_degree, spp, skip) > 160 * 1024)
    return PS_ERR_UNSUPPORTED;
  const AdapterDims dm{n_views, rp, spp, (sh_degree + 1) * (sh_degree + 1),
                       head ? head[0] : 1, head ? head[1] : 1, head ? head[2] : 1};
  dim3 grid((rp + kEntries - 1) / kEntries, n_views), block(kWave);
  const size_t sm = adapter_lds(sh_degree, spp, skip);
#define PS_GO2(D, SK)                                                                         \
  do {                                                                                        \
    static const bool lds_ok_ = (hipFuncSetAttribute((const void*)adapter_backward_kernel<D, SK>,\
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true); (void)lds_ok_;           \
    hipLaunchKernelGGL((adapter_backward_kernel<D, SK>), grid, block, sm, st, dm, smin, smax, \
                       eps, views, coords, depths, raw, d_means, d_cov, d_harmonics, d_raw,   \
                       d_depths, d_coords);                                                   \
  } while (0)
#define PS_GO(D) do { if (skip) PS_GO2(D, 2); else PS_GO2(D, 0); } while (0)
  switch (sh_degree) {
    case 0: PS_GO(0); break; case 1: PS_GO(1); break; case 2: PS_GO(2); break;
    case 3: PS_GO(3); break; default: PS_GO(4); break;
  }
#undef PS_GO
#undef PS_GO2
  return PS_OK;
}

}  // namespace ps
