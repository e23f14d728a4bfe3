// Epipolar sampling geometry (path A): per (batch, casting view, other view, ray) the epipolar
// segment of the ray in the other view, then per sample the 2-D sample point, the depth of
// the sample along the casting ray and its relative disparity -- everything
// EpipolarSampler.forward + get_depth compute before the feature gather:
//   /root/reference/src/model/encoder/epipolar/epipolar_sampler.py:51-123   (a5)
//   /root/reference/src/geometry/epipolar_lines.py:55-251, 264-292           (a4, a6)
//   /root/reference/src/geometry/projection.py:47-56, 74-137, 176-230        (a1, a2, a6)
//   /root/reference/src/model/encoder/epipolar/conversions.py:17-27          (a7)
// in ONE launch instead of ~60 elementwise kernels, 16 boolean-mask scatters (each a host
// sync) and a batched 3x3 lstsq over 1.8 M systems.
//
// BUILT WITH -ffp-contract=off.  The integer-valued outputs (overlap flag, frame-hit
// selectors, and downstream the bilinear corner indices, which are a function of xy_sample)
// must match the reference bit-for-bit, so every expression follows the reference's
// evaluation order: matrix-vector products are sequential FMA chains (what the CPU einsum
// does), everything else is one IEEE rounding per elementwise op.
#include "raster_common.h"

namespace ps {

struct V3 { float x, y, z; };

__device__ __forceinline__ float chain3(float m0, float m1, float m2, float x0, float x1, float x2) {
  return fmaf(m2, x2, fmaf(m1, x1, m0 * x0));
}
__device__ __forceinline__ float chain4(float m0, float m1, float m2, float m3, float x0, float x1,
                                        float x2, float x3) {
  return fmaf(m3, x3, fmaf(m2, x2, fmaf(m1, x1, m0 * x0)));
}

constexpr float kEps = 1e-6f;            // float32(1e-6), the reference's comparison slack
constexpr float kOnePlusEps = 1.000001f; // float32(1 + 1e-6)

__device__ __forceinline__ bool in_bounds(float x, float y) {
  return (x >= -kEps) && (y >= -kEps) && (x <= kOnePlusEps) && (y <= kOnePlusEps);
}

struct Hit { float t, x, y; bool valid; };

// intersection of the camera-space ray (o, d) with the image border {dim = value}
__device__ __forceinline__ Hit frame_hit(const float* k, V3 o, V3 d, int dim, float value) {
  const float fs = dim == 0 ? k[0] : k[4], fo = dim == 0 ? k[4] : k[0];
  const float cs = dim == 0 ? k[2] : k[5], co = dim == 0 ? k[5] : k[2];
  const float os = dim == 0 ? o.x : o.y, oo = dim == 0 ? o.y : o.x;
  const float ds = dim == 0 ? d.x : d.y, dd = dim == 0 ? d.y : d.x;
  const float oz = o.z, dz = d.z;
  const float c = (value - cs) / fs;
  const float t = (c * oz - os) / (ds - c * dz);
  const float num = fo * (oo * (c * dz - ds) + dd * (os - c * oz));
  const float den = dz * os - ds * oz;
  const float other = co + num / den;
  Hit h;
  h.t = t;
  h.x = dim == 0 ? value : other;
  h.y = dim == 0 ? other : value;
  const float pz = o.z + t * d.z;
  h.valid = in_bounds(h.x, h.y) && (pz > -kEps) && (t > -kEps);
  return h;
}

__device__ __forceinline__ float nan_to_num(float v, float pinf, float ninf) {
  if (v != v) return 0.f;
  if (v == __builtin_inff()) return pinf;
  if (v == -__builtin_inff()) return ninf;
  return v;
}

// projection of the camera-space point o + t d
__device__ __forceinline__ Hit point_hit(const float* k, V3 o, V3 d, float t) {
  const float X = o.x + t * d.x, Y = o.y + t * d.y, Z = o.z + t * d.z;
  const float den = Z + 1.1920928955078125e-07f;   // finfo(float32).eps
  const float px = nan_to_num(X / den, 1e8f, -1e8f), py = nan_to_num(Y / den, 1e8f, -1e8f),
              pz = nan_to_num(Z / den, 1e8f, -1e8f);
  Hit h;
  h.t = t;
  h.x = chain3(k[0], k[1], k[2], px, py, pz);
  h.y = chain3(k[3], k[4], k[5], px, py, pz);
  h.valid = in_bounds(h.x, h.y) && (Z > -kEps) && (t > -kEps);
  return h;
}

// normalised world ray through normalised pixel (x, y) of a camera (c2w, k_inv)
__device__ __forceinline__ V3 world_dir(const float* c2w, const float* ki, float x, float y) {
  float dx = chain3(ki[0], ki[1], ki[2], x, y, 1.0f);
  float dy = chain3(ki[3], ki[4], ki[5], x, y, 1.0f);
  float dz = chain3(ki[6], ki[7], ki[8], x, y, 1.0f);
  const float n = sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
  dx = dx / n; dy = dy / n; dz = dz / n;
  V3 w;
  w.x = chain3(c2w[0], c2w[1], c2w[2], dx, dy, dz);
  w.y = chain3(c2w[4], c2w[5], c2w[6], dx, dy, dz);
  w.z = chain3(c2w[8], c2w[9], c2w[10], dx, dy, dz);
  return w;
}

struct EpiDims { int b, v, h, w, s; };

// one thread per (b, v, ov, ray)
__global__ void __launch_bounds__(256)
epipolar_geometry_kernel(EpiDims dm, const float* __restrict__ c2w /*[b,v,16]*/,
                         const float* __restrict__ w2c /*[b,v,16]*/,
                         const float* __restrict__ kmat /*[b,v,9]*/,
                         const float* __restrict__ kinv /*[b,v,9]*/,
                         const float* __restrict__ near, const float* __restrict__ far,
                         float* __restrict__ origins /*[b,v,r,3]*/,
                         float* __restrict__ directions /*[b,v,r,3]*/,
                         float* __restrict__ seg /*[b,v,ov,r,6]: xy_min, xy_max, t_min, t_max*/,
                         uint8_t* __restrict__ flags /*[b,v,ov,r]: bit0 overlaps, 1 near ok,
                                                        2 far ok, 3-4 sel_min, 5-6 sel_max*/,
                         float* __restrict__ xy_sample /*[b,v,ov,r,s,2]*/,
                         float* __restrict__ depth /*[b,v,ov,r,s] raw*/,
                         float* __restrict__ rel_disp /*[b,v,ov,r,s] clipped + converted*/) {
  const int R = dm.h * dm.w, ovn = dm.v - 1;
  const size_t total = (size_t)dm.b * dm.v * ovn * R;
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int r = (int)(gid % R);
  const int ov = (int)((gid / R) % ovn);
  const int v = (int)((gid / ((size_t)R * ovn)) % dm.v);
  const int b = (int)(gid / ((size_t)R * ovn * dm.v));
  const int o_view = ov < v ? ov : ov + 1;          // index_v[v][ov]
  const int bv = b * dm.v + v, bo = b * dm.v + o_view;

  const int row = r / dm.w, col = r % dm.w;
  const float x = ((float)col + 0.5f) / (float)dm.w, y = ((float)row + 0.5f) / (float)dm.h;
  const float* C = c2w + 16 * bv;
  const V3 dirw = world_dir(C, kinv + 9 * bv, x, y);
  const V3 org = {C[3], C[7], C[11]};
  if (ov == 0) {
    float* po = origins + ((size_t)bv * R + r) * 3;
    float* pd = directions + ((size_t)bv * R + r) * 3;
    po[0] = org.x; po[1] = org.y; po[2] = org.z;
    pd[0] = dirw.x; pd[1] = dirw.y; pd[2] = dirw.z;
  }

  // into the other camera
  const float* Wm = w2c + 16 * bo;
  const float* K = kmat + 9 * bo;
  V3 o, d;
  o.x = chain4(Wm[0], Wm[1], Wm[2], Wm[3], org.x, org.y, org.z, 1.0f);
  o.y = chain4(Wm[4], Wm[5], Wm[6], Wm[7], org.x, org.y, org.z, 1.0f);
  o.z = chain4(Wm[8], Wm[9], Wm[10], Wm[11], org.x, org.y, org.z, 1.0f);
  d.x = chain3(Wm[0], Wm[1], Wm[2], dirw.x, dirw.y, dirw.z);
  d.y = chain3(Wm[4], Wm[5], Wm[6], dirw.x, dirw.y, dirw.z);
  d.z = chain3(Wm[8], Wm[9], Wm[10], dirw.x, dirw.y, dirw.z);

  Hit hits[4] = {frame_hit(K, o, d, 0, 0.0f), frame_hit(K, o, d, 0, 1.0f),
                 frame_hit(K, o, d, 1, 0.0f), frame_hit(K, o, d, 1, 1.0f)};
  // argmin / argmax of t over valid hits; invalid -> +/-inf; ties and all-invalid -> first
  int smin = 0, smax = 0;
  {
    const float inf = __builtin_inff();
    float bmin = hits[0].valid ? hits[0].t : inf, bmax = hits[0].valid ? hits[0].t : -inf;
#pragma unroll
    for (int i = 1; i < 4; ++i) {
      const float tmn = hits[i].valid ? hits[i].t : inf, tmx = hits[i].valid ? hits[i].t : -inf;
      if (tmn < bmin) { bmin = tmn; smin = i; }
      if (tmx > bmax) { bmax = tmx; smax = i; }
    }
  }
  const float nr = near[bv], fr = far[bv];
  const Hit pn = point_hit(K, o, d, nr), pf = point_hit(K, o, d, fr);
  Hit fmin = hits[0], fmax = hits[0];
#pragma unroll
  for (int i = 1; i < 4; ++i) { if (smin == i) fmin = hits[i]; if (smax == i) fmax = hits[i]; }
  if (!fmin.valid) fmin.t = __builtin_inff();        // the reference returns the masked t
  if (!fmax.valid) fmax.t = -__builtin_inff();
  const Hit start = pn.valid ? pn : fmin, end = pf.valid ? pf : fmax;
  const bool overlaps = start.valid && end.valid;

  const size_t ro = (((size_t)bv * ovn + ov) * R + r);
  float* ps = seg + ro * 6;
  ps[0] = start.x; ps[1] = start.y; ps[2] = end.x; ps[3] = end.y; ps[4] = start.t; ps[5] = end.t;
  flags[ro] = (uint8_t)((overlaps ? 1 : 0) | (pn.valid ? 2 : 0) | (pf.valid ? 4 : 0) |
                        (smin << 3) | (smax << 5));

  // samples
  const float m = overlaps ? 1.0f : 0.0f;
  const float ax = nan_to_num(start.x, 0.f, 0.f) * m, ay = nan_to_num(start.y, 0.f, 0.f) * m;
  const float bx = nan_to_num(end.x, 0.f, 0.f) * m, by = nan_to_num(end.y, 0.f, 0.f) * m;
  const float* C2 = c2w + 16 * bo;
  const float* KI2 = kinv + 9 * bo;
  const V3 o2 = {C2[3], C2[7], C2[11]};
  const float wx = org.x - o2.x, wy = org.y - o2.y, wz = org.z - o2.z;
  const float aw = dirw.x * wx + dirw.y * wy + dirw.z * wz;
  const float eps_d = 1e-10f;
  const float disp_near = 1.0f / (nr + eps_d), disp_far = 1.0f / (fr + eps_d);
  for (int i = 0; i < dm.s; ++i) {
    const float frac = ((float)i + 0.5f) / (float)dm.s;
    const float sx = ax + frac * (bx - ax), sy = ay + frac * (by - ay);
    const size_t so = ro * dm.s + i;
    xy_sample[2 * so] = sx; xy_sample[2 * so + 1] = sy;
    // depth: least-squares meeting point of the casting ray and the ray through the sample
    const V3 d2 = world_dir(C2, KI2, sx, sy);
    const float ab = dirw.x * d2.x + dirw.y * d2.y + dirw.z * d2.z;
    float dep;
    if (ab > 1.0f - 1e-5f) {
      const float ex = 1e10f - org.x, ey = 1e10f - org.y, ez = 1e10f - org.z;
      dep = sqrtf(ex * ex + ey * ey + ez * ez);
    } else {
      const float bw = d2.x * wx + d2.y * wy + d2.z * wz;
      const float den = 1.0f - ab * ab;
      const float t1 = (ab * bw - aw) / den, t2 = (bw - ab * aw) / den;
      // p - org = 0.5 * ((org + t1 a) + (o2 + t2 b)) - org
      const float qx = 0.5f * (t1 * dirw.x + t2 * d2.x - wx);
      const float qy = 0.5f * (t1 * dirw.y + t2 * d2.y - wy);
      const float qz = 0.5f * (t1 * dirw.z + t2 * d2.z - wz);
      dep = sqrtf(qx * qx + qy * qy + qz * qz);
    }
    depth[so] = dep;
    const float dc = fminf(fmaxf(dep, nr), fr);
    const float disp = 1.0f / (dc + eps_d);
    rel_disp[so] = 1.0f - (disp - disp_far) / (disp_near - disp_far + eps_d);
  }
}

void launch_epipolar_geometry(int b, int v, int h, int w, int s, const float* c2w,
                              const float* w2c, const float* kmat, const float* kinv,
                              const float* near, const float* far, float* origins,
                              float* directions, float* seg, uint8_t* flags, float* xy_sample,
                              float* depth, float* rel_disp, hipStream_t st) {
  EpiDims dm = {b, v, h, w, s};
  const size_t total = (size_t)b * v * (v - 1) * h * w;
  hipLaunchKernelGGL(epipolar_geometry_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     st, dm, c2w, w2c, kmat, kinv, near, far, origins, directions, seg, flags,
                     xy_sample, depth, rel_disp);
}

}  // namespace ps
