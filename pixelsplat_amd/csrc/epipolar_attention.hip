// Fused epipolar gather + single-query cross-attention (path A).
//
// The reference materialises kv = grid_sample(features) + Linear(PE(depth)) as a
// [b,v,ov,r,s,c] tensor (0.94 GB at the paper config), projects EVERY token through the
// 128 -> 1024 `to_kv` linear (481 GFLOP per layer) although each ray has ONE query token:
//   /root/reference/src/model/encoder/epipolar/epipolar_sampler.py:97-111   gather
//   /root/reference/src/model/encoder/epipolar/epipolar_transformer.py:113-142 depth enc., kv
//   /root/reference/src/model/transformer/attention.py:54-70                 attention
// Here nothing of size (rays x tokens x channels) ever reaches HBM.  With q~_h = W_k,h^T q_h
// (folded on the host by a plain GEMM) the score of token i is
//     s_{h,i} = scale * ( q~_h . feat_i + u_h . pe_i + e_{h,ov(i)} ),   u_h = W_d^T q~_h,
// and the context is sum_i a_{h,i} kv_i = fbar_h + W_d pbar_h + b_d + sum_ov abar_{h,ov} emb_ov,
// so a wave only has to produce, per ray and head, fbar (c), pbar (2*octaves), abar (ov) and
// the attention weights.  One wave64 per ray, lanes <-> channels (coalesced 512-byte NHWC
// corner reads), the gathered tokens live in LDS between the score and the context pass.
#include "raster_common.h"

namespace ps {

constexpr int kMaxHeads = 4;
constexpr int kAttnWaves = 2;   // waves (rays) per block

#define PS_DPP4(ctrl)                        \
  "v_add_f32_dpp %0, %0, %0 " ctrl "\n"      \
  "v_add_f32_dpp %1, %1, %1 " ctrl "\n"      \
  "v_add_f32_dpp %2, %2, %2 " ctrl "\n"      \
  "v_add_f32_dpp %3, %3, %3 " ctrl "\n"
// wave64 sums of four values (totals in lane 63); the 2-wait-state VALU->DPP hazard is
// covered by the s_nops because only four chains interleave here
__device__ __forceinline__ void wave_sum4_to_lane63(float& a, float& b, float& c, float& d) {
  asm volatile(
      "s_nop 1\n"
      PS_DPP4("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
      PS_DPP4("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1")
      PS_DPP4("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
      PS_DPP4("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
      PS_DPP4("row_bcast:15 row_mask:0xa bank_mask:0xf")
      PS_DPP4("row_bcast:31 row_mask:0xc bank_mask:0xf")
      "s_nop 1\n"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
#undef PS_DPP4

__device__ __forceinline__ float wave_max_all(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

struct Corner { int x0, y0; float wx, wy; };

// grid_sample(bilinear, zeros, align_corners=False) addressing of normalised (x, y) in [0,1]
// (the reference passes grid = 2 xy - 1; ATen unnormalises with ((g + 1) * size - 1) / 2)
__device__ __forceinline__ Corner corner_of(float x, float y, int w, int h) {
  const float gx = 2.0f * x - 1.0f, gy = 2.0f * y - 1.0f;
  const float ix = ((gx + 1.0f) * (float)w - 1.0f) / 2.0f;
  const float iy = ((gy + 1.0f) * (float)h - 1.0f) / 2.0f;
  const float fx = floorf(ix), fy = floorf(iy);
  Corner c;
  c.x0 = (int)fx; c.y0 = (int)fy; c.wx = ix - fx; c.wy = iy - fy;
  return c;
}

// bilinear gather of CPL consecutive channels starting at c0 from an NHWC map
template <int CPL>
__device__ __forceinline__ void gather(const float* __restrict__ img, int h, int w, int c, int c0,
                                       bool lane_on, Corner k, float* out) {
#pragma unroll
  for (int i = 0; i < CPL; ++i) out[i] = 0.f;
  if (!lane_on) return;
  const float w00 = (1.f - k.wx) * (1.f - k.wy), w10 = k.wx * (1.f - k.wy);
  const float w01 = (1.f - k.wx) * k.wy, w11 = k.wx * k.wy;
  const bool xin0 = k.x0 >= 0 && k.x0 < w, xin1 = k.x0 + 1 >= 0 && k.x0 + 1 < w;
  const bool yin0 = k.y0 >= 0 && k.y0 < h, yin1 = k.y0 + 1 >= 0 && k.y0 + 1 < h;
  auto add = [&](bool in, int xx, int yy, float wt) {
    if (!in) return;
    const float* p = img + ((size_t)yy * w + xx) * c + c0;
#pragma unroll
    for (int i = 0; i < CPL; ++i) out[i] = fmaf(p[i], wt, out[i]);
  };
  add(xin0 && yin0, k.x0, k.y0, w00);
  add(xin1 && yin0, k.x0 + 1, k.y0, w10);
  add(xin0 && yin1, k.x0, k.y0 + 1, w01);
  add(xin1 && yin1, k.x0 + 1, k.y0 + 1, w11);
}

__device__ __forceinline__ float pe_value(float rd, int p) {
  // sin(rd * 2 pi 2^(p/2) + (p & 1) pi/2)   (positional_encoding.py:14-33).  The product is
  // rounded before the phase is added, as in the reference: at 2 pi 2^9 an FMA would move
  // the argument by up to 1.2e-4.
#pragma clang fp contract(off)
  const float freq = 6.2831854820251465f * (float)(1 << (p >> 1));
  const float phase = (p & 1) ? 1.5707963705062866f : 0.0f;
  return sinf(rd * freq + phase);
}

// ------------------------------------------------------------------------------------
// materialised gather (EpipolarSampling.features for visualisers / the unfused fallback)
// ------------------------------------------------------------------------------------
template <int CPL>
__global__ void __launch_bounds__(256)
epipolar_gather_kernel(AttnDims dm, const float* __restrict__ fmap,
                       const float* __restrict__ xy, const uint8_t* __restrict__ flags,
                       float* __restrict__ out) {
  const int R = dm.h * dm.w, ovn = dm.v - 1;
  const size_t n_tok = (size_t)dm.b * dm.v * ovn * R * dm.s;
  const size_t tok = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tok >= n_tok) return;
  const int lane = threadIdx.x & 63;
  const size_t ro = tok / dm.s;
  const int ov = (int)((ro / R) % ovn);
  const size_t bv = ro / ((size_t)R * ovn);
  const int v = (int)(bv % dm.v);
  const int src = (int)(bv - v) + (ov < v ? ov : ov + 1);
  const int c0 = lane * CPL;
  const bool on = c0 < dm.c && (flags[ro] & 1);
  float f[CPL];
  gather<CPL>(fmap + (size_t)src * R * dm.c, dm.h, dm.w, dm.c, c0, on,
              corner_of(xy[2 * tok], xy[2 * tok + 1], dm.w, dm.h), f);
  if (c0 < dm.c) {
#pragma unroll
    for (int i = 0; i < CPL; ++i) out[tok * dm.c + c0 + i] = f[i];
  }
}

// ------------------------------------------------------------------------------------
// attention forward
// ------------------------------------------------------------------------------------
template <int CPL>
__global__ void __launch_bounds__(kAttnWaves* kWave)
epipolar_attn_forward_kernel(AttnDims dm, const float* __restrict__ fmap,
                             const float* __restrict__ xy, const uint8_t* __restrict__ flags,
                             const float* __restrict__ rd, const float* __restrict__ qt,
                             const float* __restrict__ u, const float* __restrict__ e,
                             float scale, float* __restrict__ fbar, float* __restrict__ pbar,
                             float* __restrict__ abar, float* __restrict__ attn) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int R = dm.h * dm.w, ovn = dm.v - 1, T = dm.s * ovn, P = 2 * dm.octaves, H = dm.heads;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t ray = (size_t)blockIdx.x * kAttnWaves + wv;      // (b v r)
  if (ray >= (size_t)dm.b * dm.v * R) return;
  float* featS = smem + (size_t)wv * (T * dm.c + T * P + kMaxHeads * T);
  float* peS = featS + T * dm.c;
  float* scS = peS + T * P;
  const int r = (int)(ray % R);
  const size_t bv = ray / R;
  const int v = (int)(bv % dm.v);
  const size_t bbase = bv - v;
  const int c0 = lane * CPL;
  const bool lane_c = c0 < dm.c;

  float q[kMaxHeads][CPL], uu[kMaxHeads];
#pragma unroll
  for (int hh = 0; hh < kMaxHeads; ++hh) {
    uu[hh] = (hh < H && lane < P) ? u[(ray * H + hh) * P + lane] : 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i)
      q[hh][i] = (hh < H && lane_c) ? qt[(ray * H + hh) * dm.c + c0 + i] : 0.f;
  }

  // pass 1: gather every token once, scores for all heads
  for (int t = 0; t < T; ++t) {
    const int si = t / ovn, ov = t % ovn;
    const size_t ro = (bv * ovn + ov) * R + r;
    const size_t so = ro * dm.s + si;
    const int src = (int)bbase + (ov < v ? ov : ov + 1);
    const bool ok = flags[ro] & 1;
    float f[CPL];
    gather<CPL>(fmap + (size_t)src * R * dm.c, dm.h, dm.w, dm.c, c0, lane_c && ok,
                corner_of(xy[2 * so], xy[2 * so + 1], dm.w, dm.h), f);
    const float pe = lane < P ? pe_value(rd[so], lane) : 0.f;
    float part[kMaxHeads];
#pragma unroll
    for (int hh = 0; hh < kMaxHeads; ++hh) {
      float a = uu[hh] * pe;
#pragma unroll
      for (int i = 0; i < CPL; ++i) a = fmaf(q[hh][i], f[i], a);
      if (e != nullptr && lane == 0 && hh < H) a += e[(ray * H + hh) * ovn + ov];
      part[hh] = a;
    }
    wave_sum4_to_lane63(part[0], part[1], part[2], part[3]);
    if (lane == 63) {
#pragma unroll
      for (int hh = 0; hh < kMaxHeads; ++hh) scS[hh * T + t] = part[hh] * scale;
    }
    if (lane_c) {
#pragma unroll
      for (int i = 0; i < CPL; ++i) featS[t * dm.c + c0 + i] = f[i];
    }
    if (lane < P) peS[t * P + lane] = pe;
  }
  wave_lds_sync();

  // softmax over the T tokens (lanes <-> tokens, T <= 128)
  for (int hh = 0; hh < H; ++hh) {
    const float s0 = lane < T ? scS[hh * T + lane] : -__builtin_inff();
    const float s1 = lane + 64 < T ? scS[hh * T + lane + 64] : -__builtin_inff();
    const float mx = wave_max_all(fmaxf(s0, s1));
    const float e0 = lane < T ? __expf(s0 - mx) : 0.f;
    const float e1 = lane + 64 < T ? __expf(s1 - mx) : 0.f;
    const float inv = 1.0f / wave_sum_all(e0 + e1);
    if (lane < T) { scS[hh * T + lane] = e0 * inv; attn[(ray * H + hh) * T + lane] = e0 * inv; }
    if (lane + 64 < T) {
      scS[hh * T + lane + 64] = e1 * inv; attn[(ray * H + hh) * T + lane + 64] = e1 * inv;
    }
  }
  wave_lds_sync();

  // pass 2: context
  float acc[kMaxHeads][CPL], pacc[kMaxHeads], aacc[kMaxHeads];
#pragma unroll
  for (int hh = 0; hh < kMaxHeads; ++hh) {
    pacc[hh] = 0.f; aacc[hh] = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) acc[hh][i] = 0.f;
  }
  for (int t = 0; t < T; ++t) {
    float f[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) f[i] = lane_c ? featS[t * dm.c + c0 + i] : 0.f;
    const float pe = lane < P ? peS[t * P + lane] : 0.f;
    const bool mine = (t % ovn) == lane;     // lanes < ovn collect the per-view attention mass
#pragma unroll
    for (int hh = 0; hh < kMaxHeads; ++hh) {
      const float a = hh < H ? scS[hh * T + t] : 0.f;
#pragma unroll
      for (int i = 0; i < CPL; ++i) acc[hh][i] = fmaf(a, f[i], acc[hh][i]);
      pacc[hh] = fmaf(a, pe, pacc[hh]);
      aacc[hh] += mine ? a : 0.f;
    }
  }
  for (int hh = 0; hh < H; ++hh) {
    if (lane_c) {
#pragma unroll
      for (int i = 0; i < CPL; ++i) fbar[(ray * H + hh) * dm.c + c0 + i] = acc[hh][i];
    }
    if (lane < P) pbar[(ray * H + hh) * P + lane] = pacc[hh];
    if (lane < ovn) abar[(ray * H + hh) * ovn + lane] = aacc[hh];
  }
}

// ------------------------------------------------------------------------------------
// attention backward, per ray: dq~, du, de and the per-token coefficients the feature-map
// scatter needs (ds = scale * a (da - sum a da))
// ------------------------------------------------------------------------------------
template <int CPL>
__global__ void __launch_bounds__(kAttnWaves* kWave)
epipolar_attn_backward_kernel(AttnDims dm, const float* __restrict__ fmap,
                              const float* __restrict__ xy, const uint8_t* __restrict__ flags,
                              const float* __restrict__ rd, const float* __restrict__ attn,
                              const float* __restrict__ dfbar, const float* __restrict__ dpbar,
                              const float* __restrict__ dabar, float scale,
                              float* __restrict__ dqt, float* __restrict__ du,
                              float* __restrict__ de, float* __restrict__ ds_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int R = dm.h * dm.w, ovn = dm.v - 1, T = dm.s * ovn, P = 2 * dm.octaves, H = dm.heads;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t ray = (size_t)blockIdx.x * kAttnWaves + wv;
  if (ray >= (size_t)dm.b * dm.v * R) return;
  float* featS = smem + (size_t)wv * (T * dm.c + T * P + kMaxHeads * T);
  float* peS = featS + T * dm.c;
  float* daS = peS + T * P;
  const int r = (int)(ray % R);
  const size_t bv = ray / R;
  const int v = (int)(bv % dm.v);
  const size_t bbase = bv - v;
  const int c0 = lane * CPL;
  const bool lane_c = c0 < dm.c;

  float g[kMaxHeads][CPL], gp[kMaxHeads];
#pragma unroll
  for (int hh = 0; hh < kMaxHeads; ++hh) {
    gp[hh] = (hh < H && lane < P) ? dpbar[(ray * H + hh) * P + lane] : 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i)
      g[hh][i] = (hh < H && lane_c) ? dfbar[(ray * H + hh) * dm.c + c0 + i] : 0.f;
  }
  // da_{h,t} = dfbar_h . feat_t + dpbar_h . pe_t + dabar_{h,ov(t)}
  for (int t = 0; t < T; ++t) {
    const int si = t / ovn, ov = t % ovn;
    const size_t ro = (bv * ovn + ov) * R + r;
    const size_t so = ro * dm.s + si;
    const int src = (int)bbase + (ov < v ? ov : ov + 1);
    const bool ok = flags[ro] & 1;
    float f[CPL];
    gather<CPL>(fmap + (size_t)src * R * dm.c, dm.h, dm.w, dm.c, c0, lane_c && ok,
                corner_of(xy[2 * so], xy[2 * so + 1], dm.w, dm.h), f);
    const float pe = lane < P ? pe_value(rd[so], lane) : 0.f;
    float part[kMaxHeads];
#pragma unroll
    for (int hh = 0; hh < kMaxHeads; ++hh) {
      float a = gp[hh] * pe;
#pragma unroll
      for (int i = 0; i < CPL; ++i) a = fmaf(g[hh][i], f[i], a);
      if (lane == 0 && hh < H) a += dabar[(ray * H + hh) * ovn + ov];
      part[hh] = a;
    }
    wave_sum4_to_lane63(part[0], part[1], part[2], part[3]);
    if (lane == 63) {
#pragma unroll
      for (int hh = 0; hh < kMaxHeads; ++hh) daS[hh * T + t] = part[hh];
    }
    if (lane_c) {
#pragma unroll
      for (int i = 0; i < CPL; ++i) featS[t * dm.c + c0 + i] = f[i];
    }
    if (lane < P) peS[t * P + lane] = pe;
  }
  wave_lds_sync();
  // softmax backward: ds = scale * a * (da - sum_t a da)
  for (int hh = 0; hh < H; ++hh) {
    const float a0 = lane < T ? attn[(ray * H + hh) * T + lane] : 0.f;
    const float a1 = lane + 64 < T ? attn[(ray * H + hh) * T + lane + 64] : 0.f;
    const float d0 = lane < T ? daS[hh * T + lane] : 0.f;
    const float d1 = lane + 64 < T ? daS[hh * T + lane + 64] : 0.f;
    const float dot = wave_sum_all(a0 * d0 + a1 * d1);
    if (lane < T) {
      const float dsv = scale * a0 * (d0 - dot);
      daS[hh * T + lane] = dsv; ds_out[(ray * H + hh) * T + lane] = dsv;
    }
    if (lane + 64 < T) {
      const float dsv = scale * a1 * (d1 - dot);
      daS[hh * T + lane + 64] = dsv; ds_out[(ray * H + hh) * T + lane + 64] = dsv;
    }
  }
  wave_lds_sync();
  float acc[kMaxHeads][CPL], pacc[kMaxHeads], eacc[kMaxHeads];
#pragma unroll
  for (int hh = 0; hh < kMaxHeads; ++hh) {
    pacc[hh] = 0.f; eacc[hh] = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) acc[hh][i] = 0.f;
  }
  for (int t = 0; t < T; ++t) {
    float f[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) f[i] = lane_c ? featS[t * dm.c + c0 + i] : 0.f;
    const float pe = lane < P ? peS[t * P + lane] : 0.f;
    const bool mine = (t % ovn) == lane;
#pragma unroll
    for (int hh = 0; hh < kMaxHeads; ++hh) {
      const float dsv = hh < H ? daS[hh * T + t] : 0.f;
#pragma unroll
      for (int i = 0; i < CPL; ++i) acc[hh][i] = fmaf(dsv, f[i], acc[hh][i]);
      pacc[hh] = fmaf(dsv, pe, pacc[hh]);
      eacc[hh] += mine ? dsv : 0.f;
    }
  }
  for (int hh = 0; hh < H; ++hh) {
    if (lane_c) {
#pragma unroll
      for (int i = 0; i < CPL; ++i) dqt[(ray * H + hh) * dm.c + c0 + i] = acc[hh][i];
    }
    if (lane < P) du[(ray * H + hh) * P + lane] = pacc[hh];
    if (lane < ovn) de[(ray * H + hh) * ovn + lane] = eacc[hh];
  }
}

// ------------------------------------------------------------------------------------
// feature-map gradient: dF[src][y][x][c] += w_corner * sum_h (a dfbar_h[c] + ds q~_h[c])
// One block owns (source image, slice of CS channels) and keeps that slice of the WHOLE
// gradient image in LDS (h*w*CS floats, up to 128 KB of the CU's 160 KB): every token of
// every ray that samples this image is splatted with LDS float atomics, then the slice is
// written out once -- no global atomics (grid_sample's backward issues ~1e9 of them).
// ------------------------------------------------------------------------------------
template <int CS>
__global__ void __launch_bounds__(256)
epipolar_dfmap_kernel(AttnDims dm, const float* __restrict__ xy,
                      const uint8_t* __restrict__ flags, const float* __restrict__ attn,
                      const float* __restrict__ ds, const float* __restrict__ dfbar,
                      const float* __restrict__ qt, float* __restrict__ dfmap) {
  extern __shared__ __attribute__((aligned(16))) float tile[];   // [h*w][CS]
  const int R = dm.h * dm.w, ovn = dm.v - 1, T = dm.s * ovn, H = dm.heads;
  const int n_slices = (dm.c + CS - 1) / CS;
  const int slice = blockIdx.x % n_slices;
  const int src_bv = blockIdx.x / n_slices;          // (b, source view)
  const int b = src_bv / dm.v, sv = src_bv % dm.v;
  const int cbase = slice * CS;
  for (int i = threadIdx.x; i < R * CS; i += blockDim.x) tile[i] = 0.f;
  __syncthreads();
  // casting views v != sv; their ov index pointing at sv is ov = sv < v ? sv : sv - 1
  const size_t pairs_per_view = (size_t)R * dm.s;
  for (int v = 0; v < dm.v; ++v) {
    if (v == sv) continue;
    const int ov = sv < v ? sv : sv - 1;
    const size_t bv = (size_t)b * dm.v + v;
    for (size_t pr = threadIdx.x; pr < pairs_per_view; pr += blockDim.x) {
      const int r = (int)(pr / dm.s), si = (int)(pr % dm.s);
      const size_t ro = (bv * ovn + ov) * R + r;
      if (!(flags[ro] & 1)) continue;
      const size_t so = ro * dm.s + si;
      const size_t ray = bv * R + r;
      const int t = si * ovn + ov;
      float df[CS];
#pragma unroll
      for (int k = 0; k < CS; ++k) df[k] = 0.f;
      for (int hh = 0; hh < H; ++hh) {
        const float a = attn[(ray * H + hh) * T + t], d = ds[(ray * H + hh) * T + t];
        const float* gq = dfbar + (ray * H + hh) * dm.c + cbase;
        const float* qq = qt + (ray * H + hh) * dm.c + cbase;
#pragma unroll
        for (int k = 0; k < CS; ++k)
          if (cbase + k < dm.c) df[k] = fmaf(a, gq[k], fmaf(d, qq[k], df[k]));
      }
      const Corner kq = corner_of(xy[2 * so], xy[2 * so + 1], dm.w, dm.h);
      const float wts[4] = {(1.f - kq.wx) * (1.f - kq.wy), kq.wx * (1.f - kq.wy),
                            (1.f - kq.wx) * kq.wy, kq.wx * kq.wy};
#pragma unroll
      for (int cr = 0; cr < 4; ++cr) {
        const int xx = kq.x0 + (cr & 1), yy = kq.y0 + (cr >> 1);
        if (xx < 0 || xx >= dm.w || yy < 0 || yy >= dm.h) continue;
        float* dst = tile + ((size_t)yy * dm.w + xx) * CS;
#pragma unroll
        for (int k = 0; k < CS; ++k) atomicAdd(dst + k, wts[cr] * df[k]);
      }
    }
  }
  __syncthreads();
  float* out = dfmap + (size_t)src_bv * R * dm.c;
  for (int i = threadIdx.x; i < R * CS; i += blockDim.x) {
    const int pix = i / CS, k = i % CS;
    if (cbase + k < dm.c) out[(size_t)pix * dm.c + cbase + k] = tile[i];
  }
}

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
static size_t attn_smem(const AttnDims& dm) {
  const int T = dm.s * (dm.v - 1), P = 2 * dm.octaves;
  return (size_t)kAttnWaves * (T * dm.c + T * P + kMaxHeads * T) * sizeof(float);
}

int launch_epipolar_gather(const AttnDims& dm, const float* fmap, const float* xy,
                           const uint8_t* flags, float* out, hipStream_t st) {
  const size_t n_tok = (size_t)dm.b * dm.v * (dm.v - 1) * dm.h * dm.w * dm.s;
  dim3 grid((unsigned)((n_tok + 3) / 4)), block(256);
  if (dm.c <= 64) hipLaunchKernelGGL(epipolar_gather_kernel<1>, grid, block, 0, st, dm, fmap, xy, flags, out);
  else if (dm.c <= 128) hipLaunchKernelGGL(epipolar_gather_kernel<2>, grid, block, 0, st, dm, fmap, xy, flags, out);
  else if (dm.c <= 256) hipLaunchKernelGGL(epipolar_gather_kernel<4>, grid, block, 0, st, dm, fmap, xy, flags, out);
  else return PS_ERR_UNSUPPORTED;
  return PS_OK;
}

int launch_epipolar_attn_forward(const AttnDims& dm, const float* fmap, const float* xy,
                                 const uint8_t* flags, const float* rd, const float* qt,
                                 const float* u, const float* e, float scale, float* fbar,
                                 float* pbar, float* abar, float* attn, hipStream_t st) {
  const size_t rays = (size_t)dm.b * dm.v * dm.h * dm.w;
  dim3 grid((unsigned)((rays + kAttnWaves - 1) / kAttnWaves)), block(kAttnWaves * kWave);
  const size_t sm = attn_smem(dm);
  if (sm > 160 * 1024 || dm.heads > kMaxHeads || dm.s * (dm.v - 1) > 128 || 2 * dm.octaves > 64)
    return PS_ERR_UNSUPPORTED;
#define PS_GO(CPL)                                                                              \
  do {                                                                                          \
    (void)hipFuncSetAttribute((const void*)epipolar_attn_forward_kernel<CPL>,                   \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);             \
    hipLaunchKernelGGL(epipolar_attn_forward_kernel<CPL>, grid, block, sm, st, dm, fmap, xy,    \
                       flags, rd, qt, u, e, scale, fbar, pbar, abar, attn);                     \
  } while (0)
  if (dm.c <= 64) PS_GO(1); else if (dm.c <= 128) PS_GO(2); else if (dm.c <= 256) PS_GO(4);
  else return PS_ERR_UNSUPPORTED;
#undef PS_GO
  return PS_OK;
}

int launch_epipolar_attn_backward(const AttnDims& dm, const float* fmap, const float* xy,
                                  const uint8_t* flags, const float* rd, const float* qt,
                                  const float* attn, const float* dfbar, const float* dpbar,
                                  const float* dabar, float scale, float* dqt, float* du,
                                  float* de, float* ds, float* dfmap, hipStream_t st) {
  const size_t rays = (size_t)dm.b * dm.v * dm.h * dm.w;
  dim3 grid((unsigned)((rays + kAttnWaves - 1) / kAttnWaves)), block(kAttnWaves * kWave);
  const size_t sm = attn_smem(dm);
  if (sm > 160 * 1024 || dm.heads > kMaxHeads || dm.s * (dm.v - 1) > 128 || 2 * dm.octaves > 64)
    return PS_ERR_UNSUPPORTED;
#define PS_GO(CPL)                                                                              \
  do {                                                                                          \
    (void)hipFuncSetAttribute((const void*)epipolar_attn_backward_kernel<CPL>,                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm);             \
    hipLaunchKernelGGL(epipolar_attn_backward_kernel<CPL>, grid, block, sm, st, dm, fmap, xy,   \
                       flags, rd, attn, dfbar, dpbar, dabar, scale, dqt, du, de, ds);           \
  } while (0)
  if (dm.c <= 64) PS_GO(1); else if (dm.c <= 128) PS_GO(2); else if (dm.c <= 256) PS_GO(4);
  else return PS_ERR_UNSUPPORTED;
#undef PS_GO
  if (dfmap != nullptr) {
    // channel slice so that h*w*CS floats fit in 128 KB of LDS
    const size_t R = (size_t)dm.h * dm.w;
    int cs = 8;
    while (cs > 1 && R * cs * 4 > 128 * 1024) cs >>= 1;
    if (R * cs * 4 > 160 * 1024) return PS_ERR_UNSUPPORTED;
    const int n_slices = (dm.c + cs - 1) / cs;
    dim3 g2((unsigned)(dm.b * dm.v * n_slices)), b2(256);
    const size_t sm2 = R * cs * 4;
#define PS_DF(CS)                                                                               \
  do {                                                                                          \
    (void)hipFuncSetAttribute((const void*)epipolar_dfmap_kernel<CS>,                           \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm2);            \
    hipLaunchKernelGGL(epipolar_dfmap_kernel<CS>, g2, b2, sm2, st, dm, xy, flags, attn, ds,     \
                       dfbar, qt, dfmap);                                                       \
  } while (0)
    if (cs == 8) PS_DF(8); else if (cs == 4) PS_DF(4); else if (cs == 2) PS_DF(2); else PS_DF(1);
#undef PS_DF
  }
  return PS_OK;
}

}  // namespace ps
