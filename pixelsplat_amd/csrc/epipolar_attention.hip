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

#include <cstdlib>

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

// sin(x) for x in [0, ~4000]: two-constant Cody-Waite reduction to [-pi/4, pi/4] (the FMA
// keeps n*C exact) and the cephes sinf/cosf polynomials; |error| < 2e-7.  libm's sinf costs
// ~150 instructions at these magnitudes, this ~25.
__device__ __forceinline__ float sin_reduced(float x) {
  const float n = rintf(x * 0.63661977236758134f);
  float r = fmaf(-n, 1.5707963705062866f, x);
  r = fmaf(-n, -4.3711388286737929e-8f, r);
  const float z = r * r;
  const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f),
                        z * r, r);
  const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z,
                             4.166664568298827e-2f), z * z, fmaf(-0.5f, z, 1.0f));
  const int q = (int)n;
  const float v = (q & 1) ? cp : sp;
  return (q & 2) ? -v : v;
}

__device__ __forceinline__ float pe_value(float rd, int p) {
  // sin(rd * 2 pi 2^(p/2) + (p & 1) pi/2)   (positional_encoding.py:14-33).  The product is
  // rounded before the phase is added, as in the reference: at 2 pi 2^9 an FMA would move
  // the argument by up to 1.2e-4.
  const float freq = 6.2831854820251465f * (float)(1 << (p >> 1));
  const float phase = (p & 1) ? 1.5707963705062866f : 0.0f;
  float arg;
  {
#pragma clang fp contract(off)
    arg = rd * freq + phase;
  }
  return sin_reduced(arg);
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
// fused attention, one wave64 per ray, tokens streamed in chunks of 8 (online softmax)
// ------------------------------------------------------------------------------------
// Wave-private LDS (floats): token records [T][8] | rd [T] | raw scores [4][T] |
// feature chunk [8][4*S4] | encoding chunk [8][Ps] | folded query [4][c]  -- 9.5 KB at the
// paper config (4 waves/SIMD; the query in registers cost 64 VGPRs and one wave), so the
// occupancy is set by registers (the whole-ray staging of the previous version needed 20 KB
// and ran at 2 waves/SIMD, 30 % of the VALU issue rate).
//
// Lane roles per phase:
//   A  lanes <-> tokens         token record: 4 clamped corner pixels + 4 bilinear weights
//   per chunk of 8 tokens:
//   B  lanes <-> (token, p)     positional encoding (range-reduced sin, ~25 instructions)
//   C  lanes <-> (token, 4 ch)  gather, 64/LPT tokens per step, 16-byte loads
//   D  lanes <-> (token, slice) lane = 8*token + j owns the channel quads j, j+8, ... of its
//                               token: partial dot products against the folded query held in
//                               registers, then a 3-step DPP sum over j (8 lanes)
//   E  8 score lanes            chunk max / sum by DPP, running max and normaliser
//   F  lanes <-> channels       context accumulation; weights via v_readlane
// The feature rows use a stride of S4 quads with S4 = 8 (mod 16): the 16 lanes of one pass
// of a 128-bit LDS read then cover all 64 banks.
template <int CPL> struct LaneVec;
template <> struct LaneVec<1> { using type = float; };
template <> struct LaneVec<2> { using type = float2; };
template <> struct LaneVec<4> { using type = float4; };

template <int CPL>
__device__ __forceinline__ void load_cpl(const float* __restrict__ p, float* out) {
  using V = typename LaneVec<CPL>::type;
  const V q = *reinterpret_cast<const V*>(p);
  const float* f = reinterpret_cast<const float*>(&q);
#pragma unroll
  for (int i = 0; i < CPL; ++i) out[i] = f[i];
}

constexpr int kChunk = 8;       // tokens per chunk
constexpr int kUPL = 3;         // encoding dims per lane in phase D (P <= 24)

__host__ __device__ inline int pe_stride(int P) { return (P + 3) & ~3; }
__host__ __device__ inline int feat_stride_quads(int c) {
  int s4 = c / 4;
  while ((s4 & 15) != 8) ++s4;
  return s4;
}
__host__ __device__ inline size_t wave_lds_floats(int T, int c, int P) {
  return (size_t)T * (8 + 1 + kMaxHeads) + (size_t)kChunk * (4 * feat_stride_quads(c) + pe_stride(P)) +
         (size_t)kMaxHeads * c;
}

struct RayCtx {
  int ray, r, v, ovn, T, P, Ps, fs, H;
  size_t bv, bbase;
  float *tokS, *rdS, *scS, *featS, *peS, *qS;
};

#define PS_DPP4(op, ctrl)                \
  op " %0, %0, %0 " ctrl "\n"            \
  op " %1, %1, %1 " ctrl "\n"            \
  op " %2, %2, %2 " ctrl "\n"            \
  op " %3, %3, %3 " ctrl "\n"
// sums over each aligned group of 8 lanes, result in the group's last lane (the other lanes
// of the group end up with partial sums that may include the neighbouring group: unused)
__device__ __forceinline__ void group8_sum4(float& a, float& b, float& c, float& d) {
  asm volatile(
      "s_nop 1\n"
      PS_DPP4("v_add_f32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1")
      PS_DPP4("v_add_f32_dpp", "row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1")
      PS_DPP4("v_add_f32_dpp", "row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1")
      "s_nop 1\n"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
// reductions over the eight lanes 7, 15, ..., 63 (result in lane 63)
__device__ __forceinline__ void lanes8_sum4_to_lane63(float& a, float& b, float& c, float& d) {
  asm volatile(
      "s_nop 1\n"
      PS_DPP4("v_add_f32_dpp", "row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1")
      PS_DPP4("v_add_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf")
      PS_DPP4("v_add_f32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf")
      "s_nop 1\n"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void lanes8_max4_to_lane63(float& a, float& b, float& c, float& d) {
  asm volatile(
      "s_nop 1\n"
      PS_DPP4("v_max_f32_dpp", "row_shr:8 row_mask:0xf bank_mask:0xf")
      PS_DPP4("v_max_f32_dpp", "row_bcast:15 row_mask:0xa bank_mask:0xf")
      PS_DPP4("v_max_f32_dpp", "row_bcast:31 row_mask:0xc bank_mask:0xf")
      "s_nop 1\n"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
#undef PS_DPP4

__device__ __forceinline__ float lane_bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// marks a wave-uniform value as such (it then lives in an SGPR)
__device__ __forceinline__ float uniform(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

__device__ __forceinline__ bool ray_setup(const AttnDims& dm, float* smem, RayCtx& k) {
  const int R = dm.h * dm.w;
  k.ovn = dm.v - 1; k.T = dm.s * k.ovn; k.P = 2 * dm.octaves; k.Ps = pe_stride(k.P);
  k.fs = 4 * feat_stride_quads(dm.c); k.H = dm.heads;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  k.ray = (int)blockIdx.x * kAttnWaves + wv;
  if (k.ray >= dm.b * dm.v * R) return false;
  k.tokS = smem + (size_t)wv * wave_lds_floats(k.T, dm.c, k.P);
  k.rdS = k.tokS + (size_t)k.T * 8;
  k.scS = k.rdS + k.T;
  k.featS = k.scS + (size_t)kMaxHeads * k.T;
  k.peS = k.featS + (size_t)kChunk * k.fs;
  k.qS = k.peS + (size_t)kChunk * k.Ps;
  k.r = k.ray % R;
  k.bv = (size_t)(k.ray / R);
  k.v = (int)(k.bv % dm.v);
  k.bbase = k.bv - k.v;
  return true;
}

// phases B, C: encodings and gathered features of tokens [t0, t0 + 8) into the chunk buffers
template <int CK>
__device__ __forceinline__ void stage_chunk(const AttnDims& dm, const RayCtx& k, int lane, int t0,
                                            const float* __restrict__ fmap) {
  constexpr int LPT = CK / 4, TPS = kWave / LPT;
  {
    // encodings: lane = 8 * token + j computes the dims j, j + 8, j + 16 of its token (the mapping of
    // phase D): the token's disparity is read once and no index arithmetic is left per value (a loop
    // over the chunk's kChunk * P values in lane order spent 12 of its 39 instructions per trip on
    // idx -> (token, dim))
    static_assert(kChunk * 8 == kWave, "lane = 8 * token + j");
    const int tl = lane >> 3, j = lane & 7;
    const float rdv = k.rdS[min(t0 + tl, k.T - 1)];
    float* dst = k.peS + tl * k.Ps + j;
#pragma unroll
    for (int i = 0; i < kUPL; ++i)            // (8 * kUPL >= P: attn_dims_ok)
      if (j + 8 * i < k.P) dst[8 * i] = pe_value(rdv, j + 8 * i);
  }
  const int lt = lane / LPT, ch = (lane % LPT) * 4;
  // TPS tokens per pass; the corner loads of NB passes are issued together (8 x 16 bytes per
  // lane in flight at c = 128): a pass-at-a-time loop waits one full load latency per pass,
  // 16 serialized latencies per ray
  constexpr int PASSES = (kChunk + TPS - 1) / TPS;
  constexpr int NB = PASSES >= 2 ? 2 : 1;
  const char* fm = reinterpret_cast<const char*>(fmap);
  // 32-bit byte offsets from the (wave-uniform) map pointer: scalar base + vector offset
  // addressing instead of a 64-bit multiply-add per corner (the maps are < 4 GB, checked)
  const uint32_t cb = 4u * (uint32_t)dm.c, chb = 4u * (uint32_t)ch;
#pragma unroll
  for (int s0 = 0; s0 < kChunk; s0 += NB * TPS) {
    float4 p[NB][4], wt[NB];
    bool on[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int tl = s0 + u * TPS + lt;
      const int t = t0 + tl;
      on[u] = tl < kChunk && t < k.T && ch < dm.c;
      const int tc = on[u] ? t : 0;                     // clamped: the loads are unconditional
      const int4 off = *reinterpret_cast<const int4*>(k.tokS + tc * 8);
      wt[u] = *reinterpret_cast<const float4*>(k.tokS + tc * 8 + 4);
      const uint32_t chq = on[u] ? chb : 0u;
      p[u][0] = *reinterpret_cast<const float4*>(fm + ((uint32_t)off.x * cb + chq));
      p[u][1] = *reinterpret_cast<const float4*>(fm + ((uint32_t)off.y * cb + chq));
      p[u][2] = *reinterpret_cast<const float4*>(fm + ((uint32_t)off.z * cb + chq));
      p[u][3] = *reinterpret_cast<const float4*>(fm + ((uint32_t)off.w * cb + chq));
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      const int tl = s0 + u * TPS + lt;
      float4 f;
      f.x = fmaf(p[u][3].x, wt[u].w, fmaf(p[u][2].x, wt[u].z, fmaf(p[u][1].x, wt[u].y, p[u][0].x * wt[u].x)));
      f.y = fmaf(p[u][3].y, wt[u].w, fmaf(p[u][2].y, wt[u].z, fmaf(p[u][1].y, wt[u].y, p[u][0].y * wt[u].x)));
      f.z = fmaf(p[u][3].z, wt[u].w, fmaf(p[u][2].z, wt[u].z, fmaf(p[u][1].z, wt[u].y, p[u][0].z * wt[u].x)));
      f.w = fmaf(p[u][3].w, wt[u].w, fmaf(p[u][2].w, wt[u].z, fmaf(p[u][1].w, wt[u].y, p[u][0].w * wt[u].x)));
      if (on[u]) *reinterpret_cast<float4*>(k.featS + tl * k.fs + ch) = f;
    }
  }
  wave_lds_sync();
}

// The "query" of phase D: the channel part goes to LDS ([h][c]; the 8 token lanes of a slice
// read the same address, a broadcast), the encoding part stays in registers: lane = 8*tl + j
// holds the dims j, j+8, ... of urow for every head.
struct QueryRegs {
  float u[kMaxHeads][kUPL];
};

// Phase A (token records of the whole ray: corner offsets + bilinear weights [T][8], relative
// disparity [T]) and the "query" of phase D (channel part to LDS [h][c] -- the 8 token lanes of
// a slice read the same address, a broadcast; encoding part in registers: lane = 8 tl + j holds
// the dims j, j + 8, ... of urow for every head) in ONE memory round trip.  Written as two plain
// staging loops the compiler emitted one s_waitcnt vmcnt(0) per load (flags -> xy -> rd, then one
// per head row of the query): 8 dependent round trips before the first token chunk, on a kernel
// whose waves spend 40-60 % of their life in s_waitcnt (SQ_WAIT_ANY).  Every address below depends on the ray index only, so all loads --
// token records of lanes' tokens t = lane, lane + 64, the H query rows, the encoding rows and
// (backward) the ray's attention weights -- are issued back to back, unconditionally (clamped
// addresses, values masked afterwards), and consumed after that.
template <bool WITH_AW, int NT>     // NT = token rows per lane: 1 (T <= 64) or 2 (T <= 128)
__device__ __forceinline__ void ray_prologue(const AttnDims& dm, const RayCtx& k, int lane,
                                             const float* __restrict__ xy,
                                             const uint8_t* __restrict__ flags,
                                             const float* __restrict__ rd,
                                             const float* __restrict__ qrow, int hs_q,
                                             const float* __restrict__ urow, int hs_u,
                                             QueryRegs& Q, const float* __restrict__ attn_row,
                                             float (&aw)[2][kMaxHeads]) {
  const int R = dm.h * dm.w;
  // ---- issue ----
  uint32_t fl[NT];
  float2 pxy[NT];
  float rdv[NT];
  int srcv[NT];
  // (no branch in here, not even a wave-uniform "T > 64": a value that is live across a branch
  // is materialised at the join, i.e. waited for, and the batch falls apart again)
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const int t = lane + u * kWave;
    const int tc = t < k.T ? t : 0;
    const int si = tc / k.ovn, ov = tc - si * k.ovn;
    const size_t ro = (k.bv * k.ovn + ov) * R + k.r;
    const size_t so = ro * dm.s + si;
    srcv[u] = (int)k.bbase + (ov < k.v ? ov : ov + 1);
    fl[u] = flags[ro];
    pxy[u] = *reinterpret_cast<const float2*>(xy + 2 * so);
    rdv[u] = rd[so];
  }
  const int qi = lane * 4;
  const bool qin = qi < dm.c;                      // c <= 256: one float4 per lane and head row
  float4 qv[kMaxHeads];
#pragma unroll
  for (int hh = 0; hh < kMaxHeads; ++hh)
    qv[hh] = *reinterpret_cast<const float4*>(qrow + (hh < k.H ? hh : 0) * hs_q + (qin ? qi : 0));
  const int j = lane & 7;
  // (urow is a valid row even for P = 0: the host passes the query row then)
#pragma unroll
  for (int hh = 0; hh < kMaxHeads; ++hh)
#pragma unroll
    for (int i = 0; i < kUPL; ++i) {
      const int p = j + 8 * i;
      Q.u[hh][i] = urow[(hh < k.H ? hh : 0) * hs_u + (p < k.P ? p : 0)];
    }
  if (WITH_AW) {
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
      for (int hh = 0; hh < kMaxHeads; ++hh) {
        const int t = lane + u * kWave;
        const bool ok = t < k.T && hh < k.H;
        aw[u][hh] = attn_row[(size_t)(ok ? hh : 0) * k.T + (ok ? t : 0)];
      }
  }
  // ---- consume ----
#pragma unroll
  for (int u = 0; u < NT; ++u) {
    const int t = lane + u * kWave;
    if (t < k.T) {
      const bool ok = fl[u] & 1u;
      const Corner cq = corner_of(pxy[u].x, pxy[u].y, dm.w, dm.h);
      const bool xin0 = cq.x0 >= 0 && cq.x0 < dm.w, xin1 = cq.x0 + 1 >= 0 && cq.x0 + 1 < dm.w;
      const bool yin0 = cq.y0 >= 0 && cq.y0 < dm.h, yin1 = cq.y0 + 1 >= 0 && cq.y0 + 1 < dm.h;
      const int xa = min(max(cq.x0, 0), dm.w - 1), xb = min(max(cq.x0 + 1, 0), dm.w - 1);
      const int ya = min(max(cq.y0, 0), dm.h - 1), yb = min(max(cq.y0 + 1, 0), dm.h - 1);
      const int base = srcv[u] * R;
      const int4 off = make_int4(base + ya * dm.w + xa, base + ya * dm.w + xb,
                                 base + yb * dm.w + xa, base + yb * dm.w + xb);
      float4 wt;
      wt.x = (ok && xin0 && yin0) ? (1.f - cq.wx) * (1.f - cq.wy) : 0.f;
      wt.y = (ok && xin1 && yin0) ? cq.wx * (1.f - cq.wy) : 0.f;
      wt.z = (ok && xin0 && yin1) ? (1.f - cq.wx) * cq.wy : 0.f;
      wt.w = (ok && xin1 && yin1) ? cq.wx * cq.wy : 0.f;
      *reinterpret_cast<int4*>(k.tokS + t * 8) = off;
      *reinterpret_cast<float4*>(k.tokS + t * 8 + 4) = wt;
      k.rdS[t] = rdv[u];
    }
  }
  if (qin) {
#pragma unroll
    for (int hh = 0; hh < kMaxHeads; ++hh)       // absent heads are zero rows: no per-head branches
      *reinterpret_cast<float4*>(k.qS + hh * dm.c + qi) =
          hh < k.H ? qv[hh] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int hh = 0; hh < kMaxHeads; ++hh)
#pragma unroll
    for (int i = 0; i < kUPL; ++i)
      Q.u[hh][i] = (j + 8 * i < k.P && hh < k.H) ? Q.u[hh][i] : 0.f;
  if (WITH_AW) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int hh = 0; hh < kMaxHeads; ++hh)
        aw[u][hh] = (u < NT && lane + u * kWave < k.T && hh < k.H) ? aw[u][hh] : 0.f;
  }
  wave_lds_sync();
}

// phase D: out[h] = qrow_h . feat_t + urow_h . pe_t + erow_{h, ov(t)} for the chunk's token
// tl = lane / 8, valid in the lanes with j == 7
template <int CK>
__device__ __forceinline__ void chunk_scores(const AttnDims& dm, const RayCtx& k, int lane, int t0,
                                             const QueryRegs& Q, const float (&ev)[kMaxHeads],
                                             bool with_view_term, float (&out)[kMaxHeads]) {
  const int tl = lane >> 3, j = lane & 7;
  const float* frow = k.featS + tl * k.fs;
  // even / odd components accumulate in the two halves of a register pair: the packed FMAs
  // then take the (x, y) and (z, w) halves of the 16-byte LDS reads in place (the scalar form
  // is packed by the compiler across HEADS, which costs a v_mov per operand to build the pairs)
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 acc2[kMaxHeads];
#pragma unroll
  for (int hh = 0; hh < kMaxHeads; ++hh) acc2[hh] = v2{0.f, 0.f};
#pragma unroll
  for (int i = 0; i < CK / 32; ++i) {
    const int ch = 4 * (j + 8 * i);
    if (ch < dm.c) {
      const float4 f = *reinterpret_cast<const float4*>(frow + ch);
      const v2 f01 = v2{f.x, f.y}, f23 = v2{f.z, f.w};
#pragma unroll
      for (int hh = 0; hh < kMaxHeads; ++hh) {
        const float4 q = *reinterpret_cast<const float4*>(k.qS + hh * dm.c + ch);
        acc2[hh] = v2{q.x, q.y} * f01 + acc2[hh];
        acc2[hh] = v2{q.z, q.w} * f23 + acc2[hh];
      }
    }
  }
  float acc[kMaxHeads];
#pragma unroll
  for (int hh = 0; hh < kMaxHeads; ++hh) acc[hh] = acc2[hh].x + acc2[hh].y;
  const float* prow = k.peS + tl * k.Ps;
#pragma unroll
  for (int i = 0; i < kUPL; ++i) {
    const int p = j + 8 * i;
    const float pe = p < k.P ? prow[p] : 0.f;
#pragma unroll
    for (int hh = 0; hh < kMaxHeads; ++hh) acc[hh] = fmaf(Q.u[hh][i], pe, acc[hh]);
  }
#pragma unroll
  for (int hh = 0; hh < kMaxHeads; ++hh)      // view term, once per token: lane j == 0
    acc[hh] += (with_view_term && j == 0 && hh < k.H) ? ev[hh] : 0.f;
  group8_sum4(acc[0], acc[1], acc[2], acc[3]);
#pragma unroll
  for (int hh = 0; hh < kMaxHeads; ++hh) out[hh] = acc[hh];
}

// the per-view term erow_{h, ov(t)} of the chunk's tokens, fetched BEFORE the chunk's gathers so
// that its latency runs under theirs.  Raw values: the load is unconditional (`safe` is any
// readable row when there is no view term) and chunk_scores() masks it -- a select here would be
// a use, i.e. a wait, in front of the gathers.
__device__ __forceinline__ void load_view_term(const RayCtx& k, int lane, int t0,
                                               const float* __restrict__ erow, int hs_e,
                                               const float* __restrict__ safe,
                                               float (&ev)[kMaxHeads]) {
  const int tl = lane >> 3;
  const int ov = min(t0 + tl, k.T - 1) % k.ovn;
  const float* src = erow != nullptr ? erow : safe;
#pragma unroll
  for (int hh = 0; hh < kMaxHeads; ++hh)
    ev[hh] = src[(erow != nullptr && hh < k.H) ? hh * hs_e + ov : 0];
}

// phase F accumulators: lanes <-> channels
template <int CK>
struct ContextRegs {
  static constexpr int CPL = CK >= 64 ? CK / 64 : 1;
  float f[kMaxHeads][CPL], p[kMaxHeads], o[kMaxHeads];
  __device__ __forceinline__ void clear() {
#pragma unroll
    for (int hh = 0; hh < kMaxHeads; ++hh) {
      p[hh] = 0.f; o[hh] = 0.f;
#pragma unroll
      for (int i = 0; i < CPL; ++i) f[hh][i] = 0.f;
    }
  }
  __device__ __forceinline__ void scale(const float (&s)[kMaxHeads]) {
#pragma unroll
    for (int hh = 0; hh < kMaxHeads; ++hh) {
      p[hh] *= s[hh]; o[hh] *= s[hh];
#pragma unroll
      for (int i = 0; i < CPL; ++i) f[hh][i] *= s[hh];
    }
  }
};

// phase F: acc += sum over the chunk's tokens of wgt_{h,t} (feat_t | pe_t | [ov(t) == lane]);
// wgt lives in lane 8*tl + 7
template <int CK, bool WITH_O>
__device__ __forceinline__ void chunk_context(const AttnDims& dm, const RayCtx& k, int lane, int t0,
                                              const float (&wgt)[kMaxHeads],
                                              ContextRegs<CK>& A) {
  constexpr int CPL = ContextRegs<CK>::CPL;
  const int c0 = lane * CPL;
  const int fo = c0 < dm.c ? c0 : 0, po = lane < k.P ? lane : 0;
  const int n = min(kChunk, k.T - t0);
  int ov = t0 % k.ovn;
  for (int tl = 0; tl < n; ++tl) {
    float f[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) f[i] = k.featS[tl * k.fs + fo + i];
    const float pe = k.peS[tl * k.Ps + po];
    const bool mine = ov == lane;
#pragma unroll
    for (int hh = 0; hh < kMaxHeads; ++hh) {
      const float a = lane_bcast(wgt[hh], 8 * tl + 7);
#pragma unroll
      for (int i = 0; i < CPL; ++i) A.f[hh][i] = fmaf(a, f[i], A.f[hh][i]);
      A.p[hh] = fmaf(a, pe, A.p[hh]);
      // compile-time: as a run-time flag this became 20 selects / moves per token, executed
      // (and thrown away) in every 2-view run, where there is no view embedding
      if (WITH_O) A.o[hh] += mine ? a : 0.f;
    }
    ov = ov + 1 == k.ovn ? 0 : ov + 1;
  }
}

template <int CK, bool WITH_O, int NT>
__global__ void __launch_bounds__(kAttnWaves* kWave)
epipolar_attn_forward_kernel(AttnDims dm, const float* __restrict__ fmap,
                             const float* __restrict__ xy, const uint8_t* __restrict__ flags,
                             const float* __restrict__ rd, const float* __restrict__ qt,
                             const float* __restrict__ u, const float* __restrict__ e,
                             float scale, float* __restrict__ fbar, float* __restrict__ pbar,
                             float* __restrict__ abar, float* __restrict__ attn) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int CPL = ContextRegs<CK>::CPL;
  RayCtx k;
  if (!ray_setup(dm, smem, k)) return;
  const int lane = threadIdx.x & 63;
  const size_t rh = (size_t)k.ray * k.H;
  QueryRegs Q;
  const size_t ray = (size_t)k.ray;
  float aw_unused[2][kMaxHeads];
  ray_prologue<false, NT>(dm, k, lane, xy, flags, rd, qt + ray * dm.ld_q, dm.hs_q, u + ray * dm.ld_u,
                      dm.hs_u, Q, nullptr, aw_unused);
  const float* erow = e ? e + ray * dm.ld_e : nullptr;

  ContextRegs<CK> A;
  A.clear();
  float m_run[kMaxHeads], l_run[kMaxHeads];
#pragma unroll
  for (int hh = 0; hh < kMaxHeads; ++hh) { m_run[hh] = -__builtin_inff(); l_run[hh] = 0.f; }
  const bool score_lane = (lane & 7) == 7;

  for (int t0 = 0; t0 < k.T; t0 += kChunk) {
    // (WITH_O = "e is given", decided by the launcher: without a view term its loads, the token -> view
    // index arithmetic in front of them and the selects in chunk_scores are not compiled at all)
    float ev[kMaxHeads] = {0.f, 0.f, 0.f, 0.f};
    if (WITH_O) load_view_term(k, lane, t0, erow, dm.hs_e, qt + ray * dm.ld_q, ev);
    stage_chunk<CK>(dm, k, lane, t0, fmap);
    float sc[kMaxHeads];
    chunk_scores<CK>(dm, k, lane, t0, Q, ev, WITH_O && erow != nullptr, sc);
    const int t = t0 + (lane >> 3);
    const bool live = score_lane && t < k.T;
    float mx[kMaxHeads];
#pragma unroll
    for (int hh = 0; hh < kMaxHeads; ++hh) {
      sc[hh] = live ? sc[hh] * scale : -__builtin_inff();
      if (live && hh < k.H) k.scS[hh * k.T + t] = sc[hh];
      mx[hh] = sc[hh];
    }
    lanes8_max4_to_lane63(mx[0], mx[1], mx[2], mx[3]);
    float corr[kMaxHeads], ps[kMaxHeads];
#pragma unroll
    for (int hh = 0; hh < kMaxHeads; ++hh) {
      const float m_new = uniform(fmaxf(m_run[hh], lane_bcast(mx[hh], 63)));
      corr[hh] = uniform(__expf(m_run[hh] - m_new));   // exp(-inf) = 0 on the first chunk
      m_run[hh] = m_new;
      sc[hh] = live ? __expf(sc[hh] - m_new) : 0.f;
      ps[hh] = sc[hh];
    }
    lanes8_sum4_to_lane63(ps[0], ps[1], ps[2], ps[3]);
#pragma unroll
    for (int hh = 0; hh < kMaxHeads; ++hh)
      l_run[hh] = uniform(fmaf(l_run[hh], corr[hh], lane_bcast(ps[hh], 63)));
    A.scale(corr);
    chunk_context<CK, WITH_O>(dm, k, lane, t0, sc, A);
    wave_lds_sync();        // the chunk buffers are overwritten by the next iteration
  }

  const int c0 = lane * CPL;
#pragma unroll
  for (int hh = 0; hh < kMaxHeads; ++hh) {
    if (hh >= k.H) break;
    const float inv = 1.0f / l_run[hh];
    if (c0 < dm.c) {
#pragma unroll
      for (int i = 0; i < CPL; ++i) fbar[ray * dm.ld_f + hh * dm.hs_f + c0 + i] = A.f[hh][i] * inv;
    }
    if (lane < k.P) pbar[ray * dm.ld_p + hh * dm.hs_p + lane] = A.p[hh] * inv;
    if (e != nullptr && lane < k.ovn) abar[ray * dm.ld_a + hh * dm.hs_a + lane] = A.o[hh] * inv;
    if (lane < dm.pad_out) {      // the caller's padding behind the head's last block
      float* tail = e != nullptr ? abar + ray * dm.ld_a + hh * dm.hs_a + k.ovn
                                 : pbar + ray * dm.ld_p + hh * dm.hs_p + k.P;
      tail[lane] = 0.f;
    }
    for (int t = lane; t < k.T; t += kWave)
      attn[(rh + hh) * k.T + t] = __expf(k.scS[hh * k.T + t] - m_run[hh]) * inv;
  }
}

// ------------------------------------------------------------------------------------
// attention backward, per ray, one pass over the tokens:
//   da_t = dfbar . feat_t + dpbar . pe_t + dabar_{ov(t)},   dot = sum_t a_t da_t,
//   ds_t = scale a_t (da_t - dot)                 (needed by the feature-map gradient)
//   dq~  = sum_t ds_t feat_t = scale (sum_t a_t da_t feat_t - dot fbar)   and likewise du, de
// so the weighted sums use w_t = a_t da_t, known per chunk, and the forward outputs
// (fbar, pbar, abar) close the expression -- no second gather.
// ------------------------------------------------------------------------------------
template <int CK, bool WITH_O, int NT>
__global__ void __launch_bounds__(kAttnWaves* kWave)
epipolar_attn_backward_kernel(AttnDims dm, const float* __restrict__ fmap,
                              const float* __restrict__ xy, const uint8_t* __restrict__ flags,
                              const float* __restrict__ rd, const float* __restrict__ attn,
                              const float* __restrict__ fbar, const float* __restrict__ pbar,
                              const float* __restrict__ abar, const float* __restrict__ dfbar,
                              const float* __restrict__ dpbar, const float* __restrict__ dabar,
                              float scale, float* __restrict__ dqt, float* __restrict__ du,
                              float* __restrict__ de, float* __restrict__ ds_out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int CPL = ContextRegs<CK>::CPL;
  RayCtx k;
  if (!ray_setup(dm, smem, k)) return;
  const int lane = threadIdx.x & 63;
  const size_t rh = (size_t)k.ray * k.H;
  QueryRegs Q;
  const size_t ray = (size_t)k.ray;
  // the ray's attention weights, lanes <-> tokens (T <= 128: two per lane), loaded ONCE, with
  // everything else the ray needs up front (ray_prologue): a load inside the chunk loop put one
  // more dependent global latency into every chunk
  float aw[2][kMaxHeads];
  ray_prologue<true, NT>(dm, k, lane, xy, flags, rd, dfbar + ray * dm.ld_f, dm.hs_f,
                     dpbar + ray * dm.ld_p, dm.hs_p, Q, attn + rh * k.T, aw);
  const float* erow = dabar ? dabar + ray * dm.ld_a : nullptr;

  ContextRegs<CK> A;
  A.clear();
  float dot[kMaxHeads] = {0.f, 0.f, 0.f, 0.f};
  const bool score_lane = (lane & 7) == 7;
  for (int t0 = 0; t0 < k.T; t0 += kChunk) {
    // (WITH_O = "dabar or de is given", decided by the launcher)
    float ev[kMaxHeads] = {0.f, 0.f, 0.f, 0.f};
    if (WITH_O) load_view_term(k, lane, t0, erow, dm.hs_a, dfbar + ray * dm.ld_f, ev);
    stage_chunk<CK>(dm, k, lane, t0, fmap);
    float da[kMaxHeads];
    chunk_scores<CK>(dm, k, lane, t0, Q, ev, WITH_O && erow != nullptr, da);
    const int t = t0 + (lane >> 3);
    const bool live = score_lane && t < k.T;
#pragma unroll
    for (int hh = 0; hh < kMaxHeads; ++hh) {
      const bool on = live && hh < k.H;
      // token t sits in lane t % 64, register t / 64 (t0 is wave-uniform, a chunk never
      // straddles the two registers)
      const float a_src = t0 < kWave ? aw[0][hh] : aw[1][hh];
      const float a_t = __shfl(a_src, t & (kWave - 1));
      const float a = on ? a_t : 0.f;
      if (on) k.scS[hh * k.T + t] = da[hh];
      da[hh] = a * da[hh];
      if (!on) da[hh] = 0.f;
      dot[hh] += da[hh];
    }
    chunk_context<CK, WITH_O>(dm, k, lane, t0, da, A);
    wave_lds_sync();
  }
  // the forward outputs that close the expressions: every load issued before the first use
  // (one dependent load -> store pair per head and output was 12 round trips at the end of
  // every wave)
  const int c0 = lane * CPL;
  const bool c_in = c0 < dm.c, p_in = lane < k.P, o_in = de != nullptr && lane < k.ovn;
  float fb[kMaxHeads][CPL], pb[kMaxHeads], ab[kMaxHeads];
  const float* ab_src = abar != nullptr ? abar : pbar;      // any readable row when absent
#pragma unroll
  for (int hh = 0; hh < kMaxHeads; ++hh) {
    const int hs = hh < k.H ? hh : 0;
    load_cpl<CPL>(fbar + ray * dm.ld_f + hs * dm.hs_f + (c_in ? c0 : 0), fb[hh]);
    pb[hh] = pbar[ray * dm.ld_p + hs * dm.hs_p + (p_in ? lane : 0)];
    ab[hh] = ab_src[abar != nullptr ? ray * dm.ld_a + hs * dm.hs_a + (o_in ? lane : 0)
                                    : ray * dm.ld_p];
  }
  wave_sum4_to_lane63(dot[0], dot[1], dot[2], dot[3]);
#pragma unroll
  for (int hh = 0; hh < kMaxHeads; ++hh) {
    if (hh >= k.H) break;
    const float d = lane_bcast(dot[hh], 63);
    if (c_in) {
#pragma unroll
      for (int i = 0; i < CPL; ++i)
        dqt[ray * dm.ld_q + hh * dm.hs_q + c0 + i] = scale * (A.f[hh][i] - d * fb[hh][i]);
    }
    if (p_in) du[ray * dm.ld_u + hh * dm.hs_u + lane] = scale * (A.p[hh] - d * pb[hh]);
    if (o_in) {
      // abar is only produced when the view embedding exists; otherwise it is the softmax mass
      // of the single other view, i.e. one
      de[ray * dm.ld_e + hh * dm.hs_e + lane] =
          scale * (A.o[hh] - d * (abar != nullptr ? ab[hh] : 1.0f));
    }
    if (lane < dm.pad_in) {
      float* tail = de != nullptr ? de + ray * dm.ld_e + hh * dm.hs_e + k.ovn
                                  : du + ray * dm.ld_u + hh * dm.hs_u + k.P;
      tail[lane] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = lane + u * kWave;
      if (t < k.T) ds_out[(rh + hh) * k.T + t] = scale * aw[u][hh] * (k.scS[hh * k.T + t] - d);
    }
  }
}

// ------------------------------------------------------------------------------------
// feature-map gradient: dF[src][y][x][c] += w_corner * sum_h (a dfbar_h[c] + ds q~_h[c])
//
// grid_sample's backward is a scatter of (rays x tokens x 4 corners x c) float atomics
// (~1e9 per layer at the paper config).  Measured on MI355X, neither flavour of float atomic
// is usable at that count: global atomics reach 21-161 G/s, and a wave-level LDS
// ds_add_f32 takes ~193 clocks whatever the address pattern (0.33 lanes/clk per CU,
// tools/lds_atomic_microbench.hip) against ~10 clocks for a plain ds_read/add/ds_write.
// So the scatter is turned into an owner-computes pass with NO atomics:
//   * one wave owns a TS x TS pixel tile of one source image, all channels, in LDS
//     (TS*TS pixels * c floats + dummy slots); lanes <-> channels, so one LDS read-modify-write touches 64
//     distinct addresses and a wave is the only writer of its tile;
//   * the wave culls the rays of the casting views against its tile with a packed pixel
//     bounding box of each ray's samples (64 rays per test, one ballot), then, for a hit,
//     computes the corner records of the ray's samples lanes <-> samples and walks only the
//     samples that really touch the tile (second ballot);
//   * corners outside the tile/image are redirected to a dummy pixel with weight 0 (it stays
//     zero), so the four read-modify-writes of a sample are branch free and independent.
// The summation order is fixed (view, ray, sample, corner): the gradient is bit-reproducible,
// which atomics never were.  Every pixel is written exactly once; dfmap needs no memset.
// ------------------------------------------------------------------------------------
// `tile_work` (optional): per (source image, TS x TS tile) a work estimate for the two-pass
// gather, which orders its blocks by it (longest first): the number of ray boxes that overlap
// the tile, long thin boxes of slanted lines weighted down.  Neighbouring rays have nearly the
// same box, so the counts are pre-aggregated in an LDS histogram per block of 256 rays (global
// atomics straight from the rays cost 0.33 ms at configs[3]) and only its non-zero bins go out.
template <int TS>
__global__ void __launch_bounds__(256)
epipolar_ray_box_kernel(AttnDims dm, const float* __restrict__ xy,
                        const uint8_t* __restrict__ flags, uint32_t* __restrict__ boxes,
                        uint32_t* __restrict__ tile_work) {
  extern __shared__ uint32_t hist[];                // [tiles] when tile_work != nullptr
  const int R = dm.h * dm.w, ovn = dm.v - 1;
  const int tiles_x = (dm.w + TS - 1) / TS, tiles_y = (dm.h + TS - 1) / TS;
  const int tiles = tiles_x * tiles_y;
  const size_t n = (size_t)dm.b * dm.v * ovn * R;
  const size_t ro_first = (size_t)blockIdx.x * blockDim.x;
  const size_t ro = ro_first + threadIdx.x;
  // all rays of the block look into the same source image?  (always when 256 divides h*w)
  const size_t ro_last = min(ro_first + blockDim.x, n) - 1;
  const bool one_src = tile_work != nullptr && ro_first / R == ro_last / R;
  if (one_src) {
    for (int i = threadIdx.x; i < tiles; i += blockDim.x) hist[i] = 0u;
    __syncthreads();
  }
  size_t src = 0;
  if (ro < n) {
    uint32_t box = 0x00FF00FFu;                       // empty: min 255 > max 0
    if (flags[ro] & 1) {
      const float2 p0 = *reinterpret_cast<const float2*>(xy + 2 * (ro * dm.s));
      const float2 p1 = *reinterpret_cast<const float2*>(xy + 2 * (ro * dm.s + dm.s - 1));
      const Corner a = corner_of(p0.x, p0.y, dm.w, dm.h), c = corner_of(p1.x, p1.y, dm.w, dm.h);
      // the samples lie on the segment between the first and the last one; one pixel of slack
      // for the rounding of the interpolation, then the +1 corner
      const int x0 = max(min(a.x0, c.x0) - 1, 0), x1 = min(max(a.x0, c.x0) + 2, dm.w - 1);
      const int y0 = max(min(a.y0, c.y0) - 1, 0), y1 = min(max(a.y0, c.y0) + 2, dm.h - 1);
      if (x0 <= x1 && y0 <= y1) {
        box = (uint32_t)x0 | ((uint32_t)x1 << 8) | ((uint32_t)y0 << 16) | ((uint32_t)y1 << 24);
        if (tile_work != nullptr) {
          // source image of this ray: (b, other view ov of casting view v)
          const int ov = (int)((ro / R) % ovn);
          const size_t bv = ro / ((size_t)R * ovn);
          const int v = (int)(bv % dm.v);
          src = bv - v + (ov < v ? ov : ov + 1);
          const int tx0 = x0 / TS, tx1 = x1 / TS, ty0 = y0 / TS, ty1 = y1 / TS;
          const int nx = tx1 - tx0 + 1, ny = ty1 - ty0 + 1;
          // ~16 x (tiles the segment crosses / tiles in its box); only the ORDER of the blocks
          // depends on the estimate
          const uint32_t wgt = (uint32_t)max(1, 16 / min(nx, ny));
          uint32_t* tw = one_src ? hist : tile_work + src * (size_t)tiles;
          for (int ty = ty0; ty <= ty1; ++ty)
            for (int tx = tx0; tx <= tx1; ++tx) atomicAdd(tw + ty * tiles_x + tx, wgt);
        }
      }
    }
    boxes[ro] = box;
  }
  if (one_src) {
    __syncthreads();
    // the block's source image: that of its first ray
    const int ov = (int)((ro_first / R) % ovn);
    const size_t bv = ro_first / ((size_t)R * ovn);
    const int v = (int)(bv % dm.v);
    uint32_t* tw = tile_work + (bv - v + (ov < v ? ov : ov + 1)) * (size_t)tiles;
    for (int i = threadIdx.x; i < tiles; i += blockDim.x)
      if (hist[i] != 0u) atomicAdd(tw + i, hist[i]);
  }
}

// Longest-first block order for the gather: a coarse (power-of-two buckets) descending sort of
// the n_work tiles by their work estimate, one block.  The order inside a bucket depends on the
// atomics' arrival order; it changes the schedule, never a result.
__global__ void __launch_bounds__(1024)
epipolar_tile_order_kernel(int n_work, const uint32_t* __restrict__ tile_work,
                           uint32_t* __restrict__ order) {
  __shared__ uint32_t count[32], base[32];
  if (threadIdx.x < 32) count[threadIdx.x] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n_work; i += blockDim.x)
    atomicAdd(&count[31 - __clz((int)(tile_work[i] | 1u))], 1u);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int bkt = 31; bkt >= 0; --bkt) { base[bkt] = run; run += count[bkt]; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_work; i += blockDim.x) {
    const int bkt = 31 - __clz((int)(tile_work[i] | 1u));
    order[atomicAdd(&base[bkt], 1u)] = (uint32_t)i;
  }
}

__device__ __forceinline__ int lane_bcast_i(int v, int lane) {
  return __builtin_amdgcn_readlane(v, lane);
}

constexpr int kDfWaves = 4;   // waves per tile, each with a private copy (rays interleaved)
constexpr int kDfChunk = 2048; // rays culled per pass (bounds the per-wave hit list)
constexpr int kMaxFgradLayers = 2;

// The attention layers of one encoder share the geometry (same samples, same features): their
// feature-map gradients differ only in the per-sample coefficients, so NL layers are
// scattered in ONE pass -- the cull, the corner records and the LDS read-modify-writes are
// paid once, only the coefficient rows and the FMAs that build df scale with NL.
struct FgradLayers {
  const float* attn[kMaxFgradLayers];
  const float* ds[kMaxFgradLayers];
  const float* dfbar[kMaxFgradLayers];
  const float* qt[kMaxFgradLayers];
};

template <int CPL, int TS, int NL>
__global__ void __launch_bounds__(kDfWaves* kWave)
epipolar_dfmap_kernel(AttnDims dm, int n_work, const float* __restrict__ xy,
                      const uint32_t* __restrict__ boxes, FgradLayers L,
                      float* __restrict__ dfmap) {
  extern __shared__ __attribute__((aligned(16))) float tiles[];  // [kDfWaves][TS*TS pixels + dummy][c]
  using V = typename LaneVec<CPL>::type;
  const int R = dm.h * dm.w, ovn = dm.v - 1, T = dm.s * ovn, H = dm.heads;
  const int tiles_x = (dm.w + TS - 1) / TS, tiles_y = (dm.h + TS - 1) / TS;
  // XCD-aware order: hardware hands block i to XCD i % 8, each with its own 4 MB L2.  A ray's
  // rows (dfbar, q~: 4 KB) are needed by every tile its line crosses, so neighbouring tiles
  // of one image must share an L2: XCD x works on the x-th contiguous eighth of the tiles
  // (measured: 3 GB -> see DESIGN.md of memory-side fetches per launch before the remap).
  const int per_xcd = (int)gridDim.x / 8;
  const int work = ((int)blockIdx.x % 8) * per_xcd + (int)blockIdx.x / 8;
  if (work >= n_work) return;
  const int tile_id = work % (tiles_x * tiles_y);
  const int src_bv = work / (tiles_x * tiles_y);                // (b, source view)
  const int b = src_bv / dm.v, sv = src_bv % dm.v;
  const int tx0 = (tile_id % tiles_x) * TS, ty0 = (tile_id / tiles_x) * TS;
  const int tx1 = min(tx0 + TS, dm.w) - 1, ty1 = min(ty0 + TS, dm.h) - 1;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c0 = lane * CPL;
  const bool lane_c = c0 < dm.c;
  const int cl = lane_c ? c0 : 0;
  const int tile_floats = TS * TS * dm.c + max(dm.c, kWave * CPL);   // + dummy slots
  float* tile = tiles + (size_t)wv * tile_floats;
  float* lane_base = tile + (lane_c ? 0 : TS * TS * dm.c) + c0;
  const int lane_mul = lane_c ? 1 : 0;
  for (int i = lane; i < tile_floats; i += kWave) tile[i] = 0.f;
  wave_lds_sync();

  // per-wave list of the rays whose box overlaps the tile (uint16 ray index inside the
  // current super-chunk of kDfChunk rays), behind the tile copies
  uint16_t* list = reinterpret_cast<uint16_t*>(tiles + (size_t)kDfWaves * tile_floats) +
                   (size_t)wv * (kDfChunk / kDfWaves);
  const int ngroups = (dm.s + kWave - 1) / kWave;                // 1 when s <= 64

  // per-item registers: sample positions and coefficients (lanes <-> samples) are prefetched
  // one item ahead; the coefficient rows (lanes <-> channels) too when there is one layer,
  // with two they are loaded at the start of the item to stay under 128 VGPRs
  struct Coef {
    float2 p;
    float av[NL][kMaxHeads], dv[NL][kMaxHeads];
  };
  struct Rows {
    float gq[NL][kMaxHeads][CPL], qq[NL][kMaxHeads][CPL];
  };
  constexpr bool kPrefetchRows = NL == 1;

  for (int v = 0; v < dm.v; ++v) {
    if (v == sv) continue;
    const int ov = sv < v ? sv : sv - 1;
    const size_t bv = (size_t)b * dm.v + v;
    const size_t ro0 = (bv * ovn + ov) * R;
    for (int chunk0 = 0; chunk0 < R; chunk0 += kDfChunk) {
      // 1. cull: lanes <-> rays, matched rays appended to the wave's list
      int count = 0;
      for (int r0 = chunk0 + wv * kWave; r0 < min(chunk0 + kDfChunk, R); r0 += kDfWaves * kWave) {
        const int r = r0 + lane;
        const uint32_t box = r < R ? boxes[ro0 + r] : 0x00FF00FFu;
        const int bx0 = box & 255, bx1 = (box >> 8) & 255, by0 = (box >> 16) & 255, by1 = box >> 24;
        bool hit = bx0 <= tx1 && bx1 >= tx0 && by0 <= ty1 && by1 >= ty0;
        if (hit) {
          // exact test: a sample touches the tile iff its pixel position lies in
          // [tx0 - 1, tx1 + 1) x [ty0 - 1, ty1 + 1); the samples lie on the segment between
          // the first and the last one (Liang-Barsky clip, 0.02 px of slack for rounding)
          const size_t so = (ro0 + r) * dm.s;
          const float2 p0 = *reinterpret_cast<const float2*>(xy + 2 * so);
          const float2 p1 = *reinterpret_cast<const float2*>(xy + 2 * (so + dm.s - 1));
          const Corner ca = corner_of(p0.x, p0.y, dm.w, dm.h), cb = corner_of(p1.x, p1.y, dm.w, dm.h);
          const float ax = (float)ca.x0 + ca.wx, ay = (float)ca.y0 + ca.wy;
          const float dx = (float)cb.x0 + cb.wx - ax, dy = (float)cb.y0 + cb.wy - ay;
          const float xlo = (float)tx0 - 1.02f, xhi = (float)tx1 + 1.02f;
          const float ylo = (float)ty0 - 1.02f, yhi = (float)ty1 + 1.02f;
          float ta = 0.f, tb = 1.f;
          auto clip = [&](float pp, float qq) {       // pp * t <= qq
            if (pp == 0.f) { if (qq < 0.f) tb = -1.f; return; }
            const float rr = qq / pp;
            if (pp < 0.f) ta = fmaxf(ta, rr); else tb = fminf(tb, rr);
          };
          clip(-dx, ax - xlo); clip(dx, xhi - ax); clip(-dy, ay - ylo); clip(dy, yhi - ay);
          hit = ta <= tb;
        }
        const uint64_t m = __ballot(hit);
        if (hit) list[count + __popcll(m & lanemask_lt())] = (uint16_t)(r - chunk0);
        count += __popcll(m);
      }
      wave_lds_sync();
      if (count == 0) continue;
      // 2. walk the list; the loads of item i+1 are in flight while item i is processed
      //    (a matched ray costs two dependent global-load latencies otherwise)
      auto fetch_coef = [&](int item, Coef& x) {
        const int rr = chunk0 + list[item / ngroups];
        const int si = min((item % ngroups) * kWave + lane, dm.s - 1);
        const size_t ray = bv * R + rr;
        x.p = *reinterpret_cast<const float2*>(xy + 2 * ((ro0 + rr) * dm.s + si));
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
          for (int hh = 0; hh < kMaxHeads; ++hh) {
            const size_t rowh = ray * H + (hh < H ? hh : 0);
            // absent heads (hh >= H) get zero coefficients: the sample loop below has no
            // per-head branches
            const float a_ = L.attn[l][rowh * T + si * ovn + ov];
            const float d_ = L.ds[l][rowh * T + si * ovn + ov];
            x.av[l][hh] = hh < H ? a_ : 0.f;
            x.dv[l][hh] = hh < H ? d_ : 0.f;
          }
      };
      auto fetch_rows = [&](int item, Rows& x) {
        const size_t ray = bv * R + chunk0 + list[item / ngroups];
#pragma unroll
        for (int l = 0; l < NL; ++l)
#pragma unroll
          for (int hh = 0; hh < kMaxHeads; ++hh) {
            const int hs = hh < H ? hh : 0;
            load_cpl<CPL>(L.dfbar[l] + ray * dm.ld_f + hs * dm.hs_f + cl, x.gq[l][hh]);
            load_cpl<CPL>(L.qt[l] + ray * dm.ld_q + hs * dm.hs_q + cl, x.qq[l][hh]);
          }
      };
      const int n_items = count * ngroups;
      Coef cur;
      Rows rows, rows_nxt;
      fetch_coef(0, cur);
      if (kPrefetchRows) fetch_rows(0, rows);
      for (int item = 0; item < n_items; ++item) {
        Coef nxt;
        if (!kPrefetchRows) fetch_rows(item, rows);
        fetch_coef(min(item + 1, n_items - 1), nxt);
        if (kPrefetchRows) fetch_rows(min(item + 1, n_items - 1), rows_nxt);
        // lanes <-> samples: corner records relative to this tile
        const bool tok = (item % ngroups) * kWave + lane < dm.s;
        const Corner kq = corner_of(cur.p.x, cur.p.y, dm.w, dm.h);
        int off[4]; float wt[4]; bool any = false;
#pragma unroll
        for (int cr = 0; cr < 4; ++cr) {
          const int xx = kq.x0 + (cr & 1), yy = kq.y0 + (cr >> 1);
          const bool in = tok && xx >= tx0 && xx <= tx1 && yy >= ty0 && yy <= ty1;
          const float wx = (cr & 1) ? kq.wx : 1.f - kq.wx, wy = (cr >> 1) ? kq.wy : 1.f - kq.wy;
          off[cr] = (in ? (yy - ty0) * TS + (xx - tx0) : TS * TS) * dm.c;
          wt[cr] = in ? wx * wy : 0.f;
          any |= in;
        }
        uint64_t toks = __ballot(any);
        // Every lane runs this loop: the token lanes' records are read with v_readlane, and a
        // value whose only use sits inside a divergent branch may be computed only for the
        // lanes that take it.  Lanes past the last channel work on a private dummy slot.
        while (toks) {
          const int tl = __builtin_ctzll(toks);
          toks &= toks - 1;
          float df[CPL];
#pragma unroll
          for (int i = 0; i < CPL; ++i) df[i] = 0.f;
#pragma unroll
          for (int l = 0; l < NL; ++l)
#pragma unroll
            for (int hh = 0; hh < kMaxHeads; ++hh) {
              const float a = lane_bcast(cur.av[l][hh], tl), d = lane_bcast(cur.dv[l][hh], tl);
#pragma unroll
              for (int i = 0; i < CPL; ++i)
                df[i] = fmaf(a, rows.gq[l][hh][i], fmaf(d, rows.qq[l][hh][i], df[i]));
            }
          // four distinct pixels (or the dummy): reads first, then the writes
          V* dst[4]; V val[4];
#pragma unroll
          for (int cr = 0; cr < 4; ++cr) {
            // lanes past the last channel: multiplier 0 and a base inside the dummy slots
            dst[cr] = reinterpret_cast<V*>(lane_base + lane_bcast_i(off[cr], tl) * lane_mul);
            val[cr] = *dst[cr];
          }
#pragma unroll
          for (int cr = 0; cr < 4; ++cr) {
            const float wgt = lane_bcast(wt[cr], tl);
            float* f = reinterpret_cast<float*>(&val[cr]);
#pragma unroll
            for (int i = 0; i < CPL; ++i) f[i] = fmaf(wgt, df[i], f[i]);
            *dst[cr] = val[cr];
          }
        }
        cur = nxt;
        if (kPrefetchRows) rows = rows_nxt;
      }
      wave_lds_sync();
    }
  }
  __syncthreads();
  // fixed-order sum of the private copies, coalesced rows out
  float* out = dfmap + (size_t)src_bv * R * dm.c;
  const int tw = tx1 - tx0 + 1, th = ty1 - ty0 + 1;
  for (int i = threadIdx.x; i < tw * th * dm.c; i += kDfWaves * kWave) {
    const int pix = i / dm.c, ch = i - pix * dm.c;
    const int py = pix / tw, px = pix - py * tw;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < kDfWaves; ++k)
      acc += tiles[(size_t)k * tile_floats + (py * TS + px) * dm.c + ch];
    out[((size_t)(ty0 + py) * dm.w + tx0 + px) * dm.c + ch] = acc;
  }
}

// ------------------------------------------------------------------------------------
// Two-pass variant of the feature-map gradient.
//
// The single-pass kernel above reloads a ray's 16 coefficient rows (dfbar_h, q~_h of both layers:
// 8 KB at c = 128) and the coefficients of all its samples (4.6 KB) for EVERY tile its line
// crosses (~10 tiles): 14 GB of L2-level reads and 4.7 GB on the memory side per launch at the
// paper config for 0.47 GB of rows, and 80 wave instructions per (token, tile) visit of which
// 48 rebuild the token's gradient vector from the rows.
//   pass 1 (token_grad_kernel): one wave per (ray, other view) builds, ONCE per token,
//       g_t[c] = sum over layers, heads of a_t dfbar[c] + ds_t q~[c]
//     -- the gradient w.r.t. the gathered feature of that token -- and streams it out
//     (b v ov r s c floats: what the reference's autograd holds as d(kv); 0.94 GB at the paper
//     config, written and read once);
//   pass 2 (dfmap_list_gather_kernel, further down): the same tile-owner scatter as above, but over
//     per-tile token lists built by a binning pass; per token ONE 512-byte row of g_t is loaded.
// Same fixed summation order per pixel (view, ray, sample, corner) => bit-reproducible; the
// value differs from the single pass only by the association of the layer / head sum.
// ------------------------------------------------------------------------------------
template <int CPL, int NL>
__global__ void __launch_bounds__(256)
epipolar_token_grad_kernel(AttnDims dm, const uint8_t* __restrict__ flags, FgradLayers L,
                           float* __restrict__ tg) {
  using V = typename LaneVec<CPL>::type;
  const int R = dm.h * dm.w, ovn = dm.v - 1, T = dm.s * ovn, H = dm.heads;
  const size_t n_ro = (size_t)dm.b * dm.v * ovn * R;
  const size_t ro = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);      // (bv, ov, r)
  if (ro >= n_ro) return;
  if (!(flags[ro] & 1)) return;            // no overlap: every weight is zero, never read
  const int lane = threadIdx.x & 63;
  const int r = (int)(ro % R);
  const int ov = (int)((ro / R) % ovn);
  const size_t bv = ro / ((size_t)R * ovn);
  const size_t ray = bv * R + r;
  const int c0 = lane * CPL;
  const bool lane_c = c0 < dm.c;
  const int cl = lane_c ? c0 : 0;
  float gq[NL][kMaxHeads][CPL], qq[NL][kMaxHeads][CPL];
#pragma unroll
  for (int l = 0; l < NL; ++l)
#pragma unroll
    for (int hh = 0; hh < kMaxHeads; ++hh) {
      const int hs = hh < H ? hh : 0;
      load_cpl<CPL>(L.dfbar[l] + ray * dm.ld_f + hs * dm.hs_f + cl, gq[l][hh]);
      load_cpl<CPL>(L.qt[l] + ray * dm.ld_q + hs * dm.hs_q + cl, qq[l][hh]);
    }
  for (int s0 = 0; s0 < dm.s; s0 += kWave) {
    // lanes <-> samples: the coefficients of this (ray, other view)
    const int si = min(s0 + lane, dm.s - 1);
    float av[NL][kMaxHeads], dv[NL][kMaxHeads];
#pragma unroll
    for (int l = 0; l < NL; ++l)
#pragma unroll
      for (int hh = 0; hh < kMaxHeads; ++hh) {
        const size_t rowh = ray * H + (hh < H ? hh : 0);
        const float a_ = L.attn[l][rowh * T + si * ovn + ov];
        const float d_ = L.ds[l][rowh * T + si * ovn + ov];
        av[l][hh] = hh < H ? a_ : 0.f;      // absent heads: zero coefficients, no branches below
        dv[l][hh] = hh < H ? d_ : 0.f;
      }
    const int n = min(kWave, dm.s - s0);
    for (int tl = 0; tl < n; ++tl) {        // lanes <-> channels
      float df[CPL];
#pragma unroll
      for (int i = 0; i < CPL; ++i) df[i] = 0.f;
#pragma unroll
      for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int hh = 0; hh < kMaxHeads; ++hh) {
          const float a = lane_bcast(av[l][hh], tl), d = lane_bcast(dv[l][hh], tl);
#pragma unroll
          for (int i = 0; i < CPL; ++i) df[i] = fmaf(a, gq[l][hh][i], fmaf(d, qq[l][hh][i], df[i]));
        }
      if (lane_c) {
        V out;
        float* f = reinterpret_cast<float*>(&out);
#pragma unroll
        for (int i = 0; i < CPL; ++i) f[i] = df[i];
        *reinterpret_cast<V*>(tg + (ro * dm.s + s0 + tl) * dm.c + c0) = out;
      }
    }
  }
}

// ------------------------------------------------------------------------------------
// Binned gather (round 3): the per-tile token lists are BUILT once per flush by a stable count /
// scan / fill pass over the tokens -- the rasterizer's binning scheme -- instead of being
// re-discovered by every tile with a sweep over all rays of the casting view.
//
//   count   a block = 256 consecutive tokens (ray-major, sample-minor) of one (casting view, other
//           view) pair: every token's bilinear footprint touches 1, 2 or 4 tiles of the source map;
//           an LDS histogram per block, written out as one row cnt[src][block][tile];
//   scan    per (source map, tile): exclusive prefix over the map's blocks (in place), the tile's
//           length; then one block orders the tiles longest first and turns the lengths into
//           global offsets;
//   fill    the same blocks again: per-wave counts -> per-(wave, tile) start = tile offset + block
//           prefix + earlier waves; inside a wave the rank of a token among the lanes with the same
//           tile (wave64 match by ballots, as in the radix sort), footprint slot by slot;
//   gather  a block per tile walks ITS list, a contiguous quarter per wave: lanes <-> list entries
//           for the sample position / corner records, then row by row as before.
// The order inside a tile's list -- (casting view, block, wave, footprint slot, lane) -- depends on
// the geometry of that source map only: the gradient is bit-reproducible and a batched launch equals
// single-scene launches bit for bit (tests/test_epipolar_gpu.py).
// ------------------------------------------------------------------------------------
constexpr int kBinTokens = 256;          // tokens per count / fill block
constexpr int kBinTileBits = 12;         // the fill's wave match compares this many bits of a tile index
constexpr size_t kBinFillLdsMax = 64 * 1024;   // dynamic LDS the fill kernel may ask for
constexpr int kDfTokGroup = 4;           // token rows in flight per wave in the list gather

struct BinDims {
  int tiles_x, tiles_y, tiles;           // TS x TS tiles per map
  int blocks_per_pair;                   // count / fill blocks per (casting view, other view) pair
  int blocks_per_src;                    // = (v - 1) * blocks_per_pair
  int n_src;                             // b * v source maps
};
template <int TS>
__host__ __device__ inline BinDims make_bin_dims(const AttnDims& dm) {
  BinDims bd;
  bd.tiles_x = (dm.w + TS - 1) / TS; bd.tiles_y = (dm.h + TS - 1) / TS;
  bd.tiles = bd.tiles_x * bd.tiles_y;
  bd.blocks_per_pair = (dm.h * dm.w * dm.s + kBinTokens - 1) / kBinTokens;
  bd.blocks_per_src = (dm.v - 1) * bd.blocks_per_pair;
  bd.n_src = dm.b * dm.v;
  return bd;
}

// the (up to 4) distinct tiles of a token's bilinear footprint; returns their number
template <int TS>
__device__ __forceinline__ int footprint_tiles(const AttnDims& dm, const BinDims& bd, float x, float y,
                                               int (&tile)[4]) {
  const Corner k = corner_of(x, y, dm.w, dm.h);
  const bool xin0 = k.x0 >= 0 && k.x0 < dm.w, xin1 = k.x0 + 1 >= 0 && k.x0 + 1 < dm.w;
  const bool yin0 = k.y0 >= 0 && k.y0 < dm.h, yin1 = k.y0 + 1 >= 0 && k.y0 + 1 < dm.h;
  const int tx0 = xin0 ? k.x0 / TS : -1, tx1 = xin1 ? (k.x0 + 1) / TS : -1;
  const int ty0 = yin0 ? k.y0 / TS : -1, ty1 = yin1 ? (k.y0 + 1) / TS : -1;
  // candidate tile columns / rows (deduplicated: the two corners of a side share a tile unless the
  // side crosses a tile boundary)
  int txs[2] = {0, 0}, tys[2] = {0, 0}, nx = 0, ny = 0;
  if (tx0 >= 0) txs[nx++] = tx0;
  if (tx1 >= 0 && tx1 != tx0) txs[nx++] = tx1;
  if (ty0 >= 0) tys[ny++] = ty0;
  if (ty1 >= 0 && ty1 != ty0) tys[ny++] = ty1;
  int n = 0;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool on = i < nx && j < ny;
      tile[2 * j + i] = on ? tys[j < ny ? j : 0] * bd.tiles_x + txs[i < nx ? i : 0] : -1;
      n += on;
    }
  return n;
}

// token `idx` (0 .. R*s) of block `blk` of pair (bv, ov): its source map, its row in cnt, its tiles
struct BinToken { bool on; uint32_t tok; int tile[4]; };
template <int TS>
__device__ __forceinline__ BinToken bin_token(const AttnDims& dm, const BinDims& bd, int pair, int blk,
                                              int t, const float* __restrict__ xy,
                                              const uint8_t* __restrict__ flags) {
  const int R = dm.h * dm.w;
  const int idx = blk * kBinTokens + t;                 // (ray, sample) of the pair
  BinToken o;
  o.on = idx < R * dm.s;
  const int r = o.on ? idx / dm.s : 0, smp = o.on ? idx - r * dm.s : 0;
  const size_t ro = (size_t)pair * R + r;
  o.tok = (uint32_t)(ro * dm.s + smp);
  o.on = o.on && (flags[ro] & 1);
  const float2 p = *reinterpret_cast<const float2*>(xy + 2 * (size_t)o.tok);
  const int n = footprint_tiles<TS>(dm, bd, p.x, p.y, o.tile);
  (void)n;
  if (!o.on) { o.tile[0] = o.tile[1] = o.tile[2] = o.tile[3] = -1; }
  return o;
}

// source map and position of a pair's blocks among the blocks of that map: pair = (bv, ov) with
// bv = b * V + v (casting view v), ov the index of the other view; the maps' lists run over the
// casting views in increasing v
__device__ __forceinline__ void pair_source(const AttnDims& dm, int pair, int& src, int& slot) {
  const int ovn = dm.v - 1;
  const int bv = pair / ovn, ov = pair - bv * ovn;
  const int v = bv % dm.v, b0 = bv - v;
  const int sv = ov < v ? ov : ov + 1;                  // the other (source) view
  src = b0 + sv;
  slot = v < sv ? v : v - 1;                            // index of v among the views != sv
}

template <int TS>
__global__ void __launch_bounds__(kBinTokens)
epipolar_bin_count_kernel(AttnDims dm, const float* __restrict__ xy,
                          const uint8_t* __restrict__ flags, uint32_t* __restrict__ cnt) {
  extern __shared__ uint32_t hist[];                    // [tiles]
  const BinDims bd = make_bin_dims<TS>(dm);
  const int pair = blockIdx.x / bd.blocks_per_pair, blk = blockIdx.x % bd.blocks_per_pair;
  for (int i = threadIdx.x; i < bd.tiles; i += kBinTokens) hist[i] = 0u;
  __syncthreads();
  const BinToken k = bin_token<TS>(dm, bd, pair, blk, threadIdx.x, xy, flags);
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (k.tile[j] >= 0) atomicAdd(&hist[k.tile[j]], 1u);
  __syncthreads();
  int src, slot;
  pair_source(dm, pair, src, slot);
  uint32_t* row = cnt + ((size_t)src * bd.blocks_per_src + (size_t)slot * bd.blocks_per_pair + blk) * bd.tiles;
  for (int i = threadIdx.x; i < bd.tiles; i += kBinTokens) row[i] = hist[i];
}

// (source map, tile): exclusive prefix of its counts over the map's blocks, in place.  Four threads
// per tile, each owning a quarter of the blocks (a single thread per tile is a chain of
// blocks_per_src / 16 dependent round trips: 28 us at configs[1]); a block = 64 tiles x 4 segments,
// the tile index fastest so that the loads of a warp are coalesced rows of cnt.
__global__ void __launch_bounds__(256)
epipolar_bin_scan_kernel(int n_src, int tiles, int blocks_per_src, uint32_t* __restrict__ cnt,
                         uint32_t* __restrict__ tile_len) {
  __shared__ uint32_t seg_sum[4][64];
  const int lt = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lt;                   // (source map, tile)
  const bool on = i < n_src * tiles;
  const int src = on ? i / tiles : 0, tile = on ? i - src * tiles : 0;
  uint32_t* col = cnt + (size_t)src * blocks_per_src * tiles + tile;
  const int per = (blocks_per_src + 3) / 4;
  const int b_lo = min(seg * per, blocks_per_src), b_hi = min(b_lo + per, blocks_per_src);
  uint32_t sum = 0;
  for (int b0 = b_lo; b0 < b_hi; b0 += 16) {
    uint32_t t16[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) t16[k] = col[(size_t)min(b0 + k, blocks_per_src - 1) * tiles];
#pragma unroll
    for (int k = 0; k < 16; ++k) sum += b0 + k < b_hi ? t16[k] : 0u;
  }
  seg_sum[seg][lt] = sum;
  __syncthreads();
  uint32_t run = 0;
  for (int s2 = 0; s2 < seg; ++s2) run += seg_sum[s2][lt];
  const uint32_t total = seg_sum[0][lt] + seg_sum[1][lt] + seg_sum[2][lt] + seg_sum[3][lt];
  if (!on) return;
  for (int b0 = b_lo; b0 < b_hi; b0 += 16) {
    uint32_t t16[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) t16[k] = col[(size_t)min(b0 + k, blocks_per_src - 1) * tiles];
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (b0 + k < b_hi) { col[(size_t)(b0 + k) * tiles] = run; run += t16[k]; }
  }
  if (seg == 0) tile_len[i] = total;
}

// one block: global exclusive offsets of the n tiles' lists (tile index order)
__global__ void __launch_bounds__(1024)
epipolar_bin_offsets_kernel(int n, const uint32_t* __restrict__ tile_len, uint32_t* __restrict__ tile_off) {
  __shared__ uint32_t part[1024];
  const int t = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int i0 = t * per, i1 = min(i0 + per, n);
  uint32_t sum = 0;
  for (int i = i0; i < i1; ++i) sum += tile_len[i];
  part[t] = sum;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const uint32_t y = t >= off ? part[t - off] : 0u;
    __syncthreads();
    part[t] += y;
    __syncthreads();
  }
  uint32_t run = part[t] - sum;
  for (int i = i0; i < i1; ++i) { tile_off[i] = run; run += tile_len[i]; }
}

template <int TS>
__global__ void __launch_bounds__(kBinTokens)
epipolar_bin_fill_kernel(AttnDims dm, const float* __restrict__ xy, const uint8_t* __restrict__ flags,
                         const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ tile_off,
                         uint32_t* __restrict__ list) {
  extern __shared__ uint32_t lds[];                     // [4 waves][tiles] counts, then running starts
  constexpr int NW = kBinTokens / kWave;
  const BinDims bd = make_bin_dims<TS>(dm);
  const int pair = blockIdx.x / bd.blocks_per_pair, blk = blockIdx.x % bd.blocks_per_pair;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < NW * bd.tiles; i += kBinTokens) lds[i] = 0u;
  __syncthreads();
  const BinToken k = bin_token<TS>(dm, bd, pair, blk, threadIdx.x, xy, flags);
  uint32_t* mine = lds + (size_t)w * bd.tiles;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (k.tile[j] >= 0) atomicAdd(&mine[k.tile[j]], 1u);
  __syncthreads();
  int src, slot;
  pair_source(dm, pair, src, slot);
  const uint32_t* row = cnt + ((size_t)src * bd.blocks_per_src + (size_t)slot * bd.blocks_per_pair + blk) * bd.tiles;
  const uint32_t* toff = tile_off + (size_t)src * bd.tiles;
  for (int i = threadIdx.x; i < bd.tiles; i += kBinTokens) {
    uint32_t run = toff[i] + row[i];                    // list start of (tile, this block)
#pragma unroll
    for (int ww = 0; ww < NW; ++ww) {
      const uint32_t c = lds[(size_t)ww * bd.tiles + i];
      lds[(size_t)ww * bd.tiles + i] = run;             // start of (tile, wave)
      run += c;
    }
  }
  __syncthreads();
  // footprint slot by slot: rank among the wave's lanes with the same tile (match by ballots over
  // the kBinTileBits bits of a tile index; the launcher refuses maps with more tiles), the leader
  // advances the start
  const uint64_t lt = lanemask_lt();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool on = k.tile[j] >= 0;
    const uint32_t tl = on ? (uint32_t)k.tile[j] : 0u;
    uint64_t mask = __ballot(on);
#pragma unroll
    for (int bit = 0; bit < kBinTileBits; ++bit) {
      const bool b1 = (tl >> bit) & 1u;
      const uint64_t bal = __ballot(b1);
      mask &= b1 ? bal : ~bal;
    }
    if (on) {
      const uint32_t start = mine[tl];
      const uint32_t r = (uint32_t)__popcll(mask & lt);
      list[start + r] = k.tok;
      if (r == 0) mine[tl] = start + (uint32_t)__popcll(mask);
    }
    wave_lds_sync();
  }
}

template <int CPL, int TS>
__global__ void __launch_bounds__(kDfWaves* kWave)
epipolar_dfmap_list_gather_kernel(AttnDims dm, int n_work, const float* __restrict__ xy,
                                  const uint32_t* __restrict__ tile_len,
                                  const uint32_t* __restrict__ tile_off,
                                  const uint32_t* __restrict__ list, const float* __restrict__ tg,
                                  const uint32_t* __restrict__ order, float* __restrict__ dfmap) {
  extern __shared__ __attribute__((aligned(16))) float tiles[];  // [kDfWaves][TS*TS pixels + dummy][c]
  using V = typename LaneVec<CPL>::type;
  const int R = dm.h * dm.w;
  const int tiles_x = (dm.w + TS - 1) / TS, tiles_y = (dm.h + TS - 1) / TS;
  if ((int)blockIdx.x >= n_work) return;
  const int work = (int)order[blockIdx.x];                       // longest list first
  const int tile_id = work % (tiles_x * tiles_y);
  const int src_bv = work / (tiles_x * tiles_y);                // (b, source view)
  const int tx0 = (tile_id % tiles_x) * TS, ty0 = (tile_id / tiles_x) * TS;
  const int tx1 = min(tx0 + TS, dm.w) - 1, ty1 = min(ty0 + TS, dm.h) - 1;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (wave-uniform: say so)
  const int c0 = lane * CPL;
  const bool lane_c = c0 < dm.c;
  const int cl = lane_c ? c0 : 0;
  const int tile_floats = TS * TS * dm.c + max(dm.c, kWave * CPL);   // + dummy slots
  float* tile = tiles + (size_t)wv * tile_floats;
  float* lane_base = tile + (lane_c ? 0 : TS * TS * dm.c) + c0;
  const int lane_mul = lane_c ? 1 : 0;
  for (int i = lane; i < tile_floats; i += kWave) tile[i] = 0.f;
  wave_lds_sync();

  // this wave's contiguous quarter of the tile's list
  const uint32_t len = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile_len[work]);
  const uint32_t off0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile_off[work]);
  const uint32_t per = (len + kDfWaves - 1) / kDfWaves;
  const uint32_t i_begin = min(len, (uint32_t)wv * per), i_end = min(len, ((uint32_t)wv + 1) * per);
  const uint32_t* mine = list + off0;
  for (uint32_t i0 = i_begin; i0 < i_end; i0 += kWave) {
    const uint32_t n_here = min((uint32_t)kWave, i_end - i0);
    // lanes <-> list entries: token, sample position, the four corner records relative to the tile
    const bool tok_on = (uint32_t)lane < n_here;
    const uint32_t tok = mine[tok_on ? i0 + lane : i_begin];
    const float2 p = *reinterpret_cast<const float2*>(xy + 2 * (size_t)tok);
    const Corner kq = corner_of(p.x, p.y, dm.w, dm.h);
    int off[4]; float wt[4];
#pragma unroll
    for (int cr = 0; cr < 4; ++cr) {
      const int xx = kq.x0 + (cr & 1), yy = kq.y0 + (cr >> 1);
      const bool in = tok_on && xx >= tx0 && xx <= tx1 && yy >= ty0 && yy <= ty1;
      const float wx = (cr & 1) ? kq.wx : 1.f - kq.wx, wy = (cr >> 1) ? kq.wy : 1.f - kq.wy;
      off[cr] = (in ? (yy - ty0) * TS + (xx - tx0) : TS * TS) * dm.c;
      wt[cr] = in ? wx * wy : 0.f;
    }
    // rows kDfTokGroup at a time: all loads of a group issued before the first read-modify-write
    for (uint32_t t0 = 0; t0 < n_here; t0 += kDfTokGroup) {
      float df[kDfTokGroup][CPL];
#pragma unroll
      for (int q = 0; q < kDfTokGroup; ++q) {
        const uint32_t tq = min(t0 + q, n_here - 1);            // (clamped: loaded, not used)
        const uint32_t tk = (uint32_t)__builtin_amdgcn_readlane((int)tok, (int)tq);
        load_cpl<CPL>(tg + (size_t)tk * dm.c + cl, df[q]);
      }
#pragma unroll
      for (int q = 0; q < kDfTokGroup; ++q) {
        if (t0 + q < n_here) {
          const int tl = (int)(t0 + q);
          V* dst[4]; V val[4];
#pragma unroll
          for (int cr = 0; cr < 4; ++cr) {
            dst[cr] = reinterpret_cast<V*>(lane_base + lane_bcast_i(off[cr], tl) * lane_mul);
            val[cr] = *dst[cr];
          }
#pragma unroll
          for (int cr = 0; cr < 4; ++cr) {
            const float wgt = lane_bcast(wt[cr], tl);
            float* f = reinterpret_cast<float*>(&val[cr]);
#pragma unroll
            for (int i = 0; i < CPL; ++i) f[i] = fmaf(wgt, df[q][i], f[i]);
            *dst[cr] = val[cr];
          }
        }
      }
    }
  }
  __syncthreads();
  float* out = dfmap + (size_t)src_bv * R * dm.c;
  const int tw = tx1 - tx0 + 1, th = ty1 - ty0 + 1;
  for (int i = threadIdx.x; i < tw * th * dm.c; i += kDfWaves * kWave) {
    const int pix = i / dm.c, ch = i - pix * dm.c;
    const int py = pix / tw, px = pix - py * tw;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < kDfWaves; ++k)
      acc += tiles[(size_t)k * tile_floats + (py * TS + px) * dm.c + ch];
    out[((size_t)(ty0 + py) * dm.w + tx0 + px) * dm.c + ch] = acc;
  }
}

// uint32 words of scratch the binned gather needs (layout: cnt | tile_len | tile_off | order | list)
size_t epipolar_bin_words(const AttnDims& dm) {
  const BinDims bd = make_bin_dims<4>(dm);
  const size_t n_tiles = (size_t)bd.n_src * bd.tiles;
  const size_t tokens = (size_t)dm.b * dm.v * (dm.v - 1) * dm.h * dm.w * dm.s;
  return (size_t)bd.n_src * bd.blocks_per_src * bd.tiles + 3 * n_tiles + 4 * tokens;
}

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
static size_t attn_smem(const AttnDims& dm) {
  return (size_t)kAttnWaves * wave_lds_floats(dm.s * (dm.v - 1), dm.c, 2 * dm.octaves) *
         sizeof(float);
}

static bool attn_dims_ok(const AttnDims& dm) {
  return attn_smem(dm) <= 160 * 1024 && dm.heads >= 1 && dm.heads <= kMaxHeads &&
         dm.s * (dm.v - 1) <= 128 && dm.s * (dm.v - 1) >= 1 && 2 * dm.octaves <= 8 * kUPL &&
         dm.octaves >= 1 && dm.c % 4 == 0 && dm.c >= 4 && dm.c <= 256 && dm.ld_q % 4 == 0 &&
         dm.ld_f % 4 == 0 && dm.hs_q % 4 == 0 && dm.hs_f % 4 == 0 &&
         (size_t)dm.b * dm.v * dm.h * dm.w < (size_t)1 << 30 &&
         (size_t)dm.b * dm.v * dm.h * dm.w * dm.c * 4 < ((size_t)1 << 32);
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

// channel class of the attention kernels
#define PS_BY_LPT(GO)                                            \
  do {                                                           \
    if (dm.c <= 32) GO(32); else if (dm.c <= 64) GO(64);         \
    else if (dm.c <= 128) GO(128); else GO(256);                 \
  } while (0)

int launch_epipolar_attn_forward(const AttnDims& dm, const float* fmap, const float* xy,
                                 const uint8_t* flags, const float* rd, const float* qt,
                                 const float* u, const float* e, float scale, float* fbar,
                                 float* pbar, float* abar, float* attn, hipStream_t st) {
  if (!attn_dims_ok(dm)) return PS_ERR_UNSUPPORTED;
  const size_t rays = (size_t)dm.b * dm.v * dm.h * dm.w;
  dim3 grid((unsigned)((rays + kAttnWaves - 1) / kAttnWaves)), block(kAttnWaves * kWave);
  const size_t sm = attn_smem(dm);
  const bool two_rows = dm.s * (dm.v - 1) > kWave;   // tokens per ray > 64: two token rows per lane
#define PS_GO2(L, O, N)                                                                         \
  do {                                                                                          \
    static const bool lds_ok_ = (hipFuncSetAttribute(                                           \
        (const void*)epipolar_attn_forward_kernel<L, O, N>,                                     \
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true); (void)lds_ok_;          \
    hipLaunchKernelGGL((epipolar_attn_forward_kernel<L, O, N>), grid, block, sm, st, dm, fmap,  \
                       xy, flags, rd, qt, u, e, scale, fbar, pbar, abar, attn);                 \
  } while (0)
#define PS_GO1(L, O) do { if (two_rows) PS_GO2(L, O, 2); else PS_GO2(L, O, 1); } while (0)
#define PS_GO(L) do { if (e != nullptr) PS_GO1(L, true); else PS_GO1(L, false); } while (0)
  PS_BY_LPT(PS_GO);
#undef PS_GO
#undef PS_GO1
#undef PS_GO2
  return PS_OK;
}

int launch_epipolar_attn_backward(const AttnDims& dm, const float* fmap, const float* xy,
                                  const uint8_t* flags, const float* rd, const float* qt,
                                  const float* attn, const float* fbar, const float* pbar,
                                  const float* abar, const float* dfbar, const float* dpbar,
                                  const float* dabar, float scale, float* dqt, float* du,
                                  float* de, float* ds, hipStream_t st) {
  if (!attn_dims_ok(dm)) return PS_ERR_UNSUPPORTED;
  const size_t rays = (size_t)dm.b * dm.v * dm.h * dm.w;
  dim3 grid((unsigned)((rays + kAttnWaves - 1) / kAttnWaves)), block(kAttnWaves * kWave);
  const size_t sm = attn_smem(dm);
  const bool two_rows = dm.s * (dm.v - 1) > kWave;   // tokens per ray > 64: two token rows per lane
#define PS_GO2(L, O, N)                                                                         \
  do {                                                                                          \
    static const bool lds_ok_ = (hipFuncSetAttribute(                                           \
        (const void*)epipolar_attn_backward_kernel<L, O, N>,                                    \
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024), true); (void)lds_ok_;          \
    hipLaunchKernelGGL((epipolar_attn_backward_kernel<L, O, N>), grid, block, sm, st, dm, fmap, \
                       xy, flags, rd, attn, fbar, pbar, abar, dfbar, dpbar, dabar, scale, dqt,  \
                       du, de, ds);                                                             \
  } while (0)
#define PS_GO1(L, O) do { if (two_rows) PS_GO2(L, O, 2); else PS_GO2(L, O, 1); } while (0)
#define PS_GO(L) do { if (de != nullptr || dabar != nullptr) PS_GO1(L, true); else PS_GO1(L, false); } while (0)
  PS_BY_LPT(PS_GO);
#undef PS_GO
#undef PS_GO1
#undef PS_GO2
  return PS_OK;
}

int launch_epipolar_feature_grad(const AttnDims& dm, int n_layers, const float* xy,
                                 const uint8_t* flags, const float* const* qt,
                                 const float* const* attn, const float* const* dfbar,
                                 const float* const* ds, float* dfmap, uint32_t* boxes,
                                 float* token_grad, int phases, hipStream_t st) {
  // phases (two-pass scheme only): bit 0 = build the per-tile token lists (geometry only: may run long
  // before the gradients exist, on another stream), bit 1 = token gradients + list gather
  if (!attn_dims_ok(dm)) return PS_ERR_UNSUPPORTED;
  if (boxes == nullptr || dm.w > 255 || dm.h > 255) return PS_ERR_BAD_ARG;
  FgradLayers L{};
  if (phases & 2) {
    if (n_layers < 1 || n_layers > kMaxFgradLayers) return PS_ERR_UNSUPPORTED;
    for (int l = 0; l < kMaxFgradLayers; ++l) {
      const int s = l < n_layers ? l : 0;
      L.attn[l] = attn[s]; L.ds[l] = ds[s]; L.dfbar[l] = dfbar[s]; L.qt[l] = qt[s];
      if (!L.attn[l] || !L.ds[l] || !L.dfbar[l] || !L.qt[l]) return PS_ERR_BAD_ARG;
    }
  }
  const size_t n_ro = (size_t)dm.b * dm.v * dm.h * dm.w * (dm.v - 1);
  constexpr int TS = 4;
  const int tiles = ((dm.w + TS - 1) / TS) * ((dm.h + TS - 1) / TS);
  const int n_work = dm.b * dm.v * tiles;
  dim3 g2((unsigned)((n_work + 7) / 8 * 8)), b2(kDfWaves * kWave);
  const int cpl = dm.c <= 64 ? 1 : dm.c <= 128 ? 2 : 4;
  const size_t tile_floats = (size_t)TS * TS * dm.c + (dm.c > kWave * cpl ? dm.c : kWave * cpl);
  const size_t sm2 = (size_t)kDfWaves * tile_floats * sizeof(float) + kDfChunk * sizeof(uint16_t);
  if (token_grad != nullptr || phases == 1) {     // two passes: token gradients once, then the tile gather
    dim3 g1((unsigned)((n_ro + 3) / 4)), b1(256);
    // binned gather: the per-tile token lists first (count / scan / offsets + order / fill), from the
    // geometry alone; `boxes` is the scratch: cnt | tile_len | tile_off | order | list
    const BinDims bd = make_bin_dims<TS>(dm);
    // the fill ranks lanes by kBinTileBits bits of the tile index and keeps one start per (wave, tile)
    // in LDS: both bound the tile count (today implied by w, h <= 255 with TS = 4; checked here so that
    // raising either limit cannot silently give colliding ranks)
    if (bd.tiles > (1 << kBinTileBits) ||
        (size_t)(kBinTokens / kWave) * bd.tiles * sizeof(uint32_t) > kBinFillLdsMax)
      return PS_ERR_UNSUPPORTED;
    uint32_t* cnt = boxes;
    uint32_t* tile_len = cnt + (size_t)bd.n_src * bd.blocks_per_src * bd.tiles;
    uint32_t* tile_off = tile_len + n_work;
    uint32_t* order = tile_off + n_work;
    uint32_t* list = order + n_work;
    const unsigned bin_blocks = (unsigned)(dm.b * dm.v * (dm.v - 1) * bd.blocks_per_pair);
    if (phases & 1) {
      hipLaunchKernelGGL(epipolar_bin_count_kernel<TS>, dim3(bin_blocks), dim3(kBinTokens),
                         (size_t)bd.tiles * sizeof(uint32_t), st, dm, xy, flags, cnt);
      hipLaunchKernelGGL(epipolar_bin_scan_kernel, dim3((unsigned)((n_work + 63) / 64)), dim3(256), 0, st,
                         bd.n_src, bd.tiles, bd.blocks_per_src, cnt, tile_len);
      hipLaunchKernelGGL(epipolar_bin_offsets_kernel, dim3(1), dim3(1024), 0, st, n_work, tile_len, tile_off);
      hipLaunchKernelGGL(epipolar_tile_order_kernel, dim3(1), dim3(1024), 0, st, n_work, tile_len, order);
      hipLaunchKernelGGL(epipolar_bin_fill_kernel<TS>, dim3(bin_blocks), dim3(kBinTokens),
                         (size_t)(kBinTokens / kWave) * bd.tiles * sizeof(uint32_t), st, dm, xy, flags, cnt,
                         tile_off, list);
    }
    if (!(phases & 2)) return PS_OK;
    const size_t sm3 = (size_t)kDfWaves * tile_floats * sizeof(float);
#define PS_TG(CPL)                                                                              \
  do {                                                                                          \
    if (n_layers == 1) hipLaunchKernelGGL((epipolar_token_grad_kernel<CPL, 1>), g1, b1, 0, st,  \
                                          dm, flags, L, token_grad);                            \
    else hipLaunchKernelGGL((epipolar_token_grad_kernel<CPL, 2>), g1, b1, 0, st, dm, flags, L,  \
                            token_grad);                                                        \
    hipLaunchKernelGGL((epipolar_dfmap_list_gather_kernel<CPL, TS>), g2, b2, sm3, st, dm,       \
                       n_work, xy, tile_len, tile_off, list, token_grad, order, dfmap);         \
  } while (0)
    if (dm.c <= 64) PS_TG(1); else if (dm.c <= 128) PS_TG(2); else PS_TG(4);
#undef PS_TG
    return PS_OK;
  }
  // single pass: the packed ray boxes only
  hipLaunchKernelGGL(epipolar_ray_box_kernel<TS>, dim3((unsigned)((n_ro + 255) / 256)), dim3(256), 0, st,
                     dm, xy, flags, boxes, (uint32_t*)nullptr);
#define PS_DF(CPL, NL)                                                                          \
  hipLaunchKernelGGL((epipolar_dfmap_kernel<CPL, TS, NL>), g2, b2, sm2, st, dm, n_work, xy,     \
                     boxes, L, dfmap)
#define PS_DFN(CPL)                                                                             \
  do { if (n_layers == 1) PS_DF(CPL, 1); else PS_DF(CPL, 2); } while (0)
  if (dm.c <= 64) PS_DFN(1); else if (dm.c <= 128) PS_DFN(2); else PS_DFN(4);
#undef PS_DFN
#undef PS_DF
  return PS_OK;
}

}  // namespace ps
