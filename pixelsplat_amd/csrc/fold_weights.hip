// Folded weights of one epipolar cross-attention layer: with
//   A = [I_c; W_d^T; E; 0]   (rows: identity, depth-encoding weights, view embeddings, padding)
//   K_h = A W_k,h^T,  V_h = A W_v,h^T                                        [Lh x dh]
//   w_in[h]  = K_h W_q,h          [Lh x d]       (attention.py:45-47 to_q / to_kv folded)
//   w_o_t[h] = V_h W_out,h        [Lh x d_out]   (to_out folded), W_out,h[j][o] = w_out[o][h dh + j]
//   bias     = b_out + w_out (W_v b_d)
// and the gradients of all seven parameter tensors.  These are a few MFLOP; as ~40 tiny
// library launches per layer and direction they cost ~0.4 ms of launch-paced GPU idle per
// step, here they are two launches forward and two backward (one thread per output element,
// loops over the short reduction dimension).
#include "raster_common.h"

namespace ps {

namespace {

struct FoldDims {
  int heads, dh, c, d, d_out, P, ov, lh;   // inner = heads * dh
};

struct FoldParams {
  const float* w_q;      // [inner][d]
  const float* w_kv;     // [2 inner][c]
  const float* w_out;    // [d_out][inner]
  const float* b_out;    // [d_out] or nullptr
  const float* depth_w;  // [c][P]
  const float* depth_b;  // [c]
  const float* view_emb; // [ov][c] or nullptr
};

__device__ __forceinline__ float a_entry(const FoldDims& m, const FoldParams& p, int r, int i) {
  if (r < m.c) return r == i ? 1.f : 0.f;
  if (r < m.c + m.P) return p.depth_w[i * m.P + (r - m.c)];
  if (r < m.c + m.P + m.ov) return p.view_emb[(r - m.c - m.P) * m.c + i];
  return 0.f;
}

// sum over the four lanes that share one output element
__device__ __forceinline__ float quad_sum(float v) {
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  return v;
}

}  // namespace

// K, V [heads][lh][dh] and vb [inner] = W_v b_d.  Identity rows of A are plain copies; the
// 2 octaves + other_views dense rows and vb are 128-term dot products, four lanes each.
__global__ __launch_bounds__(256) void fold_kv_kernel(FoldDims m, FoldParams p,
                                                      float* __restrict__ K, float* __restrict__ V,
                                                      float* __restrict__ vb) {
  const int inner = m.heads * m.dh;
  const int n_copy = m.heads * m.lh * m.dh;          // one thread per element: copy or zero
  const int dense = m.P + m.ov;
  const int n_dense = m.heads * dense * m.dh;        // four threads per element
  int t = blockIdx.x * 256 + threadIdx.x;
  if (t < n_copy) {
    const int j = t % m.dh, r = (t / m.dh) % m.lh, h = t / (m.dh * m.lh);
    if (r < m.c) {
      K[t] = p.w_kv[(size_t)(h * m.dh + j) * m.c + r];
      V[t] = p.w_kv[(size_t)(inner + h * m.dh + j) * m.c + r];
    } else if (r >= m.c + dense) {
      K[t] = 0.f; V[t] = 0.f;
    }
    return;
  }
  t -= n_copy;
  const int sub = t & 3;
  t >>= 2;
  if (t < n_dense) {
    const int j = t % m.dh, rr = (t / m.dh) % dense, h = t / (m.dh * dense);
    const float* wk = p.w_kv + (size_t)(h * m.dh + j) * m.c;
    const float* wv = p.w_kv + (size_t)(inner + h * m.dh + j) * m.c;
    const float* arow = rr < m.P ? p.depth_w + rr : p.view_emb + (size_t)(rr - m.P) * m.c;
    const int astride = rr < m.P ? m.P : 1;
    float k = 0.f, v = 0.f;
#pragma unroll 8
    for (int i = sub; i < m.c; i += 4) {
      const float a = arow[(size_t)i * astride];
      k = fmaf(a, wk[i], k); v = fmaf(a, wv[i], v);
    }
    k += __shfl_xor(k, 1); k += __shfl_xor(k, 2);
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2);
    if (sub == 0) {
      const size_t o = ((size_t)h * m.lh + m.c + rr) * m.dh + j;
      K[o] = k; V[o] = v;
    }
    return;
  }
  t -= n_dense;
  if (t < inner) {
    const float* wv = p.w_kv + (size_t)(inner + t) * m.c;
    float s = 0.f;
#pragma unroll 8
    for (int i = sub; i < m.c; i += 4) s = fmaf(wv[i], p.depth_b[i], s);
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
    if (sub == 0) vb[t] = s;
  }
}

// w_in [heads lh][d], w_o_t [heads lh][d_out], bias [d_out]
__global__ __launch_bounds__(256) void fold_out_kernel(FoldDims m, FoldParams p,
                                                       const float* __restrict__ K,
                                                       const float* __restrict__ V,
                                                       const float* __restrict__ vb,
                                                       float* __restrict__ w_in,
                                                       float* __restrict__ w_o_t,
                                                       float* __restrict__ bias) {
  const int inner = m.heads * m.dh;
  const int n_in = m.heads * m.lh * m.d, n_o = m.heads * m.lh * m.d_out;
  int t = blockIdx.x * 256 + threadIdx.x;
  if (t < n_in) {
    const int k = t % m.d, hr = t / m.d, h = hr / m.lh;
    const float* kk = K + (size_t)hr * m.dh;
    float s = 0.f;
#pragma unroll 8
    for (int j = 0; j < m.dh; ++j) s = fmaf(kk[j], p.w_q[(size_t)(h * m.dh + j) * m.d + k], s);
    w_in[t] = s;
    return;
  }
  t -= n_in;
  if (t < n_o) {
    const int o = t % m.d_out, hr = t / m.d_out, h = hr / m.lh;
    const float* vv = V + (size_t)hr * m.dh;
    const float* wo = p.w_out + (size_t)o * inner + h * m.dh;
    float s = 0.f;
#pragma unroll 8
    for (int j = 0; j < m.dh; ++j) s = fmaf(vv[j], wo[j], s);
    w_o_t[t] = s;
    return;
  }
  t -= n_o;   // n_in + n_o is a multiple of 4: the 4-lane groups below stay aligned
  const int sub = t & 3;
  t >>= 2;
  if (t < m.d_out) {
    float s = 0.f;
#pragma unroll 8
    for (int ii = sub; ii < inner; ii += 4) s = fmaf(p.w_out[(size_t)t * inner + ii], vb[ii], s);
    s = quad_sum(s);
    if (sub == 0) bias[t] = s + (p.b_out ? p.b_out[t] : 0.f);
  }
}

// dK, dV [heads][lh][dh] and dvb [inner]
__global__ __launch_bounds__(256) void fold_back_kv_kernel(FoldDims m, FoldParams p,
                                                           const float* __restrict__ d_w_in,
                                                           const float* __restrict__ d_w_o_t,
                                                           const float* __restrict__ d_bias,
                                                           float* __restrict__ dK,
                                                           float* __restrict__ dV,
                                                           float* __restrict__ dvb) {
  const int inner = m.heads * m.dh;
  const int n_kv = m.heads * m.lh * m.dh;
  const int T = blockIdx.x * 256 + threadIdx.x, sub = T & 3;   // four lanes per element
  int t = T >> 2;
  if (t < n_kv) {
    const int j = t % m.dh, hr = t / m.dh, h = hr / m.lh;
    const float* gi = d_w_in + (size_t)hr * m.d;
    const float* wq = p.w_q + (size_t)(h * m.dh + j) * m.d;
    float a = 0.f;
#pragma unroll 8
    for (int k = sub; k < m.d; k += 4) a = fmaf(gi[k], wq[k], a);
    const float* go = d_w_o_t + (size_t)hr * m.d_out;
    float b = 0.f;
#pragma unroll 8
    for (int o = sub; o < m.d_out; o += 4)
      b = fmaf(go[o], p.w_out[(size_t)o * inner + h * m.dh + j], b);
    a = quad_sum(a); b = quad_sum(b);
    if (sub == 0) { dK[t] = a; dV[t] = b; }
    return;
  }
  t -= n_kv;
  if (t < inner) {
    float s = 0.f;
#pragma unroll 8
    for (int o = sub; o < m.d_out; o += 4) s = fmaf(p.w_out[(size_t)o * inner + t], d_bias[o], s);
    s = quad_sum(s);
    if (sub == 0) dvb[t] = s;
  }
}

struct FoldGrads {
  float* w_q; float* w_kv; float* w_out; float* b_out; float* depth_w; float* depth_b;
  float* view_emb;   // b_out / view_emb may be nullptr
};

// Four lanes share one output element (the reduction index is split 4 ways and summed with two
// quad shuffles): with one thread per output this kernel was a chain of ~150 dependent global
// round trips on ~1 block per CU (240 us; 64 us unrolled).
__global__ __launch_bounds__(256) void fold_back_params_kernel(
    FoldDims m, FoldParams p, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ vb, const float* __restrict__ dK, const float* __restrict__ dV,
    const float* __restrict__ dvb, const float* __restrict__ d_w_in,
    const float* __restrict__ d_w_o_t, const float* __restrict__ d_bias, FoldGrads g) {
  const int inner = m.heads * m.dh;
  const int T = blockIdx.x * 256 + threadIdx.x, sub = T & 3;
  int t = T >> 2;
  // d w_q[(h dh + j)][k] = sum_r K[h][r][j] d_w_in[h lh + r][k]
  if (t < inner * m.d) {
    const int k = t % m.d, hj = t / m.d, h = hj / m.dh, j = hj % m.dh;
    float s = 0.f;
#pragma unroll 8
    for (int r = sub; r < m.lh; r += 4)
      s = fmaf(K[((size_t)h * m.lh + r) * m.dh + j], d_w_in[((size_t)h * m.lh + r) * m.d + k], s);
    s = quad_sum(s);
    if (sub == 0) g.w_q[t] = s;
    return;
  }
  t -= inner * m.d;
  // d w_out[o][h dh + j] = sum_r V[h][r][j] d_w_o_t[h lh + r][o] + d_bias[o] vb[h dh + j]
  if (t < m.d_out * inner) {
    const int hj = t % inner, o = t / inner, h = hj / m.dh, j = hj % m.dh;
    float s = sub == 0 ? d_bias[o] * vb[hj] : 0.f;
#pragma unroll 8
    for (int r = sub; r < m.lh; r += 4)
      s = fmaf(V[((size_t)h * m.lh + r) * m.dh + j], d_w_o_t[((size_t)h * m.lh + r) * m.d_out + o], s);
    s = quad_sum(s);
    if (sub == 0) g.w_out[t] = s;
    return;
  }
  t -= m.d_out * inner;
  // d w_kv: K half rows [0, inner), V half rows [inner, 2 inner); column i:
  //   sum_r d{K,V}[h][r][j] A[r][i] = d{K,V}[h][i][j] + sum_{r >= c} ... (+ dvb[hj] b_d[i] for V)
  if (t < 2 * inner * m.c) {
    const int i = t % m.c, row = t / m.c;
    const bool is_v = row >= inner;
    const int hj = is_v ? row - inner : row, h = hj / m.dh, j = hj % m.dh;
    const float* src = is_v ? dV : dK;
    float s = 0.f;
    if (sub == 0) {
      s = src[((size_t)h * m.lh + i) * m.dh + j];
      if (is_v) s = fmaf(dvb[hj], p.depth_b[i], s);
    }
#pragma unroll 4
    for (int r = m.c + sub; r < m.c + m.P + m.ov; r += 4)
      s = fmaf(src[((size_t)h * m.lh + r) * m.dh + j], a_entry(m, p, r, i), s);
    s = quad_sum(s);
    if (sub == 0) g.w_kv[t] = s;
    return;
  }
  t -= 2 * inner * m.c;
  // rows c .. c+P+ov of dA = sum_h dK_h W_k,h + dV_h W_v,h  ->  depth_w^T / view_emb
  if (t < (m.P + m.ov) * m.c) {
    const int i = t % m.c, rr = t / m.c, r = m.c + rr;
    float s = 0.f;
#pragma unroll 4
    for (int hj = sub; hj < inner; hj += 4) {
      const int h = hj / m.dh, j = hj - h * m.dh;
      const size_t e = ((size_t)h * m.lh + r) * m.dh + j;
      s = fmaf(dK[e], p.w_kv[(size_t)hj * m.c + i], s);
      s = fmaf(dV[e], p.w_kv[(size_t)(inner + hj) * m.c + i], s);
    }
    s = quad_sum(s);
    if (sub == 0) {
      if (rr < m.P) g.depth_w[i * m.P + rr] = s;
      else if (g.view_emb) g.view_emb[(rr - m.P) * m.c + i] = s;
    }
    return;
  }
  t -= (m.P + m.ov) * m.c;
  if (t < m.c) {   // d depth_b = W_v^T dvb
    float s = 0.f;
#pragma unroll 8
    for (int ii = sub; ii < inner; ii += 4)
      s = fmaf(p.w_kv[(size_t)(inner + ii) * m.c + t], dvb[ii], s);
    s = quad_sum(s);
    if (sub == 0) g.depth_b[t] = s;
    return;
  }
  t -= m.c;
  if (t < m.d_out && g.b_out && sub == 0) g.b_out[t] = d_bias[t];
}

namespace {
FoldDims make_dims(const PsFoldDesc& d) {
  return {d.heads, d.head_dim, d.kv_dim, d.q_dim, d.out_dim, 2 * d.octaves, d.other_views,
          (d.kv_dim + 2 * d.octaves + d.other_views + 3) & ~3};
}
unsigned blocks_for(long n) { return (unsigned)((n + 255) / 256); }
}  // namespace

size_t fold_scratch_floats(const PsFoldDesc& d) {
  const FoldDims m = make_dims(d);
  return (size_t)2 * m.heads * m.lh * m.dh + (size_t)m.heads * m.dh;   // K | V | vb
}

int launch_fold_forward(const PsFoldDesc& d, const float* w_q, const float* w_kv,
                        const float* w_out, const float* b_out, const float* depth_w,
                        const float* depth_b, const float* view_emb, float* w_in, float* w_o_t,
                        float* bias, float* scratch, hipStream_t st) {
  const FoldDims m = make_dims(d);
  const FoldParams p{w_q, w_kv, w_out, b_out, depth_w, depth_b, view_emb};
  const long n_kv = (long)m.heads * m.lh * m.dh, inner = (long)m.heads * m.dh;
  float* K = scratch; float* V = K + n_kv; float* vb = V + n_kv;
  // n_kv is a multiple of 4 (lh is), so the 4-lane groups of the dense part stay aligned
  fold_kv_kernel<<<blocks_for(n_kv + 4 * ((long)m.heads * (m.P + m.ov) * m.dh + inner)), 256, 0, st>>>(
      m, p, K, V, vb);
  const long n2 = (long)m.heads * m.lh * (m.d + m.d_out) + 4L * m.d_out;
  fold_out_kernel<<<blocks_for(n2), 256, 0, st>>>(m, p, K, V, vb, w_in, w_o_t, bias);
  return PS_OK;
}

int launch_fold_backward(const PsFoldDesc& d, const float* w_q, const float* w_kv,
                         const float* w_out, const float* b_out, const float* depth_w,
                         const float* depth_b, const float* view_emb, const float* scratch,
                         const float* d_w_in, const float* d_w_o_t, const float* d_bias,
                         float* back_scratch, float* g_w_q, float* g_w_kv, float* g_w_out,
                         float* g_b_out, float* g_depth_w, float* g_depth_b, float* g_view_emb,
                         hipStream_t st) {
  const FoldDims m = make_dims(d);
  const FoldParams p{w_q, w_kv, w_out, b_out, depth_w, depth_b, view_emb};
  const long n_kv = (long)m.heads * m.lh * m.dh, inner = (long)m.heads * m.dh;
  const float* K = scratch; const float* V = K + n_kv; const float* vb = V + n_kv;
  float* dK = back_scratch; float* dV = dK + n_kv; float* dvb = dV + n_kv;
  fold_back_kv_kernel<<<blocks_for(4 * (n_kv + inner)), 256, 0, st>>>(m, p, d_w_in, d_w_o_t, d_bias, dK,
                                                                 dV, dvb);
  const long n2 = inner * m.d + (long)m.d_out * inner + 2 * inner * m.c +
                  (long)(m.P + m.ov) * m.c + m.c + m.d_out;
  const FoldGrads g{g_w_q, g_w_kv, g_w_out, g_b_out, g_depth_w, g_depth_b, g_view_emb};
  fold_back_params_kernel<<<blocks_for(4 * n2), 256, 0, st>>>(m, p, K, V, vb, dK, dV, dvb, d_w_in,
                                                         d_w_o_t, d_bias, g);
  return PS_OK;
}

}  // namespace ps
