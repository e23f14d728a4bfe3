// Shared declarations of the gfx950 rasterizer kernels (internal; the public surface is
// include/pixelsplat_hip.h).  wave = 64 lanes everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pixelsplat_hip.h"

namespace ps {

constexpr int kWave = 64;
constexpr int kTile = 16;               // 16x16 pixel tiles (bin parity with the reference)
constexpr uint32_t kCulledKey = 0xFFFFFFFFu;
// One 64-byte LINE per (view, Gaussian): the 48-byte record + the pair's 16-byte cell window (cell_window.h) behind
// it.  Round 6: the tile kernels gather both for the same list entries -- as two arrays that was two memory lines per
// entry (a 48-byte record straddling sectors and a 16-byte window using a quarter of its own), now it is one.
constexpr int kRecFloats = 16;
constexpr int kRecWindow = 3;           // the window is the record's fourth 16-byte word
constexpr int kGradFloats = 9;          // dxy(2) dconic(3) dopacity(1) drgb(3)
constexpr int kSlotFloats = 12;         // per-(tile, entry) gradient slot: 9 used, 16-byte aligned
constexpr int kInvSlots = 4;            // Gaussians touching <= 4 tiles use slots, larger ones atomics
// records[7] of a Gaussian touching <= kInvSlots tiles: bit 31 | (rect width - 1) << 29 |
// ymin << 15 | xmin (tile units); 0 for larger rects (and for tile grids over 16383 rows)
constexpr uint32_t kSmallFlag = 0x80000000u;
// A tile list of at least kSplitMin entries is walked by the backward as TWO tasks: entries above the
// split point (a multiple of the refine batch) and entries up to it; the forward leaves every pixel's
// state at the split point in state.checkpoint (raster_tiles.hip).  0 = not split.
constexpr uint32_t kSplitMin = 256;
__host__ __device__ inline uint32_t split_point(uint32_t l_count) {
  return l_count >= kSplitMin ? ((l_count >> 1) & ~63u) : 0u;
}

// sort geometry
constexpr int kSortThreads = 1024;      // 16 waves x 4 items: the per-wave ranking chain is the
constexpr int kSortItems = 4;           // latency of the scatter kernel (16 items: 62 us per pass)
constexpr int kSortChunk = kSortThreads * kSortItems;  // 4096 keys per block
constexpr int kBinChunk = 1024;                         // sorted entries per binning block

struct Dims {
  int S, vps, V, G, H, W, gx, gy, tiles;
  size_t N;   // V * G
  size_t P;   // H * W
  int nblk;   // sort blocks per view
  int nbin;   // binning blocks per view
};

__host__ __device__ inline Dims make_dims(const PsRasterDesc& d) {
  Dims m;
  m.S = d.n_scenes; m.vps = d.views_per_scene; m.V = m.S * m.vps; m.G = d.n_gaussians;
  m.H = d.height; m.W = d.width;
  m.gx = (m.W + kTile - 1) / kTile; m.gy = (m.H + kTile - 1) / kTile; m.tiles = m.gx * m.gy;
  m.N = (size_t)m.V * m.G; m.P = (size_t)m.H * m.W;
  m.nblk = (m.G + kSortChunk - 1) / kSortChunk;
  m.nbin = (m.G + kBinChunk - 1) / kBinChunk;
  return m;
}

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

struct TempLayout {
  size_t keys_a, keys_b, vals_a, vals_b, block_hist, pass_info, bin_counts, grad2d, total;
};

inline TempLayout make_temp_layout(const PsRasterDesc& d) {
  Dims m = make_dims(d);
  TempLayout t; size_t o = 0;
  t.keys_a = o; o = align_up(o + m.N * 4);
  t.keys_b = o; o = align_up(o + m.N * 4);
  t.vals_a = o; o = align_up(o + m.N * 4);
  t.vals_b = o; o = align_up(o + m.N * 4);
  t.block_hist = o; o = align_up(o + (size_t)m.V * 512 * m.nblk * 4);      // 9-bit digits
  t.pass_info = o; o = align_up(o + ((size_t)m.V + (size_t)m.V * m.nblk) * 4);  // pass count | block maxima
  t.bin_counts = o; o = align_up(o + (size_t)m.V * m.nbin * m.tiles * 4);
  t.grad2d = 0;  // backward reuses the buffer from offset 0
  t.total = o;
  size_t bwd = align_up(m.N * kGradFloats * 4);
  if (bwd > t.total) t.total = bwd;
  return t;
}

// grad2d is the atomic accumulator of the pairs with > kInvSlots tiles (bytes [0, zeroed): ps_raster_backward
// clears the rows in use itself, ps_raster_backward_prepare clears all of it); color_grads is the compact
// per-(view, Gaussian) dL/dRGB the geometry backward hands to the SH backward (12 B rows
// instead of 3 floats out of every 36-byte grad2d row and 1 out of every 48-byte record)
struct BwdTempLayout { size_t grad2d, tile_grads, zeroed, color_grads, task_order, det_slots, rank_of, total; };
inline BwdTempLayout make_bwd_temp_layout(const PsRasterDesc& d, size_t list_capacity) {
  const Dims m = make_dims(d);
  BwdTempLayout t; size_t o = 0;
  t.grad2d = o; o = align_up(o + m.N * kGradFloats * 4);
  t.zeroed = o;   // only the atomic accumulators need clearing
  // one private slot per (view, Gaussian, tile of its <= 4-tile rect): every one is written
  // exactly once by the tile backward (values or zeros), so no memset and no position map
  t.tile_grads = o; o = align_up(o + m.N * kInvSlots * kSlotFloats * 4);
  t.color_grads = o; o = align_up(o + m.N * 3 * 4);
  // launch order of the tile backward's 2 tasks per tile, longest walk first (raster_tiles.hip)
  t.task_order = o; o = align_up(o + (size_t)m.V * m.tiles * 2 * 4);
  // PS_FLAG_DETERMINISTIC: one slot per tile-list entry for the partial gradients of the Gaussians over
  // more than kInvSlots tiles (no float atomics), and every visible pair's rank in its view's depth order
  t.det_slots = t.rank_of = o;
  if (d.flags & PS_FLAG_DETERMINISTIC) {
    t.det_slots = o; o = align_up(o + list_capacity * kSlotFloats * 4);
    t.rank_of = o; o = align_up(o + m.N * 4);
  }
  t.total = o;
  return t;
}

inline PsRasterStateLayout make_state_layout(const PsRasterDesc& d) {
  Dims m = make_dims(d);
  PsRasterStateLayout s; size_t o = 0;
  s.records = o; o = align_up(o + m.N * kRecFloats * 4);
  s.rects = o; o = align_up(o + m.N * 8);
  s.sorted_idx = o; o = align_up(o + m.N * 4);
  s.sorted_rect = o; o = align_up(o + m.N * 8);
  s.n_vis = o; o = align_up(o + (size_t)m.V * 4);
  s.final_T = o; o = align_up(o + (size_t)m.V * m.P * 4);
  s.n_contrib = o; o = align_up(o + (size_t)m.V * m.P * 4);
  s.tile_end = o; o = align_up(o + (size_t)m.V * m.tiles * 4);
  s.tile_ranges = o; o = align_up(o + (size_t)m.V * m.tiles * 8);
  s.num_rendered = o; o = align_up(o + 8);
  s.tile_order = o; o = align_up(o + (size_t)m.V * m.tiles * 4);
  s.clamp_bits = o; o = align_up(o + m.N);
  s.checkpoint = o; o = align_up(o + (size_t)m.V * m.tiles * kTile * kTile * 16);
  s.cell_windows = s.records + kRecWindow * 16;   // interleaved with the records: stride kRecFloats * 4 bytes
  s.total = o;
  return s;
}

// ---- launchers (one per translation unit) -------------------------------------------
void launch_preprocess_forward(const PsRasterDesc& d, const float* means, const float* cov,
                               const float* sh, const float* colors, const float* opacity,
                               const float* view_params, float* records, uint32_t* keys,
                               uint2* rects, int32_t* radii, uint8_t* clamp_bits, uint4* cell_windows,
                               bool geometry, bool sh_colors, hipStream_t st);

void launch_sort(const PsRasterDesc& d, uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a,
                 uint32_t* vals_b, uint32_t* block_hist, uint32_t* pass_info, uint32_t* sorted_idx,
                 const uint2* rects, uint2* sorted_rect, uint32_t* n_vis, hipStream_t st);

void launch_bin_count(const PsRasterDesc& d, const uint2* sorted_rect, const uint32_t* n_vis,
                      uint32_t* counts, uint32_t* tile_ranges, uint32_t* num_rendered,
                      uint32_t* tile_order, uint32_t* tile_end, hipStream_t st);
void launch_bin_write(const PsRasterDesc& d, const uint2* sorted_rect, const uint32_t* sorted_idx,
                      const uint32_t* n_vis, uint32_t* counts, const uint32_t* tile_ranges,
                      uint32_t* num_rendered, uint32_t* point_list,
                      uint32_t capacity, hipStream_t st);

void launch_tiles_forward(const PsRasterDesc& d, const float* records,
                          const uint32_t* tile_order, const uint32_t* tile_ranges,
                          const uint32_t* point_list,
                          uint32_t capacity, const float* view_params, float* out_color,
                          float* final_T, uint32_t* n_contrib, float4* checkpoint,
                          uint32_t* tile_end, hipStream_t st);

// the tile forward on 4x4-pixel cells, a wave64 = four 16-lane rows (raster_cells.hip)
void launch_tiles_forward_rows(const PsRasterDesc& d, const float* records, const uint4* cell_windows,
                               const uint32_t* tile_order, const uint32_t* tile_ranges,
                               const uint32_t* point_list, uint32_t capacity, const float* view_params,
                               float* out_color, float* final_T, uint32_t* n_contrib, float4* checkpoint,
                               uint32_t* tile_end, hipStream_t st);

void launch_backward_task_order(const PsRasterDesc& d, const uint32_t* tile_ranges,
                                const uint32_t* tile_end, uint32_t capacity, uint32_t* task_order,
                                hipStream_t st);
void launch_tiles_backward(const PsRasterDesc& d, const float* records, const uint4* cell_windows,
                           const uint32_t* task_order, const uint32_t* tile_ranges,
                           const uint32_t* point_list,
                           uint32_t capacity, const float* view_params, const float* final_T,
                           const uint32_t* n_contrib, const float4* checkpoint,
                           const uint32_t* tile_end, const float* dL_dcolor, float* grad2d,
                           float* tile_grads, float* det_slots, hipStream_t st);
void launch_deterministic_clear(float* det_slots, size_t entries, hipStream_t st);
// PS_FLAG_DETERMINISTIC: ranks, then the fixed-order sum of the per-entry slots into grad2d
void launch_deterministic_reduce(const PsRasterDesc& d, const int32_t* radii, const uint2* rects,
                                 const uint32_t* sorted_idx, const uint32_t* n_vis,
                                 const uint32_t* tile_ranges, const uint32_t* point_list,
                                 uint32_t capacity, const float* det_slots, uint32_t* rank_of,
                                 float* grad2d, hipStream_t st);

void launch_clear_atomic_rows(const PsRasterDesc& d, const uint2* sorted_rect, const uint32_t* sorted_idx,
                              const uint32_t* n_vis, float* grad2d, hipStream_t st);
void launch_preprocess_backward(const PsRasterDesc& d, const float* means, const float* cov,
                                const float* sh, const float* view_params, const float* records,
                                const int32_t* radii, const uint2* rects,
                                const float* tile_grads, const uint8_t* clamp_bits,
                                float* color_grads,
                                float* grad2d, float* dL_dmeans,
                                float* dL_dcov, float* dL_dsh, float* dL_dcolors,
                                float* dL_dopacity, float* dL_dmeans2D, hipStream_t st);

void launch_camera_setup(int n_views, const float* extrinsics, const float* intrinsics,
                         const float* near, const float* far, const float* bg,
                         int scale_invariant, float* view_params, hipStream_t st);

void launch_epipolar_geometry(int b, int v, int h, int w, int s, const float* c2w,
                              const float* w2c, const float* kmat, const float* kinv,
                              const float* near, const float* far, float* origins,
                              float* directions, float* seg, uint8_t* flags, float* xy_sample,
                              float* depth, float* rel_disp, hipStream_t st);

// ld_*: row strides (floats) of the per-ray arrays qt / u / e (and dqt / du / de) and
// fbar / pbar / abar (and their gradients), so that they may be column blocks of one matrix;
// inside a row heads are contiguous ([h][c], [h][P], [h][v-1]).
struct AttnDims {
  int b, v, h, w, s, c, heads, octaves, ld_q, ld_u, ld_e, ld_f, ld_p, ld_a;
  int hs_q, hs_u, hs_e, hs_f, hs_p, hs_a;   // head strides inside a row (default c, P, v-1)
  int pad_in, pad_out;                      // zero-filled floats behind a head's last block (bwd / fwd)
};
int launch_epipolar_gather(const AttnDims& dm, const float* fmap, const float* xy,
                           const uint8_t* flags, float* out, hipStream_t st);
int launch_epipolar_attn_forward(const AttnDims& dm, const float* fmap, const float* xy,
                                 const uint8_t* flags, const float* rd, const float* qt,
                                 const float* u, const float* e, float scale, float* fbar,
                                 float* pbar, float* abar, float* attn, hipStream_t st);
int launch_epipolar_attn_backward(const AttnDims& dm, const float* fmap, const float* xy,
                                  const uint8_t* flags, const float* rd, const float* qt,
                                  const float* attn, const float* fbar, const float* pbar,
                                  const float* abar, const float* dfbar, const float* dpbar,
                                  const float* dabar, float scale, float* dqt, float* du,
                                  float* de, float* ds, hipStream_t st);
size_t epipolar_bin_words(const AttnDims& dm);   // uint32 words of scratch of the binned gather
int launch_epipolar_feature_grad(const AttnDims& dm, int n_layers, const float* xy,
                                 const uint8_t* flags, const float* const* qt,
                                 const float* const* attn, const float* const* dfbar,
                                 const float* const* ds, float* dfmap, uint32_t* boxes,
                                 float* token_grad, int phases, hipStream_t st);
int launch_adapter_views(int n_views, int sh_degree, int img_h, int img_w, const float* extrinsics,
                         const float* intrinsics, const double* conj, float* views,
                         hipStream_t st);
int launch_adapter_forward(int n_views, int rp, int spp, int sh_degree, float smin, float smax,
                           float eps, const float* views, const float* coords,
                           const float* depths, const float* raw, float* means, float* cov,
                           float* harmonics, const int* head, hipStream_t st);
int launch_adapter_backward(int n_views, int rp, int spp, int sh_degree, float smin, float smax,
                            float eps, const float* views, const float* coords,
                            const float* depths, const float* raw, const float* d_means,
                            const float* d_cov, const float* d_harmonics, float* d_raw,
                            float* d_depths, float* d_coords, const int* head, hipStream_t st);
int launch_depth_sampler_forward(const PsDepthSamplerDesc& d, const float* projected,
                                 const float* near, const float* far, const float* uniforms,
                                 float* depth, float* opacity, int32_t* index, hipStream_t st);
int launch_depth_sampler_backward(const PsDepthSamplerDesc& d, const float* projected,
                                  const float* near, const float* far, const int32_t* index,
                                  const float* d_depth, const float* d_opacity,
                                  float* d_projected, hipStream_t st);
size_t image_mse_workspace_bytes(int n_images, int elems);
int launch_image_mse(int n_images, int elems, const float* pred, const float* target,
                     float grad_scale, float* grad, float* sse, float* sse_clipped,
                     void* workspace, hipStream_t st);
size_t depth_smoothness_workspace_bytes(const PsDepthLossDesc& d);
int launch_depth_smoothness_forward(const PsDepthLossDesc& d, const float* depth,
                                    const float* near, const float* far, const float* image,
                                    float* loss, void* workspace, hipStream_t st);
int launch_depth_smoothness_backward(const PsDepthLossDesc& d, const float* depth,
                                     const float* near, const float* far, const float* image,
                                     const float* d_loss, float* d_depth, hipStream_t st);
size_t fold_scratch_floats(const PsFoldDesc& d);
int launch_fold_forward(const PsFoldDesc& d, const float* w_q, const float* w_kv,
                        const float* w_out, const float* b_out, const float* depth_w,
                        const float* depth_b, const float* view_emb, float* w_in, float* w_o_t,
                        float* bias, float* scratch, hipStream_t st);
int launch_fold_backward(const PsFoldDesc& d, const float* w_q, const float* w_kv,
                         const float* w_out, const float* b_out, const float* depth_w,
                         const float* depth_b, const float* view_emb, const float* scratch,
                         const float* d_w_in, const float* d_w_o_t, const float* d_bias,
                         float* back_scratch, float* g_w_q, float* g_w_kv, float* g_w_out,
                         float* g_b_out, float* g_depth_w, float* g_depth_b, float* g_view_emb,
                         hipStream_t st);
size_t layer_norm_workspace_floats(int rows, int dim);
int launch_layer_norm_forward(int rows, int dim, float eps, const float* x, const float* gamma,
                              const float* beta, float* y, float* mean, float* rstd,
                              hipStream_t st);
int launch_layer_norm_backward(int rows, int dim, const float* x, const float* gamma,
                               const float* mean, const float* rstd, const float* dy,
                               const float* d_residual, float* dx, float* d_gamma, float* d_beta,
                               float* workspace, hipStream_t st);
void launch_camera_inverse(int n, const float* c2w, const float* k, float* w2c, float* k_inv,
                           hipStream_t st);
size_t gemm_tn_workspace_bytes(int M, int N, int K);
// colsum_a (may be null): sum_k A[k][m], M floats
int launch_gemm_tn(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                   float* C, float* colsum_a, float* workspace, hipStream_t st);

// ---- device helpers -------------------------------------------------------------------
// (block-in-view, view) of a block of a (blocks-per-view, V) grid whose blocks past a view's visible
// prefix have nothing to do (60 % of them at BASELINE configs[1]).  Blocks are dispatched x-fastest;
// enumerated view-minor -- b = linear / V, v = linear % V -- all working blocks come first and the empty
// ones trail the grid.  Measured: tile_bins 0.145 -> 0.140 ms, the sort unchanged
// (profiles/r4_grid_order_ab.txt): blocks that exit before their first memory operation are cheap wherever
// they sit, unlike the tile backward's nearly-empty one-wave tasks (profiles/r4_backward_split_ab.txt).
__device__ __forceinline__ void view_minor_block(int& b, int& v) {
  const int lin = (int)(blockIdx.y * gridDim.x + blockIdx.x);
  b = lin / (int)gridDim.y; v = lin % (int)gridDim.y;
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (kWave - 1)); }

__device__ __forceinline__ uint64_t lanemask_lt() {
  return (1ull << lane_id()) - 1ull;
}

// LDS hand-off between lanes of ONE wave (no s_barrier: the four waves of a block are
// independent here).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// global -> LDS copy of nflt floats by one wave.  All 16-byte loads are issued before the
// first LDS store: a load/store-per-iteration loop serialises one memory latency per
// iteration (this was 10-19 round trips per wave in the SH colour kernels).
template <int MAXV4>
__device__ __forceinline__ void stage_slab(const float* __restrict__ src, float* slab, int nflt,
                                           int lane) {
  if ((reinterpret_cast<size_t>(src) & 15) == 0 && nflt <= MAXV4 * kWave * 4) {
    const float4* s4 = reinterpret_cast<const float4*>(src);
    const int n4 = nflt >> 2;
    float4 r[MAXV4];
#pragma unroll
    for (int u = 0; u < MAXV4; ++u) {
      const int i = lane + u * kWave;
      r[u] = i < n4 ? s4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < MAXV4; ++u) {
      const int i = lane + u * kWave;
      if (i < n4) *reinterpret_cast<float4*>(slab + i * 4) = r[u];
    }
    for (int e = n4 * 4 + lane; e < nflt; e += kWave) slab[e] = src[e];
  } else {
    for (int i = lane; i < nflt; i += kWave) slab[i] = src[i];
  }
}

// rect packed as uint2: .x = xmin | ymin << 16, .y = xmax | ymax << 16 (in tiles)
__device__ __forceinline__ bool rect_covers(uint2 r, uint32_t tx, uint32_t ty) {
  const uint32_t xmin = r.x & 0xFFFFu, ymin = r.x >> 16, xmax = r.y & 0xFFFFu, ymax = r.y >> 16;
  return (xmin <= tx) & (tx < xmax) & (ymin <= ty) & (ty < ymax);
}

// Same test in 4 integer ops (fields are < 2^15, so a borrow out of the low half only
// happens when the low half already failed): precompute T = tx | ty << 16, T1 = T + 0x00010001.
__device__ __forceinline__ bool rect_covers_packed(uint2 r, uint32_t T, uint32_t T1) {
  return (((T - r.x) | (r.y - T1)) & 0x80008000u) == 0u;
}

}  // namespace ps
