/*
 * pixelsplat_hip.h -- C ABI of libpixelsplat_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary of pixelSplat's hot path.  Plain pointers and sizes only: every
 * buffer is DEVICE memory owned by the caller (PyTorch's caching allocator in practice),
 * `stream` is a hipStream_t passed as void*, the library keeps no state between calls,
 * never throws / aborts / synchronises the device, and returns 0 or a negative PsStatus.
 *
 * (B) rasterizer.  Replaces what the reference reaches through the third-party module
 *     `diff_gaussian_rasterization` (GaussianRasterizer.forward / autograd backward):
 *       /root/reference/src/model/decoder/cuda_splatting.py:5-8     import
 *       /root/reference/src/model/decoder/cuda_splatting.py:99-124  per-view call (render_cuda)
 *       /root/reference/src/model/decoder/cuda_splatting.py:192-217 per-view call (orthographic)
 *     and, in its batched form (n_views > 1, views_per_scene > 1, PS_SH_G3K / PS_COV_33,
 *     per-view scene_scale), the whole body of
 *       /root/reference/src/model/decoder/cuda_splatting.py:47-127  render_cuda
 *       /root/reference/src/model/decoder/decoder_splatting_cuda.py:46-57 (no v-fold repeat)
 *
 * (A) epipolar sampler + attention: ps_epipolar_* (see below; added as that path lands).
 */
#ifndef PIXELSPLAT_HIP_H
#define PIXELSPLAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum PsStatus {
  PS_OK = 0,
  PS_ERR_BAD_ARG = -1,     /* null pointer / inconsistent descriptor            */
  PS_ERR_WORKSPACE = -2,   /* state/temp buffer smaller than ps_raster_*_bytes  */
  PS_ERR_LAUNCH = -3,      /* hipGetLastError() != hipSuccess after a launch    */
  PS_ERR_UNSUPPORTED = -4, /* e.g. sh_degree > 4, image wider than 32767 tiles  */
  PS_ERR_CAPACITY = -5     /* D (sum of tile-list lengths) exceeds the point-list capacity */
} PsStatus;

/* layout selectors */
#define PS_SH_GK3 0  /* sh[G][K][3]  -- the rasterizer boundary (cuda_splatting.py:75)        */
#define PS_SH_G3K 1  /* sh[G][3][K]  -- reference `Gaussians.harmonics` (src/model/types.py:11) */
#define PS_COV_6 0   /* cov[G][6] = xx,xy,xz,yy,yz,zz  (cuda_splatting.py:115,123)            */
#define PS_COV_33 1  /* cov[G][3][3] -- reference `Gaussians.covariances`; upper triangle read,
                        gradient written to the upper triangle (lower = 0), as autograd of
                        cov[:, triu_row, triu_col] would                                        */

/* per-view parameter block: PS_VIEW_STRIDE floats, device memory, one per view.
 * Matrices are the TRANSPOSED (row-vector) 4x4s of cuda_splatting.py:84-87, i.e. element
 * [4*c + r] is row r / column c of the column-vector matrix. */
#define PS_VIEW_STRIDE 48
#define PS_VIEW_VIEWMATRIX 0   /* 16: world -> camera                                */
#define PS_VIEW_PROJMATRIX 16  /* 16: world -> clip (view @ projection)              */
#define PS_VIEW_CAMPOS 32      /*  3: camera position (after scene_scale)            */
#define PS_VIEW_TANFOVX 35
#define PS_VIEW_TANFOVY 36
#define PS_VIEW_BG 37          /*  3: background colour                              */
#define PS_VIEW_SCALE 40       /*  1: scene_scale s: means*s, cov*s^2 on load (the
                                      scale-invariant renorm of cuda_splatting.py:64-71;
                                      1.0 for the plain per-view boundary)            */

/* PsRasterDesc.flags: the caller has already zeroed the backward temp buffer with
 * ps_raster_backward_prepare, so ps_raster_backward skips its own clearing.  Since round 3 that
 * clearing is a 30 us kernel over the accumulator rows actually in use (pairs covering more than
 * four tiles) instead of a memset of the whole buffer: neither the flag nor the prepare call buys
 * anything any more; both are kept so that existing hosts keep working. */
#define PS_FLAG_BWD_TEMP_ZEROED 1
/* ps_raster_forward_plan leaves the SH -> RGB evaluation to a later ps_raster_forward_colors
 * (any time before ps_raster_forward_tiles): a host that reads D back after the plan can queue
 * it behind that copy, so the GPU evaluates colours while the host waits and allocates. */
#define PS_FLAG_DEFER_SH_COLORS 2
/* ps_raster_backward: deterministic gradient accumulation.  Gaussians whose rect covers more than
 * four tiles normally sum their per-tile partial gradients through float atomics (summation order =
 * the order in which the tiles' waves retire: run-to-run differences in the last bits).  With this flag
 * every (tile, Gaussian) partial is stored in a slot of its own -- one per tile-list entry -- and summed by
 * a second kernel in a fixed order (tiles of the rect, row-major): two runs are bitwise equal.  Costs
 * 48 bytes per list entry + 4 bytes per (view, Gaussian) of backward scratch
 * (ps_raster_backward_temp_bytes accounts for it) and a cleared slot array per call: a mode for parity /
 * reproducibility runs ("race detection"), not the benchmarked path. */
#define PS_FLAG_DETERMINISTIC 4

typedef struct PsRasterDesc {
  int32_t n_scenes;        /* S: independent Gaussian sets                           */
  int32_t views_per_scene; /* views rendered from each set; view v uses scene v / views_per_scene */
  int32_t n_gaussians;     /* G per scene                                            */
  int32_t height, width;
  int32_t sh_degree;       /* active degree 0..4                                     */
  int32_t sh_coeffs;       /* K stored per channel (>= (deg+1)^2); 0 => colors_precomp */
  int32_t sh_layout;       /* PS_SH_*                                                */
  int32_t cov_layout;      /* PS_COV_*                                               */
  int32_t flags;           /* PS_FLAG_* bits, 0 by default                           */
  /* algorithm constants; ps_raster_default_desc fills them */
  float near_cull;    /* 0.2   */
  float guard;        /* 1.3   */
  float lowpass;      /* 0.3   */
  float w_eps;        /* 1e-7  */
  float lambda_floor; /* 0.1   */
  float alpha_max;    /* 0.99  */
  float alpha_min;    /* 1/255 */
  float t_min;        /* 1e-4  */
  float det2_eps;     /* 1e-7  */
} PsRasterDesc;

/* byte offsets of the arrays inside the `state` buffer (for tests / debugging).
 * V = n_scenes*views_per_scene, N = V*G, P = H*W, T = tiles per view. */
typedef struct PsRasterStateLayout {
  size_t records;     /* float[N][16], one 64-byte line per pair: px,py,conx,cony | conz,opacity,depth,packed small-rect
                         origin (u32) | r,g,b,clamp bits | the pair's cell window (4 x u32, see cell_windows)      */
  size_t rects;       /* uint16[N][4]: tile rect xmin,ymin,xmax,ymax                   */
  size_t sorted_idx;  /* uint32[N]: per view, Gaussian ids in (depth, id) order; first n_vis valid */
  size_t sorted_rect; /* uint16[N][4]: rects permuted into sorted order                */
  size_t n_vis;       /* uint32[V]                                                     */
  size_t final_T;     /* float[V][P]                                                   */
  size_t n_contrib;   /* uint32[V][P]: 1-based index (within the tile's list) of the last contributor */
  size_t tile_end;    /* uint32[V][T]: the tile's last contributor = max n_contrib over its pixels (the tile forward's
                         waves max into it; the backward orders its tasks by it and starts its walks there) */
  size_t tile_ranges; /* uint32[V][T][2]: (start, count) of the tile's list in point_list     */
  size_t num_rendered;/* uint32[2]: D = sum of tile counts, overflow flag (D > capacity)      */
  size_t tile_order;  /* uint32[V*T]: (view,tile) ids, longest list first (launch order)      */
  size_t clamp_bits;  /* uint8[N]: bit c set = SH colour channel c was clamped at 0 (visible entries) */
  size_t checkpoint;  /* float[V][T][4][64][4]: per pixel (quadrant, lane) of a tile whose list is split in two
                         for the backward: transmittance after the list's first half and the colour composited
                         BEHIND it, divided by that transmittance (forward -> backward)             */
  size_t cell_windows;/* = records + 48: the fourth 16-byte word of every record line, STRIDE 64 bytes (version 7; a
                         separate array in version 6).  uint32[4]: which 4x4-pixel cells the pair can reach with alpha >= alpha_min (visible
                         entries; csrc/cell_window.h: a 64-bit mask over an 8x8 window of cells + its anchor,
                         or a cell range) -- the tile forward's per-row cull (csrc/raster_cells.hip) and the
                         quadrant mask of the tile backward's refine (forward -> backward)          */
  size_t total;
} PsRasterStateLayout;

void ps_raster_default_desc(PsRasterDesc* desc);
size_t ps_raster_state_bytes(const PsRasterDesc* desc); /* lives from forward to backward */
size_t ps_raster_temp_bytes(const PsRasterDesc* desc);  /* scratch of one forward call    */
size_t ps_raster_backward_temp_bytes(const PsRasterDesc* desc, size_t list_capacity);
int ps_raster_state_layout(const PsRasterDesc* desc, PsRasterStateLayout* out);

/* Forward.  Replaces GaussianRasterizer.forward (cuda_splatting.py:117-124).
 *   means      float[S][G][3]
 *   cov        float[S][G][6] or [S][G][3][3]   (cov_layout)
 *   sh         float[S][G][K][3] or [S][G][3][K] (sh_layout), or NULL
 *   colors     float[V][G][3] per-VIEW precomputed colours, or NULL (exactly one of sh/colors)
 *   opacity    float[S][G]
 *   view_params float[V][PS_VIEW_STRIDE]
 *   out_color  float[V][3][H][W]      out_radii int32[V][G]
 *   point_list uint32[list_capacity]: receives the per-(view,tile) lists of Gaussian ids in
 *              blend order (tile t of view v at state.tile_ranges[v][t] = (start,count)) --
 *              the bit-exact counterpart of the reference's sorted point list + ranges.
 *
 * The list length D is data dependent (the reference reads it back per view and resizes,
 * SURVEY.md 2.1 "num_rendered").  Two ways to size point_list:
 *   - ps_raster_forward_plan  (preprocess, depth sort, tile counts; D -> state.num_rendered)
 *     ps_raster_check         (ONE host read-back of D per batch of views)
 *     ps_raster_forward_render(bins + blend) with list_capacity >= D; or
 *   - ps_raster_forward with a capacity guess and no host sync at all: if D > capacity the
 *     overflow flag in state.num_rendered is raised, the image is invalid, and
 *     ps_raster_check / ps_raster_backward return PS_ERR_CAPACITY.
 * `temp` must be the same buffer for _plan and _render (it carries the tile counts).
 */
int ps_raster_forward(const PsRasterDesc* desc, const float* means, const float* cov,
                      const float* sh, const float* colors, const float* opacity,
                      const float* view_params, float* out_color, int32_t* out_radii,
                      void* state, size_t state_bytes, void* temp, size_t temp_bytes,
                      uint32_t* point_list, size_t list_capacity, void* stream);
int ps_raster_forward_plan(const PsRasterDesc* desc, const float* means, const float* cov,
                           const float* sh, const float* colors, const float* opacity,
                           const float* view_params, int32_t* out_radii, void* state,
                           size_t state_bytes, void* temp, size_t temp_bytes, void* stream);
int ps_raster_forward_render(const PsRasterDesc* desc, const float* view_params,
                             float* out_color, void* state, size_t state_bytes, void* temp,
                             size_t temp_bytes, uint32_t* point_list, size_t list_capacity,
                             void* stream);
/* ps_raster_forward_render in its two halves (tile lists, then blending), for hosts that want
 * to put other work between them. */
int ps_raster_forward_colors(const PsRasterDesc* desc, const float* means, const float* sh,
                             const float* view_params, const int32_t* radii, void* state,
                             size_t state_bytes, void* temp, size_t temp_bytes, void* stream);
int ps_raster_forward_bins(const PsRasterDesc* desc, void* state, size_t state_bytes, void* temp,
                           size_t temp_bytes, uint32_t* point_list, size_t list_capacity,
                           void* stream);
int ps_raster_forward_tiles(const PsRasterDesc* desc, const float* view_params, float* out_color,
                            void* state, size_t state_bytes, void* temp, size_t temp_bytes,
                            const uint32_t* point_list, size_t list_capacity, void* stream);

/* Backward.  Replaces _RasterizeGaussians.backward of the external module.
 *   radii       int32[V][G]  (the forward's out_radii)
 *   dL_dcolor   float[V][3][H][W]
 *   dL_dmeans   float[S][G][3]           dL_dcov  as cov layout
 *   dL_dsh      as sh layout, or NULL    dL_dcolors float[V][G][3], or NULL
 *   dL_dopacity float[S][G]
 *   dL_dmeans2D float[V][G][3] (NDC-scaled screen-space gradient, z = 0), may be NULL
 * temp_bytes >= ps_raster_backward_temp_bytes(desc, list_capacity).  The per-(tile, Gaussian)
 * partial gradients of Gaussians that touch <= 4 tiles are written to private slots and
 * summed in a fixed order (deterministic, no float atomics); larger ones use atomics.
 * Gradients are summed over the views of a scene (what autograd's `repeat` backward does,
 * decoder_splatting_cuda.py:53-56) and include the scene_scale chain rule.
 */
int ps_raster_backward(const PsRasterDesc* desc, const float* means, const float* cov,
                       const float* sh, const float* colors, const float* opacity,
                       const float* view_params, const int32_t* radii, const float* dL_dcolor,
                       const void* state, size_t state_bytes, void* temp, size_t temp_bytes,
                       const uint32_t* point_list, size_t list_capacity, float* dL_dmeans,
                       float* dL_dcov, float* dL_dsh, float* dL_dcolors, float* dL_dopacity,
                       float* dL_dmeans2D, void* stream);

/* Reads back (synchronising the stream) D = sum of tile-list lengths of the last
 * forward / forward_plan on this state, and whether it overflowed the capacity given to
 * ps_raster_forward.  Returns PS_OK, or PS_ERR_CAPACITY. */
int ps_raster_check(const PsRasterDesc* desc, const void* state, size_t state_bytes,
                    uint64_t* num_rendered, void* stream);

/* Zeroes the backward temp buffer's accumulators (all of them).  Optional since round 3 --
 * ps_raster_backward clears what it needs itself (see PS_FLAG_BWD_TEMP_ZEROED).  Issue it any time
 * after list_capacity is known and before ps_raster_backward, on any stream; then set
 * PS_FLAG_BWD_TEMP_ZEROED in the descriptor handed to ps_raster_backward.   Not for use inside a stream capture: it is the one entry point that issues a
 * hipMemsetAsync, and a memset node replayed from a hipGraph did not clear what the eager call clears on
 * ROCm 7.2 (round 5: PS_FLAG_DETERMINISTIC cleared its slots that way at first; it uses a kernel now). */
int ps_raster_backward_prepare(const PsRasterDesc* desc, void* temp, size_t temp_bytes,
                               size_t list_capacity, void* stream);

/* Camera set-up for n_views views in one launch: everything render_cuda computes on the host
 * side before it calls the rasterizer (cuda_splatting.py:64-71 renorm by 1/near when
 * scale_invariant != 0, :80-82 tan(fov/2) from the normalised intrinsics, :17-44 projection,
 * :84-87 transposed view / full-projection, :110 camera position) -> view_params[V][48].
 *   extrinsics float[V][4][4] camera-to-world, intrinsics float[V][3][3], near/far float[V],
 *   bg float[V][3]. */
int ps_camera_setup(int32_t n_views, const float* extrinsics, const float* intrinsics,
                    const float* near, const float* far, const float* bg,
                    int32_t scale_invariant, float* view_params, void* stream);

/* ---- (A) epipolar sampler ------------------------------------------------------------
 * Sampling geometry of EpipolarSampler.forward + get_depth + depth_to_relative_disparity
 * (/root/reference/src/model/encoder/epipolar/epipolar_sampler.py:51-123,
 *  src/geometry/epipolar_lines.py:157-292, src/geometry/projection.py:74-137,176-230,
 *  src/model/encoder/epipolar/conversions.py:17-27) for all (batch, view, other view, ray)
 * in one launch.  Rays are the h x w pixel centres of the (down-scaled) feature grid.
 *   c2w, w2c   float[b][v][4][4]  camera-to-world and its inverse (row-major)
 *   k, k_inv   float[b][v][3][3]  normalised intrinsics and inverse
 *   near, far  float[b][v]
 *   origins, directions float[b][v][h*w][3]
 *   segments   float[b][v][v-1][h*w][6] = xy_min(2), xy_max(2), t_min, t_max
 *   flags      uint8[b][v][v-1][h*w]: bit0 overlaps_image, bit1 near-point valid, bit2
 *              far-point valid, bits3-4 frame selector (min t), bits5-6 selector (max t)
 *   xy_sample  float[b][v][v-1][h*w][s][2]
 *   depth      float[b][v][v-1][h*w][s]  (unclipped), rel_disparity same shape (clipped to
 *              [near, far], then 1 - (1/d - 1/far)/(1/near - 1/far))
 * Other view index: index_v[v][ov] = ov < v ? ov : ov + 1 (heterogeneous_pairings.py:9-24). */
int ps_epipolar_geometry(int32_t b, int32_t v, int32_t h, int32_t w, int32_t s,
                         const float* c2w, const float* w2c, const float* k,
                         const float* k_inv, const float* near, const float* far,
                         float* origins, float* directions, float* segments, uint8_t* flags,
                         float* xy_sample, float* depth, float* rel_disparity, void* stream);

/* Fused epipolar gather + single-query cross-attention with folded key/value projections
 * (replaces F.grid_sample at epipolar_sampler.py:98-104, the depth-encoding add at
 * epipolar_transformer.py:113-121 and Attention.forward(x, z=kv) at attention.py:54-70; the
 * dense 128x128 folds stay plain GEMMs on the host side).  R = b*v*h*w rays, T = s*(v-1)
 * tokens per ray in the reference's "(s ov)" order, P = 2*octaves.
 *   fmap   float[b*v][h][w][c]   feature maps, channels-last
 *   xy_sample, flags, rel_disparity: outputs of ps_epipolar_geometry
 *   qt     float[R][heads][c]    q~_h = W_k,h^T q_h        u float[R][heads][P] = W_d^T q~_h
 *   e      float[R][heads][v-1]  q~_h . view_embedding[perm[ov]], or NULL
 * forward outputs  fbar[R][heads][c] = sum_i a_i feat_i, pbar[R][heads][P] = sum_i a_i pe_i,
 *                  abar[R][heads][v-1] = attention mass per other view, attn[R][heads][T].
 * backward inputs  the forward's attn, fbar, pbar, abar and dfbar, dpbar, dabar (same shapes:
 *                  one pass over the tokens, dq~ = scale (sum a da feat - (sum a da) fbar));
 *                  outputs dqt, du, de, ds (scratch
 *                  [R][heads][T]) and dfmap float[b*v][h][w][c] (written, not accumulated;
 *                  may be NULL; needs ray_boxes, a uint32[b*v*(v-1)*h*w] scratch).  No atomics
 *                  of any kind: one wave owns a 4x4 pixel tile of the gradient image in LDS,
 *                  so the result is bit-reproducible.  Limits: c % 4 == 0, c <= 256,
 *                  heads <= 4, T <= 128, h, w <= 255 for the gradient. */
typedef struct PsEpipolarDesc {
  int32_t b, v, h, w, s, c, heads, octaves;
  /* row strides in floats (0 = contiguous): qt/dqt, u/du, e/de, fbar/dfbar, pbar/dpbar,
   * abar/dabar.  With ld_q = ld_u = ld_e the three inputs are column blocks of ONE matrix
   * (one GEMM produces them), likewise the outputs.  ld_q and ld_f must be multiples of 4
   * and the qt / fbar / dfbar pointers 16-byte aligned. */
  int32_t ld_q, ld_u, ld_e, ld_f, ld_p, ld_a;
  /* head strides in floats inside a row (0 = each array keeps its heads contiguous: c, P,
   * v-1).  With hs_in = c + P + (v-1) (padded to a multiple of 4) and qt, u, e pointing at
   * columns 0, c, c + P, a row is [q~_0 u_0 e_0 | q~_1 u_1 e_1 | ...]: the layout a batched
   * per-head GEMM produces without any permutation; hs_out likewise for fbar / pbar / abar. */
  int32_t hs_in, hs_out;
  /* floats of padding behind each head's LAST block that the attention kernels fill with zeros
   * themselves (tail_pad_out: behind abar, or behind pbar when e is NULL, in the forward;
   * tail_pad_in: behind de, or du when de is NULL, in the backward), 0 ... 3: with a head stride
   * rounded up to a multiple of 4 the caller's row-of-heads matrices then need no clearing pass
   * before the GEMM that consumes them. */
  int32_t tail_pad_in, tail_pad_out;
} PsEpipolarDesc;
int ps_epipolar_gather(const PsEpipolarDesc* desc, const float* fmap, const float* xy_sample,
                       const uint8_t* flags, float* features /*[b][v][v-1][h*w][s][c]*/,
                       void* stream);
int ps_epipolar_attention_forward(const PsEpipolarDesc* desc, const float* fmap,
                                  const float* xy_sample, const uint8_t* flags,
                                  const float* rel_disparity, const float* qt, const float* u,
                                  const float* e, float scale, float* fbar, float* pbar,
                                  float* abar, float* attn, void* stream);
int ps_epipolar_attention_backward(const PsEpipolarDesc* desc, const float* fmap,
                                   const float* xy_sample, const uint8_t* flags,
                                   const float* rel_disparity, const float* qt,
                                   const float* attn, const float* fbar, const float* pbar,
                                   const float* abar /* NULL when e was NULL */,
                                   const float* dfbar, const float* dpbar,
                                   const float* dabar /* may be NULL */, float scale,
                                   float* dqt, float* du, float* de /* may be NULL */, float* ds,
                                   float* dfmap,
                                   uint32_t* ray_boxes, void* stream);

/* ---- Gaussian adapter: raw network outputs -> rasterizer inputs --------
 * Replaces GaussianAdapter.forward and its autograd graph
 * (src/model/encoder/common/gaussian_adapter.py:48-95, gaussians.py:8-41,
 * src/geometry/projection.py:65-108, src/misc/sh_rotation.py:10-31 incl. the e3nn Wigner-D).
 * Entries = (pixel, surface) pairs of a view, each with spp depth samples; Gaussian index
 * g = (view * entries_per_view + entry) * spp + sample, i.e. the flattening of
 * encoder_epipolar.py:197-214, so means / covariances / harmonics are directly the
 * [S][G]... arrays of ps_raster_forward (PS_COV_33, PS_SH_G3K).
 *   views        float[n_views][192]   per-view constants from ps_gaussian_adapter_views
 *   coordinates  float[n_views][entries][2]   normalised image coordinates of the rays
 *   depths       float[n_views][entries][spp]
 *   raw          float[n_views][entries][7 + 3 (deg+1)^2]  scale(3) | quaternion xyzw(4) | SH
 *   wigner_conj  double[164]: the fixed matrices P_1..P_4 with G_x = P G_y P^T on e3nn's real
 *                harmonics (pixelsplat_amd/wigner.py)
 * Opacities pass through unchanged and are not an argument. */
#define PS_ADAPTER_VIEW_STRIDE 192
int ps_gaussian_adapter_views(int32_t n_views, int32_t sh_degree, int32_t image_h, int32_t image_w,
                              const float* extrinsics, const float* intrinsics,
                              const double* wigner_conj, float* views, void* stream);
int ps_gaussian_adapter_forward(int32_t n_views, int32_t entries_per_view, int32_t spp,
                                int32_t sh_degree, float scale_min, float scale_max, float eps,
                                const float* views, const float* coordinates, const float* depths,
                                const float* raw, float* means, float* covariances,
                                float* harmonics, void* stream);
int ps_gaussian_adapter_backward(int32_t n_views, int32_t entries_per_view, int32_t spp,
                                 int32_t sh_degree, float scale_min, float scale_max, float eps,
                                 const float* views, const float* coordinates,
                                 const float* depths, const float* raw, const float* d_means,
                                 const float* d_covariances, const float* d_harmonics,
                                 float* d_raw, float* d_depths, float* d_coordinates,
                                 void* stream);

/* The same kernels on the encoder head's own layout (encoder_epipolar.py:149-173): head_rows
 * float[n_views][image_h * image_w][surfaces][2 + 7 + 3 (deg+1)^2] is the output of the
 * `to_gaussians` linear layer as it stands -- [xy offset (2) | scale (3) | quaternion (4) | SH] --
 * and the ray of an entry is its pixel centre ((x + .5) / w, (y + .5) / h) of sample_image_grid
 * (src/geometry/projection.py:111-140) moved by (sigmoid(offset) - 0.5) pixels (:155-164): no
 * slice copy of the 82 of 84 columns, no coordinate tensor, and d_head_rows is directly the
 * gradient of the linear layer's output. */
int ps_gaussian_head_forward(int32_t n_views, int32_t image_h, int32_t image_w, int32_t surfaces,
                             int32_t spp, int32_t sh_degree, float scale_min, float scale_max,
                             float eps, const float* views, const float* depths,
                             const float* head_rows, float* means, float* covariances,
                             float* harmonics, void* stream);
int ps_gaussian_head_backward(int32_t n_views, int32_t image_h, int32_t image_w, int32_t surfaces,
                              int32_t spp, int32_t sh_degree, float scale_min, float scale_max,
                              float eps, const float* views, const float* depths,
                              const float* head_rows, const float* d_means,
                              const float* d_covariances, const float* d_harmonics,
                              float* d_head_rows, float* d_depths, void* stream);

/* ---- Depth predictor sampling ------------------------------------------
 * Replaces everything DepthPredictorMonocular.forward does after its linear projection
 * (src/model/encoder/epipolar/depth_predictor_monocular.py:52-81): the "(dpt srf c)" split,
 * softmax / sigmoid, DistributionSampler.sample + gather
 * (src/misc/discrete_probability_distribution.py:7-33, epipolar/distribution_sampler.py:11-51),
 * relative_disparity_to_depth (epipolar/conversions.py:5-14), the optional transmittance
 * opacity, and -- when opacity_exponent != 0 -- the encoder's map_pdf_to_opacity and 1/gpp
 * (encoder_epipolar.py:97-110, :170).  Rows = (view, ray, surface).
 *   projected  float[n_views][rays][2 * buckets * surfaces]   output of the nn.Linear
 *   near, far  float[n_views]
 *   uniforms   float[n_views][rays][surfaces][spp]  the torch.rand draw of the sampler;
 *              NULL iff deterministic (then sample t is the t-th most probable bucket)
 *   depth, opacity float[rows][spp];  index int32[rows][spp] (kept for the backward)
 * A ray's buckets * surfaces logits (x2 for the backward, x4 with transmittance) must fit the
 * 9 KB per-wave LDS slab -- buckets * surfaces <= 512 always does -- else PS_ERR_UNSUPPORTED. */
typedef struct PsDepthSamplerDesc {
  int32_t n_views, rays_per_view, buckets, surfaces, spp;
  int32_t deterministic;      /* 1: gather_discrete_topk, 0: sample_discrete_distribution */
  int32_t use_transmittance;  /* depth_predictor_monocular.py:71-78 */
  float opacity_exponent;     /* 2**x of encoder_epipolar.py:106-107; 0 = return the density */
  float opacity_scale;        /* 1/gaussians_per_pixel of :170; 1 for the bare module */
} PsDepthSamplerDesc;
int ps_depth_sampler_forward(const PsDepthSamplerDesc* desc, const float* projected,
                             const float* near, const float* far, const float* uniforms,
                             float* depth, float* opacity, int32_t* index, void* stream);
int ps_depth_sampler_backward(const PsDepthSamplerDesc* desc, const float* projected,
                              const float* near, const float* far, const int32_t* index,
                              const float* d_depth, const float* d_opacity, float* d_projected,
                              void* stream);

/* ---- Image-side losses -------------------------------------------------
 * ps_image_mse: LossMse (src/loss/loss_mse.py:22-31) and compute_psnr
 * (src/evaluation/metrics.py:12-19) in one pass over n_images images of `elems` floats each
 * (c*h*w): sse[i] = sum (pred - target)^2, sse_clipped[i] = the same on values clipped to
 * [0, 1] (may be NULL), grad = grad_scale * (pred - target) (may be NULL) -- with
 * grad_scale = 2 * weight / (n_images * elems) that is dLoss/dpred, the dL_dimage of
 * ps_raster_backward.  loss = weight * sum(sse) / (n_images * elems);
 * psnr[i] = -10 log10(sse_clipped[i] / elems).  Deterministic (fixed-order partial sums). */
size_t ps_image_mse_workspace_bytes(int32_t n_images, int32_t elems);
int ps_image_mse(int32_t n_images, int32_t elems, const float* pred, const float* target,
                 float grad_scale, float* grad, float* sse, float* sse_clipped, void* workspace,
                 size_t workspace_bytes, void* stream);

/* LossDepth (src/loss/loss_depth.py:26-60): depth [n_images][h][w] clamped to
 * [log near, log far] and normalised, first (or second) differences along x and y, optional
 * bilateral weights exp(-sigma * max_c diff(target)), weight * (mean|dx| + mean|dy|).
 * loss: device float[1].  backward: d_depth = d_loss[0] * dLoss/ddepth (d_loss: device float[1]). */
typedef struct PsDepthLossDesc {
  int32_t n_images, height, width, channels;
  int32_t use_second_derivative;
  int32_t use_sigma;       /* 0: sigma_image is None (target_image may be NULL) */
  float sigma_image, weight;
} PsDepthLossDesc;
size_t ps_depth_smoothness_workspace_bytes(const PsDepthLossDesc* desc);
int ps_depth_smoothness_forward(const PsDepthLossDesc* desc, const float* depth, const float* near,
                                const float* far, const float* target_image, float* loss,
                                void* workspace, size_t workspace_bytes, void* stream);
int ps_depth_smoothness_backward(const PsDepthLossDesc* desc, const float* depth,
                                 const float* near, const float* far, const float* target_image,
                                 const float* d_loss, float* d_depth, void* stream);

/* ---- Folded weights of one epipolar cross-attention layer -----------------------
 * The linear maps either side of ps_epipolar_attention_* -- to_q / to_kv / to_out of
 * src/model/transformer/attention.py:45-52, the depth-encoding Linear of
 * epipolar_transformer.py:67-70 and the view embeddings of :126-131 -- folded into the two
 * matrices the per-ray GEMMs use, rows laid out per head ([q~ (kv_dim) | u (2 octaves) |
 * e (other_views) | pad to 4] = Lh rows per head):
 *   w_in  float[heads * Lh][q_dim],  w_o_t float[heads * Lh][out_dim],  bias float[out_dim]
 * from w_q [heads*head_dim][q_dim], w_kv [2*heads*head_dim][kv_dim], w_out [out_dim][heads*head_dim],
 * b_out [out_dim] or NULL, depth_w [kv_dim][2*octaves], depth_b [kv_dim], view_emb
 * [other_views][kv_dim] or NULL (other_views = 0).  scratch (ps_fold_scratch_floats floats) is
 * written by the forward and read by the backward; the backward needs a second buffer of the
 * same size.  Gradient outputs mirror the inputs (g_b_out / g_view_emb may be NULL). */
typedef struct PsFoldDesc {
  int32_t heads, head_dim, kv_dim, q_dim, out_dim, octaves, other_views;
} PsFoldDesc;
size_t ps_fold_scratch_floats(const PsFoldDesc* desc);
int ps_fold_attention_weights(const PsFoldDesc* desc, const float* w_q, const float* w_kv,
                              const float* w_out, const float* b_out, const float* depth_w,
                              const float* depth_b, const float* view_emb, float* w_in,
                              float* w_o_t, float* bias, float* scratch, void* stream);
int ps_fold_attention_weights_backward(const PsFoldDesc* desc, const float* w_q, const float* w_kv,
                                       const float* w_out, const float* b_out,
                                       const float* depth_w, const float* depth_b,
                                       const float* view_emb, const float* scratch,
                                       const float* d_w_in, const float* d_w_o_t,
                                       const float* d_bias, float* back_scratch, float* g_w_q,
                                       float* g_w_kv, float* g_w_out, float* g_b_out,
                                       float* g_depth_w, float* g_depth_b, float* g_view_emb,
                                       void* stream);

/* LayerNorm over the last dimension of x [rows][dim] (the PreNorm of the cross-attention
 * layers, src/model/transformer/pre_norm.py:34-35), fp32, dim % 4 == 0 and dim <= 512 (else
 * PS_ERR_UNSUPPORTED).  mean / rstd [rows] are saved for the backward; the backward's
 * workspace holds ps_layer_norm_workspace_floats(rows, dim) floats; d_gamma / d_beta are
 * summed in a fixed order (deterministic).  d_residual: the gradient reaching x through the
 * residual branch of `x + f(LayerNorm(x))`, added in the same pass. */
size_t ps_layer_norm_workspace_floats(int32_t rows, int32_t dim);
int ps_layer_norm_forward(int32_t rows, int32_t dim, float eps, const float* x,
                          const float* gamma, const float* beta, float* y, float* mean,
                          float* rstd, void* stream);
int ps_layer_norm_backward(int32_t rows, int32_t dim, const float* x, const float* gamma,
                           const float* mean, const float* rstd, const float* dy,
                           const float* d_residual /* NULL, or [rows][dim] added to dx */,
                           float* dx, float* d_gamma, float* d_beta, float* workspace,
                           void* stream);

/* Feature-map gradient of n_layers (1 or 2) attention layers that share the geometry, in ONE
 * scatter pass: dfmap = sum over layers of the gradient ps_epipolar_attention_backward would
 * write for that layer (call that function with dfmap = NULL and hand its ds here).  qt, attn,
 * dfbar, ds: arrays of n_layers device pointers, laid out as in the descriptor. */
int ps_epipolar_feature_grad(const PsEpipolarDesc* desc, int32_t n_layers,
                             const float* xy_sample, const uint8_t* flags,
                             const float* const* qt, const float* const* attn,
                             const float* const* dfbar, const float* const* ds, float* dfmap,
                             uint32_t* ray_boxes, void* stream);

/* The same gradient in two passes (values equal up to the association of the layer / head sum,
 * deterministic): pass 1 writes, once per token, the gradient w.r.t. the gathered feature --
 * token_grad [b][v][v-1][h*w][s][c], ps_epipolar_token_grad_floats(desc) floats of caller-owned
 * scratch: the tensor the reference's autograd holds as d(kv), epipolar_transformer.py:121 --
 * and pass 2 scatters those rows into the maps.  A ray's 16 coefficient rows are then read once
 * instead of once per tile its line crosses: 2.5x less memory-side traffic and 1.6x less time
 * at the paper configuration. */
size_t ps_epipolar_token_grad_floats(const PsEpipolarDesc* desc);
/* uint32 words `ray_boxes` must hold for the two-pass call: the scratch of its binned gather -- per
 * (source map, block of 256 tokens, 4x4 tile) counts, per tile a length, an offset and a slot of the
 * longest-first block order, and the per-tile token lists (at most 4 entries per token); the
 * single-pass call needs only the first b*v*(v-1)*h*w words (one packed pixel box per ray and other view). */
size_t ps_epipolar_ray_box_words(const PsEpipolarDesc* desc);
int ps_epipolar_feature_grad_two_pass(const PsEpipolarDesc* desc, int32_t n_layers,
                                      const float* xy_sample, const uint8_t* flags,
                                      const float* const* qt, const float* const* attn,
                                      const float* const* dfbar, const float* const* ds,
                                      float* dfmap, uint32_t* ray_boxes, float* token_grad,
                                      void* stream);
/* The same in two calls, for callers that overlap them with other work (the reference's autograd
 * runs grid_sample's backward, epipolar_sampler.py:98-111, wherever the graph puts it; here the host
 * decides): `_bins` builds the per-tile token lists of the binned gather into `ray_boxes`
 * (ps_epipolar_ray_box_words words) from the sampling geometry ALONE -- it may run as soon as xy_sample /
 * flags exist, e.g. on a side stream during the forward pass -- and `_binned` (token gradients + list
 * gather) consumes them once the layers' gradients are there.  _two_pass == _bins then _binned. */
int ps_epipolar_feature_bins(const PsEpipolarDesc* desc, const float* xy_sample, const uint8_t* flags,
                             uint32_t* ray_boxes, void* stream);
int ps_epipolar_feature_grad_binned(const PsEpipolarDesc* desc, int32_t n_layers,
                                    const float* xy_sample, const uint8_t* flags,
                                    const float* const* qt, const float* const* attn,
                                    const float* const* dfbar, const float* const* ds,
                                    float* dfmap, const uint32_t* ray_boxes, float* token_grad,
                                    void* stream);

/* w2c[i] = c2w[i]^-1 (4x4) and k_inv[i] = k[i]^-1 (3x3) for n cameras, one launch, no host
 * sync (replaces the sampler's torch.linalg.inv calls: src/geometry/epipolar_lines.py:167,
 * src/geometry/projection.py:84). */
int ps_invert_cameras(int32_t n, const float* c2w, const float* k, float* w2c, float* k_inv,
                      void* stream);

/* C[m][n] = sum_k A[k][m] B[k][n] in fp32 (v_mfma_f32_32x32x2_f32), split over k with a
 * fixed-order reduction: the weight gradients of the folded attention matrices
 * (dW = dY^T X over all rays; autograd of the Linear layers at attention.py:36-45,
 * epipolar_transformer.py:61-66).  A, B row-major with leading dimensions lda >= m,
 * ldb >= n (multiples of 4, 16-byte aligned), m % 4 == 0, n % 4 == 0.  workspace: ps_gemm_tn_workspace_bytes. */
size_t ps_gemm_tn_workspace_bytes(int32_t m, int32_t n, int32_t k);
int ps_gemm_tn_f32(int32_t m, int32_t n, int32_t k, const float* a, int32_t lda, const float* b,
                   int32_t ldb, float* c, void* workspace, size_t workspace_bytes, void* stream);
/* Same product plus, as a by-product of the operand stream, colsum_a[m] = sum_k a[k][m] (m floats;
 * may be NULL): with a = dY this is the bias gradient of the Linear layer (autograd of
 * nn.Linear(inner_dim, dim) at attention.py:45), which otherwise costs a pass of its own over dY. */
int ps_gemm_tn_colsum_f32(int32_t m, int32_t n, int32_t k, const float* a, int32_t lda,
                          const float* b, int32_t ldb, float* c, float* colsum_a, void* workspace,
                          size_t workspace_bytes, void* stream);

/* Profiling aid for bench.py (process-global, off by default; the only mutable global in the
 * library).  When enabled every kernel group the library launches is bracketed by hipEvents
 * on the caller's stream; ps_profile_collect synchronises those events, ADDS the elapsed
 * milliseconds / launch counts per group into the caller's arrays (length
 * ps_profile_group_count()) and clears the pending list.
 * `on` is a bit set: 1 = the hipEvent timing above; 2 = a named roctx range "ps:<group>" around
 * every group on the calling thread (the tracing hooks of SURVEY.md 5: the reference has none of
 * its own, Lightning's profiler plays that role), visible to `rocprofv3 --marker-trace`; PS_ROCTX=1
 * in the environment turns the ranges on without a call.  The marker library is looked up with
 * dlopen at first use; ps_roctx_available() tells whether one was found. */
int ps_profile_enable(int on);
int ps_roctx_available(void);
int ps_profile_group_count(void);
const char* ps_profile_group_name(int group);
int ps_profile_collect(double* total_ms, int64_t* launches);

const char* ps_status_string(int status);
/* "pixelsplat_hip gfx950 <date> <time> | <file>:<hash> ..." -- one hash per translation unit over
 * its source, the shared headers and the compile flags (pixelsplat_amd/build.py). */
const char* ps_build_info(void);
/* Layout version of the descriptor structs of this header (PsRasterDesc, PsEpipolarDesc, ...): bumped
 * whenever a struct grows or a field changes meaning.  A host built against an older header would hand
 * the library shorter structs (PsEpipolarDesc grew by tail_pad_in / tail_pad_out in version 4; PsRasterStateLayout by
 * cell_windows in version 6, interleaved with the records in version 7): check
 * ps_abi_version() == PS_ABI_VERSION once after loading (pixelsplat_amd/_lib.py does). */
#define PS_ABI_VERSION 7
int ps_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PIXELSPLAT_HIP_H */
