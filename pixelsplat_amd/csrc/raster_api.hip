// extern "C" surface of libpixelsplat_hip.so -- see include/pixelsplat_hip.h.
// No torch types, no exceptions, no device synchronisation, no hidden state.
#include "raster_common.h"

#include <atomic>
#include <cstdlib>
#include <dlfcn.h>
#include <mutex>
#include <vector>

using namespace ps;

namespace {

// ---- optional per-kernel-group timing (bench.py) ----------------------------------------
enum Group { G_PRE_FWD = 0, G_SORT, G_BINS, G_TILES_FWD, G_TILES_BWD, G_PRE_BWD, G_MEMSET,
             G_EPI_GEOM, G_EPI_FWD, G_EPI_BWD, G_EPI_FGRAD, G_GEMM_TN, G_ADAPTER_FWD, G_ADAPTER_BWD,
             G_DEPTH_FWD, G_DEPTH_BWD, G_LOSS, G_TASK_ORDER, G_COUNT };
const char* kGroupNames[G_COUNT] = {"preprocess_forward", "depth_sort", "tile_bins", "tiles_forward",
                                    "tiles_backward", "preprocess_backward", "memset",
                                    "epipolar_geometry", "epipolar_attention_forward",
                                    "epipolar_attention_backward", "epipolar_feature_grad",
                                    "gemm_tn_splitk", "gaussian_adapter_forward",
                                    "gaussian_adapter_backward", "depth_sampler_forward",
                                    "depth_sampler_backward", "image_losses", "backward_task_order"};
std::atomic<int> g_profile_on{0};     // bit 0: HIP events per group, bit 1: roctx ranges per group
std::mutex g_profile_mu;
struct Pending { hipEvent_t a, b; int group; };
std::vector<Pending> g_pending;

// roctx ranges (the tracing row of SURVEY.md 5): every profile group is a named range
// "ps:<group>" on the calling thread, visible to `rocprofv3 --marker-trace` next to the kernel
// trace.  The marker library is looked up at run time (rocprofiler-sdk's roctx, else roctracer's):
// no link-time dependency, a no-op when neither is there.  On with ps_profile_enable(2|...) or
// PS_ROCTX=1 in the environment.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  // the marker library is only looked up when ranges are asked for (PS_ROCTX=1, ps_profile_enable(2),
  // or an explicit ps_roctx_available()): a host process that never profiles loads nothing
  Roctx() {
    for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1",
                             "libroctx64.so", "libroctx64.so.4"}) {
      if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (push && pop) break;
        push = nullptr; pop = nullptr;
        dlclose(h);                    // not a marker library after all
      }
    }
  }
};
Roctx& roctx() { static Roctx r; return r; }
bool roctx_env_on() {
  static const bool on = [] { const char* e = getenv("PS_ROCTX"); return e && e[0] && e[0] != '0'; }();
  return on;
}
const char* kRangeNames[] = {"ps:preprocess_forward", "ps:depth_sort", "ps:tile_bins", "ps:tiles_forward",
                             "ps:tiles_backward", "ps:preprocess_backward", "ps:memset",
                             "ps:epipolar_geometry", "ps:epipolar_attention_forward",
                             "ps:epipolar_attention_backward", "ps:epipolar_feature_grad",
                             "ps:gemm_tn_splitk", "ps:gaussian_adapter_forward",
                             "ps:gaussian_adapter_backward", "ps:depth_sampler_forward",
                             "ps:depth_sampler_backward", "ps:image_losses", "ps:backward_task_order"};
static_assert(sizeof(kRangeNames) / sizeof(kRangeNames[0]) == G_COUNT, "one range name per group");

struct Scope {
  hipEvent_t a = nullptr, b = nullptr; int group; hipStream_t st; bool on; bool range = false;
  Scope(int g, hipStream_t s) : group(g), st(s), on((g_profile_on.load() & 1) != 0) {
    if (roctx_env_on() || (g_profile_on.load() & 2)) {     // (no library lookup otherwise)
      Roctx& rx = roctx();
      if (rx.push) { rx.push(kRangeNames[g]); range = true; }
    }
    if (!on) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
    hipEventRecord(a, st);
  }
  ~Scope() {
    if (range) roctx().pop();
    if (!on) return;
    hipEventRecord(b, st);
    std::lock_guard<std::mutex> lk(g_profile_mu);
    g_pending.push_back({a, b, group});
  }
};

bool desc_ok(const PsRasterDesc* d) {
  if (!d) return false;
  if (d->n_scenes <= 0 || d->views_per_scene <= 0 || d->n_gaussians <= 0) return false;
  if (d->height <= 0 || d->width <= 0) return false;
  if (d->sh_degree < 0 || d->sh_coeffs < 0) return false;
  return true;
}

int check_launch() { return hipGetLastError() == hipSuccess ? PS_OK : PS_ERR_LAUNCH; }

}  // namespace

extern "C" {

void ps_raster_default_desc(PsRasterDesc* d) {
  if (!d) return;
  d->n_scenes = 1; d->views_per_scene = 1; d->n_gaussians = 0; d->height = 0; d->width = 0;
  d->sh_degree = 0; d->sh_coeffs = 0; d->sh_layout = PS_SH_GK3; d->cov_layout = PS_COV_6;
  d->flags = 0;
  d->near_cull = 0.2f; d->guard = 1.3f; d->lowpass = 0.3f; d->w_eps = 1e-7f;
  d->lambda_floor = 0.1f; d->alpha_max = 0.99f; d->alpha_min = 1.0f / 255.0f;
  d->t_min = 1e-4f; d->det2_eps = 1e-7f;
}

size_t ps_raster_state_bytes(const PsRasterDesc* d) {
  return desc_ok(d) ? make_state_layout(*d).total : 0;
}
size_t ps_raster_temp_bytes(const PsRasterDesc* d) {
  return desc_ok(d) ? make_temp_layout(*d).total : 0;
}
size_t ps_raster_backward_temp_bytes(const PsRasterDesc* d, size_t list_capacity) {
  return desc_ok(d) ? make_bwd_temp_layout(*d, list_capacity).total : 0;
}
int ps_raster_state_layout(const PsRasterDesc* d, PsRasterStateLayout* out) {
  if (!desc_ok(d) || !out) return PS_ERR_BAD_ARG;
  *out = make_state_layout(*d);
  return PS_OK;
}

namespace {
struct FwdPtrs {
  float* records; uint2* rects; uint32_t* sorted_idx; uint2* sorted_rect; uint32_t* n_vis;
  float* final_T; uint32_t* n_contrib; float4* checkpoint; uint32_t* tile_end; uint32_t* tile_ranges;
  uint32_t* num_rendered; uint32_t* tile_order; uint8_t* clamp_bits; uint4* cell_windows;
  uint32_t *keys_a, *keys_b, *vals_a, *vals_b, *block_hist, *pass_info, *bin_counts;
};
FwdPtrs fwd_ptrs(const PsRasterDesc& d, void* state, void* temp) {
  const PsRasterStateLayout L = make_state_layout(d);
  const TempLayout T = make_temp_layout(d);
  char* sb = (char*)state; char* tb = (char*)temp;
  FwdPtrs p;
  p.records = (float*)(sb + L.records); p.rects = (uint2*)(sb + L.rects);
  p.sorted_idx = (uint32_t*)(sb + L.sorted_idx); p.sorted_rect = (uint2*)(sb + L.sorted_rect);
  p.n_vis = (uint32_t*)(sb + L.n_vis); p.final_T = (float*)(sb + L.final_T);
  p.n_contrib = (uint32_t*)(sb + L.n_contrib); p.checkpoint = (float4*)(sb + L.checkpoint);
  p.tile_end = (uint32_t*)(sb + L.tile_end);
  p.tile_ranges = (uint32_t*)(sb + L.tile_ranges);
  p.num_rendered = (uint32_t*)(sb + L.num_rendered);
  p.tile_order = (uint32_t*)(sb + L.tile_order);
  p.clamp_bits = (uint8_t*)(sb + L.clamp_bits);
  p.cell_windows = (uint4*)(sb + L.cell_windows);
  p.keys_a = (uint32_t*)(tb + T.keys_a); p.keys_b = (uint32_t*)(tb + T.keys_b);
  p.vals_a = (uint32_t*)(tb + T.vals_a); p.vals_b = (uint32_t*)(tb + T.vals_b);
  p.block_hist = (uint32_t*)(tb + T.block_hist); p.bin_counts = (uint32_t*)(tb + T.bin_counts);
  p.pass_info = (uint32_t*)(tb + T.pass_info);
  return p;
}
int check_sizes(const PsRasterDesc& d, size_t state_bytes, size_t temp_bytes) {
  if (state_bytes < make_state_layout(d).total || temp_bytes < make_temp_layout(d).total)
    return PS_ERR_WORKSPACE;
  return PS_OK;
}
uint32_t clamp_capacity(size_t c) { return c > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)c; }
}  // namespace

int ps_raster_forward_plan(const PsRasterDesc* d, const float* means, const float* cov,
                           const float* sh, const float* colors, const float* opacity,
                           const float* view_params, int32_t* out_radii, void* state,
                           size_t state_bytes, void* temp, size_t temp_bytes, void* stream) {
  if (!desc_ok(d) || !means || !cov || !opacity || !view_params || !out_radii || !state || !temp)
    return PS_ERR_BAD_ARG;
  if ((sh == nullptr) == (colors == nullptr)) return PS_ERR_BAD_ARG;  // exactly one
  if (sh) {
    if (d->sh_degree > 4) return PS_ERR_UNSUPPORTED;
    if (d->sh_coeffs < (d->sh_degree + 1) * (d->sh_degree + 1)) return PS_ERR_BAD_ARG;
  }
  const Dims m = make_dims(*d);
  if (m.gx > 32767 || m.gy > 32767) return PS_ERR_UNSUPPORTED;
  if (int rc = check_sizes(*d, state_bytes, temp_bytes)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const FwdPtrs p = fwd_ptrs(*d, state, temp);
  {
    Scope sc(G_PRE_FWD, st);
    launch_preprocess_forward(*d, means, cov, sh, colors, opacity, view_params, p.records,
                              p.keys_a, p.rects, out_radii, p.clamp_bits, p.cell_windows, true,
                              !(d->flags & PS_FLAG_DEFER_SH_COLORS), st);
  }
  {
    Scope sc(G_SORT, st);
    launch_sort(*d, p.keys_a, p.keys_b, p.vals_a, p.vals_b, p.block_hist, p.pass_info, p.sorted_idx, p.rects,
                p.sorted_rect, p.n_vis, st);
  }
  {
    Scope sc(G_BINS, st);
    launch_bin_count(*d, p.sorted_rect, p.n_vis, p.bin_counts, p.tile_ranges, p.num_rendered,
                     p.tile_order, p.tile_end, st);
  }
  return check_launch();
}

int ps_raster_forward_colors(const PsRasterDesc* d, const float* means, const float* sh,
                             const float* view_params, const int32_t* radii, void* state,
                             size_t state_bytes, void* temp, size_t temp_bytes, void* stream) {
  if (!desc_ok(d) || !means || !sh || !view_params || !radii || !state || !temp)
    return PS_ERR_BAD_ARG;
  if (d->sh_degree > 4) return PS_ERR_UNSUPPORTED;
  if (int rc = check_sizes(*d, state_bytes, temp_bytes)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const FwdPtrs p = fwd_ptrs(*d, state, temp);
  Scope sc(G_PRE_FWD, st);
  launch_preprocess_forward(*d, means, nullptr, sh, nullptr, nullptr, view_params, p.records,
                            p.keys_a, p.rects, const_cast<int32_t*>(radii), p.clamp_bits, p.cell_windows,
                            false, true, st);
  return check_launch();
}

int ps_raster_forward_bins(const PsRasterDesc* d, void* state, size_t state_bytes, void* temp,
                           size_t temp_bytes, uint32_t* point_list, size_t list_capacity,
                           void* stream) {
  if (!desc_ok(d) || !state || !temp) return PS_ERR_BAD_ARG;
  if (!point_list && list_capacity > 0) return PS_ERR_BAD_ARG;
  if (int rc = check_sizes(*d, state_bytes, temp_bytes)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const FwdPtrs p = fwd_ptrs(*d, state, temp);
  const uint32_t cap = clamp_capacity(list_capacity);
  Scope sc(G_BINS, st);
  launch_bin_write(*d, p.sorted_rect, p.sorted_idx, p.n_vis, p.bin_counts, p.tile_ranges,
                   p.num_rendered, point_list, cap, st);
  return check_launch();
}

int ps_raster_forward_tiles(const PsRasterDesc* d, const float* view_params, float* out_color,
                            void* state, size_t state_bytes, void* temp, size_t temp_bytes,
                            const uint32_t* point_list, size_t list_capacity, void* stream) {
  if (!desc_ok(d) || !view_params || !out_color || !state || !temp) return PS_ERR_BAD_ARG;
  if (!point_list && list_capacity > 0) return PS_ERR_BAD_ARG;
  if (int rc = check_sizes(*d, state_bytes, temp_bytes)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const FwdPtrs p = fwd_ptrs(*d, state, temp);
  Scope sc(G_TILES_FWD, st);
  // PS_FORWARD_QUADRANTS=1: the round-5 8x8-quadrant forward (A/B runs: tools/ab_cells.sh); same results
  static const bool old_forward = [] { const char* e = getenv("PS_FORWARD_QUADRANTS"); return e && e[0] == '1'; }();
  if (old_forward)
    launch_tiles_forward(*d, p.records, p.tile_order, p.tile_ranges, point_list, clamp_capacity(list_capacity),
                         view_params, out_color, p.final_T, p.n_contrib, p.checkpoint, p.tile_end, st);
  else
    launch_tiles_forward_rows(*d, p.records, p.cell_windows, p.tile_order, p.tile_ranges, point_list,
                              clamp_capacity(list_capacity), view_params, out_color, p.final_T, p.n_contrib,
                              p.checkpoint, p.tile_end, st);
  return check_launch();
}

int ps_raster_forward_render(const PsRasterDesc* d, const float* view_params, float* out_color,
                             void* state, size_t state_bytes, void* temp, size_t temp_bytes,
                             uint32_t* point_list, size_t list_capacity, void* stream) {
  if (!view_params || !out_color) return PS_ERR_BAD_ARG;
  if (int rc = ps_raster_forward_bins(d, state, state_bytes, temp, temp_bytes, point_list,
                                      list_capacity, stream))
    return rc;
  return ps_raster_forward_tiles(d, view_params, out_color, state, state_bytes, temp, temp_bytes,
                                 point_list, list_capacity, stream);
}

int ps_raster_forward(const PsRasterDesc* d, const float* means, const float* cov,
                      const float* sh, const float* colors, const float* opacity,
                      const float* view_params, float* out_color, int32_t* out_radii,
                      void* state, size_t state_bytes, void* temp, size_t temp_bytes,
                      uint32_t* point_list, size_t list_capacity, void* stream) {
  if (!out_color) return PS_ERR_BAD_ARG;
  if (int rc = ps_raster_forward_plan(d, means, cov, sh, colors, opacity, view_params, out_radii,
                                      state, state_bytes, temp, temp_bytes, stream))
    return rc;
  return ps_raster_forward_render(d, view_params, out_color, state, state_bytes, temp, temp_bytes,
                                  point_list, list_capacity, stream);
}

int ps_raster_backward(const PsRasterDesc* d, const float* means, const float* cov,
                       const float* sh, const float* colors, const float* opacity,
                       const float* view_params, const int32_t* radii, const float* dL_dcolor,
                       const void* state, size_t state_bytes, void* temp, size_t temp_bytes,
                       const uint32_t* point_list, size_t list_capacity, float* dL_dmeans,
                       float* dL_dcov, float* dL_dsh, float* dL_dcolors, float* dL_dopacity,
                       float* dL_dmeans2D, void* stream) {
  (void)opacity; (void)colors;
  if (!desc_ok(d) || !means || !cov || !view_params || !radii || !dL_dcolor || !state || !temp ||
      !dL_dmeans || !dL_dcov || !dL_dopacity)
    return PS_ERR_BAD_ARG;
  if (sh && !dL_dsh) return PS_ERR_BAD_ARG;
  if (!sh && !dL_dcolors) return PS_ERR_BAD_ARG;
  const Dims m = make_dims(*d);
  const PsRasterStateLayout L = make_state_layout(*d);
  const uint32_t capacity = clamp_capacity(list_capacity);
  if (!point_list && capacity > 0) return PS_ERR_BAD_ARG;
  const BwdTempLayout T = make_bwd_temp_layout(*d, capacity);
  if (state_bytes < L.total || temp_bytes < T.total) return PS_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const char* sb = (const char*)state; char* tb = (char*)temp;
  const float* records = (const float*)(sb + L.records);
  const uint2* rects = (const uint2*)(sb + L.rects);
  const uint32_t* tile_ranges = (const uint32_t*)(sb + L.tile_ranges);
  const uint32_t* tile_end = (const uint32_t*)(sb + L.tile_end);
  const float* final_T = (const float*)(sb + L.final_T);
  const uint32_t* n_contrib = (const uint32_t*)(sb + L.n_contrib);
  const float4* checkpoint = (const float4*)(sb + L.checkpoint);
  float* grad2d = (float*)(tb + T.grad2d);
  float* tile_grads = (float*)(tb + T.tile_grads);
  uint32_t* task_order = (uint32_t*)(tb + T.task_order);
  const bool deterministic = (d->flags & PS_FLAG_DETERMINISTIC) != 0;
  float* det_slots = deterministic ? (float*)(tb + T.det_slots) : nullptr;
  {
    Scope sc(G_MEMSET, st);
    // only the grad2d rows the tile backward adds into with atomics (pairs over > kInvSlots
    // tiles); the private slots are written exactly once each and need no clearing
    if (deterministic) {      // no atomics: the per-entry slots start from zero instead
      launch_deterministic_clear(det_slots, capacity, st);
    } else if (!(d->flags & PS_FLAG_BWD_TEMP_ZEROED)) {
      launch_clear_atomic_rows(*d, (const uint2*)(sb + L.sorted_rect), (const uint32_t*)(sb + L.sorted_idx),
                               (const uint32_t*)(sb + L.n_vis), grad2d, st);
    }
  }
  {
    // the tile backward's tasks, longest WALK first (the forward left every tile's last contributor);
    // a group of its own (it used to be timed inside `memset`)
    Scope sc(G_TASK_ORDER, st);
    launch_backward_task_order(*d, tile_ranges, tile_end, capacity, task_order, st);
  }
  {
    Scope sc(G_TILES_BWD, st);
    launch_tiles_backward(*d, records, (const uint4*)(sb + L.cell_windows), task_order, tile_ranges, point_list, capacity, view_params,
                          final_T, n_contrib, checkpoint, tile_end, dL_dcolor, grad2d, tile_grads,
                          det_slots, st);
    if (deterministic)
      launch_deterministic_reduce(*d, radii, rects, (const uint32_t*)(sb + L.sorted_idx),
                                  (const uint32_t*)(sb + L.n_vis), tile_ranges, point_list, capacity,
                                  det_slots, (uint32_t*)(tb + T.rank_of), grad2d, st);
  }
  {
    Scope sc(G_PRE_BWD, st);
    launch_preprocess_backward(*d, means, cov, sh, view_params, records, radii, rects,
                               tile_grads, (const uint8_t*)(sb + L.clamp_bits),
                               (float*)(tb + T.color_grads), grad2d, dL_dmeans, dL_dcov, dL_dsh,
                               dL_dcolors, dL_dopacity, dL_dmeans2D, st);
  }
  return check_launch();
}

int ps_raster_backward_prepare(const PsRasterDesc* d, void* temp, size_t temp_bytes,
                               size_t list_capacity, void* stream) {
  if (!desc_ok(d) || !temp) return PS_ERR_BAD_ARG;
  const BwdTempLayout T = make_bwd_temp_layout(*d, clamp_capacity(list_capacity));
  if (temp_bytes < T.total) return PS_ERR_WORKSPACE;
  // not timed as a profile group: it is meant to run concurrently with other work
  if (hipMemsetAsync(temp, 0, T.zeroed, (hipStream_t)stream) != hipSuccess) return PS_ERR_LAUNCH;
  return PS_OK;
}

int ps_camera_setup(int32_t n_views, const float* extrinsics, const float* intrinsics,
                    const float* near, const float* far, const float* bg,
                    int32_t scale_invariant, float* view_params, void* stream) {
  if (n_views <= 0 || !extrinsics || !intrinsics || !near || !far || !bg || !view_params)
    return PS_ERR_BAD_ARG;
  launch_camera_setup(n_views, extrinsics, intrinsics, near, far, bg, scale_invariant,
                      view_params, (hipStream_t)stream);
  return check_launch();
}

int ps_epipolar_geometry(int32_t b, int32_t v, int32_t h, int32_t w, int32_t s,
                         const float* c2w, const float* w2c, const float* k,
                         const float* k_inv, const float* near, const float* far,
                         float* origins, float* directions, float* segments, uint8_t* flags,
                         float* xy_sample, float* depth, float* rel_disparity, void* stream) {
  if (b <= 0 || v < 2 || h <= 0 || w <= 0 || s <= 0) return PS_ERR_BAD_ARG;
  if (!c2w || !w2c || !k || !k_inv || !near || !far || !origins || !directions || !segments ||
      !flags || !xy_sample || !depth || !rel_disparity)
    return PS_ERR_BAD_ARG;
  Scope sc(G_EPI_GEOM, (hipStream_t)stream);
  launch_epipolar_geometry(b, v, h, w, s, c2w, w2c, k, k_inv, near, far, origins, directions,
                           segments, flags, xy_sample, depth, rel_disparity,
                           (hipStream_t)stream);
  return check_launch();
}

namespace {
bool epi_ok(const PsEpipolarDesc* d) {
  return d && d->b > 0 && d->v >= 2 && d->h > 0 && d->w > 0 && d->s > 0 && d->c > 0 &&
         d->heads > 0 && d->octaves > 0 && d->tail_pad_in >= 0 && d->tail_pad_in <= 3 &&
         d->tail_pad_out >= 0 && d->tail_pad_out <= 3;
}
// the zero-filled padding behind a head's last block must lie inside the head's stride: a pad without a
// custom stride, or one that reaches into the next head's block (a host passing a struct of an older
// layout hands over garbage here), would silently zero live data
bool pad_ok(const PsEpipolarDesc* d, bool with_e) {
  const int used = d->c + 2 * d->octaves + (with_e ? d->v - 1 : 0);
  auto one = [&](int pad, int hs) { return pad == 0 || (hs > 0 && used + pad <= hs); };
  return one(d->tail_pad_in, d->hs_in) && one(d->tail_pad_out, d->hs_out);
}
AttnDims to_dims(const PsEpipolarDesc* d) {
  const int P = 2 * d->octaves, ov = d->v - 1, H = d->heads;
  auto ld = [](int given, int dflt) { return given > 0 ? given : dflt; };
  return AttnDims{d->b, d->v, d->h, d->w, d->s, d->c, d->heads, d->octaves,
                  ld(d->ld_q, H * d->c), ld(d->ld_u, H * P), ld(d->ld_e, H * ov),
                  ld(d->ld_f, H * d->c), ld(d->ld_p, H * P), ld(d->ld_a, H * ov),
                  ld(d->hs_in, d->c), ld(d->hs_in, P), ld(d->hs_in, ov),
                  ld(d->hs_out, d->c), ld(d->hs_out, P), ld(d->hs_out, ov),
                  d->tail_pad_in, d->tail_pad_out};
}
}  // namespace

int ps_epipolar_gather(const PsEpipolarDesc* d, const float* fmap, const float* xy_sample,
                       const uint8_t* flags, float* features, void* stream) {
  if (!epi_ok(d) || !fmap || !xy_sample || !flags || !features) return PS_ERR_BAD_ARG;
  if (int rc = launch_epipolar_gather(to_dims(d), fmap, xy_sample, flags, features,
                                      (hipStream_t)stream))
    return rc;
  return check_launch();
}

int ps_epipolar_attention_forward(const PsEpipolarDesc* d, const float* fmap,
                                  const float* xy_sample, const uint8_t* flags,
                                  const float* rel_disparity, const float* qt, const float* u,
                                  const float* e, float scale, float* fbar, float* pbar,
                                  float* abar, float* attn, void* stream) {
  if (!epi_ok(d) || !fmap || !xy_sample || !flags || !rel_disparity || !qt || !u || !fbar ||
      !pbar || !abar || !attn || !pad_ok(d, e != nullptr))
    return PS_ERR_BAD_ARG;
  Scope sc(G_EPI_FWD, (hipStream_t)stream);
  if (int rc = launch_epipolar_attn_forward(to_dims(d), fmap, xy_sample, flags, rel_disparity,
                                            qt, u, e, scale, fbar, pbar, abar, attn,
                                            (hipStream_t)stream))
    return rc;
  return check_launch();
}

int ps_epipolar_attention_backward(const PsEpipolarDesc* d, const float* fmap,
                                   const float* xy_sample, const uint8_t* flags,
                                   const float* rel_disparity, const float* qt,
                                   const float* attn, const float* fbar, const float* pbar,
                                   const float* abar, const float* dfbar, const float* dpbar,
                                   const float* dabar, float scale, float* dqt, float* du,
                                   float* de, float* ds, float* dfmap, uint32_t* ray_boxes,
                                   void* stream) {
  if (!epi_ok(d) || !fmap || !xy_sample || !flags || !rel_disparity || !qt || !attn || !fbar ||
      !pbar || !dfbar || !dpbar || !dqt || !du || !ds || (dfmap && !ray_boxes) ||
      !pad_ok(d, de != nullptr))
    return PS_ERR_BAD_ARG;
  {
    Scope sc(G_EPI_BWD, (hipStream_t)stream);
    if (int rc = launch_epipolar_attn_backward(to_dims(d), fmap, xy_sample, flags, rel_disparity,
                                               qt, attn, fbar, pbar, abar, dfbar, dpbar, dabar,
                                               scale, dqt, du, de, ds, (hipStream_t)stream))
      return rc;
  }
  if (dfmap) {
    Scope sc(G_EPI_FGRAD, (hipStream_t)stream);
    if (int rc = launch_epipolar_feature_grad(to_dims(d), 1, xy_sample, flags, &qt, &attn, &dfbar,
                                              &ds, dfmap, ray_boxes, nullptr, 3, (hipStream_t)stream))
      return rc;
  }
  return check_launch();
}

int ps_epipolar_feature_grad(const PsEpipolarDesc* d, int32_t n_layers, const float* xy_sample,
                             const uint8_t* flags, const float* const* qt,
                             const float* const* attn, const float* const* dfbar,
                             const float* const* ds, float* dfmap, uint32_t* ray_boxes,
                             void* stream) {
  if (!epi_ok(d) || !xy_sample || !flags || !qt || !attn || !dfbar || !ds || !dfmap || !ray_boxes)
    return PS_ERR_BAD_ARG;
  Scope sc(G_EPI_FGRAD, (hipStream_t)stream);
  if (int rc = launch_epipolar_feature_grad(to_dims(d), n_layers, xy_sample, flags, qt, attn,
                                            dfbar, ds, dfmap, ray_boxes, nullptr, 3,
                                            (hipStream_t)stream))
    return rc;
  return check_launch();
}

size_t ps_epipolar_ray_box_words(const PsEpipolarDesc* d) {
  if (!epi_ok(d)) return 0;
  const size_t tiles = (size_t)((d->w + 3) / 4) * ((d->h + 3) / 4);
  const size_t boxes = (size_t)d->b * d->v * (d->v - 1) * d->h * d->w + 2 * (size_t)d->b * d->v * tiles;
  const size_t binned = epipolar_bin_words(to_dims(d));     // the binned gather's lists and counters
  return boxes > binned ? boxes : binned;
}

size_t ps_epipolar_token_grad_floats(const PsEpipolarDesc* d) {
  if (!epi_ok(d)) return 0;
  return (size_t)d->b * d->v * (d->v - 1) * d->h * d->w * d->s * d->c;
}

int ps_epipolar_feature_grad_two_pass(const PsEpipolarDesc* d, int32_t n_layers,
                                      const float* xy_sample, const uint8_t* flags,
                                      const float* const* qt, const float* const* attn,
                                      const float* const* dfbar, const float* const* ds,
                                      float* dfmap, uint32_t* ray_boxes, float* token_grad,
                                      void* stream) {
  if (!epi_ok(d) || !xy_sample || !flags || !qt || !attn || !dfbar || !ds || !dfmap ||
      !ray_boxes || !token_grad)
    return PS_ERR_BAD_ARG;
  Scope sc(G_EPI_FGRAD, (hipStream_t)stream);
  if (int rc = launch_epipolar_feature_grad(to_dims(d), n_layers, xy_sample, flags, qt, attn,
                                            dfbar, ds, dfmap, ray_boxes, token_grad, 3,
                                            (hipStream_t)stream))
    return rc;
  return check_launch();
}

int ps_epipolar_feature_bins(const PsEpipolarDesc* d, const float* xy_sample, const uint8_t* flags,
                             uint32_t* ray_boxes, void* stream) {
  if (!epi_ok(d) || !xy_sample || !flags || !ray_boxes) return PS_ERR_BAD_ARG;
  Scope sc(G_EPI_FGRAD, (hipStream_t)stream);
  if (int rc = launch_epipolar_feature_grad(to_dims(d), 0, xy_sample, flags, nullptr, nullptr, nullptr,
                                            nullptr, nullptr, ray_boxes, nullptr, 1, (hipStream_t)stream))
    return rc;
  return check_launch();
}

int ps_epipolar_feature_grad_binned(const PsEpipolarDesc* d, int32_t n_layers, const float* xy_sample,
                                    const uint8_t* flags, const float* const* qt,
                                    const float* const* attn, const float* const* dfbar,
                                    const float* const* ds, float* dfmap, const uint32_t* ray_boxes,
                                    float* token_grad, void* stream) {
  if (!epi_ok(d) || !xy_sample || !flags || !qt || !attn || !dfbar || !ds || !dfmap || !ray_boxes ||
      !token_grad)
    return PS_ERR_BAD_ARG;
  Scope sc(G_EPI_FGRAD, (hipStream_t)stream);
  if (int rc = launch_epipolar_feature_grad(to_dims(d), n_layers, xy_sample, flags, qt, attn, dfbar, ds,
                                            dfmap, const_cast<uint32_t*>(ray_boxes), token_grad, 2,
                                            (hipStream_t)stream))
    return rc;
  return check_launch();
}

int ps_gaussian_adapter_views(int32_t n_views, int32_t sh_degree, int32_t image_h, int32_t image_w,
                              const float* extrinsics, const float* intrinsics,
                              const double* wigner_conj, float* views, void* stream) {
  if (n_views <= 0 || sh_degree < 0 || sh_degree > 4 || image_h <= 0 || image_w <= 0 ||
      !extrinsics || !intrinsics || !wigner_conj || !views)
    return PS_ERR_BAD_ARG;
  Scope sc(G_ADAPTER_FWD, (hipStream_t)stream);
  if (int rc = launch_adapter_views(n_views, sh_degree, image_h, image_w, extrinsics, intrinsics,
                                    wigner_conj, views, (hipStream_t)stream))
    return rc;
  return check_launch();
}

int ps_gaussian_adapter_forward(int32_t n_views, int32_t entries_per_view, int32_t spp,
                                int32_t sh_degree, float scale_min, float scale_max, float eps,
                                const float* views, const float* coordinates, const float* depths,
                                const float* raw, float* means, float* covariances,
                                float* harmonics, void* stream) {
  if (n_views <= 0 || entries_per_view <= 0 || spp <= 0 || !views || !coordinates || !depths ||
      !raw || !means || !covariances || !harmonics)
    return PS_ERR_BAD_ARG;
  Scope sc(G_ADAPTER_FWD, (hipStream_t)stream);
  if (int rc = launch_adapter_forward(n_views, entries_per_view, spp, sh_degree, scale_min,
                                      scale_max, eps, views, coordinates, depths, raw, means,
                                      covariances, harmonics, nullptr, (hipStream_t)stream))
    return rc;
  return check_launch();
}

int ps_gaussian_adapter_backward(int32_t n_views, int32_t entries_per_view, int32_t spp,
                                 int32_t sh_degree, float scale_min, float scale_max, float eps,
                                 const float* views, const float* coordinates,
                                 const float* depths, const float* raw, const float* d_means,
                                 const float* d_covariances, const float* d_harmonics,
                                 float* d_raw, float* d_depths, float* d_coordinates,
                                 void* stream) {
  if (n_views <= 0 || entries_per_view <= 0 || spp <= 0 || !views || !coordinates || !depths ||
      !raw || !d_means || !d_covariances || !d_harmonics || !d_raw || !d_depths || !d_coordinates)
    return PS_ERR_BAD_ARG;
  Scope sc(G_ADAPTER_BWD, (hipStream_t)stream);
  if (int rc = launch_adapter_backward(n_views, entries_per_view, spp, sh_degree, scale_min,
                                       scale_max, eps, views, coordinates, depths, raw, d_means,
                                       d_covariances, d_harmonics, d_raw, d_depths, d_coordinates,
                                       nullptr, (hipStream_t)stream))
    return rc;
  return check_launch();
}

namespace {
bool depth_desc_ok(const PsDepthSamplerDesc* d) {
  return d && d->n_views > 0 && d->rays_per_view > 0 && d->buckets > 0 && d->surfaces > 0 &&
         d->spp > 0 && (!d->deterministic || d->spp <= d->buckets) &&
         (long long)d->n_views * d->rays_per_view * d->surfaces < (1ll << 31) - 4096;
}
}  // namespace

int ps_depth_sampler_forward(const PsDepthSamplerDesc* desc, const float* projected,
                             const float* near, const float* far, const float* uniforms,
                             float* depth, float* opacity, int32_t* index, void* stream) {
  if (!depth_desc_ok(desc) || !projected || !near || !far || !depth || !opacity || !index ||
      (!desc->deterministic && !uniforms))
    return PS_ERR_BAD_ARG;
  Scope sc(G_DEPTH_FWD, (hipStream_t)stream);
  if (int rc = launch_depth_sampler_forward(*desc, projected, near, far, uniforms, depth, opacity,
                                            index, (hipStream_t)stream))
    return rc;
  return check_launch();
}

int ps_depth_sampler_backward(const PsDepthSamplerDesc* desc, const float* projected,
                              const float* near, const float* far, const int32_t* index,
                              const float* d_depth, const float* d_opacity, float* d_projected,
                              void* stream) {
  if (!depth_desc_ok(desc) || !projected || !near || !far || !index || !d_depth || !d_opacity ||
      !d_projected)
    return PS_ERR_BAD_ARG;
  Scope sc(G_DEPTH_BWD, (hipStream_t)stream);
  if (int rc = launch_depth_sampler_backward(*desc, projected, near, far, index, d_depth,
                                             d_opacity, d_projected, (hipStream_t)stream))
    return rc;
  return check_launch();
}

size_t ps_image_mse_workspace_bytes(int32_t n_images, int32_t elems) {
  if (n_images <= 0 || elems <= 0) return 0;
  return image_mse_workspace_bytes(n_images, elems);
}

int ps_image_mse(int32_t n_images, int32_t elems, const float* pred, const float* target,
                 float grad_scale, float* grad, float* sse, float* sse_clipped, void* workspace,
                 size_t workspace_bytes, void* stream) {
  if (n_images <= 0 || elems <= 0 || n_images > 65535 || !pred || !target || !sse || !workspace)
    return PS_ERR_BAD_ARG;
  if (workspace_bytes < image_mse_workspace_bytes(n_images, elems)) return PS_ERR_WORKSPACE;
  Scope sc(G_LOSS, (hipStream_t)stream);
  if (int rc = launch_image_mse(n_images, elems, pred, target, grad_scale, grad, sse, sse_clipped,
                                workspace, (hipStream_t)stream))
    return rc;
  return check_launch();
}

namespace {
bool depth_loss_desc_ok(const PsDepthLossDesc* d) {
  if (!d || d->n_images <= 0 || d->n_images > 65535 || d->channels < 0) return false;
  const int o = d->use_second_derivative ? 2 : 1;
  return d->height > o && d->width > o && (long long)d->height * d->width < (1ll << 31);
}
}  // namespace

size_t ps_depth_smoothness_workspace_bytes(const PsDepthLossDesc* desc) {
  return depth_loss_desc_ok(desc) ? depth_smoothness_workspace_bytes(*desc) : 0;
}

int ps_depth_smoothness_forward(const PsDepthLossDesc* desc, const float* depth, const float* near,
                                const float* far, const float* target_image, float* loss,
                                void* workspace, size_t workspace_bytes, void* stream) {
  if (!depth_loss_desc_ok(desc) || !depth || !near || !far || !loss || !workspace ||
      (desc->use_sigma && (!target_image || desc->channels <= 0)))
    return PS_ERR_BAD_ARG;
  if (workspace_bytes < depth_smoothness_workspace_bytes(*desc)) return PS_ERR_WORKSPACE;
  Scope sc(G_LOSS, (hipStream_t)stream);
  if (int rc = launch_depth_smoothness_forward(*desc, depth, near, far, target_image, loss,
                                               workspace, (hipStream_t)stream))
    return rc;
  return check_launch();
}

int ps_depth_smoothness_backward(const PsDepthLossDesc* desc, const float* depth,
                                 const float* near, const float* far, const float* target_image,
                                 const float* d_loss, float* d_depth, void* stream) {
  if (!depth_loss_desc_ok(desc) || !depth || !near || !far || !d_loss || !d_depth ||
      (desc->use_sigma && (!target_image || desc->channels <= 0)))
    return PS_ERR_BAD_ARG;
  Scope sc(G_LOSS, (hipStream_t)stream);
  if (int rc = launch_depth_smoothness_backward(*desc, depth, near, far, target_image, d_loss,
                                                d_depth, (hipStream_t)stream))
    return rc;
  return check_launch();
}

int ps_gaussian_head_forward(int32_t n_views, int32_t image_h, int32_t image_w, int32_t surfaces,
                             int32_t spp, int32_t sh_degree, float scale_min, float scale_max,
                             float eps, const float* views, const float* depths,
                             const float* head_rows, float* means, float* covariances,
                             float* harmonics, void* stream) {
  if (n_views <= 0 || image_h <= 0 || image_w <= 0 || surfaces <= 0 || spp <= 0 || !views ||
      !depths || !head_rows || !means || !covariances || !harmonics)
    return PS_ERR_BAD_ARG;
  const int head[3] = {surfaces, image_w, image_h};
  Scope sc(G_ADAPTER_FWD, (hipStream_t)stream);
  if (int rc = launch_adapter_forward(n_views, image_h * image_w * surfaces, spp, sh_degree,
                                      scale_min, scale_max, eps, views, nullptr, depths,
                                      head_rows, means, covariances, harmonics, head,
                                      (hipStream_t)stream))
    return rc;
  return check_launch();
}

int ps_gaussian_head_backward(int32_t n_views, int32_t image_h, int32_t image_w, int32_t surfaces,
                              int32_t spp, int32_t sh_degree, float scale_min, float scale_max,
                              float eps, const float* views, const float* depths,
                              const float* head_rows, const float* d_means,
                              const float* d_covariances, const float* d_harmonics,
                              float* d_head_rows, float* d_depths, void* stream) {
  if (n_views <= 0 || image_h <= 0 || image_w <= 0 || surfaces <= 0 || spp <= 0 || !views ||
      !depths || !head_rows || !d_means || !d_covariances || !d_harmonics || !d_head_rows ||
      !d_depths)
    return PS_ERR_BAD_ARG;
  const int head[3] = {surfaces, image_w, image_h};
  Scope sc(G_ADAPTER_BWD, (hipStream_t)stream);
  if (int rc = launch_adapter_backward(n_views, image_h * image_w * surfaces, spp, sh_degree,
                                       scale_min, scale_max, eps, views, nullptr, depths,
                                       head_rows, d_means, d_covariances, d_harmonics,
                                       d_head_rows, d_depths, nullptr, head, (hipStream_t)stream))
    return rc;
  return check_launch();
}

namespace {
bool fold_desc_ok(const PsFoldDesc* d) {
  return d && d->heads > 0 && d->head_dim > 0 && d->kv_dim > 0 && d->q_dim > 0 && d->out_dim > 0 &&
         d->octaves >= 0 && d->other_views >= 0;
}
}  // namespace

size_t ps_fold_scratch_floats(const PsFoldDesc* desc) {
  return fold_desc_ok(desc) ? fold_scratch_floats(*desc) : 0;
}

int ps_fold_attention_weights(const PsFoldDesc* desc, const float* w_q, const float* w_kv,
                              const float* w_out, const float* b_out, const float* depth_w,
                              const float* depth_b, const float* view_emb, float* w_in,
                              float* w_o_t, float* bias, float* scratch, void* stream) {
  if (!fold_desc_ok(desc) || !w_q || !w_kv || !w_out || !depth_w || !depth_b || !w_in || !w_o_t ||
      !bias || !scratch || (desc->other_views > 0 && !view_emb))
    return PS_ERR_BAD_ARG;
  if (int rc = launch_fold_forward(*desc, w_q, w_kv, w_out, b_out, depth_w, depth_b, view_emb,
                                   w_in, w_o_t, bias, scratch, (hipStream_t)stream))
    return rc;
  return check_launch();
}

int ps_fold_attention_weights_backward(const PsFoldDesc* desc, const float* w_q, const float* w_kv,
                                       const float* w_out, const float* b_out,
                                       const float* depth_w, const float* depth_b,
                                       const float* view_emb, const float* scratch,
                                       const float* d_w_in, const float* d_w_o_t,
                                       const float* d_bias, float* back_scratch, float* g_w_q,
                                       float* g_w_kv, float* g_w_out, float* g_b_out,
                                       float* g_depth_w, float* g_depth_b, float* g_view_emb,
                                       void* stream) {
  if (!fold_desc_ok(desc) || !w_q || !w_kv || !w_out || !depth_w || !depth_b || !scratch ||
      !d_w_in || !d_w_o_t || !d_bias || !back_scratch || !g_w_q || !g_w_kv || !g_w_out ||
      !g_depth_w || !g_depth_b || (desc->other_views > 0 && (!view_emb || !g_view_emb)))
    return PS_ERR_BAD_ARG;
  if (int rc = launch_fold_backward(*desc, w_q, w_kv, w_out, b_out, depth_w, depth_b, view_emb,
                                    scratch, d_w_in, d_w_o_t, d_bias, back_scratch, g_w_q, g_w_kv,
                                    g_w_out, g_b_out, g_depth_w, g_depth_b, g_view_emb,
                                    (hipStream_t)stream))
    return rc;
  return check_launch();
}

size_t ps_layer_norm_workspace_floats(int32_t rows, int32_t dim) {
  return rows > 0 && dim > 0 ? layer_norm_workspace_floats(rows, dim) : 0;
}

int ps_layer_norm_forward(int32_t rows, int32_t dim, float eps, const float* x,
                          const float* gamma, const float* beta, float* y, float* mean,
                          float* rstd, void* stream) {
  if (rows <= 0 || dim <= 0 || !x || !gamma || !beta || !y || !mean || !rstd) return PS_ERR_BAD_ARG;
  if (int rc = launch_layer_norm_forward(rows, dim, eps, x, gamma, beta, y, mean, rstd,
                                         (hipStream_t)stream))
    return rc;
  return check_launch();
}

int ps_layer_norm_backward(int32_t rows, int32_t dim, const float* x, const float* gamma,
                           const float* mean, const float* rstd, const float* dy,
                           const float* d_residual, float* dx, float* d_gamma, float* d_beta,
                           float* workspace, void* stream) {
  if (rows <= 0 || dim <= 0 || !x || !gamma || !mean || !rstd || !dy || !dx || !d_gamma ||
      !d_beta || !workspace)
    return PS_ERR_BAD_ARG;
  if (int rc = launch_layer_norm_backward(rows, dim, x, gamma, mean, rstd, dy, d_residual, dx, d_gamma, d_beta,
                                          workspace, (hipStream_t)stream))
    return rc;
  return check_launch();
}

int ps_invert_cameras(int32_t n, const float* c2w, const float* k, float* w2c, float* k_inv,
                      void* stream) {
  if (n <= 0 || !c2w || !k || !w2c || !k_inv) return PS_ERR_BAD_ARG;
  launch_camera_inverse(n, c2w, k, w2c, k_inv, (hipStream_t)stream);
  return check_launch();
}

size_t ps_gemm_tn_workspace_bytes(int32_t m, int32_t n, int32_t k) {
  return (m > 0 && n > 0 && k > 0) ? gemm_tn_workspace_bytes(m, n, k) : 0;
}

int ps_gemm_tn_colsum_f32(int32_t m, int32_t n, int32_t k, const float* a, int32_t lda,
                          const float* b, int32_t ldb, float* c, float* colsum_a, void* workspace,
                          size_t workspace_bytes, void* stream) {
  if (m <= 0 || n <= 0 || k <= 0 || !a || !b || !c || !workspace || lda < m || ldb < n)
    return PS_ERR_BAD_ARG;
  if (workspace_bytes < gemm_tn_workspace_bytes(m, n, k)) return PS_ERR_WORKSPACE;
  Scope sc(G_GEMM_TN, (hipStream_t)stream);
  if (int rc = launch_gemm_tn(m, n, k, a, lda, b, ldb, c, colsum_a, (float*)workspace,
                              (hipStream_t)stream))
    return rc;
  return check_launch();
}

int ps_gemm_tn_f32(int32_t m, int32_t n, int32_t k, const float* a, int32_t lda, const float* b,
                   int32_t ldb, float* c, void* workspace, size_t workspace_bytes, void* stream) {
  return ps_gemm_tn_colsum_f32(m, n, k, a, lda, b, ldb, c, nullptr, workspace, workspace_bytes, stream);
}

int ps_raster_check(const PsRasterDesc* d, const void* state, size_t state_bytes,
                    uint64_t* num_rendered, void* stream) {
  if (!desc_ok(d) || !state) return PS_ERR_BAD_ARG;
  const PsRasterStateLayout L = make_state_layout(*d);
  if (state_bytes < L.total) return PS_ERR_WORKSPACE;
  uint32_t host[2] = {0, 0};
  hipStream_t st = (hipStream_t)stream;
  if (hipMemcpyAsync(host, (const char*)state + L.num_rendered, 8, hipMemcpyDeviceToHost, st) !=
          hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess)
    return PS_ERR_LAUNCH;
  if (num_rendered) *num_rendered = host[0];
  return host[1] ? PS_ERR_CAPACITY : PS_OK;
}

int ps_profile_enable(int on) { g_profile_on.store(on & 3); return PS_OK; }
int ps_roctx_available(void) { return roctx().push != nullptr; }
int ps_profile_group_count(void) { return G_COUNT; }
const char* ps_profile_group_name(int g) { return (g >= 0 && g < G_COUNT) ? kGroupNames[g] : ""; }
int ps_profile_collect(double* total_ms, int64_t* launches) {
  std::vector<Pending> p;
  {
    std::lock_guard<std::mutex> lk(g_profile_mu);
    p.swap(g_pending);
  }
  int rc = PS_OK;
  for (auto& e : p) {
    float ms = 0.f;
    if (hipEventSynchronize(e.b) != hipSuccess || hipEventElapsedTime(&ms, e.a, e.b) != hipSuccess)
      rc = PS_ERR_LAUNCH;
    else if (total_ms && launches) { total_ms[e.group] += ms; launches[e.group] += 1; }
    hipEventDestroy(e.a); hipEventDestroy(e.b);
  }
  return rc;
}

int ps_abi_version(void) { return PS_ABI_VERSION; }

const char* ps_status_string(int status) {
  switch (status) {
    case PS_OK: return "ok";
    case PS_ERR_BAD_ARG: return "bad argument (null pointer or inconsistent descriptor)";
    case PS_ERR_WORKSPACE: return "state/temp buffer too small";
    case PS_ERR_LAUNCH: return "HIP launch failed";
    case PS_ERR_UNSUPPORTED: return "unsupported configuration";
    case PS_ERR_CAPACITY: return "tile point list overflowed (raise PsRasterDesc.list_factor)";
    default: return "unknown status";
  }
}

// PS_BUILD_HASHES: "<file>:<hash12> ..." of every translation unit (source + headers + flags), put
// there by pixelsplat_amd/build.py / tools/build_variant.sh.  Counter summaries under profiles/ carry
// the same string; bench.py pairs a live kernel time with committed counters only when the
// kernel's translation unit has the same hash in both.
#ifndef PS_BUILD_HASHES
#define PS_BUILD_HASHES "unhashed"
#endif
const char* ps_build_info(void) { return "pixelsplat_hip gfx950 " __DATE__ " " __TIME__ " | " PS_BUILD_HASHES; }

}  // extern "C"
