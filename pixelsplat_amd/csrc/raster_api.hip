// extern "C" surface of libpixelsplat_hip.so -- see include/pixelsplat_hip.h.
// No torch types, no exceptions, no device synchronisation, no hidden state.
#include "raster_common.h"

#include <atomic>
#include <mutex>
#include <vector>

using namespace ps;

namespace {

// ---- optional per-kernel-group timing (bench.py) ----------------------------------------
enum Group { G_PRE_FWD = 0, G_SORT, G_TILES_FWD, G_TILES_BWD, G_PRE_BWD, G_MEMSET, G_COUNT };
const char* kGroupNames[G_COUNT] = {"preprocess_forward", "depth_sort", "tiles_forward",
                                    "tiles_backward", "preprocess_backward", "memset"};
std::atomic<int> g_profile_on{0};
std::mutex g_profile_mu;
struct Pending { hipEvent_t a, b; int group; };
std::vector<Pending> g_pending;

struct Scope {
  hipEvent_t a = nullptr, b = nullptr; int group; hipStream_t st; bool on;
  Scope(int g, hipStream_t s) : group(g), st(s), on(g_profile_on.load() != 0) {
    if (!on) return;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { on = false; return; }
    hipEventRecord(a, st);
  }
  ~Scope() {
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
  d->reserved = 0;
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
int ps_raster_state_layout(const PsRasterDesc* d, PsRasterStateLayout* out) {
  if (!desc_ok(d) || !out) return PS_ERR_BAD_ARG;
  *out = make_state_layout(*d);
  return PS_OK;
}

int ps_raster_forward(const PsRasterDesc* d, const float* means, const float* cov,
                      const float* sh, const float* colors, const float* opacity,
                      const float* view_params, float* out_color, int32_t* out_radii,
                      void* state, size_t state_bytes, void* temp, size_t temp_bytes,
                      void* stream) {
  if (!desc_ok(d) || !means || !cov || !opacity || !view_params || !out_color || !out_radii ||
      !state || !temp)
    return PS_ERR_BAD_ARG;
  if ((sh == nullptr) == (colors == nullptr)) return PS_ERR_BAD_ARG;  // exactly one
  if (sh) {
    if (d->sh_degree > 4) return PS_ERR_UNSUPPORTED;
    if (d->sh_coeffs < (d->sh_degree + 1) * (d->sh_degree + 1)) return PS_ERR_BAD_ARG;
  }
  const Dims m = make_dims(*d);
  if (m.gx > 65535 || m.gy > 65535) return PS_ERR_UNSUPPORTED;
  const PsRasterStateLayout L = make_state_layout(*d);
  const TempLayout T = make_temp_layout(*d);
  if (state_bytes < L.total || temp_bytes < T.total) return PS_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  char* sb = (char*)state; char* tb = (char*)temp;
  float* records = (float*)(sb + L.records);
  uint2* rects = (uint2*)(sb + L.rects);
  uint32_t* sorted_idx = (uint32_t*)(sb + L.sorted_idx);
  uint2* sorted_rect = (uint2*)(sb + L.sorted_rect);
  uint32_t* n_vis = (uint32_t*)(sb + L.n_vis);
  float* final_T = (float*)(sb + L.final_T);
  uint32_t* n_contrib = (uint32_t*)(sb + L.n_contrib);
  uint32_t* tile_end = (uint32_t*)(sb + L.tile_end);
  uint32_t* keys_a = (uint32_t*)(tb + T.keys_a);
  uint32_t* keys_b = (uint32_t*)(tb + T.keys_b);
  uint32_t* vals_a = (uint32_t*)(tb + T.vals_a);
  uint32_t* vals_b = (uint32_t*)(tb + T.vals_b);
  uint32_t* block_hist = (uint32_t*)(tb + T.block_hist);

  {
    Scope sc(G_MEMSET, st);
    if (hipMemsetAsync(n_vis, 0, (size_t)m.V * 4, st) != hipSuccess) return PS_ERR_LAUNCH;
  }
  {
    Scope sc(G_PRE_FWD, st);
    launch_preprocess_forward(*d, means, cov, sh, colors, opacity, view_params, records, keys_a,
                              rects, out_radii, n_vis, st);
  }
  {
    Scope sc(G_SORT, st);
    launch_sort(*d, keys_a, keys_b, vals_a, vals_b, block_hist, sorted_idx, rects, sorted_rect,
                n_vis, st);
  }
  {
    Scope sc(G_TILES_FWD, st);
    launch_tiles_forward(*d, records, sorted_idx, sorted_rect, n_vis, view_params, out_color,
                         final_T, n_contrib, tile_end, st);
  }
  return check_launch();
}

int ps_raster_backward(const PsRasterDesc* d, const float* means, const float* cov,
                       const float* sh, const float* colors, const float* opacity,
                       const float* view_params, const int32_t* radii, const float* dL_dcolor,
                       const void* state, size_t state_bytes, void* temp, size_t temp_bytes,
                       float* dL_dmeans, float* dL_dcov, float* dL_dsh, float* dL_dcolors,
                       float* dL_dopacity, float* dL_dmeans2D, void* stream) {
  (void)opacity; (void)colors;
  if (!desc_ok(d) || !means || !cov || !view_params || !radii || !dL_dcolor || !state || !temp ||
      !dL_dmeans || !dL_dcov || !dL_dopacity)
    return PS_ERR_BAD_ARG;
  if (sh && !dL_dsh) return PS_ERR_BAD_ARG;
  if (!sh && !dL_dcolors) return PS_ERR_BAD_ARG;
  const Dims m = make_dims(*d);
  const PsRasterStateLayout L = make_state_layout(*d);
  const TempLayout T = make_temp_layout(*d);
  if (state_bytes < L.total || temp_bytes < T.total) return PS_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const char* sb = (const char*)state; char* tb = (char*)temp;
  const float* records = (const float*)(sb + L.records);
  const uint32_t* sorted_idx = (const uint32_t*)(sb + L.sorted_idx);
  const uint2* sorted_rect = (const uint2*)(sb + L.sorted_rect);
  const float* final_T = (const float*)(sb + L.final_T);
  const uint32_t* n_contrib = (const uint32_t*)(sb + L.n_contrib);
  const uint32_t* tile_end = (const uint32_t*)(sb + L.tile_end);
  float* grad2d = (float*)(tb + T.grad2d);
  {
    Scope sc(G_MEMSET, st);
    if (hipMemsetAsync(grad2d, 0, m.N * kGradFloats * 4, st) != hipSuccess) return PS_ERR_LAUNCH;
  }
  {
    Scope sc(G_TILES_BWD, st);
    launch_tiles_backward(*d, records, sorted_idx, sorted_rect, view_params, final_T, n_contrib,
                          tile_end, dL_dcolor, grad2d, st);
  }
  {
    Scope sc(G_PRE_BWD, st);
    launch_preprocess_backward(*d, means, cov, sh, view_params, records, radii, grad2d,
                               dL_dmeans, dL_dcov, dL_dsh, dL_dcolors, dL_dopacity, dL_dmeans2D,
                               st);
  }
  return check_launch();
}

int ps_raster_export_bins(const PsRasterDesc* d, const void* state, size_t state_bytes,
                          uint32_t* tile_counts, const uint32_t* tile_offsets,
                          uint32_t* point_list, size_t capacity, void* stream) {
  if (!desc_ok(d) || !state) return PS_ERR_BAD_ARG;
  const PsRasterStateLayout L = make_state_layout(*d);
  if (state_bytes < L.total) return PS_ERR_WORKSPACE;
  const char* sb = (const char*)state;
  launch_export_bins(*d, (const uint32_t*)(sb + L.sorted_idx), (const uint2*)(sb + L.sorted_rect),
                     (const uint32_t*)(sb + L.n_vis), tile_counts, tile_offsets, point_list,
                     capacity, (hipStream_t)stream);
  return check_launch();
}

int ps_profile_enable(int on) { g_profile_on.store(on ? 1 : 0); return PS_OK; }
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

const char* ps_status_string(int status) {
  switch (status) {
    case PS_OK: return "ok";
    case PS_ERR_BAD_ARG: return "bad argument (null pointer or inconsistent descriptor)";
    case PS_ERR_WORKSPACE: return "state/temp buffer too small";
    case PS_ERR_LAUNCH: return "HIP launch failed";
    case PS_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown status";
  }
}

const char* ps_build_info(void) { return "pixelsplat_hip gfx950 " __DATE__ " " __TIME__; }

}  // extern "C"
