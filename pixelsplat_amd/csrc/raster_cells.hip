// Tile forward on 4x4-pixel CELLS with 16-lane rows (round 6).
//
// The 8x8-quadrant kernels (raster_tiles.hip) evaluate 64 lanes per (entry, quadrant) pair for ~16
// pixels that take the entry: lane efficiency 0.26 at BASELINE configs[1].  Here a tile is sixteen 4x4
// cells and a wave64 is FOUR independent 16-lane rows, each walking the queue of ITS cell -- the list
// entries whose alpha >= 1/255 ellipse can reach that cell -- so one wave instruction blends four
// different (entry, cell) pairs: 39 M pairs / 4 = 10 M wave steps instead of 17 M quadrant evaluations
// (oracle walk, oracle/raster_ref.py::box_evaluations), lane efficiency 0.46 (0.44 with the rows' queue
// lengths unequal).
//
// One wave per tile QUADRANT (the four waves of a block = one tile), one pixel per lane, row r = cell r of
// the quadrant.  The wave runs a two-stage pipeline over the tile's bin, all of it in LDS:
//   scan    64 list entries at a time, one per lane: the pair's 16-byte cell window (cell_window.h; the
//           preprocess wrote it into the pair's 64-byte record line, which the blend's record gather then finds fetched) -> the 4-bit mask of the quadrant's cells -> four wave64 ballots append
//           (Gaussian id, list position) to the four rows' RINGS in list order (rank by mbcnt; the ring
//           cursors are wave-uniform scalars).  The list indices run two batches ahead of the scan and the
//           windows one.
//   blend   16 items per row at a time: every lane takes one item of its row's ring and gathers its
//           record (in flight while the previous chunk is blended), the row's 16 records are staged in
//           LDS and the 16 blend steps read them back row-uniformly.  A row whose ring runs dry blends
//           null items; the wave steps as often as its longest row needs.
// No refine of whole records, no quadrant branches, no ring compaction in the walk, and no queue ever
// touches HBM (the first build of this kernel read per-cell queues a separate kernel had written to HBM:
// 0.43 ms for the blend + 0.33 ms for the queue kernel, one latency-bound wave per tile;
// profiles/r6_forward_cells_ab.txt).
//
// Per-pixel arithmetic is that of raster_tiles.hip's forward, instruction for instruction: a culled
// entry is an entry every pixel of the cell would have skipped (alpha < 1/255), so images are unchanged.
// Replaces renderCUDA (forward) of the external rasterizer (call site
// /root/reference/src/model/decoder/cuda_splatting.py:117-124).
#include "raster_common.h"
#include "cell_window.h"

#include <cstdlib>

namespace ps {

namespace {

constexpr float kLog2eC = 1.4426950408889634f;
constexpr uint32_t kNullItem = 0xFFFFFFFFu;
constexpr int kRing = 128;            // items per row ring: a scan batch (up to 64 items) fits behind 64 unread ones

typedef float f32x2c __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float fast_exp2c(float x) { return __builtin_amdgcn_exp2f(x); }

// [row][item] {gx,gy,A,B} {C,opacity,r,g} {b,list position,-,-}.  Two float4 of padding per row: the four rows
// of a wave read four different items with ONE instruction -- with a row stride of 768 bytes (0 mod 32 banks)
// that was a four-way bank conflict on every read (SQ_LDS_BANK_CONFLICT = 48 % of the LDS-active cycles of the
// first build); 800 bytes puts the rows 8 banks apart.
struct RowsLds {
  uint2 ring[4][kRing];               // (Gaussian id, 1-based position in the tile's list)
  float4 rec[4][16 * 3 + 2];
};

}  // namespace

// WAVES: waves per SIMD the register allocation aims at; UNROLL: items per trip of the blend loop
template <int WAVES, int UNROLL>
__global__ void __launch_bounds__(256, WAVES)
tiles_forward_rows_kernel(PsRasterDesc d, const float* __restrict__ records,
                          const uint4* __restrict__ cell_windows,
                          const uint32_t* __restrict__ tile_order,
                          const uint32_t* __restrict__ tile_ranges,
                          const uint32_t* __restrict__ point_list, uint32_t capacity,
                          const float* __restrict__ view_params, float* __restrict__ out_color,
                          float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                          float4* __restrict__ checkpoint, uint32_t* __restrict__ tile_end) {
  __shared__ RowsLds lds_all[4];
  const int G = d.n_gaussians, H = d.height, W = d.width;
  const int gxn = (W + kTile - 1) / kTile, gyn = (H + kTile - 1) / kTile;
  const int tiles = gxn * gyn;
  const int V = d.n_scenes * d.views_per_scene;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slot_global = blockIdx.x * 4 + w;
  if (slot_global >= V * tiles * 4) return;
  // longest lists first; the four quadrants of a tile are the four waves of one block
  const int tile_global = (int)tile_order[slot_global >> 2];
  const int quad = slot_global & 3;
  RowsLds& lds = lds_all[w];
  const int v = tile_global / tiles, t = tile_global % tiles;
  const int tx = t % gxn, ty = t / gxn;
  const float* recs = records + (size_t)v * G * kRecFloats;
  const uint4* wins = cell_windows + (size_t)v * G * (kRecFloats / 4);     // (one per record line)
  uint32_t l_start = tile_ranges[2 * (size_t)tile_global];
  uint32_t l_count = tile_ranges[2 * (size_t)tile_global + 1];
  if (l_start > capacity) l_start = capacity;                       // overflowed step: stay in
  if (l_count > capacity - l_start) l_count = capacity - l_start;   // bounds (flag is raised)
  l_start = __builtin_amdgcn_readfirstlane(l_start);
  l_count = __builtin_amdgcn_readfirstlane(l_count);
  const uint32_t* list = point_list + l_start;
  // the backward walks a long list as two tasks (raster_common.h: split_point): the wave leaves its pixels'
  // state at the split point.  The scan stops there until the rings have run dry, so the state is exactly
  // "after entry ck_at" for every row at once.
  const uint32_t ck_at = split_point(l_count);

  // row r of the wave = cell (2 (quad & 1) + (r & 1), 2 (quad >> 1) + (r >> 1)) of the tile
  const int row = lane >> 4, it = lane & 15;
  const int cx = 2 * (quad & 1) + (row & 1), cy = 2 * (quad >> 1) + (row >> 1);
  const int qcx = 4 * tx + 2 * (quad & 1), qcy = 4 * ty + 2 * (quad >> 1);    // the quadrant's first cell
  const int px = tx * kTile + cx * 4 + (lane & 3), py = ty * kTile + cy * 4 + ((lane >> 2) & 3);
  const float pxf = (float)px, pyf = (float)py;
  const bool live = px < W && py < H;

  const float alpha_max = d.alpha_max, alpha_min = d.alpha_min, t_min = d.t_min;
  float Ts = live ? 1.f : -1.f;       // T while live, -T once stopped (raster_tiles.hip)
  float C0 = 0.f, C1 = 0.f, C2 = 0.f;
  uint32_t last = 0;                  // 1-based list position of the pixel's last contributor
  float4 ck = make_float4(0.f, 0.f, 0.f, 0.f);
  bool ck_written = false;            // (wave-uniform)

  // ---- scan: list -> rings ----
  uint32_t head[4] = {0u, 0u, 0u, 0u}, tail[4] = {0u, 0u, 0u, 0u};   // items taken / appended (wave-uniform)
  auto load_id = [&](uint32_t first) -> uint32_t {
    const uint32_t e = first + (uint32_t)lane;
    return list[e < l_count ? e : 0u];
  };
  // list indices of the next three batches, windows of the next two: at the start of a wave, where no blend
  // covers a latency yet, the first two scans find their windows loaded
  uint32_t id_a = 0, id_b = 0, id_c = 0;
  uint4 win_a = make_uint4(0u, 0u, 0u, 0u), win_b = win_a;
  if (l_count > 0) {
    id_a = load_id(0u); id_b = load_id(64u); id_c = load_id(128u);
    win_a = wins[(size_t)id_a * (kRecFloats / 4)]; win_b = wins[(size_t)id_b * (kRecFloats / 4)];
  }
  uint32_t scan = 0;                  // next list entry to scan (a multiple of 64)
  auto scan_batch = [&]() {
    const uint32_t id = id_a; const uint4 win = win_a;
    id_a = id_b; id_b = id_c; win_a = win_b;
    win_b = wins[(size_t)id_b * (kRecFloats / 4)];     // the windows of the batch after the next
    id_c = load_id(scan + 192u);                       // the list indices of the one after that
    // (large footprints -- over 8 cells -- are rare: their range test runs only in a batch that holds one)
    uint32_t mq = __builtin_amdgcn_ballot_w64(win.w != 0u) != 0ull ? quad_cell_mask<true>(win, qcx, qcy)
                                                                    : quad_cell_mask<false>(win, qcx, qcy);
    mq = scan + (uint32_t)lane < l_count ? mq : 0u;                 // bit r = row r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool hit = (mq >> r) & 1u;
      const uint64_t b = __ballot(hit);
      if (hit) {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
        lds.ring[r][(tail[r] + rank) & (kRing - 1)] = make_uint2(id, scan + (uint32_t)lane + 1u);
      }
      tail[r] = __builtin_amdgcn_readfirstlane(tail[r] + (uint32_t)__popcll(b));
    }
    scan += 64u;
  };
  auto occupancy = [&]() -> uint32_t {
    const uint32_t a = tail[0] - head[0], b = tail[1] - head[1], c = tail[2] - head[2], e = tail[3] - head[3];
    const uint32_t ab = a > b ? a : b, ce = c > e ? c : e;
    return __builtin_amdgcn_readfirstlane(ab > ce ? ab : ce);
  };

  // ---- blend: rings -> pixels ----
  uint2 item = make_uint2(kNullItem, 0u);   // the pending chunk: this lane's item and its record (in flight)
  float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0, g2 = g0;
  bool pend = false;                  // (wave-uniform)
  auto take = [&]() {
    uint32_t n[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { const uint32_t a = tail[r] - head[r]; n[r] = a < 16u ? a : 16u; }
    const uint32_t my_n = row < 2 ? (row == 0 ? n[0] : n[1]) : (row == 2 ? n[2] : n[3]);
    const uint32_t my_h = row < 2 ? (row == 0 ? head[0] : head[1]) : (row == 2 ? head[2] : head[3]);
    item = lds.ring[row][(my_h + (uint32_t)it) & (kRing - 1)];
    if ((uint32_t)it >= my_n) item = make_uint2(kNullItem, 0u);
#pragma unroll
    for (int r = 0; r < 4; ++r) head[r] = __builtin_amdgcn_readfirstlane(head[r] + n[r]);
    const float4* rp = reinterpret_cast<const float4*>(recs + (size_t)(item.x == kNullItem ? 0u : item.x) * kRecFloats);
    g0 = rp[0]; g1 = rp[1]; g2 = rp[2];
  };
  auto stage = [&]() {
    const bool null = item.x == kNullItem;
    const float A = -0.5f * kLog2eC * g0.z, B = -kLog2eC * g0.w, Cq = -0.5f * kLog2eC * g1.x;
    float4* st = &lds.rec[row][it * 3];
    // a null item: opacity -1 -> alpha < alpha_min for every pixel (record 0, which its lane gathered, may never
    // have been written)
    st[0] = null ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(g0.x, g0.y, A, B);
    st[1] = null ? make_float4(0.f, -1.f, 0.f, 0.f) : make_float4(Cq, g1.y, g2.x, g2.y);
    st[2] = make_float4(null ? 0.f : g2.z, __uint_as_float(item.y), 0.f, 0.f);
  };
  auto blend16 = [&]() {
#pragma unroll UNROLL
    for (int j = 0; j < 16; ++j) {
      const float4 q0 = lds.rec[row][j * 3], q1 = lds.rec[row][j * 3 + 1];
      const float2 q2 = *reinterpret_cast<const float2*>(&lds.rec[row][j * 3 + 2]);
      const f32x2c dd = f32x2c{q0.x, q0.y} - f32x2c{pxf, pyf};           // (dx, dy)
      const f32x2c bc = f32x2c{q0.w, q1.x} * f32x2c{dd.y, dd.y};         // (B dy, C dy)
      const float pw = fmaf(dd.x, fmaf(q0.z, dd.x, bc.x), dd.y * bc.y);  // power * log2(e)
      const float alpha = fminf(alpha_max, q1.y * fast_exp2c(pw));
      const bool ok = (pw <= 0.f) & (alpha >= alpha_min);
      const float ale = ok ? alpha : 0.f;          // 0 => every update below is a no-op
      float Tp;
      asm("v_max_f32 %0, 0, %1" : "=v"(Tp) : "v"(Ts));
      const f32x2c tw = f32x2c{Tp, Tp} * f32x2c{1.f - ale, ale};         // (T (1 - a), T a)
      const bool stop = tw.x < t_min;
      const float wgt = stop ? 0.f : tw.y;
      Ts = stop ? -fabsf(Ts) : tw.x;
      C0 = fmaf(q1.z, wgt, C0);
      C1 = fmaf(q1.w, wgt, C1);
      C2 = fmaf(q2.x, wgt, C2);
      last = (ok & !stop) ? __float_as_uint(q2.y) : last;
    }
  };

  bool ck_done = ck_at == 0u;         // (wave-uniform) nothing to leave, or left already
  for (;;) {
    const uint32_t limit = ck_done ? l_count : ck_at;
    if (scan < limit && occupancy() <= (uint32_t)(kRing - 64)) scan_batch();
    const bool src_done = scan >= limit;
    const uint32_t occ = occupancy();
    const bool take_ok = occ >= 16u || (src_done && occ != 0u);
    if (pend) stage();
    wave_lds_sync();                  // the scan's ring items and the staged records are visible
    const bool had = pend;
    pend = false;
    if (take_ok) { take(); pend = true; }        // (its gathers fly while the staged chunk is blended)
    if (had) {
      blend16();
      wave_lds_sync();
      if (!__any(Ts > 0.f)) break;               // every pixel of the quadrant has stopped
    }
    if (!pend && src_done) {                     // rings dry, nothing in flight
      if (ck_done) break;
      ck = make_float4(Ts, C0, C1, C2);          // the rows' state at the tile's split point
      ck_written = true;
      ck_done = true;
    }
  }
  const float T = fabsf(Ts);

  // epilogue
  const float* bg = view_params + (size_t)v * PS_VIEW_STRIDE + PS_VIEW_BG;
  const float bg0 = bg[0], bg1 = bg[1], bg2 = bg[2];
  const size_t P = (size_t)H * W;
  if (live) {
    const size_t pix = (size_t)py * W + px;
    float* oc = out_color + (size_t)v * 3 * P;
    oc[pix] = C0 + T * bg0;
    oc[P + pix] = C1 + T * bg1;
    oc[2 * P + pix] = C2 + T * bg2;
    final_T[(size_t)v * P + pix] = T;
    n_contrib[(size_t)v * P + pix] = last;
  }
  if (ck_written) {   // (T at the split, colour behind the split / that T): what the backward's front task starts from
    const int qx = (row & 1) * 4 + (lane & 3), qy = (row >> 1) * 4 + ((lane >> 2) & 3);
    const float inv = ck.x > 0.f ? 1.f / ck.x : 0.f;       // stopped before the split: never read
    checkpoint[((size_t)tile_global * 4 + quad) * kWave + qy * 8 + qx] =
        make_float4(ck.x, (C0 - ck.y) * inv, (C1 - ck.z) * inv, (C2 - ck.w) * inv);
  }
  // the tile's last contributor (the four quadrant waves max into it; cleared by the binning's scan kernel)
  uint32_t max_c = live ? last : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const uint32_t u = __shfl_xor(max_c, o); max_c = u > max_c ? u : max_c; }
  if (lane == 0 && max_c != 0u) atomicMax(&tile_end[tile_global], max_c);
}

void launch_tiles_forward_rows(const PsRasterDesc& d, const float* records, const uint4* cell_windows,
                               const uint32_t* tile_order, const uint32_t* tile_ranges,
                               const uint32_t* point_list, uint32_t capacity, const float* view_params,
                               float* out_color, float* final_T, uint32_t* n_contrib, float4* checkpoint,
                               uint32_t* tile_end, hipStream_t st) {
  const Dims m = make_dims(d);
  const int total = m.V * m.tiles;      // one block of four waves (quadrants) per tile
  static const int variant = [] { const char* e = getenv("PS_CELLS_VARIANT"); return e ? atoi(e) : 0; }();
  auto go = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(total), dim3(256), 0, st, d, records, cell_windows, tile_order,
                       tile_ranges, point_list, capacity, view_params, out_color, final_T, n_contrib,
                       checkpoint, tile_end);
  };
  switch (variant) {
    case 1: go(tiles_forward_rows_kernel<5, 16>); break;
    case 2: go(tiles_forward_rows_kernel<4, 8>); break;
    case 3: go(tiles_forward_rows_kernel<5, 4>); break;
    default: go(tiles_forward_rows_kernel<5, 8>); break;
  }
}

}  // namespace ps
