// Tile binning: turns each view's depth-sorted Gaussian sequence into per-tile lists that
// are bit-identical to the reference's bins (sorted (tile | depth) point list + tile ranges,
// SURVEY.md A.2) without ever materialising or sorting the duplicated 64-bit keys.
//
// A block owns a chunk of 1024 consecutive entries of one view's depth-sorted sequence
// (rects staged in LDS); thread t owns tile t (t, t+256, ... for larger images) and walks
// the chunk IN ORDER, so every tile list inherits the (depth, id) order for free -- a
// stable 1-to-many partition with no atomics.  Two passes over the same 8 bytes/entry:
//   count : counts[v][b][t]
//   scan  : per tile exclusive prefix over blocks; tile totals -> tile_ranges, D
//   write : point_list[tile_start + block_off + running] = gaussian id
// The LDS reads are wave-uniform (broadcast); the work is ~256 rect tests per entry, i.e.
// cheap integer VALU instead of the reference's 5-6 radix passes over D 12-byte pairs.
#include "raster_common.h"

#include <cstdlib>

namespace ps {

template <bool WRITE>
__global__ void __launch_bounds__(256)
bin_kernel(PsRasterDesc d, const uint2* __restrict__ sorted_rect,
           const uint32_t* __restrict__ sorted_idx, const uint32_t* __restrict__ n_vis,
           uint32_t* __restrict__ counts /*[V][nb][tiles]: count, then block offset*/,
           const uint32_t* __restrict__ tile_ranges /*[V][tiles][2]*/,
           uint32_t* __restrict__ point_list, uint32_t capacity) {
  __shared__ uint2 s_rect[kBinChunk];
  __shared__ uint32_t s_idx[WRITE ? kBinChunk : 1];
  const Dims m = make_dims(d);
  int v, b;
  view_minor_block(b, v);
  const uint32_t n = n_vis[v];
  const uint32_t base = (uint32_t)b * kBinChunk;
  if (base >= n) return;  // bins past the visible prefix are never read
  const uint32_t cnt = n - base < (uint32_t)kBinChunk ? n - base : (uint32_t)kBinChunk;
  const size_t vo = (size_t)v * m.G;
  for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
    s_rect[i] = sorted_rect[vo + base + i];
    if (WRITE) s_idx[i] = sorted_idx[vo + base + i];
  }
  __syncthreads();
  // A wave owns 64 consecutive tiles = a band of tile rows.  Entries are first tested
  // against the band, 64 at a time (one entry per lane, one ballot); only the survivors
  // (a rect spans 1-2 tile rows, so ~1 in 4 at 256x256) are tested by every lane against
  // its own tile.  Survivors are visited in ascending order, so the lists stay depth-sorted.
  const int lane = threadIdx.x & 63;
  for (int t0 = (int)threadIdx.x - lane; t0 < m.tiles; t0 += blockDim.x) {
    const int t = t0 + lane;
    const bool tile_ok = t < m.tiles;
    const uint32_t Tt = (uint32_t)(t % m.gx) | ((uint32_t)(t / m.gx) << 16);
    const uint32_t T1 = Tt + 0x00010001u;
    const uint32_t band_lo = (uint32_t)(t0 / m.gx);
    const uint32_t band_hi = (uint32_t)(min(t0 + 63, m.tiles - 1) / m.gx);
    uint32_t* c = counts + ((size_t)v * m.nbin + b) * m.tiles + (tile_ok ? t : 0);
    uint32_t k = 0;
    uint32_t off = 0;
    if (WRITE && tile_ok) off = tile_ranges[2 * ((size_t)v * m.tiles + t)] + *c;
    for (uint32_t i0 = 0; i0 < cnt; i0 += 64) {
      const uint32_t e = i0 + lane;
      bool in_band = false;
      if (e < cnt) {
        const uint2 r = s_rect[e];
        in_band = (r.x >> 16) <= band_hi && (r.y >> 16) > band_lo;
      }
      uint64_t hits = __ballot(in_band);
      while (hits) {
        const uint32_t i = i0 + (uint32_t)__builtin_ctzll(hits);
        hits &= hits - 1;
        const uint2 r = s_rect[i];
        if (tile_ok && rect_covers_packed(r, Tt, T1)) {
          if (!WRITE) {
            ++k;
          } else {
            if (off < capacity) point_list[off] = s_idx[i];
            ++off;
          }
        }
      }
    }
    if (!WRITE && tile_ok) *c = k;
  }
}


// Write pass, entry-parallel (round 6).  bin_kernel<true> above walks every entry of the chunk with 64 lanes = 64
// tiles of which ~5 are covered: 77 M wave instructions for 13.9 M list entries at BASELINE configs[1], 0.22 ms.
// Here the chunk's (entry, tile) pairs are EXPANDED into LDS in entry-major order (a thread owns four consecutive
// entries; a block scan of their tile counts places them) and stably counting-sorted by tile id, the ranking being
// the radix sort's (raster_sort.hip): per 64 pairs, ceil(log2 tiles) ballots build the mask of lanes on the same
// tile, rank = popcount below me, a per-wave LDS counter per tile carries the running count across the wave's
// batches, and a prefix over the waves turns ranks into positions.  Entry-major order + a stable sort by tile =
// every tile list in (depth, id) order: the same bytes bin_kernel<true> writes.  A chunk whose pairs do not fit
// kPairCap is processed in pieces of consecutive entries (the per-tile positions carry over).
constexpr int kPairCap = 4096;              // pairs per piece (16 KB of LDS)
constexpr int kWriteTilesMax = 1024;        // per-tile LDS words: (1 + waves) x tiles (images up to 512 x 512)

// NT threads: NT / 64 waves rank the piece's pairs side by side, kBinChunk / NT consecutive entries per thread
template <int NT>
__global__ void __launch_bounds__(NT)
bin_write_pairs_kernel(PsRasterDesc d, const uint2* __restrict__ sorted_rect,
                       const uint32_t* __restrict__ sorted_idx, const uint32_t* __restrict__ n_vis,
                       const uint32_t* __restrict__ counts /*[V][nb][tiles]: block offsets*/,
                       const uint32_t* __restrict__ tile_ranges, uint32_t* __restrict__ point_list,
                       uint32_t capacity, int tile_bits) {
  constexpr int NW = NT / 64, EPT = kBinChunk / NT, kPairBatches = kPairCap / NT;
  extern __shared__ uint32_t dyn[];         // pos[tiles] | cnt[NW][tiles]
  __shared__ uint32_t s_idx[kBinChunk];
  __shared__ uint32_t s_pre[kBinChunk + 1];      // exclusive prefix of the entries' tile counts
  __shared__ uint32_t s_pair[kPairCap];          // tile << 16 | entry (local)
  __shared__ uint32_t s_scan[NW];
  __shared__ uint32_t s_end;
  const Dims m = make_dims(d);
  uint32_t* const pos = dyn;
  uint32_t* const cntw = dyn + m.tiles;
  int v, b;
  view_minor_block(b, v);
  const uint32_t n = n_vis[v];
  const uint32_t base = (uint32_t)b * kBinChunk;
  if (base >= n) return;
  const uint32_t cnt = n - base < (uint32_t)kBinChunk ? n - base : (uint32_t)kBinChunk;
  const size_t vo = (size_t)v * m.G;
  const int t = threadIdx.x, w = t >> 6, lane = t & 63;
  // the thread's four consecutive entries; every load first
  uint2 r4[EPT]; uint32_t i4[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const uint32_t e = (uint32_t)EPT * t + k;
    const uint32_t ec = e < cnt ? e : 0u;
    r4[k] = sorted_rect[vo + base + ec];
    i4[k] = sorted_idx[vo + base + ec];
  }
  for (int i = t; i < m.tiles; i += NT)
    pos[i] = tile_ranges[2 * ((size_t)v * m.tiles + i)] + counts[((size_t)v * m.nbin + b) * m.tiles + i];
  uint32_t c4[EPT], mine = 0;
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const uint32_t e = (uint32_t)EPT * t + k;
    const uint32_t xmin = r4[k].x & 0xFFFFu, ymin = r4[k].x >> 16, xmax = r4[k].y & 0xFFFFu, ymax = r4[k].y >> 16;
    c4[k] = (e < cnt && xmax > xmin && ymax > ymin) ? (xmax - xmin) * (ymax - ymin) : 0u;
    s_idx[e] = i4[k];
    mine += c4[k];
  }
  // block exclusive scan of `mine` (wave scan by DPP-free shuffles, then the four wave totals)
  uint32_t inc = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const uint32_t u = __shfl_up(inc, o); if (lane >= o) inc += u; }
  if (lane == 63) s_scan[w] = inc;
  __syncthreads();
  uint32_t wave_base = 0;
#pragma unroll
  for (int i = 0; i < NW; ++i) if (i < w) wave_base += s_scan[i];
  uint32_t run = wave_base + inc - mine;
#pragma unroll
  for (int k = 0; k < EPT; ++k) { s_pre[EPT * t + k] = run; run += c4[k]; }
  if (t == NT - 1) s_pre[kBinChunk] = run;
  __syncthreads();
  const uint32_t total = s_pre[kBinChunk];
  const uint64_t lt = lanemask_lt();

  uint32_t first = 0;                          // first entry of the piece
  while (first < cnt) {
    // the piece: entries [first, end) with at most kPairCap pairs (at least one entry: tiles <= kPairCap)
    const uint32_t p0 = s_pre[first];
    if (t == 0) s_end = cnt;
    __syncthreads();
    if (total - p0 > (uint32_t)kPairCap) {
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const uint32_t e = (uint32_t)EPT * t + k;
        if (e >= first && e < cnt && s_pre[e + 1] - p0 > (uint32_t)kPairCap) atomicMin(&s_end, e);
      }
      __syncthreads();
    }
    const uint32_t end = s_end;
    const uint32_t npairs = s_pre[end] - p0;
    for (int i = t; i < NW * m.tiles; i += NT) cntw[i] = 0u;
    // expand
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const uint32_t e = (uint32_t)EPT * t + k;
      if (e >= first && e < end && c4[k] != 0u) {
        const uint32_t xmin = r4[k].x & 0xFFFFu, ymin = r4[k].x >> 16, xmax = r4[k].y & 0xFFFFu, ymax = r4[k].y >> 16;
        uint32_t q = s_pre[e] - p0;
        for (uint32_t ty = ymin; ty < ymax; ++ty)
          for (uint32_t tx = xmin; tx < xmax; ++tx) s_pair[q++] = ((ty * (uint32_t)m.gx + tx) << 16) | e;
      }
    }
    __syncthreads();
    // rank: wave w takes the pairs [w, w + 1) * per of the piece, 64 at a time, in order
    const uint32_t per = ((npairs + (uint32_t)NT - 1u) / (uint32_t)NT) * 64u;
    uint32_t pr[kPairBatches], rk[kPairBatches];
#pragma unroll
    for (int j = 0; j < kPairBatches; ++j) {
      const uint32_t q = (uint32_t)w * per + (uint32_t)j * 64u + (uint32_t)lane;
      const bool valid = (uint32_t)j * 64u < per && q < npairs;       // (first clause: wave-uniform)
      pr[j] = valid ? s_pair[q] : 0xFFFFFFFFu;
      rk[j] = 0u;
      if ((uint32_t)j * 64u < per) {
        const uint32_t tile = pr[j] >> 16;
        uint64_t mask = __ballot(valid);
        for (int bit = 0; bit < tile_bits; ++bit) {
          const bool one = (tile >> bit) & 1u;
          const uint64_t bal = __ballot(one);
          mask &= one ? bal : ~bal;
        }
        if (valid) {
          const uint32_t prefix = cntw[w * m.tiles + tile];
          const uint32_t r = (uint32_t)__popcll(mask & lt);
          rk[j] = prefix + r;
          if (r == 0u) cntw[w * m.tiles + tile] = prefix + (uint32_t)__popcll(mask);
        }
        wave_lds_sync();
      }
    }
    __syncthreads();
    // per tile: the waves' counts -> starts; the tile's position moves on by the piece's total
    for (int i = t; i < m.tiles; i += NT) {
      uint32_t run_t = pos[i];
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) { const uint32_t c = cntw[ww * m.tiles + i]; cntw[ww * m.tiles + i] = run_t; run_t += c; }
      pos[i] = run_t;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kPairBatches; ++j) {
      if (pr[j] != 0xFFFFFFFFu) {
        const uint32_t at = cntw[w * m.tiles + (pr[j] >> 16)] + rk[j];
        if (at < capacity) point_list[at] = s_idx[pr[j] & 0xFFFFu];
      }
    }
    __syncthreads();
    first = end;
  }
}

// Count pass as a summed-area table: a rect covering tiles [xmin, xmax) x [ymin, ymax) adds
// +1 / -1 / -1 / +1 at its four corners of a (gy + 1) x (gx + 1) grid in LDS (integer LDS
// atomics), and the 2-D inclusive prefix sum of that grid is the number of the block's entries
// covering each tile -- the same integers the ballot walk of bin_kernel<false> produces
// (entries x tiles / 64 tests: 116 us at configs[1]) for O(entries + tiles) work.
constexpr int kGridMax = 8192;   // grid cells that fit the LDS budget (images up to ~1400 px)

__global__ void __launch_bounds__(256)
bin_count_grid_kernel(PsRasterDesc d, const uint2* __restrict__ sorted_rect,
                      const uint32_t* __restrict__ n_vis, uint32_t* __restrict__ counts) {
  extern __shared__ int grid[];   // [(gy + 1)][(gx + 1)]
  const Dims m = make_dims(d);
  int v, b;
  view_minor_block(b, v);
  const uint32_t n = n_vis[v];
  const uint32_t base = (uint32_t)b * kBinChunk;
  if (base >= n) return;  // bins past the visible prefix are never read
  const uint32_t cnt = n - base < (uint32_t)kBinChunk ? n - base : (uint32_t)kBinChunk;
  const int gw = m.gx + 1, cells = gw * (m.gy + 1);
  for (int i = threadIdx.x; i < cells; i += blockDim.x) grid[i] = 0;
  __syncthreads();
  const size_t vo = (size_t)v * m.G;
  for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
    const uint2 r = sorted_rect[vo + base + i];
    const int xmin = r.x & 0xFFFFu, ymin = r.x >> 16, xmax = r.y & 0xFFFFu, ymax = r.y >> 16;
    if (xmax > xmin && ymax > ymin) {
      atomicAdd(&grid[ymin * gw + xmin], 1);
      atomicSub(&grid[ymin * gw + xmax], 1);
      atomicSub(&grid[ymax * gw + xmin], 1);
      atomicAdd(&grid[ymax * gw + xmax], 1);
    }
  }
  __syncthreads();
  for (int y = threadIdx.x; y <= m.gy; y += blockDim.x) {   // prefix along x, one row per thread
    int run = 0;
    for (int x = 0; x < gw; ++x) { run += grid[y * gw + x]; grid[y * gw + x] = run; }
  }
  __syncthreads();
  for (int x = threadIdx.x; x < gw; x += blockDim.x) {      // prefix along y, one column each
    int run = 0;
    for (int y = 0; y <= m.gy; ++y) { run += grid[y * gw + x]; grid[y * gw + x] = run; }
  }
  __syncthreads();
  uint32_t* c = counts + ((size_t)v * m.nbin + b) * m.tiles;
  for (int t = threadIdx.x; t < m.tiles; t += blockDim.x)
    c[t] = (uint32_t)grid[(t / m.gx) * gw + (t % m.gx)];
}

// per (view, tile): exclusive prefix of counts over the view's blocks (in place) and the
// tile total
__global__ void __launch_bounds__(256)
bin_scan_blocks_kernel(PsRasterDesc d, const uint32_t* __restrict__ n_vis,
                       uint32_t* __restrict__ counts, uint32_t* __restrict__ tile_ranges) {
  const Dims m = make_dims(d);
  const int vt = blockIdx.x * blockDim.x + threadIdx.x;
  if (vt >= m.V * m.tiles) return;
  const int v = vt / m.tiles, t = vt % m.tiles;
  // only the bins holding visible entries were counted; 16 loads in flight per step
  const int nb = (int)((n_vis[v] + kBinChunk - 1) / kBinChunk);
  uint32_t* col = counts + (size_t)v * m.nbin * m.tiles + t;
  uint32_t run = 0;
  for (int b0 = 0; b0 < nb; b0 += 16) {
    uint32_t c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = b0 + i < nb ? col[(size_t)(b0 + i) * m.tiles] : 0u;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (b0 + i < nb) col[(size_t)(b0 + i) * m.tiles] = run;
      run += c[i];
    }
  }
  tile_ranges[2 * (size_t)vt + 1] = run;
}

// single block: exclusive scan of the V*tiles tile totals -> tile starts, D, overflow flag
__global__ void __launch_bounds__(1024)
bin_scan_tiles_kernel(PsRasterDesc d, uint32_t* __restrict__ tile_ranges,
                      uint32_t* __restrict__ num_rendered /*[2]: D, overflow*/,
                      uint32_t* __restrict__ tile_order,
                      uint32_t* __restrict__ tile_end /* cleared: the tile forward's waves max into it */) {
  __shared__ uint32_t part[1024];
  __shared__ uint32_t hist[1024];
  __shared__ uint32_t s_max;
  const Dims m = make_dims(d);
  const int total = m.V * m.tiles;
  const int per = (total + 1023) / 1024;
  const int lo = threadIdx.x * per, hi = lo + per < total ? lo + per : total;
  // A single block: pure latency.  The thread's (up to kPer) tile totals are read ONCE, branch
  // free, and kept in registers for all five uses (sum, starts, max, histogram, order); read
  // where used they were ~5 x `per` dependent loads per thread.  More tiles than 1024 x kPer:
  // the totals are re-read (same results).
  constexpr int kPer = 8;
  const bool fits = per <= kPer;                                  // uniform
  uint32_t cnt[kPer];
#pragma unroll
  for (int u = 0; u < kPer; ++u)
    cnt[u] = tile_ranges[2 * (size_t)min(lo + u, total - 1) + 1];   // (unused when !fits)
  uint32_t sum = 0;
#pragma unroll
  for (int u = 0; u < kPer; ++u)
    if (fits && lo + u < hi) sum += cnt[u];
  if (!fits) for (int i = lo; i < hi; ++i) sum += tile_ranges[2 * (size_t)i + 1];
  part[threadIdx.x] = sum;
  __syncthreads();
  uint32_t x = sum;
  for (int off = 1; off < 1024; off <<= 1) {
    const uint32_t y = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    x += y; part[threadIdx.x] = x;
    __syncthreads();
  }
  uint32_t run = x - sum;
  if (fits) {
#pragma unroll
    for (int u = 0; u < kPer; ++u)
      if (lo + u < hi) { tile_ranges[2 * (size_t)(lo + u)] = run; run += cnt[u]; }
  } else {
    for (int i = lo; i < hi; ++i) {
      tile_ranges[2 * (size_t)i] = run;
      run += tile_ranges[2 * (size_t)i + 1];
    }
  }
  if (threadIdx.x == 1023) {
    num_rendered[0] = x;
    num_rendered[1] = 0u;
  }
  for (int i = lo; i < hi; ++i) tile_end[i] = 0u;
  // Longest-list-first launch order for the tile kernels (one wave per tile, dispatched in
  // block order => LPT scheduling): counting sort of the tiles by list length, 1024 buckets.
  if (threadIdx.x == 0) s_max = 1u;
  hist[threadIdx.x] = 0u;
  __syncthreads();
  uint32_t mx = 0;
  if (fits) {
#pragma unroll
    for (int u = 0; u < kPer; ++u)
      if (lo + u < hi) mx = cnt[u] > mx ? cnt[u] : mx;
  } else {
    for (int i = lo; i < hi; ++i) { const uint32_t c = tile_ranges[2 * (size_t)i + 1]; mx = c > mx ? c : mx; }
  }
  atomicMax(&s_max, mx);
  __syncthreads();
  const uint32_t maxc = s_max;
  auto bucket = [&](uint32_t c) -> uint32_t {   // 0 = longest
    return 1023u - (uint32_t)(((uint64_t)c * 1023ull) / maxc);
  };
  if (fits) {
#pragma unroll
    for (int u = 0; u < kPer; ++u)
      if (lo + u < hi) atomicAdd(&hist[bucket(cnt[u])], 1u);
  } else {
    for (int i = lo; i < hi; ++i) atomicAdd(&hist[bucket(tile_ranges[2 * (size_t)i + 1])], 1u);
  }
  __syncthreads();
  const uint32_t hsum = hist[threadIdx.x];
  part[threadIdx.x] = hsum;
  __syncthreads();
  uint32_t hx = hsum;
  for (int off = 1; off < 1024; off <<= 1) {
    const uint32_t y = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    hx += y; part[threadIdx.x] = hx;
    __syncthreads();
  }
  hist[threadIdx.x] = hx - hsum;   // exclusive start of the bucket
  __syncthreads();
  if (fits) {
#pragma unroll
    for (int u = 0; u < kPer; ++u)
      if (lo + u < hi) {
        const uint32_t pos = atomicAdd(&hist[bucket(cnt[u])], 1u);
        tile_order[pos] = (uint32_t)(lo + u);
      }
  } else {
    for (int i = lo; i < hi; ++i) {
      const uint32_t pos = atomicAdd(&hist[bucket(tile_ranges[2 * (size_t)i + 1])], 1u);
      tile_order[pos] = (uint32_t)i;
    }
  }
}

__global__ void bin_flag_kernel(uint32_t* __restrict__ num_rendered, uint32_t capacity) {
  if (threadIdx.x == 0) num_rendered[1] = num_rendered[0] > capacity ? 1u : 0u;
}

// count + scans: everything that does not need the point list (whose size, D, they produce)
void launch_bin_count(const PsRasterDesc& d, const uint2* sorted_rect, const uint32_t* n_vis,
                      uint32_t* counts, uint32_t* tile_ranges, uint32_t* num_rendered,
                      uint32_t* tile_order, uint32_t* tile_end, hipStream_t st) {
  const Dims m = make_dims(d);
  dim3 grid(m.nbin, m.V);
  const int cells = (m.gx + 1) * (m.gy + 1);
  if (cells <= kGridMax)
    hipLaunchKernelGGL(bin_count_grid_kernel, grid, dim3(256), cells * sizeof(int), st, d,
                       sorted_rect, n_vis, counts);
  else
    hipLaunchKernelGGL(bin_kernel<false>, grid, dim3(256), 0, st, d, sorted_rect,
                       (const uint32_t*)nullptr, n_vis, counts, (const uint32_t*)nullptr,
                       (uint32_t*)nullptr, 0u);
  hipLaunchKernelGGL(bin_scan_blocks_kernel, dim3((m.V * m.tiles + 255) / 256), dim3(256), 0, st,
                     d, n_vis, counts, tile_ranges);
  hipLaunchKernelGGL(bin_scan_tiles_kernel, dim3(1), dim3(1024), 0, st, d, tile_ranges,
                     num_rendered, tile_order, tile_end);
}

void launch_bin_write(const PsRasterDesc& d, const uint2* sorted_rect, const uint32_t* sorted_idx,
                      const uint32_t* n_vis, uint32_t* counts, const uint32_t* tile_ranges,
                      uint32_t* num_rendered, uint32_t* point_list, uint32_t capacity,
                      hipStream_t st) {
  const Dims m = make_dims(d);
  dim3 grid(m.nbin, m.V);
  hipLaunchKernelGGL(bin_flag_kernel, dim3(1), dim3(64), 0, st, num_rendered, capacity);
  // PS_BIN_WRITE_WALK=1: the round-2 tile-parallel walk (A/B runs; also what larger tile grids take)
  static const bool walk = [] { const char* e = getenv("PS_BIN_WRITE_WALK"); return e && e[0] == '1'; }();
  if (!walk && m.tiles <= kWriteTilesMax && m.tiles <= kPairCap) {
    int bits = 0;
    while ((1 << bits) < m.tiles) ++bits;
    static const int nt = [] { const char* e = getenv("PS_BIN_WRITE_THREADS"); return e ? atoi(e) : 512; }();
    if (nt == 256)
      hipLaunchKernelGGL(bin_write_pairs_kernel<256>, grid, dim3(256), 5 * m.tiles * sizeof(uint32_t), st, d,
                         sorted_rect, sorted_idx, n_vis, counts, tile_ranges, point_list, capacity, bits);
    else if (nt == 1024 && m.tiles <= 256)     // (static + dynamic LDS within 64 KB)
      hipLaunchKernelGGL(bin_write_pairs_kernel<1024>, grid, dim3(1024), 17 * m.tiles * sizeof(uint32_t), st, d,
                         sorted_rect, sorted_idx, n_vis, counts, tile_ranges, point_list, capacity, bits);
    else
      hipLaunchKernelGGL(bin_write_pairs_kernel<512>, grid, dim3(512), 9 * m.tiles * sizeof(uint32_t), st, d,
                         sorted_rect, sorted_idx, n_vis, counts, tile_ranges, point_list, capacity, bits);
  } else {
    hipLaunchKernelGGL(bin_kernel<true>, grid, dim3(256), 0, st, d, sorted_rect, sorted_idx, n_vis,
                       counts, tile_ranges, point_list, capacity);
  }
}

}  // namespace ps
