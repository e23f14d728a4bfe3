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
  hipLaunchKernelGGL(bin_kernel<true>, grid, dim3(256), 0, st, d, sorted_rect, sorted_idx, n_vis,
                     counts, tile_ranges, point_list, capacity);
}

}  // namespace ps
