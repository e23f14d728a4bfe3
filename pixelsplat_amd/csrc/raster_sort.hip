// Per-view stable LSD radix sort of (depth bits -> Gaussian id), 4 passes x 8 bits, all
// views in one launch per pass.  Culled Gaussians carry key 0xFFFFFFFF: the first pass drops
// them (it is also the compaction: n_vis[v] = its histogram total), the other three passes
// only touch the n_vis[v] visible entries (40 % of the Gaussians at the paper config).
// Stability + ids emitted in ascending order give exactly the order
// of the reference's (tile | depth) stable sort restricted to any tile, so
// the per-tile lists the tile kernels walk are bit-identical to the reference's bins.
//
// Pass = histogram (per 4096-key block) -> per-view scan -> stable scatter.  Ranking inside a
// block is wave-synchronous: 8 ballots build the mask of lanes that share my digit
// (wave64 "match"), rank = popcount below me, a per-wave LDS counter carries the running
// digit offsets across the wave's 16 sequential 64-key rows.
#include "raster_common.h"

namespace ps {


template <bool FIRST>
__global__ void __launch_bounds__(kSortThreads)
sort_hist_kernel(const uint32_t* __restrict__ keys, uint32_t* __restrict__ block_hist, int G,
                 int nblk, int shift, const uint32_t* __restrict__ n_vis) {
  __shared__ uint32_t h[256];
  const int v = blockIdx.y, blk = blockIdx.x, t = threadIdx.x;
  const int limit = FIRST ? G : (int)n_vis[v];
  const int base = blk * kSortChunk;
  if (base < limit) {
    if (t < 256) h[t] = 0;
    __syncthreads();
    const uint32_t* k = keys + (size_t)v * G;
    // all loads first, branch free (clamped index): a load inside `if (p < limit)` is waited
    // for at the end of its branch, i.e. kSortItems dependent round trips per thread
    uint32_t key[kSortItems];
#pragma unroll
    for (int i = 0; i < kSortItems; ++i) {
      const int p = base + i * kSortThreads + t;
      key[i] = k[p < limit ? p : base];
    }
#pragma unroll
    for (int i = 0; i < kSortItems; ++i) {
      const int p = base + i * kSortThreads + t;
      if (p < limit && (!FIRST || key[i] != kCulledKey)) atomicAdd(&h[(key[i] >> shift) & 0xFFu], 1u);
    }
    __syncthreads();
  }
  if (t < 256)
    block_hist[((size_t)v * nblk + blk) * 256 + t] = base < limit ? h[t] : 0u;   // [v][blk][digit]
}

// one block per view: exclusive scan of block_hist[v] in (digit-major, block-minor) order.
// Storage is [v][block][digit], so a thread (= digit) walking its blocks reads coalesced
// 1 KB rows, 16 at a time so that the loads of a group are in flight together (a plain
// running loop over a digit-major array was a chain of nblk dependent, uncoalesced loads).
constexpr int kScanRegs = 128;   // block histograms a thread keeps in registers (G <= 524 288)

__global__ void __launch_bounds__(256)
sort_scan_kernel(uint32_t* __restrict__ block_hist, int nblk, uint32_t* __restrict__ n_vis) {
  __shared__ uint32_t tot[256];
  const int v = blockIdx.x, dgt = threadIdx.x;
  uint32_t* row = block_hist + (size_t)v * nblk * 256 + dgt;     // element b at row[b * 256]
  // One block per view does this, so it is pure latency.  Up to kScanRegs blocks: ONE batch of
  // loads (branch free, clamped), the running sums stay in registers across the digit scan, one
  // batch of stores -- 2 memory round trips instead of 12 (load 16 / store 16, twice over).
  const bool fits = nblk <= kScanRegs;                            // uniform
  uint32_t c[kScanRegs];
  uint32_t sum = 0;
  if (fits) {
#pragma unroll
    for (int i = 0; i < kScanRegs; ++i) c[i] = row[(size_t)(i < nblk ? i : 0) * 256];
#pragma unroll
    for (int i = 0; i < kScanRegs; ++i) {
      const uint32_t x = i < nblk ? c[i] : 0u;
      c[i] = sum;                                                  // exclusive prefix over blocks
      sum += x;
    }
  } else {
    for (int b0 = 0; b0 < nblk; b0 += 16) {
      uint32_t t16[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) t16[i] = row[(size_t)min(b0 + i, nblk - 1) * 256];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (b0 + i < nblk) { row[(size_t)(b0 + i) * 256] = sum; sum += t16[i]; }
      }
    }
  }
  tot[dgt] = sum;
  __syncthreads();
  // exclusive scan over 256 digit totals (Hillis-Steele in LDS)
  uint32_t x = sum;
  for (int off = 1; off < 256; off <<= 1) {
    const uint32_t y = (dgt >= off) ? tot[dgt - off] : 0u;
    __syncthreads();
    x += y; tot[dgt] = x;
    __syncthreads();
  }
  const uint32_t excl = x - sum;
  if (fits) {
#pragma unroll
    for (int i = 0; i < kScanRegs; ++i)
      if (i < nblk) row[(size_t)i * 256] = c[i] + excl;
  } else {
    for (int b0 = 0; b0 < nblk; b0 += 16) {
      uint32_t t16[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) t16[i] = row[(size_t)min(b0 + i, nblk - 1) * 256];
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (b0 + i < nblk) row[(size_t)(b0 + i) * 256] = t16[i] + excl;
    }
  }
  // first pass only: its histogram skipped the culled keys, so the grand total is the number
  // of visible Gaussians of the view -- n_vis without any atomics
  if (n_vis != nullptr && dgt == 255) n_vis[v] = x;
}

// LAST: the final pass also writes sorted_rect[pos] = rects[id] (the tile rects in depth order
// for the binning kernels): the id is in a register here, so the gather's loads fly under the
// ranking instead of being a kernel of their own (57 us of dependent 8-byte gathers).
template <bool IOTA_VALS, bool LAST = false>
__global__ void __launch_bounds__(kSortThreads)
sort_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                    const uint32_t* __restrict__ block_hist, int G, int nblk, int shift,
                    const uint32_t* __restrict__ n_vis, const uint2* __restrict__ rects = nullptr,
                    uint2* __restrict__ sorted_rect = nullptr) {
  constexpr int NW = kSortThreads / kWave;
  __shared__ uint32_t cnt[NW][256];   // per-wave running digit counts
  __shared__ uint32_t base[NW][256];  // global start of (wave, digit)
  const int v = blockIdx.y, blk = blockIdx.x, t = threadIdx.x;
  const int w = t >> 6, lane = t & 63;
  const int limit = IOTA_VALS ? G : (int)n_vis[v];
  if (blk * kSortChunk >= limit) return;
  for (int i = t; i < NW * 256; i += kSortThreads) (&cnt[0][0])[i] = 0;
  __syncthreads();

  const size_t vo = (size_t)v * G;
  const int start = blk * kSortChunk + w * (kSortChunk / NW);
  uint32_t key[kSortItems], val[kSortItems], rank[kSortItems];
  uint2 rc[LAST ? kSortItems : 1];
  const uint64_t lt = lanemask_lt();
  // Every global load of the block's items is issued before the ranking starts, branch free
  // (clamped index, masked afterwards): keys and values in one round trip, the rect gathers of
  // the last pass in a second one that runs under the ranking.  Loading inside the per-item
  // loop made each item wait for its own loads: kSortItems dependent round trips per wave.
  const int p_safe = blk * kSortChunk;               // < limit (checked above)
#pragma unroll
  for (int i = 0; i < kSortItems; ++i) {
    const int p = start + i * kWave + lane;
    const int pc = p < limit ? p : p_safe;
    key[i] = keys_in[vo + pc];
    val[i] = IOTA_VALS ? (uint32_t)pc : vals_in[vo + pc];
  }
  // the block's digit offsets (used after the ranking by threads 0..255; every thread loads)
  const uint32_t digit_base = block_hist[((size_t)v * nblk + blk) * 256 + (t & 255)];
  if (LAST) {
#pragma unroll
    for (int i = 0; i < kSortItems; ++i) rc[i] = rects[vo + val[i]];
  }
#pragma unroll
  for (int i = 0; i < kSortItems; ++i) {
    const int p = start + i * kWave + lane;
    bool valid = p < limit;
    if (IOTA_VALS) valid = valid && key[i] != kCulledKey;
    const uint32_t dg = (key[i] >> shift) & 0xFFu;
    uint64_t mask = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (dg >> b) & 1u;
      const uint64_t bal = __ballot(bit);
      mask &= bit ? bal : ~bal;
    }
    if (valid) {
      const uint32_t prefix = cnt[w][dg];
      const uint32_t r = (uint32_t)__popcll(mask & lt);
      rank[i] = prefix + r;
      if (r == 0) cnt[w][dg] = prefix + (uint32_t)__popcll(mask);
    } else {
      rank[i] = kCulledKey;       // dropped
    }
    wave_lds_sync();
  }
  __syncthreads();
  if (t < 256) {
    uint32_t b = digit_base;
#pragma unroll
    for (int i = 0; i < NW; ++i) { base[i][t] = b; b += cnt[i][t]; }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kSortItems; ++i) {
    if (rank[i] != kCulledKey) {
      const uint32_t dg = (key[i] >> shift) & 0xFFu;
      const uint32_t pos = base[w][dg] + rank[i];
      if (!LAST) keys_out[vo + pos] = key[i];   // nobody reads the keys after the last pass
      vals_out[vo + pos] = val[i];
      if (LAST) sorted_rect[vo + pos] = rc[i];
    }
  }
}

void launch_sort(const PsRasterDesc& d, uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a,
                 uint32_t* vals_b, uint32_t* block_hist, uint32_t* sorted_idx,
                 const uint2* rects, uint2* sorted_rect, uint32_t* n_vis, hipStream_t st) {
  const Dims m = make_dims(d);
  dim3 grid(m.nblk, m.V), block(kSortThreads);
  uint32_t* kin = keys_a; uint32_t* kout = keys_b;
  uint32_t* vin = nullptr; uint32_t* vout = vals_b;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = pass * 8;
    if (pass == 0)
      hipLaunchKernelGGL(sort_hist_kernel<true>, grid, block, 0, st, kin, block_hist, m.G, m.nblk,
                         shift, n_vis);
    else
      hipLaunchKernelGGL(sort_hist_kernel<false>, grid, block, 0, st, kin, block_hist, m.G, m.nblk,
                         shift, n_vis);
    hipLaunchKernelGGL(sort_scan_kernel, dim3(m.V), dim3(256), 0, st, block_hist, m.nblk,
                       pass == 0 ? n_vis : (uint32_t*)nullptr);
    if (pass == 0)
      hipLaunchKernelGGL((sort_scatter_kernel<true, false>), grid, block, 0, st, kin, vin, kout,
                         vout, block_hist, m.G, m.nblk, shift, n_vis, (const uint2*)nullptr,
                         (uint2*)nullptr);
    else if (pass < 3)
      hipLaunchKernelGGL((sort_scatter_kernel<false, false>), grid, block, 0, st, kin, vin, kout,
                         vout, block_hist, m.G, m.nblk, shift, n_vis, (const uint2*)nullptr,
                         (uint2*)nullptr);
    else
      hipLaunchKernelGGL((sort_scatter_kernel<false, true>), grid, block, 0, st, kin, vin, kout,
                         vout, block_hist, m.G, m.nblk, shift, n_vis, rects, sorted_rect);
    // ping-pong: keys a<->b ; vals: (iota)->b->a->b->sorted_idx
    uint32_t* tk = kin; kin = kout; kout = tk;
    vin = vout;
    vout = (pass == 0) ? vals_a : (pass == 1) ? vals_b : sorted_idx;
  }
}

}  // namespace ps
