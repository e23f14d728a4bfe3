// Per-view stable LSD radix sort of (depth bits -> Gaussian id), all views in one launch per pass.
// Digits are 9 bits of (key - kmin), kmin = the bit pattern of the near-cull depth (every visible key
// lies above it): the depths of a view span < 2^27 bit patterns in every configuration seen (0.2 .. 600
// in units of the near plane: 2^26.2), so THREE passes sort a view; a view whose keys span more gets a
// fourth pass over the top 5 bits -- decided per view on the device (the first pass's histogram kernel
// leaves each block's largest key, the scan kernel turns them into the view's pass count), no host
// round trip: the fourth pass's kernels are always launched and return at once for 3-pass views.
// Round 3 ran 4 passes x 8 bits whatever the keys: 0.33 -> 0.27 ms at BASELINE configs[1].
// Culled Gaussians carry key 0xFFFFFFFF: the first pass drops them (it is also the compaction:
// n_vis[v] = its histogram total), the later passes only touch the n_vis[v] visible entries (40 % of the
// Gaussians at the paper config).
// Stability + ids emitted in ascending order give exactly the order
// of the reference's (tile | depth) stable sort restricted to any tile, so
// the per-tile lists the tile kernels walk are bit-identical to the reference's bins.
//
// Pass = histogram (per 4096-key block) -> per-view scan -> stable scatter.  Ranking inside a
// block is wave-synchronous: 9 ballots build the mask of lanes that share my digit
// (wave64 "match"), rank = popcount below me, a per-wave LDS counter carries the running
// digit offsets across the wave's 16 sequential 64-key rows.
#include "raster_common.h"

#include <cstring>

namespace ps {

constexpr int kDigitBits = 9;
constexpr int kDigits = 1 << kDigitBits;          // 512
constexpr uint32_t kDigitMask = kDigits - 1;
constexpr int kMaxPasses = 4;                     // 9 + 9 + 9 + 5 bits
static_assert(kDigits <= kSortThreads / 2, "one thread per digit in the histogram / scan kernels");

// pass_info[v] = number of passes view v needs (3 or 4); written by the first pass's scan kernel
__device__ __forceinline__ uint32_t digit_of(uint32_t key, uint32_t kmin, int shift) {
  return ((key - kmin) >> shift) & kDigitMask;
}

template <bool FIRST>
__global__ void __launch_bounds__(kSortThreads)
sort_hist_kernel(const uint32_t* __restrict__ keys, uint32_t* __restrict__ block_hist, int G,
                 int nblk, int shift, uint32_t kmin, int pass, const uint32_t* __restrict__ n_vis,
                 uint32_t* __restrict__ pass_info /* [V] pass count | [V][nblk] block maxima */) {
  __shared__ uint32_t h[kDigits];
  __shared__ uint32_t s_max;
  const int t = threadIdx.x;
  int v, blk;
  view_minor_block(blk, v);
  if (!FIRST && pass >= (int)pass_info[v]) return;      // (a 3-pass view has no fourth pass)
  const int limit = FIRST ? G : (int)n_vis[v];
  const int base = blk * kSortChunk;
  if (FIRST && t == 0) s_max = 0u;
  if (base < limit) {
    if (t < kDigits) h[t] = 0;
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
      if (p < limit && (!FIRST || key[i] != kCulledKey)) atomicAdd(&h[digit_of(key[i], kmin, shift)], 1u);
    }
    if (FIRST) {     // the block's largest visible key (wave maximum, one LDS atomic per wave)
      uint32_t mx = 0;
#pragma unroll
      for (int i = 0; i < kSortItems; ++i) {
        const int p = base + i * kSortThreads + t;
        if (p < limit && key[i] != kCulledKey) mx = key[i] > mx ? key[i] : mx;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { const uint32_t u = __shfl_xor(mx, o); mx = u > mx ? u : mx; }
      if ((t & 63) == 0 && mx) atomicMax(&s_max, mx);
    }
    __syncthreads();
  }
  if (t < kDigits)
    block_hist[((size_t)v * nblk + blk) * kDigits + t] = base < limit ? h[t] : 0u;   // [v][blk][digit]
  if (FIRST) {
    __syncthreads();
    if (t == 0) pass_info[gridDim.y + (size_t)v * nblk + blk] = base < limit ? s_max : 0u;   // (gridDim.y = V)
  }
}

// one block per view: exclusive scan of block_hist[v] in (digit-major, block-minor) order.
// Storage is [v][block][digit], so a thread (= digit) walking its blocks reads coalesced
// 1 KB rows, 16 at a time so that the loads of a group are in flight together (a plain
// running loop over a digit-major array was a chain of nblk dependent, uncoalesced loads).
constexpr int kScanRegs = 128;   // block histograms a thread keeps in registers (G <= 524 288)

__global__ void __launch_bounds__(kDigits)
sort_scan_kernel(uint32_t* __restrict__ block_hist, int nblk, uint32_t* __restrict__ n_vis,
                 uint32_t kmin, int pass, uint32_t* __restrict__ pass_info) {
  __shared__ uint32_t tot[kDigits];
  const int v = blockIdx.x, dgt = threadIdx.x;
  if (n_vis == nullptr && pass >= (int)pass_info[v]) return;
  uint32_t* row = block_hist + (size_t)v * nblk * kDigits + dgt;     // element b at row[b * kDigits]
  // One block per view does this, so it is pure latency.  Up to kScanRegs blocks: ONE batch of
  // loads (branch free, clamped), the running sums stay in registers across the digit scan, one
  // batch of stores -- 2 memory round trips instead of 12 (load 16 / store 16, twice over).
  const bool fits = nblk <= kScanRegs;                            // uniform
  uint32_t c[kScanRegs];
  uint32_t sum = 0;
  if (fits) {
#pragma unroll
    for (int i = 0; i < kScanRegs; ++i) c[i] = row[(size_t)(i < nblk ? i : 0) * kDigits];
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
      for (int i = 0; i < 16; ++i) t16[i] = row[(size_t)min(b0 + i, nblk - 1) * kDigits];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (b0 + i < nblk) { row[(size_t)(b0 + i) * kDigits] = sum; sum += t16[i]; }
      }
    }
  }
  tot[dgt] = sum;
  __syncthreads();
  // exclusive scan over the digit totals (Hillis-Steele in LDS)
  uint32_t x = sum;
  for (int off = 1; off < kDigits; off <<= 1) {
    const uint32_t y = (dgt >= off) ? tot[dgt - off] : 0u;
    __syncthreads();
    x += y; tot[dgt] = x;
    __syncthreads();
  }
  const uint32_t excl = x - sum;
  if (fits) {
#pragma unroll
    for (int i = 0; i < kScanRegs; ++i)
      if (i < nblk) row[(size_t)i * kDigits] = c[i] + excl;
  } else {
    for (int b0 = 0; b0 < nblk; b0 += 16) {
      uint32_t t16[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) t16[i] = row[(size_t)min(b0 + i, nblk - 1) * kDigits];
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (b0 + i < nblk) row[(size_t)(b0 + i) * kDigits] = t16[i] + excl;
    }
  }
  // first pass only: its histogram skipped the culled keys, so the grand total is the number
  // of visible Gaussians of the view -- n_vis without any atomics
  if (n_vis != nullptr && dgt == kDigits - 1) n_vis[v] = x;
  // first pass only: the view's pass count from the blocks' largest keys (threads stride the blocks,
  // LDS maximum): 3 when every (key - kmin) fits 27 bits
  if (n_vis != nullptr) {
    __syncthreads();
    uint32_t mx = 0;
    for (int b = dgt; b < nblk; b += kDigits) {
      const uint32_t k = pass_info[gridDim.x + (size_t)v * nblk + b];
      mx = k > mx ? k : mx;
    }
    tot[dgt] = mx;
    __syncthreads();
    for (int off = kDigits / 2; off > 0; off >>= 1) {
      if (dgt < off) tot[dgt] = tot[dgt + off] > tot[dgt] ? tot[dgt + off] : tot[dgt];
      __syncthreads();
    }
    if (dgt == 0) {
      const uint32_t span = tot[0] > kmin ? tot[0] - kmin : 0u;
      pass_info[v] = (span >> (3 * kDigitBits)) ? 4u : 3u;
    }
  }
}

// The view's LAST pass (pass index pass_info[v] - 1: known on the device only) writes the ids to
// `sorted_idx` instead of the ping-pong buffer and also sorted_rect[pos] = rects[id] (the tile rects in
// depth order for the binning kernels): the id is in a register here, so the gather's loads fly under
// the ranking instead of being a kernel of their own (57 us of dependent 8-byte gathers).  MAYBE_LAST =
// this launch can be a view's last pass (pass index >= 2).
template <bool IOTA_VALS, bool MAYBE_LAST = false>
__global__ void __launch_bounds__(kSortThreads)
sort_scatter_kernel(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                    uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                    const uint32_t* __restrict__ block_hist, int G, int nblk, int shift,
                    uint32_t kmin, int pass, const uint32_t* __restrict__ pass_info,
                    const uint32_t* __restrict__ n_vis, const uint2* __restrict__ rects = nullptr,
                    uint2* __restrict__ sorted_rect = nullptr, uint32_t* __restrict__ sorted_idx = nullptr) {
  constexpr int NW = kSortThreads / kWave;
  // per-wave running digit counts during the ranking, then (in place) the global start of (wave, digit)
  __shared__ uint32_t cnt[NW][kDigits];
  const int t = threadIdx.x;
  int v, blk;
  view_minor_block(blk, v);
  const int w = t >> 6, lane = t & 63;
  const int limit = IOTA_VALS ? G : (int)n_vis[v];
  if (blk * kSortChunk >= limit) return;
  bool last = false;
  if (MAYBE_LAST) {
    const int np = (int)pass_info[v];
    if (pass >= np) return;                         // (a 3-pass view has no fourth pass)
    last = pass == np - 1;
  }
  {   // (16-byte stores: 8192 words by 1024 threads)
    uint4* z = reinterpret_cast<uint4*>(&cnt[0][0]);
    for (int i = t; i < NW * kDigits / 4; i += kSortThreads) z[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();

  const size_t vo = (size_t)v * G;
  const int start = blk * kSortChunk + w * (kSortChunk / NW);
  uint32_t key[kSortItems], val[kSortItems], rank[kSortItems];
  uint2 rc[MAYBE_LAST ? kSortItems : 1];
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
  // the block's digit offsets (used after the ranking by threads 0..kDigits-1; every thread loads)
  const uint32_t digit_base = block_hist[((size_t)v * nblk + blk) * kDigits + (t & kDigitMask)];
  if (MAYBE_LAST) {
    // unconditional (a load under `if (last)` is not speculated: it would be issued -- and waited for --
    // at the join, instead of flying under the ranking); wasted only in pass 2 of a 4-pass view
#pragma unroll
    for (int i = 0; i < kSortItems; ++i) rc[i] = rects[vo + val[i]];
  }
#pragma unroll
  for (int i = 0; i < kSortItems; ++i) {
    const int p = start + i * kWave + lane;
    bool valid = p < limit;
    if (IOTA_VALS) valid = valid && key[i] != kCulledKey;
    const uint32_t dg = digit_of(key[i], kmin, shift);
    uint64_t mask = __ballot(valid);
#pragma unroll
    for (int b = 0; b < kDigitBits; ++b) {
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
  // Round 6: the block's keys leave through LDS in block-sorted order.  Scattered straight from the ranking
  // lanes, every 4-byte store went to a (block, digit) run of its own -- ~8 keys per run with 512 digits -- and
  // the memory side wrote 245 MB for 35 MB of keys and values per pass (WRITE_SIZE, profiles/r6_c2_pmc_traffic.json:
  // 3.7 TB/s of partial sectors, which is what a pass waited for).  Now: per digit the waves' local offsets and the
  // block total, an exclusive scan over the digits, every item staged at its position in the block's sorted order,
  // and thread t writes staged items t, t + 1024, ...: a run of one digit leaves as consecutive lanes of one store.
  __shared__ uint32_t s_key[kSortChunk];
  __shared__ uint32_t s_val[kSortChunk];
  __shared__ uint16_t s_dig[kSortChunk];
  __shared__ uint32_t s_lstart[kDigits];      // first block-sorted position of the digit
  __shared__ uint32_t s_gbase[kDigits];       // global position of the digit's first key of this block
  __shared__ uint32_t s_scan[kDigits];
  uint32_t dtot = 0;
  if (t < kDigits) {
#pragma unroll
    for (int i = 0; i < NW; ++i) { const uint32_t c = cnt[i][t]; cnt[i][t] = dtot; dtot += c; }   // offset inside the digit
    s_gbase[t] = digit_base;
    s_scan[t] = dtot;
  }
  __syncthreads();
  uint32_t dx = dtot;
  for (int off = 1; off < kDigits; off <<= 1) {
    const uint32_t y = (t < kDigits && t >= off) ? s_scan[t - off] : 0u;
    __syncthreads();
    if (t < kDigits) { dx += y; s_scan[t] = dx; }
    __syncthreads();
  }
  if (t < kDigits) s_lstart[t] = dx - dtot;
  const uint32_t n_valid = s_scan[kDigits - 1];          // (read after the last barrier of the scan)
  __syncthreads();
  const bool is_last = MAYBE_LAST && last;
#pragma unroll
  for (int i = 0; i < kSortItems; ++i) {
    if (rank[i] != kCulledKey) {
      const uint32_t dg = digit_of(key[i], kmin, shift);
      const uint32_t in_digit = cnt[w][dg] + rank[i];
      const uint32_t lp = s_lstart[dg] + in_digit;
      s_key[lp] = key[i]; s_val[lp] = val[i]; s_dig[lp] = (uint16_t)dg;
      if (MAYBE_LAST) { if (is_last) sorted_rect[vo + s_gbase[dg] + in_digit] = rc[i]; }   // (8 bytes: from the lane that gathered it)
    }
  }
  __syncthreads();
  uint32_t* const vout = is_last ? sorted_idx : vals_out;
#pragma unroll
  for (int i = 0; i < kSortItems; ++i) {
    const uint32_t lp = (uint32_t)t + (uint32_t)i * kSortThreads;
    if (lp < n_valid) {
      const uint32_t dg = s_dig[lp];
      const uint32_t pos = s_gbase[dg] + (lp - s_lstart[dg]);
      if (!is_last) keys_out[vo + pos] = s_key[lp];     // nobody reads the keys after the last pass
      vout[vo + pos] = s_val[lp];
    }
  }
  // (the last pass's 8-byte rects still leave from the lanes that gathered them: gathering them HERE, by the lanes
  //  that write the runs, exposes the gather's latency at the end of the block -- 0.32 instead of 0.277 ms for the
  //  sort, profiles/r6_ab_sort.txt)
}

void launch_sort(const PsRasterDesc& d, uint32_t* keys_a, uint32_t* keys_b, uint32_t* vals_a,
                 uint32_t* vals_b, uint32_t* block_hist, uint32_t* pass_info, uint32_t* sorted_idx,
                 const uint2* rects, uint2* sorted_rect, uint32_t* n_vis, hipStream_t st) {
  const Dims m = make_dims(d);
  dim3 grid(m.nblk, m.V), block(kSortThreads);
  // every visible key is the bit pattern of a depth above the near-cull distance
  uint32_t kmin = 0;
  if (d.near_cull > 0.f) { const float nc = d.near_cull; memcpy(&kmin, &nc, 4); }
  uint32_t* kin = keys_a; uint32_t* kout = keys_b;
  uint32_t* vin = nullptr; uint32_t* vout = vals_b;
  for (int pass = 0; pass < kMaxPasses; ++pass) {
    const int shift = pass * kDigitBits;
    if (pass == 0)
      hipLaunchKernelGGL(sort_hist_kernel<true>, grid, block, 0, st, kin, block_hist, m.G, m.nblk,
                         shift, kmin, pass, n_vis, pass_info);
    else
      hipLaunchKernelGGL(sort_hist_kernel<false>, grid, block, 0, st, kin, block_hist, m.G, m.nblk,
                         shift, kmin, pass, n_vis, pass_info);
    hipLaunchKernelGGL(sort_scan_kernel, dim3(m.V), dim3(kDigits), 0, st, block_hist, m.nblk,
                       pass == 0 ? n_vis : (uint32_t*)nullptr, kmin, pass, pass_info);
    if (pass == 0)
      hipLaunchKernelGGL((sort_scatter_kernel<true, false>), grid, block, 0, st, kin, vin, kout,
                         vout, block_hist, m.G, m.nblk, shift, kmin, pass, pass_info, n_vis,
                         (const uint2*)nullptr, (uint2*)nullptr, (uint32_t*)nullptr);
    else if (pass == 1)
      hipLaunchKernelGGL((sort_scatter_kernel<false, false>), grid, block, 0, st, kin, vin, kout,
                         vout, block_hist, m.G, m.nblk, shift, kmin, pass, pass_info, n_vis,
                         (const uint2*)nullptr, (uint2*)nullptr, (uint32_t*)nullptr);
    else
      hipLaunchKernelGGL((sort_scatter_kernel<false, true>), grid, block, 0, st, kin, vin, kout,
                         vout, block_hist, m.G, m.nblk, shift, kmin, pass, pass_info, n_vis, rects,
                         sorted_rect, sorted_idx);
    // ping-pong: keys a<->b ; vals: (iota)->b->a->b->a (a view's last pass writes sorted_idx instead)
    uint32_t* tk = kin; kin = kout; kout = tk;
    vin = vout;
    vout = (vout == vals_b) ? vals_a : vals_b;
  }
}

}  // namespace ps
