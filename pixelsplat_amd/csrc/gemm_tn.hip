// C[M][N] = sum_k A[k][M] * B[k][N]   (A^T B, fp32, k = rays: 57 344 at the paper config)
//
// The weight gradients of the folded matrices of path (A) (pixelsplat_amd/epipolar.py) are
// this shape: a small output (592 x 128 or 128 x 592) over a very long k.  hipBLASLt's fp32
// heuristics run it at 28 TFLOP/s (268 us); the fix is all split-k.
//
// One wave64 owns a 128x128 output tile over a slice of k and keeps it in 256 accumulator
// registers (16 x v_mfma_f32_32x32x2_f32).  Both operands are k-major, which is exactly what
// the MFMA wants: lane l holds A[k = l/32][m(l%32)], so operands go from global memory straight
// into the MFMA, no LDS, no transposes.  One float4 load per lane brings 4 different m (or n)
// for the same k; MFMA (ja, jb) takes component ja of the A vector and jb of the B vector, i.e.
// it computes the rows {4 i + ja} x columns {4 i' + jb} of the tile -- an interleaved sub-tile,
// undone when the tile is stored.  Two 16-byte loads feed 16 MFMAs (1024 MFMA cycles); a single
// wave per SIMD keeps eight k-steps of operands in flight in registers.  The four waves of a
// block (= the four SIMDs of a CU) sum their tiles through LDS, the blocks' partial tiles are
// summed in a fixed order by a second kernel: deterministic.
//
// Measured (tools/ab_gemm_tn.sh, tools/prof_gemm_tn.sh; 576 x 128 x 57 344): 121 -> 102 us per
// call, partial kernel 85 us = 100 TFLOP/s of the 157 (the MFMAs alone are 61 us at 2.4 GHz,
// 10 % of them on the zero half of the fifth 128-row tile).
#include "raster_common.h"

namespace ps {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kGemmTile = 128;
constexpr int kGemmWaves = 1024;     // one 512-register wave per SIMD: 256 CUs x 4
constexpr int kGemmPrefetch = 6; // k-steps of operands in flight per wave

constexpr int kXcds = 8;

// A/B switch (tools/ab_gemm_tn.sh): 0 = round-1 mapping (block -> (tile, chunk) in launch order),
// 1 = XCD-aware, 2 = XCD-aware + 8 k-steps of operands in flight, 3 (default) = 2 with four
// waves per block that sum their partial tiles through LDS before anything is written
static int gemm_tn_variant() {
  static const int v = [] { const char* e = getenv("PS_GEMM_TN_VARIANT"); return e ? atoi(e) : 3; }();
  return v;
}

// chunks = k-slices, one per wave; a block of `wpb` waves owns `wpb` consecutive slices of one
// tile and writes ONE partial tile: `parts` partial matrices reach the reduce kernel
struct GemmPlan { int tiles_m, tiles_n, chunks, rows, wpb, parts, blocks; };
// k is cut so that tiles x chunks fills the machine once
static GemmPlan gemm_tn_plan(int M, int N, int K) {
  GemmPlan p;
  const int variant = gemm_tn_variant();
  p.wpb = variant >= 3 ? 4 : 1;
  p.tiles_m = (M + kGemmTile - 1) / kGemmTile;
  p.tiles_n = (N + kGemmTile - 1) / kGemmTile;
  const int tiles = p.tiles_m * p.tiles_n;
  int groups = kGemmWaves / p.wpb / tiles;
  // XCD-aware launch: whole rounds of 8 blocks per tile, so that the grid fits the machine once
  if (variant > 0 && groups >= kXcds) groups -= groups % kXcds;
  if (groups < 1) groups = 1;
  const int chunks = groups * p.wpb;
  int rows = (K + chunks - 1) / chunks;
  rows = (rows + 1) & ~1;                      // k advances in steps of 2
  if (rows < 16) rows = 16;
  p.rows = rows;
  p.chunks = (K + rows - 1) / rows;
  p.parts = (p.chunks + p.wpb - 1) / p.wpb;
  p.blocks = variant > 0 ? tiles * ((p.parts + kXcds - 1) / kXcds) * kXcds : tiles * p.parts;
  return p;
}

// Workgroups go to the 8 XCDs round-robin (block x -> XCD x % 8) and every XCD has its own L2.
// XCD_AWARE puts all tiles of one k-slice on the same XCD, next to each other in time: the
// operand a tile row / column shares is then fetched from HBM once and hit in that L2 by the
// others (M = 576, N = 128: the 29 MB of B were read by five different XCDs before).
//
// WPB = 4: the four waves of a block (one per SIMD, 512 registers each) take four consecutive
// k-slices of the same tile and sum their 64 KB accumulator tiles through LDS before anything
// is written (reduce-scatter, see the epilogue): 4x less partial-sum traffic (59 -> 15 MB written
// and read again at the paper shape), and the store is spread over all four waves.
template <int PF, bool XCD_AWARE, int WPB>
__global__ void __launch_bounds__(kWave * WPB)
gemm_tn_partial_kernel(int M, int N, int K, int rows, int parts, const float* __restrict__ A, int lda,
                       const float* __restrict__ B, int ldb, float* __restrict__ partial,
                       float* __restrict__ colpart) {
  const int tiles_n = (N + kGemmTile - 1) / kGemmTile, tiles_m = (M + kGemmTile - 1) / kGemmTile;
  int tile, part;
  if (XCD_AWARE) {
    const int xcd = blockIdx.x % kXcds, j = blockIdx.x / kXcds;
    tile = j % (tiles_m * tiles_n);
    part = (j / (tiles_m * tiles_n)) * kXcds + xcd;
    if (part >= parts) return;
  } else {
    tile = blockIdx.x % (tiles_m * tiles_n);
    part = blockIdx.x / (tiles_m * tiles_n);
  }
  const int m0 = (tile / tiles_n) * kGemmTile, n0 = (tile % tiles_n) * kGemmTile;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, kk = lane >> 5, q = lane & 31;
  const int chunk = part * WPB + wave;
  // a slice past the end (the last block of a tile may have fewer than WPB) has no rows: its
  // loads are clamped and zeroed, its accumulators stay 0, it still takes part in the barriers
  const int k_begin = min(chunk * rows, K), k_end = min(k_begin + rows, K);
  // this lane's 4 consecutive m (n): m0 + 4 q + j
  const int ma = m0 + 4 * q, nb = n0 + 4 * q;
  // M and N are multiples of 4, so a lane's four columns are all inside or all outside the
  // matrix.  Lanes outside read column 0 instead: their products land in tile rows / columns
  // >= M / N, which the store skips, so nothing has to be zeroed for them.  Rows past the end of
  // the slice are read from a clamped address and zeroed WHEN USED (A only: 0 x b = 0) -- zeroing
  // at load time made every load's result needed right away, and the compiler answered by
  // sinking all loads of an unrolled trip to its end and draining them (vmcnt(0)) before the
  // next trip: 25 % of the wave's time was spent in s_waitcnt (SQ_WAIT_ANY).
  const int mac = ma < M ? ma : 0, nbc = nb < N ? nb : 0;
  const float* __restrict__ a_col = A + mac;
  const float* __restrict__ b_col = B + nbc;
  auto load_a = [=](int k) {
    return *reinterpret_cast<const float4*>(a_col + (size_t)min(k, K - 1) * lda);
  };
  auto load_b = [=](int k) {
    return *reinterpret_cast<const float4*>(b_col + (size_t)min(k, K - 1) * ldb);
  };

  floatx16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // register prefetch ring, PF k-steps (of 2 rows) deep: one wave per SIMD has nobody to hide an
  // HBM miss behind, so a slot is refilled right after its MFMAs were issued and is not looked
  // at again for PF - 1 steps (~1000 MFMA cycles each).  The scheduling barrier keeps that order.
  float4 ra[PF], rb[PF];
  // by-product: the column sums of A over this slice (the bias gradient sum_k dY[k][m] when
  // A = dY); four adds per k-step that issue under the MFMAs
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    ra[u] = load_a(k_begin + 2 * u + kk);
    rb[u] = load_b(k_begin + 2 * u + kk);
    // same issue order as in the loop: the compiler's s_waitcnt at the loop head is the merge of
    // both orders, and a shuffled prologue made it vmcnt(1) (drain the ring) instead of 2 PF - 2
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int ks = k_begin; ks < k_end; ks += 2 * PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const bool live = ks + 2 * u + kk < k_end;
      const float4 a = ra[u], b = rb[u];
      const float av[4] = {live ? a.x : 0.f, live ? a.y : 0.f, live ? a.z : 0.f, live ? a.w : 0.f};
      const float bv[4] = {b.x, b.y, b.z, b.w};
      cs.x += av[0]; cs.y += av[1]; cs.z += av[2]; cs.w += av[3];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
      ra[u] = load_a(ks + 2 * (u + PF) + kk);
      rb[u] = load_b(ks + 2 * (u + PF) + kk);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // column sums: the waves of the first tile column park [slice][k parity][M] partial sums
  if (colpart != nullptr && n0 == 0 && ma < M)
    *reinterpret_cast<float4*>(colpart + ((size_t)chunk * 2 + kk) * M + ma) = cs;

  // store of the tile rows {4 r + i}: MFMA (i, j), register e of lane l is row
  // 8 (e / 4) + 4 (l / 32) + e % 4, column l % 32 of the 32x32 block, i.e. tile row 4 * row + i,
  // tile column 4 * col + j
  float* out = partial + (size_t)part * M * N;
  auto store_rows = [&](const int i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = m0 + 4 * (8 * (e / 4) + 4 * kk + (e % 4)) + i;
      if (m >= M) continue;
      const int n = n0 + 4 * q;
      float* dst = out + (size_t)m * N + n;
      if (n + 3 < N) {
        *reinterpret_cast<float4*>(dst) =
            make_float4(acc[i][0][e], acc[i][1][e], acc[i][2][e], acc[i][3][e]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (n + j < N) dst[j] = acc[i][j][e];
      }
    }
  };

  if (WPB == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) store_rows(i);
  } else {
    // Reduce-scatter through LDS, all four waves at once: wave w ends up with the sum of the four
    // partial tiles for the rows {4 r + w} (its accumulators acc[w][*]) and stores only those.
    // Every wave keeps the same (register, lane) -> tile element map, so the exchange is
    // element-wise: a wave parks the three row groups it does not own, [owner][source slot], and
    // the owner adds them to its own in source order 0, 1, 2, 3 -- the same order for every
    // element, whichever wave owns it: bit-reproducible.  Two halves (j = 0, 1 then 2, 3) of
    // 96 KB each, because 192 KB do not fit the 160 KB of LDS.
    extern __shared__ float4 xch[];   // [owner 4][slot 3][block 2][e4 4][lane 64] float4 = 96 KB
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h) __syncthreads();         // the first half has been read by everybody
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (wave == i) continue;
        const int slot = wave < i ? wave : wave - 1;
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            const floatx16& c = acc[i][2 * h + j2];
            xch[(((i * 3 + slot) * 2 + j2) * 4 + e4) * kWave + lane] =
                make_float4(c[4 * e4], c[4 * e4 + 1], c[4 * e4 + 2], c[4 * e4 + 3]);
          }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (wave != i) continue;
#pragma unroll
        for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
          for (int e4 = 0; e4 < 4; ++e4) {
            floatx16& c = acc[i][2 * h + j2];
            float4 t[3];
#pragma unroll
            for (int sl = 0; sl < 3; ++sl)
              t[sl] = xch[(((i * 3 + sl) * 2 + j2) * 4 + e4) * kWave + lane];
            const float4 own = make_float4(c[4 * e4], c[4 * e4 + 1], c[4 * e4 + 2], c[4 * e4 + 3]);
            // partial of source wave s: own for s == i, else slot s (s < i) or s - 1 (s > i)
            auto src = [&](int s_) { return s_ == i ? own : t[s_ < i ? s_ : s_ - 1]; };
            const float4 p0 = src(0), p1 = src(1), p2 = src(2), p3 = src(3);
            c[4 * e4] = ((p0.x + p1.x) + p2.x) + p3.x;
            c[4 * e4 + 1] = ((p0.y + p1.y) + p2.y) + p3.y;
            c[4 * e4 + 2] = ((p0.z + p1.z) + p2.z) + p3.z;
            c[4 * e4 + 3] = ((p0.w + p1.w) + p2.w) + p3.w;
          }
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (wave == i) store_rows(i);
  }
}

// 64 elements x 4 chunk groups per block; each thread keeps 4 loads in flight, the four
// group sums are combined in a fixed order.  Blocks past `blocks1` do the same for the second
// job (the column sums), so that the by-product costs no launch.
__global__ void __launch_bounds__(256)
gemm_tn_reduce_kernel(int n_elem, int n_chunks, const float* __restrict__ partial,
                      float* __restrict__ C, int blocks1, int n_elem2, int n_chunks2,
                      const float* __restrict__ partial2, float* __restrict__ C2) {
  __shared__ float part[4][64];
  int blk = blockIdx.x;
  if (blk >= blocks1) {
    blk -= blocks1; n_elem = n_elem2; n_chunks = n_chunks2; partial = partial2; C = C2;
  }
  const int el = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int e = blk * 64 + el;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < n_elem) {
    int c = g;
    for (; c + 12 < n_chunks; c += 16) {
      s0 += partial[(size_t)c * n_elem + e];
      s1 += partial[(size_t)(c + 4) * n_elem + e];
      s2 += partial[(size_t)(c + 8) * n_elem + e];
      s3 += partial[(size_t)(c + 12) * n_elem + e];
    }
    for (; c < n_chunks; c += 4) s0 += partial[(size_t)c * n_elem + e];
  }
  part[g][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (g == 0 && e < n_elem) C[e] = (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
}

size_t gemm_tn_workspace_bytes(int M, int N, int K) {
  const GemmPlan p = gemm_tn_plan(M, N, K);
  // partial tiles + the column-sum partials [slice][k parity][M] (slices incl. the empty ones of
  // the last block)
  return ((size_t)p.parts * M * N + (size_t)p.parts * p.wpb * 2 * M) * sizeof(float);
}

int launch_gemm_tn(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                   float* C, float* colsum_a, float* workspace, hipStream_t st) {
  // float4 loads/stores need 16-byte aligned rows
  if ((lda & 3) || (ldb & 3) || (M & 3) || (N & 3) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) ||
      ((uintptr_t)workspace & 15))
    return PS_ERR_UNSUPPORTED;
  const GemmPlan p = gemm_tn_plan(M, N, K);
  const dim3 grid((unsigned)p.blocks), block(kWave * p.wpb);
  float* colpart = colsum_a ? workspace + (size_t)p.parts * M * N : nullptr;
#define PS_GEMM_TN_LAUNCH(PF, XCD, WPB)                                                            \
  do {                                                                                             \
    constexpr size_t lds = WPB > 1 ? (size_t)4 * 3 * 2 * 4 * kWave * sizeof(float4) : 0;           \
    static const bool lds_ok_ = (lds > 0 && hipFuncSetAttribute(                                   \
        (const void*)gemm_tn_partial_kernel<PF, XCD, WPB>,                                         \
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess); (void)lds_ok_;       \
    hipLaunchKernelGGL((gemm_tn_partial_kernel<PF, XCD, WPB>), grid, block, lds, st, M, N, K,      \
                       p.rows, p.parts, A, lda, B, ldb, workspace, colpart);                       \
  } while (0)
  switch (gemm_tn_variant()) {
    case 0: PS_GEMM_TN_LAUNCH(kGemmPrefetch, false, 1); break;
    case 1: PS_GEMM_TN_LAUNCH(kGemmPrefetch, true, 1); break;
    case 2: PS_GEMM_TN_LAUNCH(8, true, 1); break;
    default: PS_GEMM_TN_LAUNCH(8, true, 4);
  }
#undef PS_GEMM_TN_LAUNCH
  const int blocks1 = (M * N + 63) / 64, blocks2 = colsum_a ? (M + 63) / 64 : 0;
  hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)(blocks1 + blocks2)), dim3(256), 0, st,
                     M * N, p.parts, workspace, C, blocks1, M, p.parts * p.wpb * 2,
                     (const float*)colpart, colsum_a);
  return PS_OK;
}

}  // namespace ps
