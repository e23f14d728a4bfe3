// C[M][N] = sum_k A[k][M] * B[k][N]   (A^T B, fp32, k = rays: 57 344 at the paper config)
//
// The weight gradients of the folded matrices of path (A) (pixelsplat_amd/epipolar.py) are
// this shape: two small outputs (512x128, 128x512, 80x128, 128x80) over a very long k.
// hipBLASLt's fp32 heuristics run it at 28 TFLOP/s (268 us); the fix is all split-k.
//
// One wave64 owns a 128x128 output tile over a chunk of kGemmChunk rows and keeps it in 256
// accumulator registers (16 x v_mfma_f32_32x32x2_f32).  Both operands are k-major, which is
// exactly what the MFMA wants: lane l holds A[k = l/32][m(l%32)], so operands go from global
// memory straight into the MFMA, no LDS, no transposes.  One float4 load per lane brings 4
// different m (or n) for the same k; MFMA (ja, jb) takes component ja of the A vector and jb
// of the B vector, i.e. it computes the rows {4 i + ja} x columns {4 i' + jb} of the tile --
// an interleaved sub-tile, undone when the partial tile is stored.  Two 16-byte loads feed
// 16 MFMAs (1024 MFMA cycles); a single wave per SIMD keeps six k-steps of operands in
// flight in registers.  Partials are summed in a fixed order by a second kernel: deterministic.
#include "raster_common.h"

namespace ps {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kGemmTile = 128;
constexpr int kGemmWaves = 1024;     // one 512-register wave per SIMD: 256 CUs x 4
constexpr int kGemmPrefetch = 6;     // k-steps of operands in flight per wave

struct GemmPlan { int tiles_m, tiles_n, chunks, rows; };
// k is cut so that tiles x chunks fills the machine once
static GemmPlan gemm_tn_plan(int M, int N, int K) {
  GemmPlan p;
  p.tiles_m = (M + kGemmTile - 1) / kGemmTile;
  p.tiles_n = (N + kGemmTile - 1) / kGemmTile;
  const int tiles = p.tiles_m * p.tiles_n;
  int chunks = kGemmWaves / tiles;
  if (chunks < 1) chunks = 1;
  int rows = (K + chunks - 1) / chunks;
  rows = (rows + 1) & ~1;                      // k advances in steps of 2
  if (rows < 16) rows = 16;
  p.rows = rows;
  p.chunks = (K + rows - 1) / rows;
  return p;
}

__global__ void __launch_bounds__(kWave)
gemm_tn_partial_kernel(int M, int N, int K, int rows, const float* __restrict__ A, int lda,
                       const float* __restrict__ B, int ldb, float* __restrict__ partial) {
  const int tiles_n = (N + kGemmTile - 1) / kGemmTile, tiles_m = (M + kGemmTile - 1) / kGemmTile;
  const int tile = blockIdx.x % (tiles_m * tiles_n), chunk = blockIdx.x / (tiles_m * tiles_n);
  const int m0 = (tile / tiles_n) * kGemmTile, n0 = (tile % tiles_n) * kGemmTile;
  const int lane = threadIdx.x, kk = lane >> 5, q = lane & 31;
  const int k_begin = chunk * rows, k_end = min(k_begin + rows, K);
  // this lane's 4 consecutive m (n): m0 + 4 q + j
  const int ma = m0 + 4 * q, nb = n0 + 4 * q;
  // M and N are multiples of 4, so a lane's four columns are all inside or all outside.
  // Loads are branch-free (clamped address + select): with loads inside divergent branches the
  // compiler waits for vmcnt(0) before every MFMA group and the prefetch ring is useless.
  const bool a_in = ma < M, b_in = nb < N;
  const int mac = a_in ? ma : 0, nbc = b_in ? nb : 0;
  auto load4 = [k_end](const float* __restrict__ base, int ld, int k, int col, bool in) {
    const int kc = k < k_end ? k : k_end - 1;
    const float4 v = *reinterpret_cast<const float4*>(base + (size_t)kc * ld + col);
    const bool ok = in && k < k_end;
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
  };

  floatx16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // register prefetch ring, kGemmPrefetch k-steps (of 2 rows) deep: one wave per SIMD has
  // nobody to hide an HBM miss behind, the loads must be ~2 us ahead of their MFMAs
  float4 ra[kGemmPrefetch], rb[kGemmPrefetch];
#pragma unroll
  for (int u = 0; u < kGemmPrefetch; ++u) {
    const int k = k_begin + 2 * u + kk;
    ra[u] = load4(A, lda, k, mac, a_in);
    rb[u] = load4(B, ldb, k, nbc, b_in);
  }
  for (int ks = k_begin; ks < k_end; ks += 2 * kGemmPrefetch) {
#pragma unroll
    for (int u = 0; u < kGemmPrefetch; ++u) {
      const float4 a = ra[u], b = rb[u];
      const int k2 = ks + 2 * (u + kGemmPrefetch) + kk;
      ra[u] = load4(A, lda, k2, mac, a_in);
      rb[u] = load4(B, ldb, k2, nbc, b_in);
      // rows past k_end were loaded as zeros: the extra MFMAs of the last group add nothing
      const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  }
  // store: MFMA (i, j), register e of lane l is row 8 (e / 4) + 4 (l / 32) + e % 4, column
  // l % 32 of the 32x32 block, i.e. tile row 4 * row + i, tile column 4 * col + j
  float* out = partial + (size_t)chunk * M * N;
#pragma unroll
  for (int i = 0; i < 4; ++i)
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
}

// 64 elements x 4 chunk groups per block; each thread keeps 4 loads in flight, the four
// group sums are combined in a fixed order
__global__ void __launch_bounds__(256)
gemm_tn_reduce_kernel(int n_elem, int n_chunks, const float* __restrict__ partial,
                      float* __restrict__ C) {
  __shared__ float part[4][64];
  const int el = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
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
  return (size_t)p.chunks * M * N * sizeof(float);
}

int launch_gemm_tn(int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                   float* C, float* workspace, hipStream_t st) {
  // float4 loads/stores need 16-byte aligned rows
  if ((lda & 3) || (ldb & 3) || (M & 3) || (N & 3) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) ||
      ((uintptr_t)workspace & 15))
    return PS_ERR_UNSUPPORTED;
  const GemmPlan p = gemm_tn_plan(M, N, K);
  hipLaunchKernelGGL(gemm_tn_partial_kernel, dim3((unsigned)(p.tiles_m * p.tiles_n * p.chunks)),
                     dim3(kWave), 0, st, M, N, K, p.rows, A, lda, B, ldb, workspace);
  hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)((M * N + 63) / 64)), dim3(256), 0, st,
                     M * N, p.chunks, workspace, C);
  return PS_OK;
}

}  // namespace ps
