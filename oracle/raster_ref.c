/*
 * ORACLE (test infrastructure, NOT product code).  See raster_ref_impl.inc for the
 * header: CPU restatement of the diff_gaussian_rasterization algorithm pixelSplat calls
 * at /root/reference/src/model/decoder/cuda_splatting.py:99-124.  PARITY UNPINNED.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int ps_oracle_par_bwd = 0;
void ps_oracle_parallel_backward(int on) { ps_oracle_par_bwd = on; }
#define PS_ATOMIC _Pragma("omp atomic")

#define REAL float
#define SUF _f32
#include "raster_ref_impl.inc"
#undef REAL
#undef SUF

#define REAL double
#define SUF _f64
#include "raster_ref_impl.inc"
#undef REAL
#undef SUF

/* ---- A.2 binning (precision independent: consumes fp32 depth bits) -----------------
 * For each visible Gaussian, for y in [ymin,ymax), x in [xmin,xmax) emit
 * key = (tile << 32) | bits(depth), value = gaussian index; stable ascending sort;
 * ranges[tile] = [first, last+1), empty tiles keep (0,0).
 */
typedef struct { uint64_t key; uint32_t idx; } PsPair;

static int pair_cmp(const void* a, const void* b) {
  const PsPair* x = (const PsPair*)a; const PsPair* y = (const PsPair*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1; /* stable: emission order = idx */
  return 0;
}

/* returns D (= sum tiles_touched). point_list must hold D entries; keys_out may be NULL */
int64_t ps_oracle_bin(int32_t G, int32_t H, int32_t W, const int32_t* radii, const int32_t* rect,
                      const float* depth, uint32_t* point_list, uint64_t* keys_out,
                      uint32_t* ranges /*[tiles,2]*/) {
  const int gx = (W + 15) / 16, gy = (H + 15) / 16;
  int64_t D = 0;
  for (int i = 0; i < G; ++i)
    if (radii[i] > 0)
      D += (int64_t)(rect[4 * i + 2] - rect[4 * i]) * (rect[4 * i + 3] - rect[4 * i + 1]);
  memset(ranges, 0, sizeof(uint32_t) * 2 * (size_t)gx * gy);
  if (D == 0) return 0;
  PsPair* pairs = (PsPair*)malloc(sizeof(PsPair) * (size_t)D);
  int64_t n = 0;
  for (int i = 0; i < G; ++i) {
    if (radii[i] <= 0) continue;
    uint32_t bits;
    memcpy(&bits, &depth[i], 4);
    for (int y = rect[4 * i + 1]; y < rect[4 * i + 3]; ++y)
      for (int x = rect[4 * i]; x < rect[4 * i + 2]; ++x) {
        pairs[n].key = ((uint64_t)(uint32_t)(y * gx + x) << 32) | bits;
        pairs[n].idx = (uint32_t)i;
        ++n;
      }
  }
  qsort(pairs, (size_t)D, sizeof(PsPair), pair_cmp);
  for (int64_t k = 0; k < D; ++k) {
    point_list[k] = pairs[k].idx;
    if (keys_out) keys_out[k] = pairs[k].key;
    const uint32_t tile = (uint32_t)(pairs[k].key >> 32);
    if (k == 0 || tile != (uint32_t)(pairs[k - 1].key >> 32)) ranges[2 * tile] = (uint32_t)k;
    if (k == D - 1 || tile != (uint32_t)(pairs[k + 1].key >> 32))
      ranges[2 * tile + 1] = (uint32_t)(k + 1);
  }
  free(pairs);
  return D;
}
