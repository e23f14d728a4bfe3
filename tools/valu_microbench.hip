// Issue-cost microbenchmark for gfx950 VALU flavours used by the tile kernels.
// One wave per SIMD (256 threads/block, 1 block/CU), 8 independent chains per op.
// Prints cycles per wave-instruction (s_memtime ticks = shader cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void __launch_bounds__(256) bench(float* out, long long* cyc, int iters) {
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 1.0f + 0.001f * (threadIdx.x + i);
  float c = 0.999f + 1e-6f * threadIdx.x;
  unsigned long long mask = 0x5555555555555555ull;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (OP == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
        REP8(X)
#undef X
      } else if (OP == 1) {
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v[i]));
        REP8(X)
#undef X
      } else if (OP == 2) {
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v[i]));
        REP8(X)
#undef X
      } else if (OP == 3) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        REP8(X)
#undef X
      } else if (OP == 4) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
        REP8(X)
#undef X
      } else if (OP == 5) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "s"(mask));
        REP8(X)
#undef X
      } else if (OP == 6) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(c) : "vcc");
        REP8(X)
#undef X
      } else if (OP == 7) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&v[(i)&6]) : "v"(*(double*)&v[(i)&6]));
        X(0) X(2) X(4) X(6) X(0) X(2) X(4) X(6)
#undef X
      } else if (OP == 8) {
#define X(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
        REP8(X)
#undef X
      } else if (OP == 9) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
        REP8(X)
#undef X
      } else if (OP == 10) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[i]));
        REP8(X)
#undef X
      } else if (OP == 11) {
#define X(i) v[i] = __shfl_xor(v[i], 16);
        REP8(X)
#undef X
      } else if (OP == 12) {
#define X(i) asm volatile("v_cmp_lt_f32 %1, %0, %2" : "+v"(v[i]), "=s"(mask) : "v"(c));
        REP8(X)
#undef X
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + (float)(mask & 1);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd, iters = 2000;
  float* out; long long* cyc;
  hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
  hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto x : h) avg += x; avg /= blocks;
  const double n = (double)iters * 64.0 * (OP == 6 ? 2 : 1);
  // s_memtime counts at a fixed 100 MHz-ish REFCLK on some parts: report raw ticks too
  printf("%-28s waves/SIMD=%d  ticks/inst=%.3f  (ticks=%.0f)\n", name, waves_per_simd, avg / n, avg);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32", w);
    run<9>("v_mul_f32", w);
    run<8>("v_min_f32", w);
    run<7>("v_pk_fma_f32", w);
    run<3>("v_exp_f32", w);
    run<4>("v_rcp_f32", w);
    run<5>("v_cndmask (sgpr mask)", w);
    run<6>("v_cmp+v_cndmask (vcc)", w);
    run<12>("v_cmp -> sgpr", w);
    run<1>("v_add_f32_dpp row_shr:1", w);
    run<2>("v_add_f32_dpp row_bcast:15", w);
    run<10>("v_mov_dpp quad_perm", w);
    run<11>("__shfl_xor 16 (bpermute)", w);
  }
  return 0;
}
