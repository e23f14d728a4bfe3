// Wall-clock VALU issue rate on gfx950: N dependent-free v_fma_f32 / v_pk_fma_f32 / v_mul_f32 per
// wave, `w` waves per SIMD on every SIMD, timed with hipEvents.  Prints wave-instructions per
// SIMD per microsecond and the implied cycles per instruction at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 1.0f + 0.001f * (threadIdx.x + i);
  float c = 0.999f + 1e-6f * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
        if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
        if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&v[i & 6]) : "v"(*(double*)&v[i & 6]));
        if (OP == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        if (OP == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(c));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
void run(const char* name, int w) {
  const int blocks = 256 * w, iters = 20000;
  float* out; hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double inst_per_simd = (double)iters * 64.0 * w;   // each block puts one wave on each SIMD
  printf("%-14s waves/SIMD=%d  %.3f ms  %.1f inst/us/SIMD  -> %.2f cycles/inst at 2.4 GHz\n", name, w, ms,
         inst_per_simd / (ms * 1e3), 2400.0 * ms * 1e3 / inst_per_simd);
  hipFree(out);
}
int main() {
  for (int w : {1, 2, 4, 8}) { run<0>("v_fma_f32", w); run<1>("v_mul_f32", w); run<2>("v_pk_fma_f32", w); run<3>("v_exp_f32", w); run<4>("v_cndmask", w); }
  return 0;
}
