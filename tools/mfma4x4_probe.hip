// Lane layout and issue rate of v_mfma_f32_4x4x1_16b_f32 on gfx950 (16 independent 4x4 outer
// products per instruction: D_b[i][j] += A_b[i] * B_b[j]).  Prints, for every (output register r,
// lane l), which A lane and which B lane feed it, and the wall-clock cycles per instruction with
// 1 / 2 / 4 waves per SIMD (4 independent accumulators per wave).
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma4x4_probe.hip -o tools/mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout(float* out) {
  const int l = threadIdx.x;
  f32x4 z = {0.f, 0.f, 0.f, 0.f};
  const f32x4 da = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), 1.0f, z, 0, 0, 0);
  const f32x4 db = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)(l + 1), z, 0, 0, 0);
  for (int r = 0; r < 4; ++r) { out[(r * 64 + l) * 2] = da[r]; out[(r * 64 + l) * 2 + 1] = db[r]; }
}

__global__ void __launch_bounds__(256) rate(float* out, int iters) {
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float a = 1.0f + 0.001f * threadIdx.x, b = 0.999f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the same loop with VALU FMAs interleaved 1:1 in the SAME wave: does the wave's VALU stream hide
// behind its own MFMAs?
__global__ void __launch_bounds__(256) rate_mixed(float* out, int iters) {
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float v[4] = {1.f, 2.f, 3.f, 4.f};
  const float a = 1.0f + 0.001f * threadIdx.x, b = 0.999f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(b));
      }
  }
  float s = v[0] + v[1] + v[2] + v[3];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename K>
static void time_kernel(const char* name, K kern, int w) {
  const int blocks = 256 * w, iters = 20000;
  float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double inst = (double)iters * 16.0 * w;    // MFMAs per SIMD (one wave of each block per SIMD)
  printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f cycles per MFMA per SIMD at 2.4 GHz\n", name, w, ms,
         2400.0 * ms * 1e3 / inst);
  hipFree(out);
}

int main() {
  float* d; hipMalloc(&d, 4 * 64 * 2 * 4);
  hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, d);
  float h[4 * 64 * 2]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("D register r, lane l  <-  A lane, B lane (values are lane+1)\n");
  for (int r = 0; r < 4; ++r) {
    printf("r=%d:", r);
    for (int l = 0; l < 64; ++l) printf(" %d/%d", (int)h[(r * 64 + l) * 2] - 1, (int)h[(r * 64 + l) * 2 + 1] - 1);
    printf("\n");
  }
  for (int w : {1, 2, 4}) { time_kernel("mfma_f32_4x4x1_16b", rate, w); time_kernel("mfma 4x4x1 + v_fma 1:1", rate_mixed, w); }
  return 0;
}
