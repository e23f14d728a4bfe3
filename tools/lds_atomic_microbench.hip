// LDS float-atomic throughput on gfx950: cycles per wave-level ds_add_f32 for several
// address patterns.  hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_microbench.hip -o /tmp/ldsmb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void __launch_bounds__(1024) bench(const int* __restrict__ idx, int n_idx, int iters,
                                              int mode, float* out, long long* cyc) {
  extern __shared__ float tile[];
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) tile[i] = 0.f;
  __syncthreads();
  const int base = idx[(blockIdx.x * blockDim.x + threadIdx.x) % n_idx];
  const long long t0 = clock64();
  int a = base;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int addr = (a + k * 4099) & 32767;
      if (mode == 0) atomicAdd(&tile[addr], 1.0f);
      else { tile[addr] += 1.0f; }
    }
    a = (a * 5 + 1) & 32767;
    if (mode == 2) a = base;
  }
  __syncthreads();
  const long long t1 = clock64();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  float s = 0.f;
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) s += tile[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  const int threads_opts[] = {64, 256, 1024};
  const char* pat_names[] = {"consecutive", "stride65", "random", "same_address", "pairs_same", "8_same"};
  for (int pat = 0; pat < 6; ++pat) {
    std::vector<int> h(1024);
    for (int i = 0; i < 1024; ++i) {
      switch (pat) {
        case 0: h[i] = i; break;
        case 1: h[i] = (i * 65) & 32767; break;
        case 2: h[i] = (int)((i * 2654435761u) >> 17) & 32767; break;
        case 3: h[i] = 7; break;
        case 4: h[i] = (i / 2) * 3; break;
        case 5: h[i] = (i / 8) * 3; break;
      }
    }
    int* d_idx; float* d_out; long long* d_cyc;
    hipMalloc(&d_idx, 4096); hipMalloc(&d_out, 1024 * 1024 * 4); hipMalloc(&d_cyc, 8 * 1024);
    hipMemcpy(d_idx, h.data(), 4096, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; mode += 2) {   // 0: pseudo-random walk, 2: fixed addresses
      for (int th : threads_opts) {
        const int iters = 2000;
        hipFuncSetAttribute((const void*)bench, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
        hipLaunchKernelGGL(bench, dim3(256), dim3(th), 131072, 0, d_idx, 1024, iters, mode, d_out, d_cyc);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(bench, dim3(256), dim3(th), 131072, 0, d_idx, 1024, iters, mode, d_out, d_cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double wave_instrs = (double)iters * 8 * (th / 64);
        printf("%-13s mode %d threads %4d: %.3f ms  -> %.1f ns per wave-atomic per CU (%.1f clk @2.4GHz), %.2f lanes/clk\n",
               pat_names[pat], mode, th, ms, ms * 1e6 / wave_instrs, ms * 1e6 / wave_instrs * 2.4,
               64.0 / (ms * 1e6 / wave_instrs * 2.4));
      }
    }
    hipFree(d_idx); hipFree(d_out); hipFree(d_cyc);
  }
  return 0;
}
