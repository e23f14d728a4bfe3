// Float-atomic throughput on gfx950: device (agent) scope vs workgroup scope into a per-XCD
// private copy (selected with HW_REG_XCC_ID), scattered vs wave-contiguous addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xF;
}

template <int MODE>  // 0: agent scattered, 1: agent contiguous, 2: wg-scope per-XCD contiguous, 3: wg-scope per-XCD scattered
__global__ void __launch_bounds__(256) k(float* buf, size_t n_elems, int iters, unsigned* xcc_seen) {
  const unsigned lane = threadIdx.x & 63;
  const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  unsigned x = xcc_id();
  if (threadIdx.x == 0) atomicOr(&xcc_seen[blockIdx.x % 64], 1u << x);
  float* base = (MODE >= 2) ? buf + (size_t)x * n_elems : buf;
  unsigned s = wave * 2654435761u + 12345u;
  for (int i = 0; i < iters; ++i) {
    s = s * 1664525u + 1013904223u;
    size_t idx;
    if (MODE == 0 || MODE == 3) {
      unsigned r = s ^ (lane * 2246822519u);
      r ^= r >> 15; r *= 2654435761u; r ^= r >> 13;
      idx = r % n_elems;                       // every lane a different random line
    } else {
      idx = ((size_t)(s % (n_elems / 128)) * 128) + lane * 2;   // 64 lanes -> 512 contiguous bytes
    }
    if (MODE >= 2) {
      __hip_atomic_fetch_add(base + idx, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (MODE == 2) __hip_atomic_fetch_add(base + idx + 1, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
      atomicAdd(base + idx, 1.0f);
      if (MODE == 1) atomicAdd(base + idx + 1, 1.0f);
    }
  }
}

__global__ void reduce8(const float* buf, size_t n, double* out) {
  double acc = 0;
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += buf[i];
  atomicAdd(out, acc);
}

template <int MODE>
void run(const char* name) {
  const size_t n = 7340032;  // 29 MB of floats (feature maps of config 2)
  const int copies = MODE >= 2 ? 8 : 1;
  float* buf; unsigned* seen; double* total;
  hipMalloc(&buf, n * copies * 4); hipMemset(buf, 0, n * copies * 4);
  hipMalloc(&seen, 64 * 4); hipMemset(seen, 0, 256); hipMalloc(&total, 8); hipMemset(total, 0, 8);
  const int blocks = 4096, iters = 512;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, n, 8, seen);
  hipMemset(buf, 0, n * copies * 4);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, n, iters, seen);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipLaunchKernelGGL(reduce8, dim3(1024), dim3(256), 0, 0, buf, n * copies, total);
  double h; hipMemcpy(&h, total, 8, hipMemcpyDeviceToHost);
  const double per_lane = (MODE == 1 || MODE == 2) ? 2.0 : 1.0;
  const double ops = (double)blocks * 256 * iters * per_lane;
  unsigned hs[64]; hipMemcpy(hs, seen, 256, hipMemcpyDeviceToHost);
  unsigned any = 0; for (int i = 0; i < 64; ++i) any |= hs[i];
  printf("%-44s %.3f ms  %.1f G atomics/s  sum_ok=%d  xcc_mask=0x%x\n", name, ms, ops / ms / 1e6,
         (int)(h == ops), any);
  hipFree(buf); hipFree(seen); hipFree(total);
}

int main() {
  run<0>("agent scope, scattered (1 line/lane)");
  run<1>("agent scope, wave-contiguous 512 B");
  run<3>("workgroup scope, per-XCD copy, scattered");
  run<2>("workgroup scope, per-XCD copy, contiguous");
  return 0;
}
