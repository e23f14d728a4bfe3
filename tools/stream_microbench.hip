// HBM streaming-read ceiling on MI355X for the access shapes the preprocess kernels use.
// hipcc --offload-arch=gfx950 -O3 tools/stream_microbench.hip -o tools/streammb
#include <hip/hip_runtime.h>
#include <cstdio>

// each block reads `per_block` consecutive float4, `unroll` loads in flight per thread
template <int UNROLL>
__global__ void read_kernel(const float4* __restrict__ src, size_t per_block, float* out) {
  const float4* p = src + (size_t)blockIdx.x * per_block;
  float acc = 0.f;
  for (size_t i = threadIdx.x; i < per_block; i += (size_t)blockDim.x * UNROLL) {
    float4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const size_t j = i + (size_t)u * blockDim.x;
      v[u] = j < per_block ? p[j] : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  if (acc == 123.456f) out[0] = acc;
}

__global__ void write_kernel(float4* __restrict__ dst, size_t per_block) {
  float4* p = dst + (size_t)blockIdx.x * per_block;
  for (size_t i = threadIdx.x; i < per_block; i += blockDim.x) p[i] = make_float4(1, 2, 3, 4);
}

template <typename F>
float time_ms(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}

int main() {
  const size_t bytes = (size_t)826 << 20;
  float4* buf; float* out; hipMalloc(&buf, bytes); hipMalloc(&out, 64);
  hipMemset(buf, 0, bytes);
  const size_t n4 = bytes / 16;
  struct Cfg { int threads; size_t per_block4; int lds; } cfgs[] = {
      {64, 1200, 0}, {64, 1200, 19200}, {256, 4800, 0}, {256, 4800, 76800}, {256, 1200, 0},
      {1024, 19200, 0}, {256, 65536, 0}, {512, 65536, 0}};
  for (auto c : cfgs) {
    const unsigned blocks = (unsigned)(n4 / c.per_block4);
    float ms1 = time_ms([&] { hipLaunchKernelGGL(read_kernel<1>, dim3(blocks), dim3(c.threads), c.lds, 0, buf, c.per_block4, out); });
    float ms4 = time_ms([&] { hipLaunchKernelGGL(read_kernel<4>, dim3(blocks), dim3(c.threads), c.lds, 0, buf, c.per_block4, out); });
    float ms8 = time_ms([&] { hipLaunchKernelGGL(read_kernel<8>, dim3(blocks), dim3(c.threads), c.lds, 0, buf, c.per_block4, out); });
    float msw = time_ms([&] { hipLaunchKernelGGL(write_kernel, dim3(blocks), dim3(c.threads), c.lds, 0, buf, c.per_block4); });
    printf("threads %4d  bytes/block %7zu  lds %6d  blocks %7u : read u1 %.2f TB/s  u4 %.2f  u8 %.2f | write %.2f TB/s\n",
           c.threads, c.per_block4 * 16, c.lds, blocks, bytes / ms1 / 1e9, bytes / ms4 / 1e9, bytes / ms8 / 1e9, bytes / msw / 1e9);
  }
  return 0;
}
