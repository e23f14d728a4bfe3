// Timing probe for the SPLAT-MAJOR tile backward VERDICT r4 (next #3b) asked to build or to show losing at equal
// work: one lane owns one list entry of a 64-entry bucket, the tile's 256 pixels stream through the wave -- lane l
// works on pixel t - l at step t, the pixel's running state (transmittance, composited colour . dL/dC) moves one
// lane up per step (DPP wave_shr:1), its constants (dL/dC, the background term, the final colour . dL/dC) are
// read from LDS by pixel index -- and the nine gradient sums of an entry stay in its lane's registers: no
// cross-lane reduction, no atomics.  256 + 63 steps per bucket.  The per-step arithmetic is the backward block of
// csrc/raster_tiles.hip in front-to-back form (same instruction classes: 1 exp, 1 rcp, the alpha tests, the nine
// accumulations); results are NOT checked -- this measures what the scheme costs per bucket at the occupancy the
// shipped kernel runs at or above (the probe needs 56 VGPRs: up to 8 waves per SIMD, which flatters it), every SIMD loaded.
//   hipcc --offload-arch=gfx950 -O3 tools/splat_major_probe.hip -o tools/splat_major_probe && tools/splat_major_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kPixels = 256, kWave = 64, kSteps = kPixels + kWave - 1;

__device__ __forceinline__ float wave_shr1(float v) {     // lane l receives lane l - 1's value (lane 0: its own)
  return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_uint(v), __float_as_uint(v), 0x138, 0xf, 0xf, false));
}

struct Entry { float gx, gy, A, B, C, o, c0, c1, c2; };

__global__ void __launch_bounds__(256, 4)
splat_major_kernel(const Entry* __restrict__ entries, const float4* __restrict__ pixel_const,
                   const float2* __restrict__ checkpoint, float* __restrict__ out, int buckets_per_wave) {
  __shared__ float4 pc[4][kPixels];      // per wave: (dL/dC r, g, b, T_final * bg . dL/dC) per pixel of the tile
  __shared__ float cgf[4][kPixels];      // final colour . dL/dC per pixel
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wave_global = blockIdx.x * 4 + w;
  float acc_out = 0.f;
  for (int b = 0; b < buckets_per_wave; ++b) {
    const size_t bucket = (size_t)wave_global * buckets_per_wave + b;
    // the tile's pixel constants -> LDS (once per bucket here; a tile's buckets could share them)
    for (int i = lane; i < kPixels; i += kWave) {
      const float4 c = pixel_const[(bucket & 1023) * kPixels + i];
      pc[w][i] = c;
      cgf[w][i] = c.x + c.y + c.z;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const Entry e = entries[bucket * kWave + lane];
    float Mx = 0, My = 0, Mxx = 0, Mxy = 0, Myy = 0, s_op = 0, s_r = 0, s_g = 0, s_b = 0;
    float T = 1.f, Pg = 0.f;             // state this lane hands to the next one
    const float2* ck = checkpoint + (bucket & 1023) * kPixels;
    for (int t = 0; t < kSteps; ++t) {
      const int p = t - lane;
      const bool active = (p >= 0) & (p < kPixels);
      const int pi = p & (kPixels - 1);
      // state in: from the previous lane; lane 0 starts a fresh pixel from the forward's checkpoint
      float Tin = wave_shr1(T), Pin = wave_shr1(Pg);
      if (lane == 0) { const float2 c = ck[pi]; Tin = c.x; Pin = c.y; }
      const float4 g = pc[w][pi];
      const float Cg = cgf[w][pi];
      const float px = (float)(pi & 15), py = (float)(pi >> 4);
      const float dx = e.gx - px, dy = e.gy - py;
      const float pw = fmaf(dx, fmaf(e.A, dx, e.B * dy), dy * (e.C * dy));
      const float G = __builtin_amdgcn_exp2f(pw);
      const float alpha = fminf(0.99f, e.o * G);
      const bool ok = active & (pw <= 0.f) & (alpha >= (1.f / 255.f));
      const float ale = ok ? alpha : 0.f;
      const float one = 1.f - ale;
      const float rcp = __builtin_amdgcn_rcpf(one);
      const float cg = fmaf(e.c2, g.z, fmaf(e.c1, g.y, e.c0 * g.x));
      const float wgt = ale * Tin;
      // dL/dalpha = T (c . g) - (suffix colour . g + T_final bg . g) / (1 - alpha)
      float dLda = Tin * cg - (Cg - Pin - wgt * cg) * rcp;
      dLda = fmaf(g.w, rcp, dLda);
      s_r = fmaf(wgt, g.x, s_r); s_g = fmaf(wgt, g.y, s_g); s_b = fmaf(wgt, g.z, s_b);
      const float q = ale * dLda;
      s_op += q;
      const float qx = q * dx, qy = q * dy;
      Mx += qx; My += qy;
      Mxx = fmaf(qx, dx, Mxx); Mxy = fmaf(qx, dy, Mxy); Myy = fmaf(qy, dy, Myy);
      Pg = fmaf(wgt, cg, Pin);
      T = Tin * one;
    }
    acc_out += Mx + My + Mxx + Mxy + Myy + s_op + s_r + s_g + s_b + T + Pg;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc_out;
}

int main() {
  const int waves = 256 * 4 * 4 * 4;          // 4 rounds of 4 waves on each of the 1024 SIMDs
  const int buckets_per_wave = 8;
  const size_t buckets = (size_t)waves * buckets_per_wave;
  std::vector<Entry> h(buckets * kWave);
  srand(1);
  auto u = [] { return (float)rand() / RAND_MAX; };
  for (auto& e : h) e = Entry{16 * u(), 16 * u(), -0.05f - 0.1f * u(), 0.02f * (u() - 0.5f), -0.05f - 0.1f * u(), 0.3f * u(), u(), u(), u()};
  std::vector<float4> hp(1024 * kPixels);
  for (auto& c : hp) c = make_float4(u() - 0.5f, u() - 0.5f, u() - 0.5f, 0.01f * u());
  std::vector<float2> hc(1024 * kPixels);
  for (auto& c : hc) c = make_float2(0.5f + 0.5f * u(), 0.1f * u());
  Entry* de; float4* dp; float2* dc; float* dout;
  hipMalloc(&de, h.size() * sizeof(Entry)); hipMalloc(&dp, hp.size() * sizeof(float4));
  hipMalloc(&dc, hc.size() * sizeof(float2)); hipMalloc(&dout, (size_t)waves * 64 * sizeof(float));
  hipMemcpy(de, h.data(), h.size() * sizeof(Entry), hipMemcpyHostToDevice);
  hipMemcpy(dp, hp.data(), hp.size() * sizeof(float4), hipMemcpyHostToDevice);
  hipMemcpy(dc, hc.data(), hc.size() * sizeof(float2), hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL(splat_major_kernel, dim3(waves / 4), dim3(256), 0, 0, de, dp, dc, dout, buckets_per_wave);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double simd_cycles = ms * 1e-3 * 2.4e9 * 1024.0 / (double)buckets;
    printf("rep %d: %zu buckets of 64 entries x 256 pixels in %.3f ms -> %.1f k SIMD cycles per bucket "
           "(%.1f cycles per pipeline step) at 2.4 GHz\n",
           rep, buckets, ms, simd_cycles / 1e3, simd_cycles / kSteps);
  }
  // the shipped pixel-major kernel, for scale: 1.63 ms for 13.9 M list entries of which 9.2 M reach a quadrant
  const double shipped = 1.63e-3 * 2.4e9 * 1024.0 / (13.94e6 / 64.0);
  printf("shipped tiles_backward at BASELINE configs[1]: %.1f k SIMD cycles per 64 LIST entries all-in (refine, blend, "
         "reduction, slots); a splat-major bucket holds 64 SURVIVING entries = 97 list entries\n", shipped / 1e3);
  return 0;
}
