// Issue-cost table of the instruction classes the blend kernels are made of, on gfx950, at the
// occupancy those kernels run at (4 waves per SIMD; also 1, 2, 8): every SIMD of the chip loaded,
// 8 independent chains per wave, wall clock (hipEvents) AND the shader clock the kernel saw
// (s_memtime at both ends of wave 0), so that cycles per instruction do not depend on the
// nominal 2.4 GHz.  The last rows are the forward and backward per-quadrant blocks of
// raster_tiles.hip restated instruction for instruction (dependent chain, as in the kernel).
//   hipcc --offload-arch=gfx950 -O3 tools/issue_model.hip -o tools/issue_model && tools/issue_model
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void __launch_bounds__(256) k(float* out, uint64_t* ticks, int iters) {
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = 1.0f + 0.001f * (threadIdx.x + i);
  float c = 0.999f + 1e-6f * threadIdx.x, e = 0.5f + 1e-6f * threadIdx.x;
  asm volatile("s_mov_b32 vcc_lo, 0x55555555\n s_mov_b32 vcc_hi, 0x55555555\n s_mov_b32 s10, 0x33333333\n s_mov_b32 s11, 0x33333333" ::: "vcc", "s10", "s11");
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        double& d = *(double*)&v[2 * i];
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(e));
        if (OP == 1) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
        if (OP == 2) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(e));
        if (OP == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(d) : "v"(*(double*)&v[(2 * i + 2) & 14]));
        if (OP == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d) : "v"(*(double*)&v[(2 * i + 2) & 14]));
        if (OP == 5) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d) : "v"(*(double*)&v[(2 * i + 2) & 14]));
        if (OP == 6) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(v[i]));
        if (OP == 7) asm volatile("v_rcp_f32_e32 %0, %0" : "+v"(v[i]));
        if (OP == 8) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(c));
        if (OP == 9) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(v[i]) : "v"(c));
        if (OP == 10) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" :: "v"(v[i]), "v"(c) : "vcc");
        if (OP == 11) asm volatile("v_cmp_lt_f32_e64 s[10:11], %0, %1" :: "v"(v[i]), "v"(c) : "s10", "s11");
        if (OP == 12) asm volatile("v_min_f32_e32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
        if (OP == 13) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v[i]));
        if (OP == 14) asm volatile("v_permlane32_swap_b32_e32 %0, %1" : "+v"(v[i]), "+v"(v[8 + i]));
        if (OP == 15) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(e));
        if (OP == 16) asm volatile("v_sub_f32_e32 %0, 1.0, %0" : "+v"(v[i]));
        if (OP == 17) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(v[i]) : "v"(c));
        if (OP == 18) asm volatile("v_cndmask_b32_e64 %0, -|%0|, %1, s[10:11]" : "+v"(v[i]) : "v"(c));
        if (OP == 19) asm volatile("v_cmp_lt_f32_e32 vcc, %1, %2\n v_cndmask_b32_e32 %0, %0, %2, vcc" : "+v"(v[i]) : "v"(v[8 + i]), "v"(c) : "vcc");
        if (OP == 20) asm volatile("v_fmac_f32_e32 %0, %1, %2\n v_mul_f32_e32 %3, %3, %1" : "+v"(v[i]), "+v"(c), "+v"(e), "+v"(v[8 + i]));
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + c + e;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

// the forward per-quadrant block (raster_tiles.hip process_entry), one dependent chain per trip:
// 4 independent "quadrants" per trip to mimic the kernel's unrolled k loop without branches
template <int WHICH>
__global__ void __launch_bounds__(256) blk(float* out, uint64_t* ticks, int iters) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  const float l = threadIdx.x & 63;
  f2 pix[4]; float Ts[4], C2[4]; f2 C01[4]; uint32_t last[4];
  float T[4], acc2[4], g2[4], Tfb[4]; f2 acc01[4], g01[4]; uint32_t nc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    pix[q] = f2{(float)((int)l & 7) + 8.f * (q & 1), (float)((int)l >> 3) + 8.f * (q >> 1)};
    Ts[q] = 1.f; C2[q] = 0.f; C01[q] = f2{0.f, 0.f}; last[q] = 0;
    T[q] = 0.3f; acc2[q] = 0.f; acc01[q] = f2{0.f, 0.f}; g01[q] = f2{0.1f, 0.2f}; g2[q] = 0.3f; Tfb[q] = 0.01f; nc[q] = 1u << 30;
  }
  float gx = 7.5f + 0.01f * blockIdx.x, gy = 8.5f, A = -0.02f, B = 0.001f, Cq = -0.03f, o = 0.5f;
  float c0 = 0.5f, c1 = 0.25f, c2 = 0.125f;
  const float amax = 0.99f, amin = 1.f / 255.f, tmin = 1e-4f;
  float Mx = 0, My = 0, Mxx = 0, Mxy = 0, Myy = 0, s_op = 0, s_b = 0; f2 s_rg = {0, 0};
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const uint32_t hidx = (uint32_t)it + 1u;
    gx += 1e-4f;            // keep the compiler from hoisting the geometry
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f2 dd = f2{gx, gy} - pix[q];
      const f2 bc = f2{B, Cq} * f2{dd.y, dd.y};
      const float pw = fmaf(dd.x, fmaf(A, dd.x, bc.x), dd.y * bc.y);
      if (WHICH == 0) {
        const float alpha = fminf(amax, o * __builtin_amdgcn_exp2f(pw));
        const bool ok = (pw <= 0.f) & (alpha >= amin);
        const float ale = ok ? alpha : 0.f;
        float Tp;
        asm("v_max_f32 %0, 0, %1" : "=v"(Tp) : "v"(Ts[q]));
        const f2 tw = f2{Tp, Tp} * f2{1.f - ale, ale};
        const bool stop = tw.x < tmin;
        const float wgt = stop ? 0.f : tw.y;
        Ts[q] = stop ? -fabsf(Ts[q]) : tw.x;
        C01[q] = f2{c0, c1} * f2{wgt, wgt} + C01[q];
        C2[q] = fmaf(c2, wgt, C2[q]);
        last[q] = (ok & !stop) ? hidx : last[q];
      } else {
        const float Gv = __builtin_amdgcn_exp2f(pw);
        float alpha;
        asm("v_min_f32 %0, %1, %2" : "=v"(alpha) : "s"(amax), "v"(o * Gv));
        const bool ok = (hidx <= nc[q]) & (pw <= 0.f) & (alpha >= amin);
        const float ale = ok ? alpha : 0.f;
        const float rcp = __builtin_amdgcn_rcpf(1.f - ale);
        const float Tn = T[q] * rcp;
        const f2 d01 = f2{c0, c1} - acc01[q];
        const float d2 = c2 - acc2[q];
        const f2 t01 = d01 * g01[q];
        float dLda = fmaf(d2, g2[q], t01.x + t01.y) * Tn;
        dLda = fmaf(Tfb[q], rcp, dLda);
        const float dch = ale * Tn;
        s_rg = f2{dch, dch} * g01[q] + s_rg;
        s_b = fmaf(dch, g2[q], s_b);
        const float gda = ok ? Gv * dLda : 0.f;
        s_op += gda;
        const float qq = o * gda;
        const f2 qxy = f2{qq, qq} * dd;
        Mx += qxy.x; My += qxy.y;
        Mxx = fmaf(qxy.x, dd.x, Mxx); Mxy = fmaf(qxy.x, dd.y, Mxy);
        Myy = fmaf(qxy.y, dd.y, Myy);
        T[q] = Tn * 0.999f + 0.0003f;      // (keeps T bounded over the loop; one extra fma)
        acc01[q] = f2{ale, ale} * d01 + acc01[q];
        acc2[q] = fmaf(ale, d2, acc2[q]);
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float s = Mx + My + Mxx + Mxy + Myy + s_op + s_b + s_rg.x + s_rg.y;
#pragma unroll
  for (int q = 0; q < 4; ++q) s += Ts[q] + C2[q] + C01[q].x + C01[q].y + (float)last[q] + T[q] + acc2[q] + acc01[q].x + acc01[q].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}



static float* g_out; static uint64_t* g_ticks;
// The per-quadrant blocks as they sit in the kernel: one quadrant after the other behind wave-uniform
// branches (no interleaving across quadrants), entry values in registers.
//   MODE 0: the long form as the compiler emits it (packed pairs + selects)
//   MODE 1: the short form, hand-written VOP2 sequence (raster_tiles.hip, round 3)
//   MODE 2: the short form left to the compiler (packed pairs where it wants them)
//   MODE 3: the hand-written short form with two quadrants interleaved instruction by instruction
template <int MODE>
__global__ void __launch_bounds__(256) seq(float* out, uint64_t* ticks, int iters, uint32_t qmask) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63;
  float pxf[4], pyf[4], Ts[4], C0[4], C1[4], C2[4]; uint32_t last[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    pxf[q] = (float)(lane & 7) + 8.f * (q & 1); pyf[q] = (float)(lane >> 3) + 8.f * (q >> 1);
    Ts[q] = 1.f; C0[q] = C1[q] = C2[q] = 0.f; last[q] = 0;
  }
  float gx = 7.5f + 0.01f * blockIdx.x, gy = 8.5f, A = -0.02f, B = 0.001f, Cq = -0.03f, o = 0.3f;
  float c0 = 0.5f, c1 = 0.25f, c2 = 0.125f;
  const float amax = 0.99f, amin = 1.f / 255.f, tmin = 1e-4f;
  const uint32_t qm = __builtin_amdgcn_readfirstlane(qmask);
  uint64_t cb[4] = {0, 0, 0, 0};
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const uint32_t hidx = (uint32_t)it + 1u;
    gx += 1e-4f;
    if (MODE == 3) {
#pragma unroll
      for (int q = 0; q < 4; q += 2) {
        if (qm & (1u << q)) {
          float dx, dy, t, u, dx2, dy2, t2, u2;
          asm volatile(
              "v_sub_f32 %[dx], %[gx], %[px]\n v_sub_f32 %[dx2], %[gx], %[px2]\n"
              "v_sub_f32 %[dy], %[gy], %[py]\n v_sub_f32 %[dy2], %[gy], %[py2]\n"
              "v_mul_f32 %[t], %[B], %[dy]\n v_mul_f32 %[t2], %[B], %[dy2]\n"
              "v_mul_f32 %[u], %[C], %[dy]\n v_mul_f32 %[u2], %[C], %[dy2]\n"
              "v_fmac_f32 %[t], %[A], %[dx]\n v_fmac_f32 %[t2], %[A], %[dx2]\n"
              "v_mul_f32 %[u], %[u], %[dy]\n v_mul_f32 %[u2], %[u2], %[dy2]\n"
              "v_fmac_f32 %[u], %[dx], %[t]\n v_fmac_f32 %[u2], %[dx2], %[t2]\n"
              "v_exp_f32 %[u], %[u]\n v_exp_f32 %[u2], %[u2]\n"
              "v_mul_f32 %[u], %[o], %[u]\n v_mul_f32 %[u2], %[o], %[u2]\n"
              "v_cmp_le_f32 vcc, %[amin], %[u]\n v_cmp_le_f32 s[10:11], %[amin], %[u2]\n"
              "v_cndmask_b32 %[u], 0, %[u], vcc\n v_cndmask_b32 %[u2], 0, %[u2], s[10:11]\n"
              "v_cndmask_b32 %[last], %[last], %[hidx], vcc\n v_cndmask_b32 %[last2], %[last2], %[hidx], s[10:11]\n"
              "v_mul_f32 %[t], %[T], %[u]\n v_mul_f32 %[t2], %[T2], %[u2]\n"
              "v_sub_f32 %[dx], 1.0, %[u]\n v_sub_f32 %[dx2], 1.0, %[u2]\n"
              "v_mul_f32 %[T], %[T], %[dx]\n v_mul_f32 %[T2], %[T2], %[dx2]\n"
              "v_fmac_f32 %[c0], %[r], %[t]\n v_fmac_f32 %[c02], %[r], %[t2]\n"
              "v_fmac_f32 %[c1], %[g], %[t]\n v_fmac_f32 %[c12], %[g], %[t2]\n"
              "v_fmac_f32 %[c2], %[b], %[t]\n v_fmac_f32 %[c22], %[b], %[t2]\n"
              : [dx] "=&v"(dx), [dy] "=&v"(dy), [t] "=&v"(t), [u] "=&v"(u), [T] "+v"(Ts[q]),
                [c0] "+v"(C0[q]), [c1] "+v"(C1[q]), [c2] "+v"(C2[q]), [last] "+v"(last[q]),
                [dx2] "=&v"(dx2), [dy2] "=&v"(dy2), [t2] "=&v"(t2), [u2] "=&v"(u2), [T2] "+v"(Ts[q + 1]),
                [c02] "+v"(C0[q + 1]), [c12] "+v"(C1[q + 1]), [c22] "+v"(C2[q + 1]), [last2] "+v"(last[q + 1])
              : [gx] "v"(gx), [gy] "v"(gy), [A] "v"(A), [B] "v"(B), [C] "v"(Cq), [o] "v"(o), [r] "v"(c0),
                [g] "v"(c1), [b] "v"(c2), [hidx] "v"(hidx), [px] "v"(pxf[q]), [py] "v"(pyf[q]),
                [px2] "v"(pxf[q + 1]), [py2] "v"(pyf[q + 1]), [amin] "s"(amin)
              : "vcc", "s10", "s11");
        }
      }
      continue;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (qm & (1u << q)) {
        if (MODE == 1) {
          float dx, dy, t, u;
          asm volatile(
              "v_sub_f32 %[dx], %[gx], %[px]\n"
              "v_sub_f32 %[dy], %[gy], %[py]\n"
              "v_mul_f32 %[t], %[B], %[dy]\n"
              "v_mul_f32 %[u], %[C], %[dy]\n"
              "v_fmac_f32 %[t], %[A], %[dx]\n"
              "v_mul_f32 %[u], %[u], %[dy]\n"
              "v_fmac_f32 %[u], %[dx], %[t]\n"
              "v_exp_f32 %[u], %[u]\n"
              "s_nop 0\n"
              "v_mul_f32 %[u], %[o], %[u]\n"
              "v_cmp_le_f32 vcc, %[amin], %[u]\n"
              "v_cndmask_b32 %[u], 0, %[u], vcc\n"
              "v_cndmask_b32 %[last], %[last], %[hidx], vcc\n"
              "v_mul_f32 %[t], %[T], %[u]\n"
              "v_sub_f32 %[dx], 1.0, %[u]\n"
              "v_mul_f32 %[T], %[T], %[dx]\n"
              "v_fmac_f32 %[c0], %[r], %[t]\n"
              "v_fmac_f32 %[c1], %[g], %[t]\n"
              "v_fmac_f32 %[c2], %[b], %[t]\n"
              : [dx] "=&v"(dx), [dy] "=&v"(dy), [t] "=&v"(t), [u] "=&v"(u), [T] "+v"(Ts[q]),
                [c0] "+v"(C0[q]), [c1] "+v"(C1[q]), [c2] "+v"(C2[q]), [last] "+v"(last[q])
              : [gx] "v"(gx), [gy] "v"(gy), [A] "v"(A), [B] "v"(B), [C] "v"(Cq), [o] "v"(o), [r] "v"(c0),
                [g] "v"(c1), [b] "v"(c2), [hidx] "v"(hidx), [px] "v"(pxf[q]), [py] "v"(pyf[q]),
                [amin] "s"(amin)
              : "vcc");
        } else {
          const f2 dd = f2{gx, gy} - f2{pxf[q], pyf[q]};
          const f2 bc = f2{B, Cq} * f2{dd.y, dd.y};
          const float pw = fmaf(dd.x, fmaf(A, dd.x, bc.x), dd.y * bc.y);
          if (MODE == 2) {
            const float alpha = o * __builtin_amdgcn_exp2f(pw);
            const bool ok = alpha >= amin;
            const float ale = ok ? alpha : 0.f;
            const f2 tw = f2{Ts[q], Ts[q]} * f2{1.f - ale, ale};
            Ts[q] = tw.x;
            C0[q] = fmaf(c0, tw.y, C0[q]); C1[q] = fmaf(c1, tw.y, C1[q]); C2[q] = fmaf(c2, tw.y, C2[q]);
            last[q] = ok ? hidx : last[q];
          } else if (MODE == 4) {
            // round 5: the long form WITHOUT scalar operations on vector-written compare masks (the two
            // s_and_b64 of the shipped block): nested selects for `ok`, "took" asked of the weight
            const float alpha = fminf(amax, o * __builtin_amdgcn_exp2f(pw));
            const float a1 = pw <= 0.f ? alpha : 0.f;
            const float ale = a1 >= amin ? a1 : 0.f;
            float Tp;
            asm("v_max_f32 %0, 0, %1" : "=v"(Tp) : "v"(Ts[q]));
            const f2 tw = f2{Tp, Tp} * f2{1.f - ale, ale};
            const bool stop = tw.x < tmin;
            const float wgt = stop ? 0.f : tw.y;
            Ts[q] = stop ? -fabsf(Ts[q]) : tw.x;
            C0[q] = fmaf(c0, wgt, C0[q]); C1[q] = fmaf(c1, wgt, C1[q]); C2[q] = fmaf(c2, wgt, C2[q]);
            last[q] = wgt > 0.f ? hidx : last[q];
          } else {
            const float alpha = fminf(amax, o * __builtin_amdgcn_exp2f(pw));
            const bool ok = (pw <= 0.f) & (alpha >= amin);
            const float ale = ok ? alpha : 0.f;
            float Tp;
            asm("v_max_f32 %0, 0, %1" : "=v"(Tp) : "v"(Ts[q]));
            const f2 tw = f2{Tp, Tp} * f2{1.f - ale, ale};
            const bool stop = tw.x < tmin;
            const float wgt = stop ? 0.f : tw.y;
            Ts[q] = stop ? -fabsf(Ts[q]) : tw.x;
            C0[q] = fmaf(c0, wgt, C0[q]); C1[q] = fmaf(c1, wgt, C1[q]); C2[q] = fmaf(c2, wgt, C2[q]);
            last[q] = (ok & !stop) ? hidx : last[q];
            if (MODE == 5) {
              // round 5's reverted "quadrant contributed" accumulation: one compare + scalar s_cmp / s_cselect / s_or
              const uint64_t bit = __builtin_amdgcn_ballot_w64(wgt > 0.f) != 0ull ? (1ull << (it & 63)) : 0ull;
              const uint64_t nb = cb[q] | bit;
              cb[q] = ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(nb >> 32)) << 32) |
                      (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)nb);
            }
          }
        }
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) s += Ts[q] + C0[q] + C1[q] + C2[q] + (float)last[q] + (float)(uint32_t)(cb[q] ^ (cb[q] >> 32));
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

// The backward's per-contributing-entry reduction as it sits in raster_tiles.hip (wave_sum9_partials: four
// v_permlane32_swap + two v_permlane16_swap folds, nine DPP adds, then the 8-lane partial sums staged in LDS by
// eight lanes) and, every 32 entries, the finalising read of the staged rows by lane j (six 16-byte LDS reads,
// the adds, three 16-byte stores to the entry's gradient slot).
__global__ void __launch_bounds__(256) red(float* out, uint64_t* ticks, int iters) {
  __shared__ float stage_all[4][32][3][8];
  float (*stage)[3][8] = stage_all[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  float a = 0.01f * lane, b = 0.02f, c = 0.03f, d = 0.04f, e = 0.05f, f = 0.06f, g = 0.07f, h = 0.08f, i = 0.09f;
  float4* slots = reinterpret_cast<float4*>(out) + (size_t)(blockIdx.x * 256 + threadIdx.x) * 3;
  float acc = 0.f;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    a += 1e-3f; e += 1e-3f; i += 1e-3f;
    auto f32 = [](float x, float y) {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
      return __uint_as_float(r[0]) + __uint_as_float(r[1]); };
    auto f16 = [](float x, float y) {
      const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
      return __uint_as_float(r[0]) + __uint_as_float(r[1]); };
    float r1 = f16(f32(a, b), f32(c, d)), r2 = f16(f32(e, f), f32(g, h)), r3 = i;
    asm volatile(
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_add_f32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_add_f32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_add_f32_dpp %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_add_f32_dpp %2, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "s_nop 1\n"
        : "+v"(r1), "+v"(r2), "+v"(r3));
    if ((lane & 7) == 7) {
      float* st = &stage[it & 31][0][lane >> 3];
      st[0] = r1; st[8] = r2; st[16] = r3;
    }
    if ((it & 31) == 31) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
      if (lane < 32) {
        const float4* sp = reinterpret_cast<const float4*>(&stage[lane][0][0]);
        const float4 u0 = sp[0], u1 = sp[1], v0 = sp[2], v1 = sp[3], w0 = sp[4], w1 = sp[5];
        slots[0] = make_float4(u0.x + u0.y, u0.z + u0.w, u1.x + u1.y, u1.z + u1.w);
        slots[1] = make_float4(v0.x + v0.y, v0.z + v0.w, v1.x + v1.y, v1.z + v1.w);
        slots[2] = make_float4((w0.x + w0.y) + (w0.z + w0.w) + (w1.x + w1.y) + (w1.z + w1.w), 0.f, 0.f, 0.f);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
    acc += r1 + r2 + r3;
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}

// round-5 probe: the same nine sums WITHOUT the DPP stage and the exec-masked staging stores -- after the two
// permlane folds every lane writes its r1 / r2 (16 partial sums per value, one value per 16-lane row) and a
// 3-step-DPP r3 to LDS with plain stores; every kEnt entries lane (e, v) sums its value's partials from LDS
// (4 x 16-byte reads + 15 adds; 8 scalars for the ninth value), hands the total over through LDS, and lane e does
// the finalising arithmetic and the three 16-byte slot stores as before.
constexpr int kEnt = 5;
__global__ void __launch_bounds__(256) red2(float* out, uint64_t* ticks, int iters) {
  __shared__ float part_all[4][kEnt][3][64];
  __shared__ float fin_all[4][kEnt][12];
  float (*part)[3][64] = part_all[threadIdx.x >> 6];
  float (*fin)[12] = fin_all[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63;
  float a = 0.01f * lane, b = 0.02f, c = 0.03f, d = 0.04f, e = 0.05f, f = 0.06f, g = 0.07f, h = 0.08f, i = 0.09f;
  float4* slots = reinterpret_cast<float4*>(out) + (size_t)(blockIdx.x * 256 + threadIdx.x) * 3;
  float acc = 0.f;
  int slot_e = 0;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    a += 1e-3f; e += 1e-3f; i += 1e-3f;
    auto f32 = [](float x, float y) {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
      return __uint_as_float(r[0]) + __uint_as_float(r[1]); };
    auto f16 = [](float x, float y) {
      const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
      return __uint_as_float(r[0]) + __uint_as_float(r[1]); };
    const float r1 = f16(f32(a, b), f32(c, d)), r2 = f16(f32(e, f), f32(g, h));
    float r3 = i;
    asm volatile(
        "s_nop 1\n"
        "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
        "s_nop 1\n"
        : "+v"(r3));
    part[slot_e][0][lane] = r1; part[slot_e][1][lane] = r2; part[slot_e][2][lane] = r3;
    acc += r1 + r2 + r3;
    if (++slot_e == kEnt) {
      slot_e = 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
      if (lane < 8 * kEnt) {                     // lane = 8 * entry + value (values 0..7: rows of r1, r2)
        const int en = lane >> 3, v = lane & 7;
        const float4* sp = reinterpret_cast<const float4*>(&part[en][v >> 2][16 * (v & 3)]);
        const float4 p0 = sp[0], p1 = sp[1], p2 = sp[2], p3 = sp[3];
        fin[en][v] = ((p0.x + p0.y) + (p0.z + p0.w)) + ((p1.x + p1.y) + (p1.z + p1.w)) +
                     ((p2.x + p2.y) + (p2.z + p2.w)) + ((p3.x + p3.y) + (p3.z + p3.w));
      } else if (lane < 9 * kEnt) {              // the ninth value: eight 8-lane partials at lanes 7, 15, ..
        const int en = lane - 8 * kEnt;
        const float* sp = &part[en][2][7];
        fin[en][8] = ((sp[0] + sp[8]) + (sp[16] + sp[24])) + ((sp[32] + sp[40]) + (sp[48] + sp[56]));
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
      if (lane < kEnt) {
        const float4* sp = reinterpret_cast<const float4*>(&fin[lane][0]);
        const float4 u = sp[0], v = sp[1], w = sp[2];
        slots[0] = make_float4((-0.3f * u.x - 0.1f * u.z) * 128.f, (-0.2f * u.z - 0.1f * u.x) * 128.f, -0.5f * u.y, -0.5f * u.w);
        slots[1] = make_float4(-0.5f * v.x, v.z / 0.3f, v.y, v.w);
        slots[2] = make_float4(w.x, 0.f, 0.f, 0.f);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1 - t0;
}


// round-6 probe: what the backward's nine sums cost PER WAVE STEP when a wave64 is four
// independent 16-lane rows, each on its own (entry, 4x4 cell) pair -- the forward's layout (csrc/raster_cells.hip)
// carried over to the backward.  Reduction + staging + the finalising read every 32 steps, nothing else (the
// per-pixel block is the `blk<1>` row above: the same 64-lane instruction stream whatever the lanes hold).
//   MODE 0  nine values x four DPP row_shr adds (1, 2, 4, 8): the row total lands in lane 15 of each row, which
//           stages its nine sums with three exec-masked LDS stores
//   MODE 1  "fold" tree inside the row: two values share a register after each level (DPP adds with bank masks
//           for the 8- and 4-lane levels, quad_perm for the last two): 8 + 4 + 3 + 1 instructions for eight values,
//           4 for the ninth; the eight totals sit in the even lanes, one LDS store stages them
//   MODE 2  "systolic": lane p works on item t - p, the nine sums travel with their item from lane to lane
//           (v_add_f32_dpp row_shr:1, zero shifted in at lane 0) and arrive complete at lane 15, which stages
//           them; the accumulating FMAs of the block become multiply + DPP add (7 more instructions)
template <int MODE>
__global__ void __launch_bounds__(256) rowred(float* out, uint64_t* ticks, int iters) {
  __shared__ float stage_all[4][32][4][12];          // [step][row][9 sums]
  float (*stage)[4][12] = stage_all[threadIdx.x >> 6];
  const int lane = threadIdx.x & 63, row = lane >> 4, it16 = lane & 15;
  float a = 0.01f * lane, b = 0.02f, c = 0.03f, d = 0.04f, e = 0.05f, f = 0.06f, g = 0.07f, h = 0.08f, i = 0.09f;
  float4* slots = reinterpret_cast<float4*>(out) + (size_t)(blockIdx.x * 256 + threadIdx.x) * 6;
  float acc = 0.f;
  float s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0, s8 = 0;   // MODE 2: the travelling sums
  const uint64_t odd2 = 0xCCCCCCCCCCCCCCCCull;       // lanes with bit 1 set
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    a += 1e-3f; e += 1e-3f; i += 1e-3f;
    if (MODE == 0) {
      float v0 = a, v1 = b, v2 = c, v3 = d, v4 = e, v5 = f, v6 = g, v7 = h, v8 = i;
#define PS_STEP(N) \
      "v_add_f32_dpp %0, %0, %0 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
      "v_add_f32_dpp %1, %1, %1 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
      "v_add_f32_dpp %2, %2, %2 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
      "v_add_f32_dpp %3, %3, %3 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
      "v_add_f32_dpp %4, %4, %4 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
      "v_add_f32_dpp %5, %5, %5 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
      "v_add_f32_dpp %6, %6, %6 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
      "v_add_f32_dpp %7, %7, %7 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n" \
      "v_add_f32_dpp %8, %8, %8 row_shr:" #N " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      asm volatile("s_nop 1\n" PS_STEP(1) PS_STEP(2) PS_STEP(4) PS_STEP(8) "s_nop 1\n"
                   : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(v8));
#undef PS_STEP
      if (it16 == 15) {
        float4* st = reinterpret_cast<float4*>(&stage[it & 31][row][0]);
        st[0] = make_float4(v0, v1, v2, v3); st[1] = make_float4(v4, v5, v6, v7); st[2].x = v8;
      }
      acc += v0;
    } else if (MODE == 1) {
      float v0 = a, v1 = b, v2 = c, v3 = d, v4 = e, v5 = f, v6 = g, v7 = h, v8 = i, t1, t2;
      asm volatile(
          "s_nop 1\n"
          // level 1: (v0,v1) (v2,v3) (v4,v5) (v6,v7) -> v0 v2 v4 v6: lanes 0-7 first value, 8-15 second
          "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n"
          "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n"
          "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n"
          "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n"
          "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xc\n"
          "v_add_f32_dpp %2, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xc\n"
          "v_add_f32_dpp %4, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n"
          "v_add_f32_dpp %6, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xc\n"
          "v_add_f32_dpp %8, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xf\n"
          // level 2: (v0,v2) (v4,v6) -> v0 v4: banks 0, 2 from the first register, 1, 3 from the second
          "v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n"
          "v_add_f32_dpp %4, %4, %4 row_shl:4 row_mask:0xf bank_mask:0x5\n"
          "v_add_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa\n"
          "v_add_f32_dpp %4, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xa\n"
          "v_add_f32_dpp %8, %8, %8 row_ror:4 row_mask:0xf bank_mask:0xf\n"
          // level 3: (v0,v4) -> v0: lanes 0,1 of a quad from v0, lanes 2,3 from v4
          "v_add_f32_dpp %9, %0, %0 quad_perm:[2,3,2,3] row_mask:0xf bank_mask:0xf\n"
          "v_add_f32_dpp %10, %4, %4 quad_perm:[0,1,0,1] row_mask:0xf bank_mask:0xf\n"
          "v_add_f32_dpp %8, %8, %8 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
          "v_cndmask_b32 %0, %9, %10, %11\n"
          // level 4
          "s_nop 1\n"
          "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
          "v_add_f32_dpp %8, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
          "s_nop 1\n"
          : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), "+v"(v8),
            "=&v"(t1), "=&v"(t2)
          : "s"(odd2));
      // even lane 2 k of a row holds total number k' (a fixed permutation of 0..7); lane 1 stores the ninth
      if ((lane & 1) == 0) stage[it & 31][row][it16 >> 1] = v0;
      if (it16 == 1) stage[it & 31][row][8] = v8;
      acc += v0;
    } else {
      // the block's nine contributions (a .. i stand for them); seven of them were the product inside an FMA
      float m0 = b * a, m1 = c * a, m2 = d * a, m3 = f * e, m4 = g * e, m5 = h * e, m6 = i * a;
      asm volatile(
          "s_nop 1\n"
          "v_add_f32_dpp %0, %0, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %1, %1, %10 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %2, %2, %11 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %3, %3, %12 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %4, %4, %13 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %5, %5, %14 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %6, %6, %15 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %7, %7, %16 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %8, %8, %17 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3), "+v"(s4), "+v"(s5), "+v"(s6), "+v"(s7), "+v"(s8)
          : "v"(m0), "v"(m1), "v"(m2), "v"(m3), "v"(m4), "v"(m5), "v"(m6), "v"(a), "v"(e));
      if (it16 == 15) {
        float4* st = reinterpret_cast<float4*>(&stage[it & 31][row][0]);
        st[0] = make_float4(s0, s1, s2, s3); st[1] = make_float4(s4, s5, s6, s7); st[2].x = s8;
      }
      acc += s0;
    }
    if ((it & 31) == 31) {     // 32 steps x 4 rows = 128 (entry, cell) results: two per lane
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const float4* sp = reinterpret_cast<const float4*>(&stage[(lane >> 2) + 16 * u][lane & 3][0]);
        const float4 u0 = sp[0], u1 = sp[1], u2 = sp[2];
        slots[3 * u + 0] = make_float4((-0.3f * u0.x - 0.1f * u0.z) * 128.f, (-0.2f * u0.z - 0.1f * u0.x) * 128.f, -0.5f * u0.y, -0.5f * u0.w);
        slots[3 * u + 1] = make_float4(-0.5f * u1.x, u1.z / 0.3f, u1.y, u1.w);
        slots[3 * u + 2] = make_float4(u2.x, 0.f, 0.f, 0.f);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
    }
  }
  const uint64_t t1c = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) ticks[0] = t1c - t0;
}

template <typename K>
void timeit_q(const char* name, K kern, int w, int iters, uint32_t qmask, double evals_per_iter) {
  const int blocks = 256 * w;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, g_out, g_ticks, iters, qmask);
  hipEventRecord(a);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, g_out, g_ticks, iters, qmask);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double evals_per_simd = (double)iters * evals_per_iter * w;
  printf("%-52s w=%d qmask=%x  %8.3f ms  %.1f cyc per quadrant evaluation @2.4GHz\n", name, w, qmask, ms,
         2400.0 * ms * 1e3 / evals_per_simd);
}
template <typename K>
void timeit(const char* name, K kern, int w, double inst_per_iter, int iters) {
  const int blocks = 256 * w;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, g_out, g_ticks, iters);
  hipEventRecord(a);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, g_out, g_ticks, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  uint64_t ticks; hipMemcpy(&ticks, g_ticks, 8, hipMemcpyDeviceToHost);
  const double inst_per_simd = (double)iters * inst_per_iter * w;   // one wave of each block per SIMD
  const double mhz = (double)ticks / (ms * 1e3);                    // s_memtime ticks per microsecond
  printf("%-34s w=%d  %8.3f ms  memtime %7.1f MHz  %.2f cyc/inst @2.4GHz  %.2f ticks/inst\n", name, w, ms, mhz,
         2400.0 * ms * 1e3 / inst_per_simd, (double)ticks / inst_per_simd);
}

#define RUN(OP, NAME, N) for (int w : {1, 4, 8}) timeit(NAME, k<OP>, w, 32.0 * N, 6000)
int main(int argc, char** argv) {
  hipMalloc(&g_out, 256 * 8 * 256 * 4); hipMalloc(&g_ticks, 64);
  if (argc > 1 && argv[1][0] == 'b') {      // `issue_model b`: the backward's block and its per-entry reduction
    hipFree(g_out); hipMalloc(&g_out, (size_t)256 * 8 * 256 * 48 + 4096);
    for (int rep = 0; rep < 2; ++rep) {
      for (int w : {2, 4}) timeit("backward block x4 quadrants / trip", blk<1>, w, 4.0, 20000);
      for (int w : {2, 4}) timeit("backward 9-sum reduction + staging / entry", red, w, 1.0, 20000);
      for (int w : {2, 4}) timeit("r5 probe: folds + plain LDS stores, sums from LDS", red2, w, 1.0, 20000);
    }
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'r') {      // `issue_model r`: the backward's reduction per wave step on 16-lane rows
    hipFree(g_out); hipMalloc(&g_out, (size_t)256 * 8 * 256 * 96 + 4096);
    for (int rep = 0; rep < 2; ++rep) {
      for (int w : {4, 5}) timeit("backward block x4 quadrants / trip", blk<1>, w, 4.0, 20000);
      for (int w : {4, 5}) timeit("shipped: 9-sum reduction + staging / entry", red, w, 1.0, 20000);
      for (int w : {4, 5}) timeit("rows: 36 DPP row_shr adds + staging / wave step", rowred<0>, w, 1.0, 20000);
      for (int w : {4, 5}) timeit("rows: fold tree (20 DPP) + staging / wave step", rowred<1>, w, 1.0, 20000);
      for (int w : {4, 5}) timeit("rows: systolic sums (9 DPP + 7 mul) / wave step", rowred<2>, w, 1.0, 20000);
    }
    return 0;
  }
  if (argc > 1 && argv[1][0] == 'f') {      // `issue_model f`: the forward block's round-5 variants only
    for (int rep = 0; rep < 2; ++rep)
      for (int w : {4, 6, 8})
        for (uint32_t qmask : {0x1u, 0x3u}) {
          const double ev = __builtin_popcount(qmask);
          timeit_q("seq long form, shipped (2 s_and on compare masks)", seq<0>, w, 20000, qmask, ev);
          timeit_q("seq long form, no scalar op on a mask (r5 probe)", seq<4>, w, 20000, qmask, ev);
          timeit_q("seq long form + contributed-bit accumulation", seq<5>, w, 20000, qmask, ev);
        }
    return 0;
  }
  RUN(0, "v_fma_f32 (VOP3, 3 regs)", 1);  RUN(1, "v_mul_f32_e32", 1);  RUN(2, "v_fmac_f32_e32", 1);
  RUN(3, "v_pk_fma_f32", 1);  RUN(4, "v_pk_mul_f32", 1);  RUN(5, "v_pk_add_f32", 1);
  RUN(6, "v_exp_f32", 1);  RUN(7, "v_rcp_f32", 1);
  RUN(8, "v_cndmask_b32_e32 (vcc)", 1);  RUN(9, "v_cndmask_b32_e64 (sgpr pair)", 1);
  RUN(10, "v_cmp_lt_f32_e32 -> vcc", 1);  RUN(11, "v_cmp_lt_f32_e64 -> sgpr pair", 1);
  RUN(12, "v_min_f32_e32", 1);  RUN(13, "v_add_f32_dpp row_shr:1", 1);  RUN(14, "v_permlane32_swap", 1);
  RUN(15, "v_max3_f32", 1);  RUN(16, "v_sub_f32_e32 1.0 - x", 1);  RUN(17, "v_mov_b32", 1);
  RUN(18, "v_cndmask_e64 with -|x|", 1);  RUN(19, "v_cmp_e32 + v_cndmask_e32 pair", 2);
  RUN(20, "v_fmac_e32 + v_mul_e32 pair", 2);
  for (int w : {1, 2, 4, 8}) timeit("forward block x4 quadrants / trip", blk<0>, w, 4.0, 20000);
  for (int w : {1, 2, 4}) timeit("backward block x4 quadrants / trip", blk<1>, w, 4.0, 20000);
  for (int w : {1, 4, 8})
    for (uint32_t qmask : {0xFu, 0x5u, 0x1u}) {
      const double ev = __builtin_popcount(qmask);
      timeit_q("seq long form (compiler)", seq<0>, w, 20000, qmask, ev);
      timeit_q("seq short form (hand-written VOP2)", seq<1>, w, 20000, qmask, ev);
      timeit_q("seq short form (compiler)", seq<2>, w, 20000, qmask, ev);
      if (qmask != 0x1u) timeit_q("seq short form, two quadrants interleaved", seq<3>, w, 20000, qmask == 0xFu ? 0x5u : 0x1u, qmask == 0xFu ? 4.0 : 2.0);
    }
  return 0;
}
