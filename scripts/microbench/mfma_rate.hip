// micro-benchmark: issue rate of v_mfma_f32_32x32x16_f16 for different accumulator patterns, 1 or 2 waves per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int PAT, bool BIG>
__global__ __launch_bounds__(512, 2) void k(float* out, long long* t, int iters) {
  if (BIG) asm volatile("v_mov_b32 v255, 0" ::: "v255");      // force a 256-register allocation per wave
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  f16x8 a0, a1, b0, b1, a2, a3, bb[8];
  for (int j = 0; j < 8; ++j) { a2[j] = (_Float16)(j + 0.125f * threadIdx.x); a3[j] = (_Float16)(3 - j); for (int i = 0; i < 8; ++i) bb[i][j] = (_Float16)(i + j + (threadIdx.x & 3)); }
  for (int j = 0; j < 8; ++j) { a0[j] = (_Float16)(threadIdx.x * 0.001f + j); a1[j] = (_Float16)(j * 0.5f); b0[j] = (_Float16)(0.25f * j); b1[j] = (_Float16)(threadIdx.x & 7); }
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    FENCE();
    if (PAT == 0) {          // 8 independent accumulators round-robin, 16 MFMAs
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) { MFMA(a0, b0, acc[i]); FENCE(); }
    } else if (PAT == 1) {   // chains of 2
#pragma unroll
      for (int i = 0; i < 8; ++i) { MFMA(a0, b0, acc[i]); FENCE(); MFMA(a1, b1, acc[i]); FENCE(); }
    } else if (PAT == 2) {   // chains of 4
#pragma unroll
      for (int i = 0; i < 4; ++i) { MFMA(a0, b0, acc[i]); FENCE(); MFMA(a1, b1, acc[i]); FENCE(); MFMA(a0, b1, acc[i]); FENCE(); MFMA(a1, b0, acc[i]); FENCE(); }
    } else if (PAT == 3) {   // one accumulator, 16 chained
#pragma unroll
      for (int i = 0; i < 16; ++i) { MFMA(a0, b0, acc[0]); FENCE(); }
    } else if (PAT == 4) {   // 2 accumulators alternating
#pragma unroll
      for (int i = 0; i < 8; ++i) { MFMA(a0, b0, acc[0]); FENCE(); MFMA(a1, b1, acc[1]); FENCE(); }
    } else if (PAT == 6) {   // K1p pattern: 2 A x 4 B operands, every MFMA a different accumulator, 16 MFMAs = 2 K-steps
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) { MFMA(a0, bb[nb], acc[nb]); FENCE(); MFMA(a1, bb[nb], acc[4 + nb]); FENCE(); }
    } else if (PAT == 7) {   // chains of 2 over K: (a0,b[nb]) then (a2,b[4+nb]) on the same accumulator
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        MFMA(a0, bb[nb], acc[nb]); FENCE(); MFMA(a2, bb[4 + nb], acc[nb]); FENCE();
        MFMA(a1, bb[nb], acc[4 + nb]); FENCE(); MFMA(a3, bb[4 + nb], acc[4 + nb]); FENCE();
      }
    } else if (PAT == 8) {   // same A for 4 consecutive MFMAs (A-stationary): a0 x b0..b3, then a1 x b0..b3; 8 accs
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) { MFMA(a0, bb[nb], acc[nb]); FENCE(); }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) { MFMA(a1, bb[nb], acc[4 + nb]); FENCE(); }
      }
    } else if (PAT == 9) {   // chains of 4 over K on one accumulator at a time, operands all different
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) {
        MFMA(a0, bb[nb], acc[nb]); FENCE(); MFMA(a1, bb[4 + nb], acc[nb]); FENCE(); MFMA(a2, bb[(nb + 1) & 3], acc[nb]); FENCE(); MFMA(a3, bb[4 + ((nb + 1) & 3)], acc[nb]); FENCE();
      }
    } else if (PAT == 5) {   // chains of 3 over 4 accumulators (K1h pattern) + 4 more
#pragma unroll
      for (int i = 0; i < 4; ++i) { MFMA(a0, b0, acc[i]); FENCE(); MFMA(a1, b0, acc[i]); FENCE(); MFMA(a0, b1, acc[i]); FENCE(); MFMA(a1, b1, acc[i]); FENCE(); }
    }
  }
  long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) t[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int PAT, bool BIG>
void run(const char* name, int threads, int blocks) {
  float* out; long long* t;
  hipMalloc(&out, 512 * 1024 * 4); hipMalloc(&t, 8 * 1024 * 8);
  hipMemset(t, 0, 8 * 1024 * 8);
  const int iters = 200;
  hipLaunchKernelGGL((k<PAT, BIG>), dim3(blocks), dim3(threads), 0, 0, out, t, iters);
  hipLaunchKernelGGL((k<PAT, BIG>), dim3(blocks), dim3(threads), 0, 0, out, t, iters);
  hipDeviceSynchronize();
  long long h[8]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  long long mx = 0; for (int i = 0; i < threads / 64; ++i) mx = h[i] > mx ? h[i] : mx;
  const int waves_per_simd = threads / 256;
  printf("%s %-46s threads %3d blocks %3d: %6.1f cycles per MFMA per SIMD (slowest wave %lld ticks)\n", BIG ? "[256 VGPR]" : "[small]   ", name, threads, blocks,
         (double)mx / (iters * 16.0 * waves_per_simd), mx);
  hipFree(out); hipFree(t);
}

int main() {
  for (int threads : {256, 512}) {
    run<0, false>("8 accumulators round-robin, same A/B", threads, 1);
    run<6, false>("K1p pattern: 2 A x 4 B, new acc every MFMA", threads, 1);
    run<8, false>("A-stationary: a0 x b0..3, a1 x b0..3", threads, 1);
    run<7, false>("chains of 2 over K, all operands differ", threads, 1);
    run<9, false>("chains of 4 over K, all operands differ", threads, 1);
  }
  return 0;
}
