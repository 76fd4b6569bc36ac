// Exact-fp32 K loop on the f32-input matrix cores (v_mfma_f32_32x32x2_f32), shared by the fixed-architecture kernel
// (hm_decoder.hip) and the any-architecture kernel (hm_decoder_any.hip).  A operand: weights pre-packed by
// hm_pack.hip: pack_stage as [row block][K group of 8][lane][4]; B operand: activations in LDS as X[k/4][q][k%4].
#pragma once
#include "hm_common.h"

namespace hm {

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

// One K-group (8 k) of the product for this wave's two row blocks: 16 exact fp32 MFMAs.
template <bool U0, bool U1>
__device__ __forceinline__ void mfma_group(f32x16 (&acc)[2][2], const f32x4& a0, const f32x4& a1, const f32x4& b0,
                                           const f32x4& b1) {
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (U0) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b0[t], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[t], b1[t], acc[0][1], 0, 0, 0);
    }
    if (U1) {
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b0[t], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[t], b1[t], acc[1][1], 0, 0, 0);
    }
  }
}

// K loop: weights (A, from L2) and activations (B, from LDS) are both fetched one K-group ahead into two statically
// named register sets (no rotation copies, so hipcc counts its waits instead of draining them), the pairs of groups
// run branch-free.  A group is 16 MFMAs x 16 passes = 1024 matrix-pipe cycles per wave, ample cover for both fetches.
template <bool U0, bool U1>
__device__ __forceinline__ void gemm_loop(f32x16 (&acc)[2][2], const f32x4* __restrict__ wp0,
                                          const f32x4* __restrict__ wp1, int n_kg,
                                          const f32x4* xs, int lane) {
  const int xo = (lane >> 5) * TQ + (lane & 31);
  const int last = n_kg - 1;
  f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  f32x4 b0, b1, d0, d1;
  if (U0) a0 = wp0[0];
  if (U1) a1 = wp1[0];
  b0 = xs[xo]; b1 = xs[xo + 32];
  int kg = 0;
  for (; kg + 2 <= n_kg; kg += 2) {
    const int k1 = kg + 1, k2 = (kg + 2 < n_kg) ? kg + 2 : last;
    if (U0) c0 = wp0[k1 * 64];
    if (U1) c1 = wp1[k1 * 64];
    d0 = xs[k1 * 2 * TQ + xo]; d1 = xs[k1 * 2 * TQ + xo + 32];
    __builtin_amdgcn_sched_barrier(0);
    mfma_group<U0, U1>(acc, a0, a1, b0, b1);
    __builtin_amdgcn_sched_barrier(0);
    if (U0) a0 = wp0[k2 * 64];
    if (U1) a1 = wp1[k2 * 64];
    b0 = xs[k2 * 2 * TQ + xo]; b1 = xs[k2 * 2 * TQ + xo + 32];
    __builtin_amdgcn_sched_barrier(0);
    mfma_group<U0, U1>(acc, c0, c1, d0, d1);
    __builtin_amdgcn_sched_barrier(0);
  }
  if (kg < n_kg) mfma_group<U0, U1>(acc, a0, a1, b0, b1);      // odd group count: the last group is already loaded
}

}  // namespace hm
