// micro-benchmark: the K loop of the plain-fp16 decoder kernel (csrc/hm_gemm_p.h, the product's own code) on its own:
// barrier / k_loop_p over 8 groups of 4 K-steps (one 512 x 512 stage of a 128-query tile) / barrier, 12 stages, streamed
// from per-wave weight streams shared by all workgroups.  Ablations by macro: -DABL_NOA (weight fetches replaced by
// register moves), -DABL_NOB (no LDS reads), both = the matrix pipe alone.  Launched with 512 threads (two waves per SIMD,
// the product) and 256 (one wave per SIMD: the four older waves alone).  Reports the K-loop end of every wave of
// workgroup 0 in stage 6, relative to the stage's first K-loop start (ideal: 8192 for a wave alone, 16384 for the pair).
//   hipcc --offload-arch=gfx950 -O3 -I hortimapping_amd/csrc [-DABL_NOA] [-DABL_NOB] [-DHM_P_CLUSTER=1] k1p_kloop.hip -o k1p_kloop
#include <hip/hip_runtime.h>
#include <stdio.h>
#if defined(ABL_NOA) || defined(ABL_NOB)
#define HM_P_ABLATE 1
#endif
#ifndef HM_P_AHEAD
#define HM_P_AHEAD 3
#endif
#include "hm_gemm_p.h"
using namespace hm_p;

__global__ __launch_bounds__(512, 2) void k(float* out, long long* t, const char* wts, int steps_per_wave, int stages, int rnd, int n_grp) {
  __shared__ f16x8 xp[64 * TQP];
#ifdef LIKE_KERNEL
  __shared__ float sc[3072];
  __shared__ float bl[9 * 512];
  sc[threadIdx.x] = 0.f; bl[threadIdx.x] = 1.f;
#endif
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 64 * TQP; i += blockDim.x) {
    f16x8 v;
    for (int j = 0; j < 8; ++j) {
      unsigned h = (unsigned)(i * 8 + j) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
      v[j] = rnd ? (_Float16)(((int)(h & 0xffff) - 32768) * (1.f / 32768.f)) : (_Float16)(0.01f * ((i + j) & 15));
    }
    xp[i] = v;
  }
  f32x16 acc[NRB][NQB];
  for (int r = 0; r < NRB; ++r) for (int nb = 0; nb < NQB; ++nb) for (int j = 0; j < 16; ++j) acc[r][nb][j] = 0.f;
  WStreamP ws;
  const int stream_bytes = steps_per_wave * 2048;
  ws.rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wts) + (size_t)w * stream_bytes, 0, stream_bytes, 0x00020000);
  ws.v0 = lane * 16; ws.v1 = lane * 16 + 1024;
  const int xo = (lane >> 5) * TQP + (lane & 31);
  ASetP a0, a1, a2, a3;
  a0.r[0] = wload_p(ws, 0, 0);    a0.r[1] = wload_p(ws, 1, 0);
  a1 = a0; a2 = a0; a3 = a0;
  if (HM_P_AHEAD >= 2) { a1.r[0] = wload_p(ws, 0, 2048); a1.r[1] = wload_p(ws, 1, 2048); }
  if (HM_P_AHEAD >= 3) { a2.r[0] = wload_p(ws, 0, 4096); a2.r[1] = wload_p(ws, 1, 4096); }
  int sq = 0;
  for (int s = 0; s < stages; ++s) {
    __syncthreads();
#ifdef LIKE_KERNEL
    for (int r = 0; r < NRB; ++r) for (int nb = 0; nb < NQB; ++nb) { asm volatile("" :: "v"(acc[r][nb])); for (int j = 0; j < 16; ++j) acc[r][nb][j] = bl[(s * 37 + j) & 511]; }
#endif
    const long long t0 = clock64();
    k_loop_p<true, true, HM_P_AHEAD>(acc, a0, a1, a2, a3, ws, sq, n_grp, xp, xo);
    sq += n_grp * 8192;
    const long long t1 = clock64();
    __syncthreads();
    if (lane == 0 && blockIdx.x == 0) { t[(s * 8 + w) * 2] = t0; t[(s * 8 + w) * 2 + 1] = t1; }
  }
  float sum = 0.f;
#ifdef LIKE_KERNEL
  sum = sc[threadIdx.x & 1023];
#endif
  for (int r = 0; r < NRB; ++r) for (int nb = 0; nb < NQB; ++nb) for (int j = 0; j < 16; ++j) sum += acc[r][nb][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

int main(int argc, char** argv) {
  setvbuf(stdout, NULL, _IONBF, 0);
  const int rnd = argc > 1;      // any argument: pseudo-random fp16 weights and activations in [-1, 1) instead of constants
  const int stages = 12, steps = stages * 32 + 8;
  float* out; long long* t; char* wts;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&t, stages * 8 * 2 * 8); hipMalloc(&wts, (size_t)8 * steps * 2048); hipMemset(wts, 0x11, (size_t)8 * steps * 2048);
  if (rnd) {
    const size_t n = (size_t)8 * steps * 1024;
    _Float16* hw = (_Float16*)malloc(n * 2);
    unsigned st = 12345u;
    for (size_t i = 0; i < n; ++i) { st = st * 1664525u + 1013904223u; hw[i] = (_Float16)(((int)(st >> 16) - 32768) * (1.f / 32768.f)); }
    hipMemcpy(wts, hw, n * 2, hipMemcpyHostToDevice);
    free(hw);
  }
  printf(rnd ? "pseudo-random operands\n" : "constant operands\n");
  for (int threads : {512, 256}) for (int blocks : {8, 256}) {
    hipMemset(t, 0, stages * 8 * 2 * 8);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, t, wts, steps, stages, rnd, 8);
    hipDeviceSynchronize();
    long long h[12 * 8 * 2]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    const int nw = threads / 64, s = 6;
    long long b = h[(s * 8) * 2];
    for (int w = 0; w < nw; ++w) b = h[(s * 8 + w) * 2] < b ? h[(s * 8 + w) * 2] : b;
    printf("%3d threads %3d blocks: K-loop end of waves:", threads, blocks);
    for (int w = 0; w < nw; ++w) printf(" %6lld", h[(s * 8 + w) * 2 + 1] - b);
    printf("\n");
  }
  return 0;
}
