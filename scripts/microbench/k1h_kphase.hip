// micro-benchmark: the K phase of the f16x3 decoder kernel (12 MFMA + 4 weight loads + 4 LDS reads per K-step and
// 2 x 2 blocks) as 8 waves of 2 x 2 blocks (two waves per SIMD, the product layout) or as 4 waves of 4 x 2 blocks (one
// wave per SIMD, accumulators 128 registers).  Same bytes, same MFMA count per CU; 16 stages, L2-resident 1 MiB weights.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)
constexpr int TQ = 64;

// NRB row blocks per wave (2 or 4), 2 query blocks.  weights: [row block 16][k 32][128 f16x8: hi 64 | lo 64]
template <int NRB, bool DISTINCT>
__global__ __launch_bounds__(NRB == 2 ? 512 : 256) void k(float* out, long long* t, const f16x8* __restrict__ wts_all, int stages) {
  __shared__ f16x8 xh[64 * TQ], xl[64 * TQ];     // 2 x 64 KiB
  constexpr int NW = 16 / NRB;                    // waves
  f32x16 acc[NRB][2];
  for (int r = 0; r < NRB; ++r) for (int q = 0; q < 2; ++q) for (int j = 0; j < 16; ++j) acc[r][q][j] = 0.f;
  for (int i = threadIdx.x; i < 64 * TQ; i += NW * 64) { f16x8 v; for (int j = 0; j < 8; ++j) v[j] = (_Float16)(0.01f * ((i + j) & 15)); xh[i] = v; xl[i] = v; }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int xo = (lane >> 5) * TQ + (lane & 31);
  const _Float16 cs = (_Float16)(1.f / 2048.f);
  for (int s = 0; s < stages; ++s) {
    const f16x8* wts = wts_all + (DISTINCT ? (size_t)s * 16 * 32 * 128 : 0);     // a fresh 1 MiB block per stage (16 MiB set) or one block
    const f16x8* wp[NRB];
#pragma unroll
    for (int r = 0; r < NRB; ++r) wp[r] = wts + (size_t)(w + NW * r) * 32 * 128 + lane;
    __syncthreads();
    long long t0 = clock64();
    f16x8 ah[3][NRB], al[3][NRB], bh[2][2], bl[2][2];
#pragma unroll
    for (int r = 0; r < NRB; ++r) { ah[0][r] = wp[r][0]; al[0][r] = wp[r][64]; ah[1][r] = wp[r][128]; al[1][r] = wp[r][128 + 64]; ah[2][r] = ah[0][r]; al[2][r] = al[0][r]; }
    bh[0][0] = xh[xo]; bh[0][1] = xh[xo + 32]; bl[0][0] = xl[xo]; bl[0][1] = xl[xo + 32];
    bh[1][0] = bh[0][0]; bh[1][1] = bh[0][1]; bl[1][0] = bl[0][0]; bl[1][1] = bl[0][1];
#pragma unroll 1
    for (int ks = 0; ks < 30; ks += 6) {
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int a = u % 3, an = (u + 2) % 3, b = u & 1, bn = b ^ 1;
        const int kb = ks + u + 1, ka = ks + u + 2 < 32 ? ks + u + 2 : 31;
        const f16x8* ph = xh + kb * 2 * TQ + xo;
        const f16x8* pl = xl + kb * 2 * TQ + xo;
        FENCE();
#pragma unroll
        for (int r = 0; r < NRB; ++r) {
          const f16x8 ac = ah[a][r] * cs;
          MFMA(ah[a][r], bh[b][0], acc[r][0]); MFMA(ah[a][r], bh[b][1], acc[r][1]);
          FENCE();
          if (r == 0) { bh[bn][0] = ph[0]; bh[bn][1] = ph[32]; }
          if (r == 1) { bl[bn][0] = pl[0]; bl[bn][1] = pl[32]; }
          FENCE();
          MFMA(ac, bl[b][0], acc[r][0]); MFMA(ac, bl[b][1], acc[r][1]);
          FENCE();
          ah[an][r] = wp[r][ka * 128];
          FENCE();
          MFMA(al[a][r], bh[b][0], acc[r][0]); MFMA(al[a][r], bh[b][1], acc[r][1]);
          FENCE();
          al[an][r] = wp[r][ka * 128 + 64];
          FENCE();
        }
      }
    }
    for (int u = 0; u < 2; ++u) for (int r = 0; r < NRB; ++r) { MFMA(ah[u][r], bh[u][0], acc[r][0]); MFMA(ah[u][r], bh[u][1], acc[r][1]); MFMA(al[u][r], bl[u][0], acc[r][0]); MFMA(al[u][r], bl[u][1], acc[r][1]); MFMA(al[u][r], bh[u][0], acc[r][0]); MFMA(al[u][r], bh[u][1], acc[r][1]); }
    long long t1 = clock64();
    __syncthreads();
    long long t2 = clock64();
    if (lane == 0 && blockIdx.x == 0) { t[(s * 8 + w) * 3 + 0] = t0; t[(s * 8 + w) * 3 + 1] = t1; t[(s * 8 + w) * 3 + 2] = t2; }
  }
  float sum = 0.f;
  for (int r = 0; r < NRB; ++r) for (int q = 0; q < 2; ++q) for (int j = 0; j < 16; ++j) sum += acc[r][q][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int NRB, bool DISTINCT>
void run(const char* name, int blocks) {
  float* out; long long* t; f16x8* wts;
  hipMalloc(&out, 512 * 1024 * 4); hipMalloc(&t, 16 * 8 * 3 * 8); hipMalloc(&wts, (size_t)16 * 16 * 32 * 128 * 16); hipMemset(wts, 0x11, (size_t)16 * 16 * 32 * 128 * 16);
  const int stages = 16, NW = 16 / NRB;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NRB, DISTINCT>), dim3(blocks), dim3(NW * 64), 0, 0, out, t, wts, stages);
  hipDeviceSynchronize();
  long long h[16 * 8 * 3]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  long long b = h[8 * 8 * 3], e = 0;
  for (int w = 0; w < NW; ++w) { b = h[(8 * 8 + w) * 3] < b ? h[(8 * 8 + w) * 3] : b; e = h[(8 * 8 + w) * 3 + 2] > e ? h[(8 * 8 + w) * 3 + 2] : e; }
  printf("%-40s %3d blocks: K phase (first start -> barrier passed) %6lld clocks, K-loop end per wave:", name, blocks, e - b);
  for (int w = 0; w < NW; ++w) printf(" %6lld", h[(8 * 8 + w) * 3 + 1] - b);
  printf("   (ideal 24576)\n");
  hipFree(out); hipFree(t); hipFree(wts);
}
int main() {
  for (int blocks : {1, 256, 1024}) {
    run<2, false>("8 waves x (2 x 2), one 1 MiB block", blocks); run<2, true>("8 waves x (2 x 2), 16 x 1 MiB blocks", blocks);
    run<4, false>("4 waves x (4 x 2), one 1 MiB block", blocks); run<4, true>("4 waves x (4 x 2), 16 x 1 MiB blocks", blocks);
  }
  return 0;
}
