// micro-benchmark 3: K1p-like K phase, features added one by one.  barrier / 32 K-steps x 8 MFMAs per wave / barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

// FEAT bit 0: in-place B refill from LDS (4 ds_read_b128 per step); bit 1: A ring of 3 from global (2 loads per step)
template <int FEAT>
__global__ __launch_bounds__(512, 2) void k(float* out, long long* t, const f16x8* __restrict__ wts, int stages) {
  __shared__ f16x8 xp[64 * 128];     // 128 KiB
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  for (int i = threadIdx.x; i < 64 * 128; i += 512) { f16x8 v; for (int j = 0; j < 8; ++j) v[j] = (_Float16)(0.01f * ((i + j) & 15)); xp[i] = v; }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int xo = (lane >> 5) * 128 + (lane & 31);
  const f16x8* wp0 = wts + (size_t)w * 32 * 128 + lane;
  const f16x8* wp1 = wts + (size_t)(w + 8) * 32 * 128 + lane;
  for (int s = 0; s < stages; ++s) {
    __syncthreads();
    long long t0 = clock64();
    f16x8 a[3][2], b[4];
    a[0][0] = wp0[0]; a[0][1] = wp1[0]; a[1][0] = wp0[128]; a[1][1] = wp1[128]; a[2][0] = a[0][0]; a[2][1] = a[0][1];
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = xp[xo + 32 * i];
#pragma unroll 1
    for (int ks = 0; ks < 30; ks += 3) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int kb = ks + u + 1, ka = ks + u + 2 < 32 ? ks + u + 2 : 31;
        const f16x8* ph = xp + kb * 2 * 128 + xo;
        FENCE();
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
          MFMA(a[u][0], b[nb], acc[nb]); MFMA(a[u][1], b[nb], acc[4 + nb]);
          FENCE();
          if (FEAT & 1) b[nb] = ph[32 * nb];
          if ((FEAT & 2) && nb == 2) a[(u + 2) % 3][0] = wp0[ka * 128];
          if ((FEAT & 2) && nb == 3) a[(u + 2) % 3][1] = wp1[ka * 128];
          FENCE();
        }
      }
    }
    for (int u = 0; u < 2; ++u) { FENCE(); for (int nb = 0; nb < 4; ++nb) { MFMA(a[u][0], b[nb], acc[nb]); MFMA(a[u][1], b[nb], acc[4 + nb]); } }
    long long t1 = clock64();
    __syncthreads();
    long long t2 = clock64();
    if (lane == 0 && blockIdx.x == 0) { t[(s * 8 + w) * 3 + 0] = t0; t[(s * 8 + w) * 3 + 1] = t1; t[(s * 8 + w) * 3 + 2] = t2; }
  }
  float sum = 0.f;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) sum += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int FEAT>
void run(const char* name, int blocks) {
  float* out; long long* t; f16x8* wts;
  hipMalloc(&out, 512 * 1024 * 4); hipMalloc(&t, 16 * 8 * 3 * 8); hipMalloc(&wts, 16 * 32 * 128 * 16); hipMemset(wts, 0x11, 16 * 32 * 128 * 16);
  const int stages = 16;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<FEAT>), dim3(blocks), dim3(512), 0, 0, out, t, wts, stages);
  hipDeviceSynchronize();
  long long h[16 * 8 * 3]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  long long b = h[8 * 8 * 3];
  for (int w = 0; w < 8; ++w) b = h[(8 * 8 + w) * 3] < b ? h[(8 * 8 + w) * 3] : b;
  printf("%-44s %3d blocks: K-loop end of waves 0..7:", name, blocks);
  for (int w = 0; w < 8; ++w) printf(" %6lld", h[(8 * 8 + w) * 3 + 1] - b);
  printf("   (ideal 8192 / 16384)\n");
  hipFree(out); hipFree(t); hipFree(wts);
}

int main() {
  for (int blocks : {1, 256}) {
    run<0>("MFMA only", blocks);
    run<1>("+ in-place B refill from LDS", blocks);
    run<2>("+ A ring from global", blocks);
    run<3>("+ both", blocks);
  }
  return 0;
}
