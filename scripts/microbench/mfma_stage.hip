// micro-benchmark 2: K1p-like phases -- barrier, 256 MFMAs per wave (2 A x 4 B operands, 8 accumulators), barrier.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int LDSKB, int VALU, int DATA>
__global__ __launch_bounds__(512, 2) void k(float* out, long long* t, int stages) {
  __shared__ float lds[LDSKB * 256];
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  f16x8 a0, a1, bb[4];
  for (int j = 0; j < 8; ++j) { a0[j] = (_Float16)(threadIdx.x * 0.001f + j); a1[j] = (_Float16)(j * 0.5f); for (int i = 0; i < 4; ++i) bb[i][j] = (_Float16)(i + j + (threadIdx.x & 3)); }
  if (DATA == 1) for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) bb[i][j] = (_Float16)(1e-6f * (1 + ((threadIdx.x + i + j) & 7)));   // fp16 subnormals
  if (DATA == 2) for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) bb[i][j] = (_Float16)0.f;
  if (DATA == 3) for (int j = 0; j < 8; ++j) { a0[j] = (_Float16)(3e-6f * (1 + j)); a1[j] = (_Float16)(2e-6f * (1 + j)); }          // subnormal A
  lds[threadIdx.x] = 1.f;
  const int w = threadIdx.x >> 6;
  float x = threadIdx.x;
  for (int s = 0; s < stages; ++s) {
    __syncthreads();
    long long t0 = clock64();
    for (int ks = 0; ks < 32; ++ks) {
      FENCE();
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) { MFMA(a0, bb[nb], acc[nb]); MFMA(a1, bb[nb], acc[4 + nb]); FENCE(); }
    }
    long long t1 = clock64();
    __syncthreads();
    long long t2 = clock64();
    if (VALU) {   // epilogue-like VALU work
      for (int i = 0; i < VALU; ++i) x = fmaf(x, 1.0001f, 0.5f);
    }
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { t[(s * 8 + w) * 3 + 0] = t0; t[(s * 8 + w) * 3 + 1] = t1; t[(s * 8 + w) * 3 + 2] = t2; }
  }
  float sum = x + lds[(threadIdx.x * 7) & 255];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) sum += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int LDSKB, int VALU, int DATA>
void run(const char* name, int blocks) {
  float* out; long long* t;
  hipMalloc(&out, 512 * 1024 * 4); hipMalloc(&t, 16 * 8 * 3 * 8);
  const int stages = 16;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<LDSKB, VALU, DATA>), dim3(blocks), dim3(512), 0, 0, out, t, stages);
  hipDeviceSynchronize();
  long long h[16 * 8 * 3]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  printf("%s, %d blocks: stage 8 per wave (loop start, loop end, after barrier) relative to the earliest start\n", name, blocks);
  long long b = h[8 * 8 * 3];
  for (int w = 0; w < 8; ++w) b = h[(8 * 8 + w) * 3] < b ? h[(8 * 8 + w) * 3] : b;
  for (int w = 0; w < 8; ++w) printf("   wave %d: %6lld %6lld %6lld\n", w, h[(8 * 8 + w) * 3] - b, h[(8 * 8 + w) * 3 + 1] - b, h[(8 * 8 + w) * 3 + 2] - b);
  hipFree(out); hipFree(t);
}

int main() {
  run<150, 0, 0>("normal operands", 1);
  run<150, 0, 1>("fp16-subnormal B operands", 1);
  run<150, 0, 2>("zero B operands", 1);
  run<150, 0, 3>("fp16-subnormal A operands", 1);
  return 0;
}
