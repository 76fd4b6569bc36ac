// micro-benchmark: what does one operand-fetch instruction cost a wave that is otherwise issuing MFMAs?
// Loop body = 8 x v_mfma_f32_32x32x16_f16 (8 independent accumulators) + N fetch instructions of ONE kind, 64 iterations,
// timed with s_memtime by every wave of workgroup 0.  One wave per SIMD (256 threads) gives the serial cost directly:
// cycles / iteration = 8 x 32 + N x cost; two waves per SIMD (512 threads) show how much of it the partner's MFMAs cover.
// (kind 4, no-VGPR addressing, measured 450-990 cycles per iteration and faulted with 256 blocks: left out of the run.)
// Kinds: 0 none, 1 global_load_dwordx4 (64-bit vaddr), 2 global_load_dwordx4 saddr + 32-bit voffset, 3 buffer_load_dwordx4
// offen, 4 buffer_load_dwordx4 with no VGPR address (descriptor ADD_TID_ENABLE), 5 global_load_lds_dwordx4 (LDS-DMA),
// 6 ds_read_b128, 7 ds_read_b64.   hipcc --offload-arch=gfx950 -O3 issue_cost.hip -o issue_cost && ./issue_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
#define FENCE() __builtin_amdgcn_sched_barrier(0)

template <int KIND, int N, int PRIO, int FOOT = 32768, int NL = 0>
__global__ __launch_bounds__(512, 2) void k(float* out, long long* t, const char* wts, int iters) {
  extern __shared__ char lds[];
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0.001f * (i & 31);
  __syncthreads();
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * (lane + j)); b[j] = (_Float16)(0.02f * (lane - j)); }
  const size_t wbase = FOOT > 32768 ? (size_t)w * (FOOT + 4096) : ((size_t)blockIdx.x * 8 + w) * 65536;   // large footprints: every block reads the same 8 streams (L2 hits)
  const char* gp = wts + wbase + lane * 16;
  uint32_t voff = lane * 16;
  const char* sbase = wts + wbase;
  // buffer descriptors: plain (num_records = whole buffer) and with ADD_TID_ENABLE (stride field = 16 bytes)
  i32x4 rs, rt;
  {
    const uint64_t p = (uint64_t)sbase;
    rs[0] = (int)(uint32_t)p; rs[1] = (int)(uint32_t)(p >> 32); rs[2] = FOOT + 4096; rs[3] = 0x00020000;
    rt[0] = rs[0]; rt[1] = (int)((uint32_t)(p >> 32) | (16u << 16)); rt[2] = 65536; rt[3] = 0x00020000 | (1 << 23);
  }
  rs[0] = __builtin_amdgcn_readfirstlane(rs[0]); rs[1] = __builtin_amdgcn_readfirstlane(rs[1]);
  rt[0] = __builtin_amdgcn_readfirstlane(rt[0]); rt[1] = __builtin_amdgcn_readfirstlane(rt[1]);
  const uint64_t sb64 = ((uint64_t)(uint32_t)rs[1] << 32) | (uint64_t)(uint32_t)rs[0];
  const uint32_t laddr = (uint32_t)(w * 4096 + lane * 16);
  f32x4 d[N > 0 ? N : 1];
  for (int i = 0; i < (N > 0 ? N : 1); ++i) d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 e[NL > 0 ? NL : 1];
  for (int i = 0; i < (NL > 0 ? NL : 1); ++i) e[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  if (PRIO == 2 && w >= 4) __builtin_amdgcn_s_setprio(1);
  const long long t0 = clock64();
  uint32_t so = 0;
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    FENCE();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      MFMA(a, b, acc[i]);
      FENCE();
      if (i < N) {
        if (KIND == 1) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(d[i]) : "v"(gp), "n"(1024 * i));
        if (KIND == 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(d[i]) : "v"(voff), "s"(sb64), "n"(1024 * i));
        if (KIND == 3) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(d[i]) : "v"(voff), "s"(rs), "n"(1024 * i));
        if (KIND == 4) asm volatile("buffer_load_dwordx4 %0, off, %1, %2 offset:%3" : "=v"(d[i]) : "s"(rt), "s"(so), "n"(1024 * i));
        if (KIND == 5) __builtin_amdgcn_global_load_lds(reinterpret_cast<const void*>(gp + 1024 * i), reinterpret_cast<__attribute__((address_space(3))) void*>(w * 8192 + 1024 * i), 16, 0, 0);
        if (KIND == 6) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d[i]) : "v"(laddr), "n"(1024 * i));
        if (KIND == 7) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(*reinterpret_cast<double*>(&d[i])) : "v"(laddr), "n"(1024 * i));
      }
      if (i < NL) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(e[i]) : "v"(laddr), "n"(1024 * i));
      FENCE();
    }
    gp += 4096; voff += 4096; so += 4096;
    if (((it + 1) & (FOOT / 4096 - 1)) == 0) { gp -= FOOT; voff -= FOOT; so -= FOOT; }
    if (KIND != 0 && KIND != 5) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(8)");
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  const long long t1 = clock64();
  if (lane == 0 && blockIdx.x == 0) { t[w * 2] = t0; t[w * 2 + 1] = t1; }
  float sum = 0.f;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) sum += acc[i][j];
  for (int i = 0; i < (N > 0 ? N : 1); ++i) sum += d[i][0] + d[i][3];
  for (int i = 0; i < (NL > 0 ? NL : 1); ++i) sum += e[i][0] + e[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int KIND, int N, int PRIO = 0, int FOOT = 32768, int NL = 0>
void run(const char* name, int threads, int blocks, float* out, long long* t, char* wts) {
  const int iters = FOOT > 32768 ? 256 : 64;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<KIND, N, PRIO, FOOT, NL>), dim3(blocks), dim3(threads), 65536, 0, out, t, wts, iters);
  hipDeviceSynchronize();
  long long h[16]; hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
  const int nw = threads / 64;
  long long b0 = h[0], e1 = h[1];
  for (int w = 0; w < nw; ++w) { if (h[2 * w] < b0) b0 = h[2 * w]; if (h[2 * w + 1] > e1) e1 = h[2 * w + 1]; }
  printf("%-46s N=%d %3d thr %3d blk: cyc/iter wave0 %6.1f  first %6.1f  last %6.1f   (8 MFMA = 256)\n", name, N, threads, blocks,
         (h[1] - h[0]) / (double)iters, (h[1] - b0) / (double)iters, (e1 - b0) / (double)iters);
}

int main() {
  setvbuf(stdout, NULL, _IONBF, 0);
  float* out; long long* t; char* wts;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&t, 256); hipMalloc(&wts, (size_t)256 * 8 * 65536); hipMemset(wts, 0x11, (size_t)256 * 8 * 65536);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<0, 0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int threads : {256, 512}) for (int blocks : {1, 256}) {
    run<0, 0>("MFMA only", threads, blocks, out, t, wts);
    run<1, 2>("global_load_dwordx4 vaddr64", threads, blocks, out, t, wts);
    run<1, 4>("global_load_dwordx4 vaddr64", threads, blocks, out, t, wts);
    run<2, 2>("global_load_dwordx4 saddr+voff", threads, blocks, out, t, wts);
    run<2, 4>("global_load_dwordx4 saddr+voff", threads, blocks, out, t, wts);
    run<3, 2>("buffer_load_dwordx4 offen", threads, blocks, out, t, wts);
    run<3, 4>("buffer_load_dwordx4 offen", threads, blocks, out, t, wts);
    run<5, 2>("global_load_lds_dwordx4 (DMA)", threads, blocks, out, t, wts);
    run<5, 4>("global_load_lds_dwordx4 (DMA)", threads, blocks, out, t, wts);
    run<6, 4>("ds_read_b128", threads, blocks, out, t, wts);
    run<6, 8>("ds_read_b128", threads, blocks, out, t, wts);
    run<7, 8>("ds_read_b64", threads, blocks, out, t, wts);
  }
  for (int blocks : {8, 256}) {
    run<1, 2, 0, 32768, 4>("vaddr64 x2 + ds_read_b128 x4, 32 KiB/wave", 512, blocks, out, t, wts);
    run<2, 2, 0, 32768, 4>("saddr x2 + ds_read_b128 x4, 32 KiB/wave", 512, blocks, out, t, wts);
    run<3, 2, 0, 32768, 4>("buffer x2 + ds_read_b128 x4, 32 KiB/wave", 512, blocks, out, t, wts);
    run<1, 2, 0, 524288, 0>("vaddr64 x2, 512 KiB/wave shared streams", 512, blocks, out, t, wts);
    run<1, 2, 0, 524288, 4>("vaddr64 x2 + ds_read_b128 x4, 512 KiB/wave shared", 512, blocks, out, t, wts);
    run<3, 2, 0, 524288, 4>("buffer x2 + ds_read_b128 x4, 512 KiB/wave shared", 512, blocks, out, t, wts);
    run<1, 1, 0, 524288, 4>("vaddr64 x1 + ds_read_b128 x4, 512 KiB/wave shared", 512, blocks, out, t, wts);
  }
  run<1, 2, 2>("global_load vaddr64, waves 4-7 at setprio 1", 512, 256, out, t, wts);
  run<6, 4, 2>("ds_read_b128, waves 4-7 at setprio 1", 512, 256, out, t, wts);
  return 0;
}
