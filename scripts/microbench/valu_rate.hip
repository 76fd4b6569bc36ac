// issue cost of the VALU instructions the decoder epilogues use, in shader clocks per wave64 instruction on one SIMD
// (two waves per SIMD, 8 independent chains per wave so that latency does not show)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP>
__global__ __launch_bounds__(512) void k(uint32_t* out, long long* t, int iters) {
  uint32_t r[8];
  for (int i = 0; i < 8; ++i) r[i] = threadIdx.x * 2654435761u + i * 40503u + 0x3c003c00u;
  uint32_t c = 0x3f800000u + threadIdx.x, d = 0x40000000u, vcc_dummy = 0;
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#define A0(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(c), "v"(d));
#define A1(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(uint64_t*)&r[i & 6]) : "v"(*(uint64_t*)&r[(i & 6)]));
#define A2(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(uint64_t*)&r[i & 6]) : "v"(*(uint64_t*)&r[(i & 6)]));
#define A3(i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
#define A4(i) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(r[i]) : "v"(c), "v"(d));
#define A5(i) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(r[i]) : "v"(c));
#define A6(i) asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(r[i]));
#define A7(i) asm volatile("v_cmp_gt_f32 vcc, %1, 0\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(r[i]) : "v"(c) : "vcc");
#define A8(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
#define A9(i) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(r[i]));
#define A10(i) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(r[i]));
#define A11(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(c));
#define A12(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(c));
#define A13(i) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(r[i]) : "v"(c));
#define A14(i) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(c));
    if (OP == 0) { REP8(A0) REP8(A0) } if (OP == 1) { REP8(A1) REP8(A1) } if (OP == 2) { REP8(A2) REP8(A2) }
    if (OP == 3) { REP8(A3) REP8(A3) } if (OP == 4) { REP8(A4) REP8(A4) } if (OP == 5) { REP8(A5) REP8(A5) }
    if (OP == 6) { REP8(A6) REP8(A6) } if (OP == 7) { REP8(A7) REP8(A7) } if (OP == 8) { REP8(A8) REP8(A8) }
    if (OP == 9) { REP8(A9) REP8(A9) } if (OP == 10) { REP8(A10) REP8(A10) } if (OP == 11) { REP8(A11) REP8(A11) }
    if (OP == 12) { REP8(A12) REP8(A12) } if (OP == 13) { REP8(A13) REP8(A13) } if (OP == 14) { REP8(A14) REP8(A14) }
  }
  long long t1 = clock64();
  uint32_t s = vcc_dummy;
  for (int i = 0; i < 8; ++i) s ^= r[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) t[0] = t1 - t0;
}
template <int OP> void run(const char* name, int per) {
  uint32_t* out; long long* t; hipMalloc(&out, 512 * 4); hipMalloc(&t, 8);
  const int iters = 500;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<OP>), dim3(1), dim3(512), 0, 0, out, t, iters);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  printf("%-44s %5.2f clocks per instruction per SIMD (2 waves)\n", name, (double)h / (iters * 16.0 * per * 2));
  hipFree(out); hipFree(t);
}
int main() {
  run<0>("v_fma_f32", 1); run<1>("v_pk_fma_f32", 1); run<2>("v_pk_mul_f32", 1); run<3>("v_cvt_pk_f16_f32", 1);
  run<4>("v_fma_mixlo_f16", 1); run<5>("v_pk_max_f16", 1); run<6>("v_bfe_i32", 1); run<7>("v_cmp_gt_f32 + v_addc_co_u32 (pair)", 2);
  run<8>("v_max_f32", 1); run<9>("v_cvt_f16_f32", 1); run<10>("v_cvt_f32_f16", 1); run<11>("v_and_b32", 1);
  run<12>("v_cndmask_b32", 1); run<13>("v_pk_mul_f16", 1); run<14>("v_mov_b32", 1);
  return 0;
}
