// check: __builtin_amdgcn_raw_buffer_load_b128 through a make_buffer_rsrc descriptor returns the same bytes as a plain load
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const char* p, u32x4* out, int* bad, int stream_bytes) {
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  char* base = const_cast<char*>(p) + (size_t)w * stream_bytes;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, stream_bytes, 0x00020000);
  int nbad = 0;
  for (int so = 0; so + 2048 <= stream_bytes; so += 2048) {
    u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, so, 0);
    u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + 1024, so, 0);
    const u32x4 ra = *reinterpret_cast<const u32x4*>(base + so + lane * 16);
    const u32x4 rb = *reinterpret_cast<const u32x4*>(base + so + lane * 16 + 1024);
    for (int j = 0; j < 4; ++j) nbad += (a[j] != ra[j]) + (b[j] != rb[j]);
  }
  atomicAdd(bad, nbad);
}
int main() {
  const int stream = 456 * 2048;
  char* p; int* bad; u32x4* out;
  hipMalloc(&p, (size_t)8 * stream); hipMalloc(&bad, 4); hipMalloc(&out, 4096);
  unsigned* h = (unsigned*)malloc((size_t)8 * stream);
  for (size_t i = 0; i < (size_t)8 * stream / 4; ++i) h[i] = (unsigned)(i * 2654435761u);
  hipMemcpy(p, h, (size_t)8 * stream, hipMemcpyHostToDevice); hipMemset(bad, 0, 4);
  hipLaunchKernelGGL(k, dim3(4), dim3(512), 0, 0, p, out, bad, stream);
  int hb = -1; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
  printf("buffer-load mismatches: %d\n", hb);
  return 0;
}
