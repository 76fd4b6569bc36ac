// Nearest-neighbour distances between two point clouds (SURVEY.md 8f "next" row 3): the primitive under the reference's
// ChamferDistance (metrics_3d/chamfer_distance.py:16-26) and PrecisionRecall (metrics_3d/precision_recall.py:13-50),
// which query an Open3D KD-tree per point (metrics_3d/metric.py:35-55 samples 1,000,000 points per mesh).  On the GPU
// the exact brute-force scan is the simple and fast formulation: every thread keeps four query points in registers,
// the reference cloud streams through LDS in tiles that all threads of the workgroup share (HBM-bound on the tile
// stream, VALU-bound on 3 sub + 3 fma + min per pair), distances in the direct (a-b)^2 form (no cancellation between
// large norms; the caller centres both clouds).  Output: Euclidean distance (not squared) to the nearest point.
#include "hm_common.h"
#include "hm_internal.h"

using namespace hm;

namespace {

constexpr int NN_THREADS = 256;
constexpr int NN_PER_THREAD = 4;
constexpr int NN_TILE = 2048;       // reference points per LDS tile (32 KiB as float4)

__global__ __launch_bounds__(NN_THREADS) void k_nn_distance(const f32x4* __restrict__ a, int na,
                                                            const f32x4* __restrict__ b, int nb,
                                                            float* __restrict__ dist) {
  __shared__ f32x4 tile[NN_TILE];
  const int base = blockIdx.x * NN_THREADS * NN_PER_THREAD + threadIdx.x;
  float ax[NN_PER_THREAD], ay[NN_PER_THREAD], az[NN_PER_THREAD], best[NN_PER_THREAD];
#pragma unroll
  for (int u = 0; u < NN_PER_THREAD; ++u) {
    const int i = base + u * NN_THREADS;
    const f32x4 p = i < na ? a[i] : f32x4{0.f, 0.f, 0.f, 0.f};
    ax[u] = p[0]; ay[u] = p[1]; az[u] = p[2];
    best[u] = INFINITY;
  }
  for (int t0 = 0; t0 < nb; t0 += NN_TILE) {
    const int cnt = nb - t0 < NN_TILE ? nb - t0 : NN_TILE;
    __syncthreads();
    for (int j = threadIdx.x; j < NN_TILE; j += NN_THREADS)
      tile[j] = j < cnt ? b[t0 + j] : f32x4{INFINITY, INFINITY, INFINITY, 0.f};   // padding never wins the min
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < NN_TILE; ++j) {
      const f32x4 q = tile[j];                           // same address for all lanes: LDS broadcast
#pragma unroll
      for (int u = 0; u < NN_PER_THREAD; ++u) {
        const float dx = ax[u] - q[0], dy = ay[u] - q[1], dz = az[u] - q[2];
        best[u] = fminf(best[u], fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
      }
    }
  }
#pragma unroll
  for (int u = 0; u < NN_PER_THREAD; ++u) {
    const int i = base + u * NN_THREADS;
    if (i < na) dist[i] = sqrtf(best[u]);
  }
}

}  // namespace

// d_a4 [na][4], d_b4 [nb][4] (xyz, w ignored) -> d_dist[na] = min_j |a_i - b_j|.  nb == 0 gives +inf.
extern "C" int hm_nn_distance(const float* d_a4, int na, const float* d_b4, int nb, float* d_dist, void* stream) {
  if (na < 0 || nb < 0 || (na > 0 && (d_a4 == nullptr || d_dist == nullptr)) || (nb > 0 && d_b4 == nullptr)) {
    hm_set_error("hm_nn_distance: bad argument"); return -1; }
  if (na == 0) return 0;
  const int per_block = NN_THREADS * NN_PER_THREAD;
  hipLaunchKernelGGL(k_nn_distance, dim3((na + per_block - 1) / per_block), dim3(NN_THREADS), 0,
                     static_cast<hipStream_t>(stream), reinterpret_cast<const f32x4*>(d_a4), na,
                     reinterpret_cast<const f32x4*>(d_b4), nb, d_dist);
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}
