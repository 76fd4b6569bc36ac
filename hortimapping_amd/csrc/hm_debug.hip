// Unit hooks: run the device functions of K5 (exp maps) and K4 (Huber weights) on caller-supplied values, so that
// the reference's golden vectors for exp_se3 / exp_sim3 (wild_completion/utils.py:220-254, 279-324, quirk cases
// included) and huber_norm_weights / get_robust_res (utils.py:327-358) are checked on the HIP path itself and not only
// through optimisation trajectories.  The functions are the ones the product kernels call (hm_device_fn.h).
#include "hm_common.h"
#include "hm_device_fn.h"

using namespace hm;

namespace {

__global__ void k_debug_exp(const float* __restrict__ tang, int n, int sim3, float* __restrict__ T) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x[7], out[16];
  for (int k = 0; k < 7; ++k) x[k] = tang[(size_t)i * 7 + k];
  exp_pose(x, sim3 != 0, out);
  for (int k = 0; k < 16; ++k) T[(size_t)i * 16 + k] = out[k];
}

__global__ void k_debug_huber(const float* __restrict__ res, int n, float th, float* __restrict__ rho,
                              float* __restrict__ robust_res) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float r = res[i];
  const float w2 = huber_rho(r, th);
  rho[i] = w2;
  if (robust_res != nullptr) robust_res[i] = sqrtf(w2) * r;     // get_robust_res returns (w r, w^2), utils.py:343-358
}

}  // namespace

extern "C" int hm_debug_exp_map(const float* d_tangents, int n, int sim3, float* d_T, void* stream) {
  if (d_tangents == nullptr || d_T == nullptr || n < 0) { hm_set_error("hm_debug_exp_map: bad argument"); return -1; }
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_debug_exp, dim3((n + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), d_tangents, n,
                     sim3, d_T);
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}

extern "C" int hm_debug_huber(const float* d_res, int n, float threshold, float* d_rho, float* d_robust_res,
                              void* stream) {
  if (d_res == nullptr || d_rho == nullptr || n < 0) { hm_set_error("hm_debug_huber: bad argument"); return -1; }
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_debug_huber, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), d_res, n,
                     threshold, d_rho, d_robust_res);
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}
