// Small device functions shared by the product kernels and the unit hooks of hm_debug.hip: the hooks run EXACTLY the
// code K5 (exp maps) and K4 (Huber weights) run, so a golden-vector test of a hook is a test of the kernel's arithmetic.
#pragma once
#include "hm_common.h"

namespace hm {

__device__ inline void mat3_mul(const float* A, const float* B, float* C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float s = 0.f;
      for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * B[k * 3 + j];
      C[i * 3 + j] = s;
    }
}

__device__ inline float det3(const float* M, int ld) {
  const double a = M[0], b = M[1], c = M[2], d = M[ld], e = M[ld + 1], f = M[ld + 2], g = M[2 * ld],
               h = M[2 * ld + 1], i = M[2 * ld + 2];
  return (float)(a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g));
}

// exp map of sim(3) / se(3), tangent order (translation, rotation[, log-scale]); returns 4x4 row-major.
// utils.py:220-254 (se3) and :279-324 (sim3).  Quirks kept: sim3 with theta > 1e-8 uses c = 0 whenever
// s <= 1e-8 (:314); the theta <= 1e-8 branch tests s == 0 exactly (:303-309).
__device__ inline void exp_pose(const float* x, bool sim3, float* T) {
  const float v[3] = {x[0], x[1], x[2]};
  const float w[3] = {x[3], x[4], x[5]};
  const float s = sim3 ? x[6] : 0.f;
  const float W[9] = {0.f, -w[2], w[1], w[2], 0.f, -w[0], -w[1], w[0], 0.f};
  float W2[9];
  mat3_mul(W, W, W2);
  const float theta = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const float th2 = theta * theta;
  const float st = sinf(theta), ct = cosf(theta);
  const float es = sim3 ? expf(s) : 1.f;
  float R[9], Jm[9];
  const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (theta <= 1e-8f) {
    float c = 1.f;
    if (sim3 && s != 0.f) c = (es - 1.f) / s;
    for (int i = 0; i < 9; ++i) { R[i] = I[i]; Jm[i] = c * I[i]; }
  } else {
    for (int i = 0; i < 9; ++i) R[i] = I[i] + W[i] * st / theta + W2[i] * (1.f - ct) / th2;
    if (sim3) {
      const float s2 = s * s;
      const float a = es * st, b = es * ct;
      const float c = (s <= 1e-8f) ? 0.f : (es - 1.f) / s;
      const float k1 = (a * s + (1.f - b) * theta) / (s2 + th2);
      const float k2 = c - ((b - 1.f) * s + a * theta) / (s2 + th2);
      for (int i = 0; i < 9; ++i) Jm[i] = c * I[i] + k1 * W[i] / theta + k2 * W2[i] / th2;
    } else {
      const float th3 = th2 * theta;
      const float k1 = (1.f - ct) / th2, k2 = (theta - st) / th3;
      for (int i = 0; i < 9; ++i) Jm[i] = I[i] + k1 * W[i] + k2 * W2[i];
    }
  }
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T[i * 4 + j] = es * R[i * 3 + j];
    T[i * 4 + 3] = Jm[i * 3 + 0] * v[0] + Jm[i * 3 + 1] * v[1] + Jm[i * 3 + 2] * v[2];
  }
  T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
}

__device__ __forceinline__ float huber_rho(float r, float th) {
  // w^2 with w = 1 inside the window, sqrt(2 b |r| - b^2)/|r| outside (utils.py:327-340).  The reference evaluates
  // sqrt(res_norm) / x AFTER replacing x == 0 by 1, with res_norm = x^2 = 0 there: a residual that is exactly zero gets
  // weight 0 (its row drops out of H as well as of b) -- reproduced, golden vector G5.
  const float a = fabsf(r);
  if (th <= 0.f) return 1.f;          // robust kernel off (before robust_iter, mask term)
  if (a == 0.f) return 0.f;
  if (a <= th) return 1.f;
  return (2.f * th * a - th * th) / (a * a);
}

}  // namespace hm
