// K5: per-instance damped Gauss-Newton step: assemble, solve, Lie-group update, convergence flags.
//
// Restates, one workgroup per fruit instance and without any host round trip (reference lines relative to
// /root/reference/wild_completion/):
//   optimizer.py:200-231  code regulariser, scale damping, Levenberg-Marquardt damping, H and b assembly
//   optimizer.py:234      delta = inverse(H) b          -> Cholesky (packed lower triangle in LDS) + one step of
//                                                          iterative refinement with an fp64 residual
//   optimizer.py:237-253  pose_known masking, exp_sim3 / exp_se3 (utils.py:220-254, 279-324, quirks kept),
//                         T <- exp(delta_p) T, z <- z + delta_c, scale / translation / rotation deltas
//   optimizer.py:273-291  iter_count and the three convergence tests (only for i > 1)
//   optimizer.py:139-141  "submap not valid" exit when no depth-render residual survived
// The unknown vector is ordered [z (L) | pose (P)] internally (the reference orders it [pose | z]); the normal
// equations are permutation-equivariant, so only rounding differs.
#include "hm_common.h"
#include "hm_internal.h"

using namespace hm;

namespace {

constexpr int MAX_E = MAX_L + 7;
constexpr int NT = 1024;

__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }   // i >= j

__device__ void mat3_mul(const float* A, const float* B, float* C) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float s = 0.f;
      for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * B[k * 3 + j];
      C[i * 3 + j] = s;
    }
}

__device__ float det3(const float* M, int ld) {
  const double a = M[0], b = M[1], c = M[2], d = M[ld], e = M[ld + 1], f = M[ld + 2], g = M[2 * ld],
               h = M[2 * ld + 1], i = M[2 * ld + 2];
  return (float)(a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g));
}

// exp map of sim(3) / se(3), tangent order (translation, rotation[, log-scale]); returns 4x4 row-major.
// utils.py:220-254 (se3) and :279-324 (sim3).  Quirks kept: sim3 with theta > 1e-8 uses c = 0 whenever
// s <= 1e-8 (:314); the theta <= 1e-8 branch tests s == 0 exactly (:303-309).
__device__ void exp_pose(const float* x, bool sim3, float* T) {
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

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

}  // namespace

constexpr int NBK = (MAX_E + 31) / 32;   // 32-wide blocks of the unknown vector (9 for E <= 263)

// Blocked triangular solves on the packed factor in LDS.  `rv` holds the right-hand side on entry and the solution
// on exit.  Diagonal 32x32 blocks are solved by one wavefront with lane-parallel column updates (no reductions on
// the serial chain); the off-diagonal updates are spread over the whole workgroup.
__device__ void solve_packed(const float* Lp, float* rv, int E, int tid) {
  const int lane = tid & 63, wv = tid >> 6;
  const int nblk = (E + 31) / 32;
  // forward: L y = r
  for (int bb = 0; bb < nblk; ++bb) {
    const int base = bb * 32;
    if (wv == 0) {
      const int i = base + (lane & 31);
      const bool ok = i < E;
      float ri = ok ? rv[i] : 0.f;
      float lrow[32];
#pragma unroll
      for (int jj = 0; jj < 32; ++jj) lrow[jj] = (ok && base + jj <= i) ? Lp[tri(i, base + jj)] : 1.f;
#pragma unroll
      for (int jj = 0; jj < 32; ++jj) {
        const float num = __shfl(ri, jj);
        const float ljj = __shfl(lrow[jj], jj);
        const float yj = num / ljj;
        if ((lane & 31) == jj) ri = yj;
        else if ((lane & 31) > jj) ri -= lrow[jj] * yj;
      }
      if (lane < 32 && ok) rv[i] = ri;
    }
    __syncthreads();
    for (int i = base + 32 + tid; i < E; i += NT) {
      float s = 0.f;
      const float* lr = Lp + tri(i, base);
#pragma unroll 8
      for (int kk = 0; kk < 32; ++kk) s = fmaf(lr[kk], rv[base + kk], s);
      rv[i] -= s;
    }
    __syncthreads();
  }
  // backward: L^T x = y
  for (int bb = nblk - 1; bb >= 0; --bb) {
    const int base = bb * 32;
    if (wv == 0) {
      const int j = base + (lane & 31);
      const bool ok = j < E;
      float sj = ok ? rv[j] : 0.f;
      for (int ii = 31; ii >= 0; --ii) {
        const int i = base + ii;
        if (i >= E) continue;
        const float lii = Lp[tri(i, i)];
        const float xi = __shfl(sj, ii) / lii;
        const float lij = (ok && j < i) ? Lp[tri(i, j)] : 0.f;
        if ((lane & 31) == ii) sj = xi;
        else if ((lane & 31) < ii) sj -= lij * xi;
      }
      if (lane < 32 && ok) rv[j] = sj;
    }
    __syncthreads();
    const int hi = (base + 32 < E) ? base + 32 : E;
    for (int j = tid; j < base; j += NT) {
      float s = 0.f;
      for (int i = base; i < hi; ++i) s = fmaf(Lp[tri(i, j)], rv[i], s);
      rv[j] -= s;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(NT) void k_solve_update(const SolveArgs a) {
  __shared__ float Lp[MAX_E * (MAX_E + 1) / 2];   // packed lower-triangular Cholesky factor (<= 139 KiB)
  __shared__ float cj[NBK * 32];
  __shared__ float bvec[NBK * 32];
  __shared__ float xvec[NBK * 32];
  __shared__ float rv[NBK * 32];
  __shared__ float diagA[NBK * 32];
  __shared__ float red[NT / 64];
  __shared__ float red2[NT / 64];
  __shared__ float sh_piv;
  __shared__ int flag;

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  if (a.active[b] == 0) return;
  if (a.V != nullptr && a.V[b] <= 0) {           // optimizer.py:139-141
    if (tid == 0) { a.active[b] = 0; a.status[b] |= HM_STATUS_INVALID; }
    return;
  }
  const int L = a.L, P = a.P, E = L + P, ld = a.ldJ;
  const float* H = a.Hext + (size_t)b * ld * ld;
  float* z = a.latent + (size_t)b * a.ld_latent;
  const int lane = tid & 63, wv = tid >> 6;

  // ---- assemble (optimizer.py:200-231) ----
  for (int i = tid; i < E; i += NT) {
    float d = H[(size_t)i * ld + i];
    if (i < L) d += a.w_code;                                   // :200-201
    if (a.scale_on && P == 7 && i == L + 6) d += a.s_damp;      // :217-218
    diagA[i] = d;
    bvec[i] = -H[(size_t)(L + 7) * ld + i] - (i < L ? a.w_code * z[i] : 0.f);   // :153,190,202-203
  }
  if (tid == 0) flag = 0;
  __syncthreads();
  if (a.lm_on) {                                                // :220-225
    if (a.lm_eye) {
      float mx = -INFINITY;
      for (int i = tid; i < E; i += NT) mx = fmaxf(mx, diagA[i]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      if (lane == 0) red[wv] = mx;
      __syncthreads();
      mx = red[0];
      for (int i = 1; i < NT / 64; ++i) mx = fmaxf(mx, red[i]);
      __syncthreads();
      for (int i = tid; i < E; i += NT) diagA[i] += a.lam0 * mx;
    } else {
      for (int i = tid; i < E; i += NT) diagA[i] += a.lam0 * diagA[i];
    }
  }
  __syncthreads();

  // ---- register-resident right-looking Cholesky: thread (ti, tk) owns A[32 bi + ti][32 bk + tk], bk <= bi ----
  const int tk = tid & 31, ti = tid >> 5;
  float A[NBK][NBK];
#pragma unroll
  for (int bi = 0; bi < NBK; ++bi)
#pragma unroll
    for (int bk = 0; bk <= bi; ++bk) {
      const int i = bi * 32 + ti, k = bk * 32 + tk;
      float v = 0.f;
      if (i < E && k <= i) v = (i == k) ? diagA[i] : H[(size_t)i * ld + k];
      A[bi][bk] = v;
    }
  if (a.dbg_A != nullptr) {
    float* Ad = a.dbg_A + (size_t)b * ld * ld;
#pragma unroll
    for (int bi = 0; bi < NBK; ++bi)
#pragma unroll
      for (int bk = 0; bk <= bi; ++bk) {
        const int i = bi * 32 + ti, k = bk * 32 + tk;
        if (i < E && k <= i) Ad[(size_t)i * ld + k] = A[bi][bk];
      }
  }
  if (a.dbg_b != nullptr)
    for (int i = tid; i < E; i += NT) a.dbg_b[(size_t)b * ld + i] = bvec[i];

#pragma unroll
  for (int bj = 0; bj < NBK; ++bj) {
    for (int tj = 0; tj < 32; ++tj) {
      const int j = bj * 32 + tj;
      if (j >= E) break;
      if (ti == tj && tk == tj) sh_piv = A[bj][bj];
      __syncthreads();
      const float piv = sh_piv;
      if (!(piv > 0.f)) flag = 1;
      const float d = sqrtf(piv);
      const float inv = 1.f / d;
      if (tk == tj) {
#pragma unroll
        for (int bi = bj; bi < NBK; ++bi) {
          const int i = bi * 32 + ti;
          if (i > j && i < E) {
            const float l = A[bi][bj] * inv;
            A[bi][bj] = l;
            cj[i] = l;
            Lp[tri(i, j)] = l;
          } else if (i == j) {
            Lp[tri(j, j)] = d;
          }
        }
      }
      __syncthreads();
      float ci[NBK], ck[NBK];
#pragma unroll
      for (int bb = bj; bb < NBK; ++bb) { ci[bb] = cj[bb * 32 + ti]; ck[bb] = cj[bb * 32 + tk]; }
#pragma unroll
      for (int bi = bj; bi < NBK; ++bi)
#pragma unroll
        for (int bk = bj; bk <= bi; ++bk) {
          const int i = bi * 32 + ti, k = bk * 32 + tk;
          if (k > j && k <= i && i < E) A[bi][bk] = fmaf(-ci[bi], ck[bk], A[bi][bk]);
        }
    }
  }
  __syncthreads();
  if (flag) {
    if (tid == 0) { a.active[b] = 0; a.status[b] |= HM_STATUS_SOLVE_FAILED; }
    return;
  }

  // ---- solve L L^T x = b (fp32), then one refinement step with an fp64 residual from the fp32 system ----
  for (int i = tid; i < NBK * 32; i += NT) { rv[i] = i < E ? bvec[i] : 0.f; xvec[i] = 0.f; }
  __syncthreads();
  solve_packed(Lp, rv, E, tid);
  for (int i = tid; i < E; i += NT) xvec[i] = rv[i];
  __syncthreads();
  for (int i = wv; i < E; i += NT / 64) {
    double s = 0.0;
    for (int k = lane; k < E; k += 64) {
      const float aik = (k == i) ? diagA[i] : (k < i ? H[(size_t)i * ld + k] : H[(size_t)k * ld + i]);
      s += (double)aik * (double)xvec[k];
    }
    s = wave_sum(s);
    if (lane == 0) rv[i] = (float)((double)bvec[i] - s);
  }
  __syncthreads();
  solve_packed(Lp, rv, E, tid);
  for (int i = tid; i < E; i += NT) xvec[i] += rv[i];
  __syncthreads();
  if (a.dbg_delta != nullptr)
    for (int i = tid; i < E; i += NT) a.dbg_delta[(size_t)b * ld + i] = xvec[i];

  // ---- convergence statistics (optimizer.py:276-288) ----
  float mg = 0.f, mc = 0.f;
  bool bad = false;
  for (int i = tid; i < E; i += NT) {
    mg = fmaxf(mg, fabsf(bvec[i]));
    if (!isfinite(xvec[i])) bad = true;
    if (i < L) {
      const float zn = z[i] + xvec[i];                              // :248
      mc = fmaxf(mc, fabsf(xvec[i] / (zn + 1e-12f)));               // :280 (post-update z)
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { mg = fmaxf(mg, __shfl_xor(mg, o)); mc = fmaxf(mc, __shfl_xor(mc, o)); }
  if (lane == 0) { red[wv] = mg; red2[wv] = mc; }
  if (bad) flag = 1;
  __syncthreads();
  if (flag) {
    if (tid == 0) { a.active[b] = 0; a.status[b] |= HM_STATUS_SOLVE_FAILED; }
    return;
  }
  for (int i = tid; i < L; i += NT) z[i] += xvec[i];

  if (tid == 0) {
    for (int i = 1; i < NT / 64; ++i) { mg = fmaxf(mg, red[i]); mc = fmaxf(mc, red2[i]); }
    mg = fmaxf(mg, red[0]); mc = fmaxf(mc, red2[0]);
    int st = 0;
    const bool late = a.iter > 1;
    bool pose_conv = false;
    if (P > 0) {
      float dp[7] = {0, 0, 0, 0, 0, 0, 0};
      for (int i = 0; i < P; ++i) dp[i] = xvec[L + i];
      const bool known = a.pose_known != nullptr && a.pose_known[b] != 0;
      if (known) for (int i = 0; i < 6; ++i) dp[i] = 0.f;             // :237-238 (scale still updated)
      float dT[16], Tn[16];
      exp_pose(dp, P == 7, dT);                                       // :242-245
      float* T = a.T_ow + (size_t)b * 16;
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
          float s = 0.f;
          for (int k = 0; k < 4; ++k) s += dT[i * 4 + k] * T[k * 4 + j];
          Tn[i * 4 + j] = s;
        }
      for (int i = 0; i < 16; ++i) T[i] = Tn[i];                      // :247
      const float cur_scale = powf(det3(Tn, 4), -1.f / 3.f);          // :250
      const float d_scale = powf(det3(dT, 4), 1.f / 3.f);             // :251
      const float d_tran = sqrtf(dT[3] * dT[3] + dT[7] * dT[7] + dT[11] * dT[11]) * cur_scale;   // :252
      const float tr = (dT[0] + dT[5] + dT[10]) * cur_scale;          // :253 (multiplies by cur_scale: quirk)
      const float d_rot = fabsf(acosf((tr - 1.f) * 0.5f)) * 180.0f / 3.14159265358979323846f;
      if (a.cur_scale != nullptr) a.cur_scale[b] = cur_scale;
      pose_conv = (!known) && (d_tran < a.eps_t) && (d_rot < a.eps_r) && (d_scale < a.eps_s) && late;   // :285
    }
    a.iter_count[b] = a.iter + 1;                                     // :273
    if (mg < a.eps_g && late) st = HM_STATUS_CONV_G;                  // :276
    else if (mc < a.eps_c && late) st = HM_STATUS_CONV_C;             // :280
    else if (pose_conv) st = HM_STATUS_CONV_P;                        // :285
    else if (a.iter == a.max_iter - 1) st = HM_STATUS_MAX_ITER;       // :289
    if (st != 0) { a.status[b] |= st; a.active[b] = 0; }
  }
}

namespace hm {

int launch_solve_update(const SolveArgs& args, int B, hipStream_t stream) {
  hipLaunchKernelGGL(k_solve_update, dim3(B), dim3(NT), 0, stream, args);
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hm
