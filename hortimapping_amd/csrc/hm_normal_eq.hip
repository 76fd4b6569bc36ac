// K4: weighted normal equations  H = sum_i c_i e_i^T e_i  on the fp32 matrix cores.
//
// Replaces the reference's per-term `torch.bmm(J^T, J).sum(0)` / `torch.bmm(J^T, r).sum(0)` which materialise an
// (n, E, E) tensor (wild_completion/optimizer.py:152-159, 189-190, 362-363).  A row is the extended vector
//     e_i = [ d r_i / d z (L) | d r_i / d pose (<= 7) | r_i ]            (ldJ = L + 8 floats, r in column L + 7)
// so one symmetric product yields H_zz, H_pz, H_pp, the gradient J^T r (row L+7) and the cost sum c r^2.
// The row weight is c_i = weight * rho_i / count with rho_i the squared Huber weight of r_i
// (wild_completion/utils.py:327-358) when the robust kernel is on for that term, else 1.
//
// One 4-wave workgroup owns one lower-triangular 32x32 block pair of one instance; the waves split the rows of up
// to three row segments round-robin in groups of 8 and their partial blocks are summed in a fixed order through LDS.
// Row segments: (SDF rows, depth-render rows, mask-render rows); the blockIdx -> (instance, pair) map keeps all
// pairs of an instance on one XCD so that the instance's rows are fetched from HBM once and re-read from that
// XCD's L2.  Fixed summation order => bitwise reproducible.
#include "hm_common.h"
#include "hm_internal.h"

using namespace hm;

struct NormalEqArgs {
  RowSegment seg[3];
  int n_seg;
  int L;
  int ldJ;         // L + 8
  int nblk;        // L/32 + 1
  int B;
  const int* active;
  float* Hext;     // [B][ldJ][ldJ], lower block pairs written
};

__device__ __forceinline__ float huber_rho(float r, float th) {
  // w^2 with w = 1 inside the window, sqrt(2 b |r| - b^2)/|r| outside (utils.py:327-340)
  const float a = fabsf(r);
  if (th <= 0.f || a <= th) return 1.f;
  return (2.f * th * a - th * th) / (a * a);
}

constexpr int KSPLIT = 4;   // waves per block pair: each takes every 4th group of 8 rows, partials reduced through LDS

__global__ __launch_bounds__(64 * KSPLIT) void k_normal_eq(const NormalEqArgs a) {
  __shared__ float part[KSPLIT - 1][16][64];
  // XCD-aware decomposition: blockIdx % 8 selects the XCD (observed dispatch rule); all pairs of an instance share it
  const int npair = a.nblk * (a.nblk + 1) / 2;
  const int xcd = blockIdx.x & 7;
  const int slot = blockIdx.x >> 3;          // index within this XCD's share
  const int inst_in_xcd = slot / npair;
  const int pair = slot % npair;
  const int b = inst_in_xcd * 8 + xcd;
  if (b >= a.B) return;
  if (a.active != nullptr && a.active[b] == 0) return;
  // pair -> (bi >= bj)
  int bi = 0;
  while ((bi + 1) * (bi + 2) / 2 <= pair) ++bi;
  const int bj = pair - bi * (bi + 1) / 2;

  const int lane = threadIdx.x & 63;
  const int ws = threadIdx.x >> 6;           // K-split index of this wave
  const int c = lane & 31, h = lane >> 5;
  const int colA = bi * 32 + c, colB = bj * 32 + c;
  const bool okA = colA < a.ldJ, okB = colB < a.ldJ;
  const int rcol = a.L + 7;

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  for (int sidx = 0; sidx < a.n_seg; ++sidx) {
    const RowSegment& sg = a.seg[sidx];
    const int n = sg.count_dev != nullptr ? sg.count_dev[b] : sg.count_const;
    const int nd = sg.norm_dev != nullptr ? sg.norm_dev[b] : n;
    if (n <= 0 || nd <= 0) continue;
    const float scale = sg.weight / (float)nd;
    const float* base = sg.rows + (size_t)b * sg.inst_stride + (size_t)sg.row_offset * a.ldJ;
    // software pipeline: the 12 loads of this wave's next row group are issued before the 4 MFMAs of the current one
    auto load_step = [&](int r0, float (&av)[4], float (&bv)[4]) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row = r0 + 4 * h + t;
        const bool ok = row < n;
        const float* rp = base + (size_t)row * a.ldJ;
        const float res = ok ? rp[rcol] : 0.f;
        const float va = (ok && okA) ? rp[colA] : 0.f;
        const float vb = (ok && okB) ? rp[colB] : 0.f;
        av[t] = va;
        bv[t] = vb * (scale * huber_rho(res, sg.robust_th));
      }
    };
    float av[4], bv[4], an[4], bn[4];
    const int stride = 8 * KSPLIT;
    load_step(8 * ws, av, bv);
    for (int r0 = 8 * ws; r0 < n; r0 += stride) {
      if (r0 + stride < n) load_step(r0 + stride, an, bn);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[t], acc, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) { av[t] = an[t]; bv[t] = bn[t]; }
    }
  }
  // fixed-order reduction of the KSPLIT partial accumulators (wave 0 adds waves 1, 2, 3 in that order)
  if (ws > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) part[ws - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (ws == 0) {
#pragma unroll
    for (int w2 = 0; w2 < KSPLIT - 1; ++w2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += part[w2][r][lane];
    float* H = a.Hext + (size_t)b * a.ldJ * a.ldJ;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const int col = bj * 32 + c;
      if (row < a.ldJ && col < a.ldJ) H[(size_t)row * a.ldJ + col] = acc[r];
    }
  }
}

namespace hm {

int launch_normal_eq(const RowSegment* segs, int n_seg, int L, int B, const int* d_active, float* d_Hext,
                     hipStream_t stream) {
  NormalEqArgs a;
  for (int i = 0; i < 3; ++i) a.seg[i] = segs[i < n_seg ? i : 0];
  a.n_seg = n_seg; a.L = L; a.ldJ = L + POSE_PAD; a.nblk = L / 32 + 1; a.B = B; a.active = d_active;
  a.Hext = d_Hext;
  const int npair = a.nblk * (a.nblk + 1) / 2;
  const int inst_per_xcd = (B + 7) / 8;
  const int grid = inst_per_xcd * npair * 8;
  hipLaunchKernelGGL(k_normal_eq, dim3(grid), dim3(64 * KSPLIT), 0, stream, a);
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hm
