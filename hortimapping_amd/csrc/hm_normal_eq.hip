// K4: weighted normal equations  H = sum_i c_i e_i^T e_i  on the fp32 matrix cores.
//
// Replaces the reference's per-term `torch.bmm(J^T, J).sum(0)` / `torch.bmm(J^T, r).sum(0)` which materialise an
// (n, E, E) tensor (wild_completion/optimizer.py:152-159, 189-190, 362-363).  A row is the extended vector
//     e_i = [ d r_i / d z (L) | d r_i / d pose (<= 7) | r_i ]            (ldJ = L + 8 floats, r in column L + 7)
// so one symmetric product yields H_zz, H_pz, H_pp, the gradient J^T r (row L+7) and the cost sum c r^2.
// The row weight is c_i = weight * rho_i / count with rho_i the squared Huber weight of r_i
// (wild_completion/utils.py:327-358) when the robust kernel is on for that term, else 1.
//
// Row segments: (SDF rows, depth-render rows, mask-render rows); the blockIdx -> (instance, pair) map keeps all
// pairs of an instance on one XCD so that the instance's rows are fetched from HBM once and re-read from that
// XCD's L2.  Fixed summation order => bitwise reproducible.
#include "hm_common.h"
#include "hm_internal.h"
#include "hm_device_fn.h"

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

constexpr int CH = 32;        // rows staged per chunk
constexpr int TW = 64;        // tile width: 2 x 2 blocks of 32 columns per workgroup (one block per wave)

// One workgroup (4 waves) owns a 64 x 64 tile = 2 x 2 lower-triangular 32x32 blocks of one instance's H and walks
// all rows of up to three row segments in chunks of CH rows: the chunk's two 64-column strips (A side, B side) go
// straight from global memory into LDS (global_load_lds_dwordx4, double buffered: the next chunk is in flight while the
// current one feeds CH / 2 MFMAs per wave), so every J element is fetched once per tile instead of once per block pair.
// Measured (rocprofv3, C2-joint): register-staged 64-row chunks 106 us, LDS-DMA 64 rows 98 us, 32 rows (four
// workgroups per CU) 91 us, 16 rows 94 us.
__global__ __launch_bounds__(256) void k_normal_eq(const NormalEqArgs a) {
  __shared__ float as[2][CH][TW];
  __shared__ float bs[2][CH][TW];
  __shared__ float cw[2][CH];
  // XCD-aware decomposition: blockIdx % 8 selects the XCD (observed dispatch rule); all tiles of an instance share it
  const int ntile = (a.nblk + 1) / 2;
  const int ntp = ntile * (ntile + 1) / 2;
  const int xcd = blockIdx.x & 7;
  const int slot = blockIdx.x >> 3;
  const int b = (slot / ntp) * 8 + xcd;
  const int tp = slot % ntp;
  if (b >= a.B) return;
  if (a.active != nullptr && a.active[b] == 0) return;
  int ti = 0;
  while ((ti + 1) * (ti + 2) / 2 <= tp) ++ti;
  const int tj = tp - ti * (ti + 1) / 2;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, h = lane >> 5;
  const int bi = 2 * ti + (wv >> 1), bj = 2 * tj + (wv & 1);
  const bool work = bi < a.nblk && bj < a.nblk && bi >= bj;          // blocks above the diagonal / past the end idle
  const int rcol = a.L + 7;
  // staging role of this thread: row (tid / 32 + 8 k), 16-byte chunk (tid % 32): chunks 0..15 A strip, 16..31 B strip
  const int srow = tid >> 5, sch = tid & 31;
  const bool sideB = sch >= 16;
  const int scol = (sideB ? tj : ti) * TW + (sch & 15) * 4;

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
    const int nchunk = (n + CH - 1) / CH;
    float sres = 0.f;
    // staging by LDS-DMA (global_load_lds_dwordx4: no register hop, no ds_write): one wave instruction moves 4 rows x 64
    // columns of a strip (lane l: row l / 16, 16-byte column chunk l % 16; LDS side linear = the row-major strip).
    // Wave wv issues CH / 8 of the chunk's CH / 2 instructions (CH / 4 per strip).  Rows past the segment end and
    // columns past the matrix are clamped to valid addresses: their weight is 0 / their H entries are never stored.
    auto dma = [&](int buf, int ch) {
#pragma unroll
      for (int k = 0; k < CH / 8; ++k) {
        const int i = (CH / 8) * wv + k;                  // instruction of the chunk
        const bool sB = i >= CH / 4;
        const int r = 4 * (i % (CH / 4)) + (lane >> 4);   // row within the chunk
        int row = ch * CH + r;
        row = row < n ? row : n - 1;
        int col = (sB ? tj : ti) * TW + (lane & 15) * 4;
        col = col + 4 <= a.ldJ ? col : a.ldJ - 4;
        const float* src = base + (size_t)row * a.ldJ + col;
        float* dst = sB ? &bs[buf][4 * (i % (CH / 4))][0] : &as[buf][4 * (i % (CH / 4))][0];
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                         (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
      }
    };
    auto fetch_w = [&](int ch) { if (tid < CH) { const int row = ch * CH + tid; sres = row < n ? base[(size_t)row * a.ldJ + rcol] : 0.f; } };
    auto stash_w = [&](int buf, int ch) { if (tid < CH) cw[buf][tid] = (ch * CH + tid < n) ? scale * huber_rho(sres, sg.robust_th) : 0.f; };
    __syncthreads();                           // previous segment's readers are done with both buffers
    dma(0, 0);
    fetch_w(0);
    stash_w(0, 0);
    for (int ch = 0; ch < nchunk; ++ch) {
      const int buf = ch & 1;
      __builtin_amdgcn_s_waitcnt(0x0070);      // this wave's DMA of chunk `ch` (and its weight loads) have landed
      __syncthreads();                         // chunk `ch` visible in LDS; buffer buf^1 free
      if (ch + 1 < nchunk) { dma(buf ^ 1, ch + 1); fetch_w(ch + 1); }
      if (work) {
#pragma unroll
        for (int g = 0; g < CH / 8; ++g) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int r = 8 * g + 4 * h + t;
            const float av = as[buf][r][(wv >> 1) * 32 + c];
            const float bv = bs[buf][r][(wv & 1) * 32 + c] * cw[buf][r];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
          }
        }
      }
      if (ch + 1 < nchunk) stash_w(buf ^ 1, ch + 1);
    }
  }
  if (work) {
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
  const int ntile = (a.nblk + 1) / 2;
  const int ntp = ntile * (ntile + 1) / 2;
  const int inst_per_xcd = (B + 7) / 8;
  const int grid = inst_per_xcd * ntp * 8;
  hipLaunchKernelGGL(k_normal_eq, dim3(grid), dim3(256), 0, stream, a);
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hm
