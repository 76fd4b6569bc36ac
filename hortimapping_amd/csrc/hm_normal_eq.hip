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

// ---------------------------------------------------------------------------------------------------------------------
// K4h (round 5): the same product on the fp16 matrix cores with SPLIT operands, for the f16x3 decoder arithmetics.
//
// k_normal_eq is bound by the fp32-input matrix cores (16 v_mfma_f32_32x32x2 per wave and 32-row chunk: 4.1 k of the 5.5 k
// clocks a chunk takes with four workgroups per CU; 0.09 ms of a 1.9 ms C2-joint iteration, 0.18 ms with 2048 surface points).
// Here every staged element is split ONCE per workgroup into fp16 hi / lo (lo scaled by 2^11, as in hm_decoder_h.hip) while
// it moves global -> registers -> LDS, in the MFMA operand layout [row / 8][column][8 rows]; a chunk then costs each wave six
// v_mfma_f32_32x32x16_f16 (192 clocks instead of 1024):
//     acc  += Ah Bh            acc2 += Ah Bl' + Al' Bh            H = (acc + 2^-11 acc2) 2^-q
// (the dropped Al Bl term is 2^-22 of the product; every fp16 x fp16 product is exact in fp32; the cross terms have their own
// accumulator, so nothing is rescaled inside the loop).  A = e_i (the extended row), B = c_i 2^q e_i with the row weight
// c_i = weight rho_i / count as in k_normal_eq and q = -ilogb(weight / count) a per-SEGMENT power of two that keeps B in the
// normal fp16 range (c_i is 1e-3 ... 1e-5); between segments the accumulators are rescaled by the exact factor
// 2^(q_next - q).  Entries beyond fp16's 65504 turn into inf -> NaN in H -> HM_STATUS_SOLVE_FAILED, the policy of the f16x3
// decoder itself (the exact-f32 arithmetic keeps k_normal_eq).  Same tiling, same XCD map, fixed summation order.
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

// hi = f16(v), lo = f16((v - hi) 2^11) for eight values: per PAIR one packed conversion, one packed multiply and the two
// mixed-precision fmas of hm_decoder_h.hip (lo = f16(hi * -2^11 + v * 2^11): an exact fp32 fma, ONE rounding -- bit-identical
// to the subtract-scale-convert form, a third of its instructions).  `w` scales the values first (row weights of the B side).
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2k __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split8(const float (&v)[8], h16x8& hi, h16x8& lo) {
  uint32_t hw[4], lw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x2k x = {v[2 * j], v[2 * j + 1]};
    const h16x2 hp = __builtin_convertvector(x, h16x2);
    const f32x2k sx = x * 2048.f;
    uint32_t hpk, d;
    __builtin_memcpy(&hpk, &hp, 4);
    const float cneg = -2048.f;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(d) : "v"(hpk), "s"(cneg), "v"(sx[0]), "v"(sx[1]));
    hw[j] = hpk; lw[j] = d;
  }
  __builtin_memcpy(&hi, hw, 16);
  __builtin_memcpy(&lo, lw, 16);
}

__global__ __launch_bounds__(256) void k_normal_eq_h(const NormalEqArgs a) {
  __shared__ h16x8 Ah[2][4][TW], Al[2][4][TW], Bh[2][4][TW], Bl[2][4][TW];     // 4 x 8 KiB
  __shared__ float cw[3][CH];
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
  const bool work = bi < a.nblk && bj < a.nblk && bi >= bj;
  const int rcol = a.L + 7;
  // staging role: the 8 rows 8 kg .. 8 kg + 7 of column `scol` of both strips (a wave reads 64 consecutive floats of a row)
  const int kg = tid >> 6, scol = tid & 63;
  int colA = ti * TW + scol, colB = tj * TW + scol;
  colA = colA < a.ldJ ? colA : a.ldJ - 1;            // columns past the matrix: valid addresses, H entries never stored
  colB = colB < a.ldJ ? colB : a.ldJ - 1;

  f32x16 acc, acc2;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc[i] = 0.f; acc2[i] = 0.f; }
  int q_cur = 0;
  bool any = false;

  for (int sidx = 0; sidx < a.n_seg; ++sidx) {
    const RowSegment& sg = a.seg[sidx];
    const int n = sg.count_dev != nullptr ? sg.count_dev[b] : sg.count_const;
    const int nd = sg.norm_dev != nullptr ? sg.norm_dev[b] : n;
    if (n <= 0 || nd <= 0) continue;
    const float scale = sg.weight / (float)nd;
    if (!(scale > 0.f)) continue;                    // a term with weight 0 adds nothing
    const int q = -ilogbf(scale);                    // scale 2^q in [1, 2)
    const float scale_q = ldexpf(scale, q);
    if (any && q != q_cur) {
      const float f = ldexpf(1.f, q - q_cur);        // exact power of two
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc[i] *= f; acc2[i] *= f; }
    }
    q_cur = q; any = true;
    const float* base = sg.rows + (size_t)b * sg.inst_stride + (size_t)sg.row_offset * a.ldJ;
    const int nchunk = (n + CH - 1) / CH;
    float va[8], vb[8];
    float sres = 0.f;
    auto fetch = [&](int ch) {                       // this thread's 2 x 8 elements of chunk `ch` (rows clamped: weight 0)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        int row = ch * CH + 8 * kg + j;
        row = row < n ? row : n - 1;
        const float* rp = base + (size_t)row * a.ldJ;
        va[j] = rp[colA];
        vb[j] = rp[colB];
      }
    };
    auto fetch_w = [&](int ch) { if (tid < CH) { const int row = ch * CH + tid; sres = row < n ? base[(size_t)row * a.ldJ + rcol] : 0.f; } };
    auto stash_w = [&](int ch) { if (tid < CH) cw[ch % 3][tid] = (ch * CH + tid < n) ? scale_q * huber_rho(sres, sg.robust_th) : 0.f; };
    auto store = [&](int buf, int ch) {              // split, weight the B side, write the operand planes of chunk `ch`
      h16x8 hi, lo;
      split8(va, hi, lo);
      Ah[buf][kg][scol] = hi; Al[buf][kg][scol] = lo;
      const float* w = cw[ch % 3] + 8 * kg;
      float vw[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) vw[j] = vb[j] * w[j];
      split8(vw, hi, lo);
      Bh[buf][kg][scol] = hi; Bl[buf][kg][scol] = lo;
    };
    __syncthreads();                                 // previous segment's readers are done with LDS
    fetch(0); fetch_w(0); stash_w(0);
    if (nchunk > 1) { fetch_w(1); stash_w(1); }
    __syncthreads();                                 // cw[0], cw[1] visible
    store(0, 0);
    for (int ch = 0; ch < nchunk; ++ch) {
      const int buf = ch & 1;
      __syncthreads();                               // planes of chunk `ch` (and cw of chunk ch + 1) visible; buf ^ 1 free
      if (ch + 1 < nchunk) fetch(ch + 1);            // global loads in flight under the MFMAs
      if (ch + 2 < nchunk) fetch_w(ch + 2);
      if (work) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const h16x8 ah = Ah[buf][2 * s2 + h][(wv >> 1) * 32 + c], al = Al[buf][2 * s2 + h][(wv >> 1) * 32 + c];
          const h16x8 bh = Bh[buf][2 * s2 + h][(wv & 1) * 32 + c], bl = Bl[buf][2 * s2 + h][(wv & 1) * 32 + c];
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2, 0, 0, 0);
          acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2, 0, 0, 0);
        }
      }
      if (ch + 2 < nchunk) stash_w(ch + 2);          // cw[(ch + 2) % 3]: last read (chunk ch - 1) before the barrier above
      if (ch + 1 < nchunk) store(buf ^ 1, ch + 1);   // reads cw[(ch + 1) % 3], published by an earlier barrier
    }
  }
  if (work) {
    float* H = a.Hext + (size_t)b * a.ldJ * a.ldJ;
    const float fin = any ? ldexpf(1.f, -q_cur) : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const int col = bj * 32 + c;
      if (row < a.ldJ && col < a.ldJ) H[(size_t)row * a.ldJ + col] = fmaf(acc2[r], 1.f / 2048.f, acc[r]) * fin;
    }
  }
}

#ifdef HM_EXPERIMENTAL      // K4w: one workgroup per instance for the whole block triangle -- half the CU-time of K4h, but 141 us of
#include "experimental/hm_normal_eq_w.inc"   // latency on B CUs: -2.7 % end to end (profiles/r05_k4_ab.txt); not in the product library
#endif

namespace hm {

int launch_normal_eq(const RowSegment* segs, int n_seg, int L, int B, const int* d_active, float* d_Hext,
                     hipStream_t stream, int split_f16) {
  NormalEqArgs a;
  for (int i = 0; i < 3; ++i) a.seg[i] = segs[i < n_seg ? i : 0];
  a.n_seg = n_seg; a.L = L; a.ldJ = L + POSE_PAD; a.nblk = L / 32 + 1; a.B = B; a.active = d_active;
  a.Hext = d_Hext;
  const int ntile = (a.nblk + 1) / 2;
  const int ntp = ntile * (ntile + 1) / 2;
  const int inst_per_xcd = (B + 7) / 8;
  const int grid = inst_per_xcd * ntp * 8;
#ifdef HM_EXPERIMENTAL
  if (split_f16 == 2) hipLaunchKernelGGL(k_normal_eq_w, dim3(B), dim3(512), 0, stream, a);
  else
#endif
  if (split_f16) hipLaunchKernelGGL(k_normal_eq_h, dim3(grid), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(k_normal_eq, dim3(grid), dim3(256), 0, stream, a);
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hm
