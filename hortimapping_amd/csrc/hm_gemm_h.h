// f16x3 arithmetic on the fp16 matrix cores (v_mfma_f32_32x32x16_f16 on hi / lo split operands; see hm_decoder_h.hip for the
// derivation): operand split + range guard of the epilogues, ReLU-mask helpers and the K loop, shared by the fixed-architecture
// kernel (hm_decoder_h.hip) and the any-architecture kernel (hm_decoder_any.hip).  A operand: weights packed by hm_pack.hip:
// pack_stage_h as [row block][K step of 16][hi | lo][lane][8]; B operand: two fp16 planes in LDS, X[k/8][q][8].
#pragma once
#include "hm_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace hm {
namespace {

constexpr float LO_SCALE = 2048.f;          // 2^11
constexpr float LO_UNSCALE = 1.f / 2048.f;

__device__ __forceinline__ f32x16 zero16h() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Epilogue arithmetic.  The epilogues run between two barriers with the matrix pipe idle and are VALU-bound (64
// accumulator values per lane and stage), so every value is kept to ~6 instructions: packed fp32 fma / mul, packed
// conversions, and the low part by ONE mixed-precision fma per value that reads the fp16 high part directly,
//     lo = f16( hi * -2^11 + v * 2^11 )          (v_fma_mixlo/hi_f16: exact fp32 fma, one rounding to fp16)
// which is bit-identical to f16((v - f32(hi)) * 2^11): the difference and both scalings are exact.
__device__ __forceinline__ uint32_t lo_pair(uint32_t hpk, float s0, float s1) {
  uint32_t d;
  const float c = -LO_SCALE;
  asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(d) : "v"(hpk), "s"(c), "v"(s0), "v"(s1));
  return d;
}

// range guard: running maximum of |hi| as packed fp16 (an overflowing value converts to +-inf and sticks)
__device__ __forceinline__ void track_max(f16x2& xm, uint32_t h0, uint32_t h1, bool take_abs) {
  if (take_abs) { h0 &= 0x7fff7fffu; h1 &= 0x7fff7fffu; }
  f16x2 a, b;
  __builtin_memcpy(&a, &h0, 4);
  __builtin_memcpy(&b, &h1, 4);
  xm = __builtin_elementwise_max(xm, __builtin_elementwise_max(a, b));
}

// split 4 consecutive-row values into the hi / scaled-lo fp16 planes (8 bytes each)
template <bool ABS>
__device__ __forceinline__ void split_store(f16x4* xh4, f16x4* xl4, int idx, const f32x2 a, const f32x2 b, f16x2& xm) {
  const f16x2 ha = __builtin_convertvector(a, f16x2), hb = __builtin_convertvector(b, f16x2);
  const f32x2 sa = a * LO_SCALE, sb = b * LO_SCALE;
  uint32_t hau, hbu;
  __builtin_memcpy(&hau, &ha, 4);
  __builtin_memcpy(&hbu, &hb, 4);
  const uint2 hh = {hau, hbu};
  const uint2 ll = {lo_pair(hau, sa[0], sa[1]), lo_pair(hbu, sb[0], sb[1])};
  track_max(xm, hau, hbu, ABS);
  reinterpret_cast<uint2*>(xh4)[idx] = hh;
  reinterpret_cast<uint2*>(xl4)[idx] = ll;
}

// ReLU masks: bit of the k-th value an epilogue processes sits at position 31 - k of its 32-bit word (the forward
// epilogue shifts the word left and adds the compare result: v_cmp_gt_f32 + v_addc_co_u32, two instructions per value).
// Processing order everywhere: g (row group of 8) outer, nb (query block) inner, j (row within the group's 4) innermost.
__device__ __forceinline__ constexpr int mask_pos(int g, int nb, int j) { return 31 - ((g * 2 + nb) * 4 + j); }
__device__ __forceinline__ void mask_push(uint32_t& bits, float val) {
  asm("v_cmp_gt_f32 vcc, %1, 0\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(val) : "vcc");
}
// x where the mask bit is set, +0 elsewhere (v_bfe_i32 + v_and_b32)
__device__ __forceinline__ float mask_keep(float x, uint32_t bits, int pos) {
  const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)bits, pos, 1);
  return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x) & m);
}

// K loop.  Weights (A, from L2) are fetched TWO K-steps ahead into a ring of three statically named register sets,
// activations (B, from LDS) one step ahead into a ring of two; the loop is unrolled by six so every set has a fixed
// name (no register rotation => hipcc emits counted waits instead of draining vmcnt/lgkmcnt each step), and
// sched_barriers pin each prefetch between the MFMAs it overlaps (hipcc otherwise sinks loads to their first use).
// The weight operand's source: the stage's packed weights as a BUFFER (descriptor in SGPRs), this lane's fixed byte offset
// in it (row block + lane, one VGPR) and a wave-uniform byte offset (K step; scalar).  It indexes like the pointer it
// replaces -- wp + k, wp[k], k in 16-byte units -- but every fetch is a buffer_load_dwordx4 whose only per-lane operand
// never changes: round-6 measurement (scripts/microbench/issue_cost.hip, profiles/r06_issue_cost.txt): with weight fetches
// on 64-bit per-lane addresses (global_load_dwordx4 v, v[a:b], off: what pointer indexing compiled to) a SIMD's two
// waves take 576-611 clocks per 16 MFMAs + 2 fetches + 4 LDS reads each, with this form 544 (512 = the matrix pipe alone).
struct WSrc {
  __amdgpu_buffer_rsrc_t rs;
  int voff, soff;
  __device__ __forceinline__ WSrc operator+(int k) const { return WSrc{rs, voff, soff + k * 16}; }
  __device__ __forceinline__ f16x8 operator[](int k) const {
    return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff + k * 16, 0));
  }
};
// base: wave-uniform start of the stage's weights; lane_unit: this lane's position in it, in 16-byte units
__device__ __forceinline__ WSrc make_wsrc(const void* base, int lane_unit) {
  return WSrc{__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7ffffff0, 0x00020000), lane_unit * 16, 0};
}

struct ASet { f16x8 h0, l0, h1, l1; };
struct BSet { f16x8 h0, h1, l0, l1; };

template <bool U0, bool U1>
__device__ __forceinline__ void load_a(ASet& a, const WSrc wp0, const WSrc wp1, int k) {
  if (U0) { a.h0 = wp0[k * 128]; a.l0 = wp0[k * 128 + 64]; }
  if (U1) { a.h1 = wp1[k * 128]; a.l1 = wp1[k * 128 + 64]; }
}

__device__ __forceinline__ void load_b(BSet& b, const f16x8* xh, const f16x8* xl, int k, int xo) {
  b.h0 = xh[k * 2 * TQ + xo]; b.h1 = xh[k * 2 * TQ + xo + 32];
  b.l0 = xl[k * 2 * TQ + xo]; b.l1 = xl[k * 2 * TQ + xo + 32];
}

#ifdef HM_EXPERIMENTAL      // timing ablations of the K step (HM_ABL_*): experimental builds only
#include "experimental/hm_ablation.inc"
#else
#define HM_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
#define HM_LDA(dst, src) dst = src
#define HM_LDB(dst, src) dst = src
#endif
#define HM_FENCE() __builtin_amdgcn_sched_barrier(0)

// One K-step with its prefetches woven BETWEEN the MFMAs (pairs of MFMAs, then one or two memory instructions): a
// wave that has to wait for a slot in the shared vector-memory / LDS queues does so in the shadow of its own MFMAs
// instead of in front of them.  Activation reads (needed next step) go first, weight fetches (needed two steps on) last.
template <bool U0, bool U1>
__device__ __forceinline__ void step_h(f32x16 (&acc)[2][2], const ASet& a, const BSet& b, ASet& an, BSet& bn,
                                       const WSrc wp0, const WSrc wp1, int ka,
                                       const f16x8* xh, const f16x8* xl, int kb, int xo) {
  const _Float16 cs = (_Float16)LO_UNSCALE;
  const WSrc w0 = wp0 + ka * 128;
  const WSrc w1 = wp1 + ka * 128;
  const f16x8* ph = xh + kb * 2 * TQ + xo;
  const f16x8* pl = xl + kb * 2 * TQ + xo;
  if (U0 && U1) {
#if defined(HM_EXPERIMENTAL) && defined(HM_ABL_NOSCALE)      // round 5 timing ablation: no v_pk_mul_f16 at all (wrong results)
    const f16x8 a0c = a.h0;
    const f16x8 a1c = a.h1;
#elif defined(HM_EXPERIMENTAL) && defined(HM_ABL_HALFSCALE)  // only the second row block's weight operand is rescaled
    const f16x8 a0c = a.h0;
    const f16x8 a1c = a.h1 * cs;
#else
    const f16x8 a0c = a.h0 * cs;
    const f16x8 a1c = a.h1 * cs;
#endif
    HM_FENCE();
    HM_MFMA(a.h0, b.h0, acc[0][0]); HM_MFMA(a.h0, b.h1, acc[0][1]);
    HM_FENCE(); HM_LDB(bn.h0, ph[0]); HM_LDB(bn.h1, ph[32]); HM_FENCE();
    HM_MFMA(a.h1, b.h0, acc[1][0]); HM_MFMA(a.h1, b.h1, acc[1][1]);
    HM_FENCE(); HM_LDB(bn.l0, pl[0]); HM_LDB(bn.l1, pl[32]); HM_FENCE();
    HM_MFMA(a0c, b.l0, acc[0][0]); HM_MFMA(a0c, b.l1, acc[0][1]);
    HM_FENCE(); HM_LDA(an.h0, w0[0]); HM_FENCE();
    HM_MFMA(a1c, b.l0, acc[1][0]); HM_MFMA(a1c, b.l1, acc[1][1]);
    HM_FENCE(); HM_LDA(an.l0, w0[64]); HM_FENCE();
    HM_MFMA(a.l0, b.h0, acc[0][0]); HM_MFMA(a.l0, b.h1, acc[0][1]);
    HM_FENCE(); HM_LDA(an.h1, w1[0]); HM_FENCE();
    HM_MFMA(a.l1, b.h0, acc[1][0]); HM_MFMA(a.l1, b.h1, acc[1][1]);
    HM_FENCE(); HM_LDA(an.l1, w1[64]); HM_FENCE();
  } else if (U0) {
    const f16x8 a0c = a.h0 * cs;
    HM_FENCE();
    HM_MFMA(a.h0, b.h0, acc[0][0]); HM_MFMA(a.h0, b.h1, acc[0][1]);
    HM_FENCE(); bn.h0 = ph[0]; bn.h1 = ph[32]; bn.l0 = pl[0]; bn.l1 = pl[32]; HM_FENCE();
    HM_MFMA(a0c, b.l0, acc[0][0]); HM_MFMA(a0c, b.l1, acc[0][1]);
    HM_FENCE(); an.h0 = w0[0]; an.l0 = w0[64]; HM_FENCE();
    HM_MFMA(a.l0, b.h0, acc[0][0]); HM_MFMA(a.l0, b.h1, acc[0][1]);
    HM_FENCE();
  } else {
    const f16x8 a1c = a.h1 * cs;
    HM_FENCE();
    HM_MFMA(a.h1, b.h0, acc[1][0]); HM_MFMA(a.h1, b.h1, acc[1][1]);
    HM_FENCE(); bn.h0 = ph[0]; bn.h1 = ph[32]; bn.l0 = pl[0]; bn.l1 = pl[32]; HM_FENCE();
    HM_MFMA(a1c, b.l0, acc[1][0]); HM_MFMA(a1c, b.l1, acc[1][1]);
    HM_FENCE(); an.h1 = w1[0]; an.l1 = w1[64]; HM_FENCE();
    HM_MFMA(a.l1, b.h0, acc[1][0]); HM_MFMA(a.l1, b.h1, acc[1][1]);
    HM_FENCE();
  }
}

// ALT (experimental builds, tune bit 4): the two waves of a SIMD take turns at `s_setprio 1`, one group of six K-steps
// (or two, `alt_shift` = 1) each, so that neither runs ahead of the other for a whole K loop (`alt_half` = 0 for the
// older half of the workgroup, 1 for the younger).
template <bool U0, bool U1, bool ALT = false>
__device__ __forceinline__ void gemm_loop_h(f32x16 (&acc)[2][2], const WSrc wp0,
                                            const WSrc wp1, int n_k16, const f16x8* xh,
                                            const f16x8* xl, int lane, int alt_half = 0, int alt_shift = 0) {
  const int xo = (lane >> 5) * TQ + (lane & 31);
  const int last = n_k16 - 1;
  ASet a0 = {}, a1 = {}, a2 = {};
  BSet b0, b1 = {};
  load_a<U0, U1>(a0, wp0, wp1, 0);
  load_a<U0, U1>(a1, wp0, wp1, last < 1 ? last : 1);
  load_b(b0, xh, xl, 0, xo);
#define HM_STEP(AS, BS, ANEXT, BNEXT, I)                                                         \
  if (HM_COND(I)) {                                                                              \
    step_h<U0, U1>(acc, AS, BS, ANEXT, BNEXT, wp0, wp1, (ks + (I) + 2 < n_k16) ? ks + (I) + 2 : last, xh, xl, \
                   (ks + (I) + 1 < n_k16) ? ks + (I) + 1 : last, xo);                            \
  }
  // full groups of six run branch-free: with a conditional per step hipcc's wait-count pass merges the "step
  // skipped" paths and emits vmcnt(1) where vmcnt(4+) is right, which cuts the two-step prefetch distance to one
  int ks = 0;
  int grp6 = 0;
#define HM_COND(I) true
  for (; ks + 6 <= n_k16; ks += 6) {
    if (ALT) {
      if ((((grp6 >> alt_shift) & 1) ^ alt_half) != 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
      ++grp6;
    }
    HM_STEP(a0, b0, a2, b1, 0)
    HM_STEP(a1, b1, a0, b0, 1)
    HM_STEP(a2, b0, a1, b1, 2)
    HM_STEP(a0, b1, a2, b0, 3)
    HM_STEP(a1, b0, a0, b1, 4)
    HM_STEP(a2, b1, a1, b0, 5)
  }
#undef HM_COND
#define HM_COND(I) (ks + (I) < n_k16)
  if (ks < n_k16) {
    HM_STEP(a0, b0, a2, b1, 0)
    HM_STEP(a1, b1, a0, b0, 1)
    HM_STEP(a2, b0, a1, b1, 2)
    HM_STEP(a0, b1, a2, b0, 3)
    HM_STEP(a1, b0, a0, b1, 4)
  }
#undef HM_COND
#undef HM_STEP
  if (ALT) __builtin_amdgcn_s_setprio(0);
}

}  // namespace
}  // namespace hm
