// K1h: the fused decoder forward + input-gradient backward of hm_decoder.hip on the fp16 matrix cores
// (v_mfma_f32_32x32x16_f16, 16x the rate of the fp32-input MFMA) with SPLIT OPERANDS so that the result keeps
// ~2^-22 relative accuracy (fp32 class):
//        W X  =  Wh Xh  +  Wh (X - Xh)  +  (W - Wh) Xh  +  O(2^-22)
// All three products accumulate into ONE fp32 accumulator because the operands are pre-scaled by powers of two:
//   weights by 2^s per stage, packed on the host as  hi = fp16(W 2^s),  lo = fp16(W 2^s - hi);
//   the low part of the activations by 2^11:          Xh = fp16(X),     Xl' = fp16((X - Xh) 2^11);
//   and the weight operand of the middle term by 2^-11 (an exact exponent shift, v_pk_mul_f16, in registers).
// Every fp16 x fp16 product is exact in fp32 and the accumulator is fp32, so the only errors are the 2^-22 second
// order term and fp32 summation rounding.  Activations live in LDS as two fp16 planes [k/8][q][8] (2 x 64 KiB).
// Tiling, register-resident ReLU masks, latent folding and the VALU side paths are those of hm_decoder.hip
// (reference: deepsdf/networks/deep_sdf_decoder.py:75-110, wild_completion/utils.py:112-193, loss.py:229-241).
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "hm_common.h"
#include "hm_internal.h"
#include "hm_gemm_h.h"

using namespace hm;

namespace {

// One decode job = a contiguous range of workgroups of a launch.  A launch carries up to two jobs (the LM iteration
// puts the SDF-term forward+backward tiles and the forward-only tiles of the ray samples into ONE grid: 1024 + ~830 tiles
// instead of 4 + 3.25 rounds of 256), so the job kind is a wave-uniform RUNTIME value:
//   mode 0  forward only (sdf values); with `mask_out` it also saves the ReLU masks of every tile (512 B per query)
//   mode 1  forward + input-gradient backward (sdf + Jacobian rows)
//   mode 2  backward only: the queries are a GATHERED subset of an earlier mode-0 job (src_slot -> its compacted query
//           index); sdf and ReLU masks come from that job's outputs, so the 8 forward stages are not recomputed.  Per
//           query the arithmetic of the backward stages is exactly that of mode 1, hence bit-identical Jacobian rows.
struct DecSeg {
  const float* pts;      // [B][n_stride][4]
  const int* n_q;        // [B]
  float* y;              // [B][n_stride] (modes 0, 1) or nullptr
  float* J;              // [B][n_stride][ldJ] (modes 1, 2)
  uint2* mask_out;       // mode 0: [B][n_stride / 64][8 layers][512 threads] or nullptr
  const uint2* mask_in;  // mode 2: the mode-0 job's mask_out
  const int* src_slot;   // mode 2: [B][n_stride] query index in the mode-0 job
  const float* y_in;     // mode 2: the mode-0 job's y  ([B][src_stride])
  int n_stride;
  int src_stride;        // mode 2: n_stride of the mode-0 job
  int mode;
  int pose_dim;
};

struct DecodeArgsH {
  DecoderDev dec;
  const int* active;
  const float* c0;
  const float* c4;
  int B;
  int ldJ;
  int n_blocks0;         // workgroups of seg[0]; the rest belongs to seg[1]
  DecSeg seg[2];
  long long* trace;    // optional shader-clock stamps of block 0, thread trace_tid (perf analysis), or nullptr:
                       // k_decoder_h [NSTAGE][4] + 1, experimental k_decoder_g [NSTAGE][8] + 1
  int trace_tid;
  int tune;            // experimental builds (HM_EXPERIMENTAL): bits 0-1 priority scheme of k_decoder_g (0 none, 1 K loops high,
                       // 2 epilogues high), bit 3 four-set weight ring in k_decoder_h; 0 in the product
};

#ifdef HM_EXPERIMENTAL
#include "experimental/hm_gemm_loop_h4.inc"
#endif

// ---- one-pass K loop (precision 2, backward stages only): G W = Gh Wh, a single fp16 MFMA pass on the hi parts.
// A K-step is 4 MFMAs (128 matrix-pipe cycles per wave) instead of 12, so the weight fetches run THREE steps ahead
// (ring of four hi-only sets) and the activation reads one; the loop is unrolled by four (branch-free groups).
struct ASet1 { f16x8 h0, h1; };
struct BSet1 { f16x8 h0, h1; };

template <bool U0, bool U1>
__device__ __forceinline__ void step_h1(f32x16 (&acc)[2][2], const ASet1& a, const BSet1& b, ASet1& an, BSet1& bn,
                                        const WSrc wp0, const WSrc wp1, int ka,
                                        const f16x8* xh, int kb, int xo) {
  const WSrc w0 = wp0 + ka * 128;
  const WSrc w1 = wp1 + ka * 128;
  const f16x8* ph = xh + kb * 2 * TQ + xo;
  HM_FENCE();
  if (U0) { HM_MFMA(a.h0, b.h0, acc[0][0]); HM_MFMA(a.h0, b.h1, acc[0][1]); }
  else    { HM_MFMA(a.h1, b.h0, acc[1][0]); HM_MFMA(a.h1, b.h1, acc[1][1]); }
  HM_FENCE(); bn.h0 = ph[0]; bn.h1 = ph[32]; HM_FENCE();
  if (U0 && U1) { HM_MFMA(a.h1, b.h0, acc[1][0]); HM_MFMA(a.h1, b.h1, acc[1][1]); }
  HM_FENCE();
  if (U0) an.h0 = w0[0];
  if (U1) an.h1 = w1[0];
  HM_FENCE();
}

template <bool U0, bool U1>
__device__ __forceinline__ void gemm_loop_h1(f32x16 (&acc)[2][2], const WSrc wp0,
                                             const WSrc wp1, int n_k16, const f16x8* xh, int lane) {
  const int xo = (lane >> 5) * TQ + (lane & 31);
  const int last = n_k16 - 1;
  ASet1 a0 = {}, a1 = {}, a2 = {}, a3 = {};
  BSet1 b0, b1 = {};
  auto lda = [&](ASet1& a, int k) {
    k = k < last ? k : last;
    if (U0) a.h0 = wp0[k * 128];
    if (U1) a.h1 = wp1[k * 128];
  };
  lda(a0, 0); lda(a1, 1); lda(a2, 2);
  b0.h0 = xh[xo]; b0.h1 = xh[xo + 32];
#define HM_STEP1(AS, BS, ANEXT, BNEXT, I)                                                          \
  if (HM_COND(I)) {                                                                                \
    step_h1<U0, U1>(acc, AS, BS, ANEXT, BNEXT, wp0, wp1, (ks + (I) + 3 < n_k16) ? ks + (I) + 3 : last, xh, \
                    (ks + (I) + 1 < n_k16) ? ks + (I) + 1 : last, xo);                             \
  }
  int ks = 0;
#define HM_COND(I) true
  for (; ks + 4 <= n_k16; ks += 4) {
    HM_STEP1(a0, b0, a3, b1, 0)
    HM_STEP1(a1, b1, a0, b0, 1)
    HM_STEP1(a2, b0, a1, b1, 2)
    HM_STEP1(a3, b1, a2, b0, 3)
  }
#undef HM_COND
#define HM_COND(I) (ks + (I) < n_k16)
  if (ks < n_k16) {
    HM_STEP1(a0, b0, a3, b1, 0)
    HM_STEP1(a1, b1, a0, b0, 1)
    HM_STEP1(a2, b0, a1, b1, 2)
  }
#undef HM_COND
#undef HM_STEP1
}

// hi plane only (one-pass stages): 4 consecutive-row values -> 8 bytes
__device__ __forceinline__ void hi_store(f16x4* xh4, int idx, const float (&v)[4]) {
  f16x4 h;
#pragma unroll
  for (int j = 0; j < 4; ++j) h[j] = (_Float16)v[j];
  xh4[idx] = h;
}

// X[k][q] as float for k = 8*grp + j (VALU side paths)
template <bool HI_ONLY = false>
__device__ __forceinline__ void load_group(const f16x8* xh, const f16x8* xl, int grp, int q, float (&x)[8]) {
  const f16x8 h = xh[grp * TQ + q];
  if (HI_ONLY) {
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (float)h[j];
  } else {
    const f16x8 l = xl[grp * TQ + q];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (float)h[j] + (float)l[j] * LO_UNSCALE;
  }
}

#define HM_MASK_CASES(OP) \
  case 0: OP(mk0); break; case 1: OP(mk1); break; case 2: OP(mk2); break; case 3: OP(mk3); break; \
  case 4: OP(mk4); break; case 5: OP(mk5); break; case 6: OP(mk6); break; default: OP(mk7); break;

// G7 = d sdf / d h7 = (1 - y^2) * w8, masked by layer 7's ReLU pattern: the input of the first backward stage, written
// into the activation planes (tail of EPI_FWD7 in modes 1, prologue of mode 2).  dyA / dyB: 1 - y^2 of this lane's two
// queries (qa, 32 + qa).
template <bool BW1>
__device__ __forceinline__ void write_g7(f16x4* xh4, f16x4* xl4, const float* bl, uint2 mk, int mb0, int mb1, int hi,
                                         int qa, float dyA, float dyB, f16x2 xm2) {
#pragma unroll
  for (int sl = 0; sl < 2; ++sl) {
    const int mb = sl == 0 ? mb0 : mb1;
    const uint32_t bits = sl == 0 ? mk.x : mk.y;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int f4 = mb * 32 + 8 * g + 4 * hi;
      const f32x4 wv = *reinterpret_cast<const f32x4*>(bl + 8 * HID + f4);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const float dy = nb == 0 ? dyA : dyB;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = mask_keep(dy * wv[j], bits, mask_pos(g, nb, j));
        if (BW1) hi_store(xh4, ((f4 >> 3) * TQ + nb * 32 + qa) * 2 + hi, v);
        else {
          f16x2 unused = xm2;      // dy * w8 cannot overflow where the forward did not: not tracked
          split_store<false>(xh4, xl4, ((f4 >> 3) * TQ + nb * 32 + qa) * 2 + hi, f32x2{v[0], v[1]}, f32x2{v[2], v[3]}, unused);
        }
      }
    }
  }
}

// ReLU-mask word of a gathered query: the saved word holds, per row group g, one nibble for the source tile's query
// block 0 (bits 31-28, 23-20, ...) and one for block 1 (27-24, 19-16, ...); move the nibbles of the source block
// (nb_src) to the positions of the destination block (nb).
__device__ __forceinline__ uint32_t mask_nibbles(uint32_t w, int nb_src, int nb) {
  const uint32_t sel = nb_src == 0 ? (w & 0xF0F0F0F0u) : (w & 0x0F0F0F0Fu);
  if (nb_src == nb) return sel;
  return nb == 0 ? (sel << 4) : (sel >> 4);
}

// TAG only names the launch site in profiles (0: the iteration's main launch, 1: the render Jacobian launch, 2: the
// decode API).  BW1: precision 2 ("f16x3f_f16b"): the forward stages (residuals, ReLU masks) run the three-pass split
// product, the eight backward stages (Jacobian rows) ONE fp16 pass on the hi parts -- 4 instead of 6 matrix passes per query.
template <int TAG, bool BW1>
__global__ __launch_bounds__(512, 2) void k_decoder_h(const DecodeArgsH a) {
  __shared__ f16x8 xh[64 * TQ];    // 64 KiB: hi plane  X[k/8][q][8]
  __shared__ f16x8 xl[64 * TQ];    // 64 KiB: lo plane (scaled by 2^11)
  __shared__ float sc[2048 + 128];
  __shared__ float bl[9 * HID];    // biases of the 8 forward stages (per-instance c0/c4 included) + lin8's weight row,
                                   // staged once per tile: the epilogues never wait on global memory

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int si = blockIdx.x >= (unsigned)a.n_blocks0 ? 1 : 0;
  const DecSeg& sg = a.seg[si];
  const int blk = si ? blockIdx.x - a.n_blocks0 : blockIdx.x;
  const int mode = sg.mode;
  const int b = (blk + blk / a.B) % a.B;          // tile-major block order, instance rotated by the tile (see hm_decoder.hip)
  const int tile = blk / a.B;
  const int q0 = tile * TQ;
  if (a.active != nullptr && a.active[b] == 0) return;
  const int nq = sg.n_q[b];
  if (q0 >= nq) return;
  const int cnt = (nq - q0 < TQ) ? nq - q0 : TQ;

  const int L = a.dec.L, m = a.dec.m, mb_zx = a.dec.mb_zx;
  const size_t qbase = (size_t)b * sg.n_stride + q0;
  const int qa = lane & 31;
  const int hi = lane >> 5;
  const f32x4* pts4 = reinterpret_cast<const f32x4*>(sg.pts);
  f16x4* xh4 = reinterpret_cast<f16x4*>(xh);
  f16x4* xl4 = reinterpret_cast<f16x4*>(xl);
  float* Jout = sg.J;
  const int mb0 = w, mb1 = w + 8;

  uint2 mk0 = {0, 0}, mk1 = {0, 0}, mk2 = {0, 0}, mk3 = {0, 0}, mk4 = {0, 0}, mk5 = {0, 0}, mk6 = {0, 0},
        mk7 = {0, 0};
  f32x16 acc[2][2];
  float gx0 = 0.f, gx1 = 0.f, gx2 = 0.f;
  float y_keep = 0.f;
  // Largest |value| this lane ever stored into the fp16 activation planes.  fp16 tops out at 65504: beyond it the hi
  // part becomes inf, the next ReLU turns the resulting NaN into 0 and the network would silently compute garbage.
  // A tile that overflowed is poisoned instead (NaN sdf, NaN Jacobian rows), which the solver reports as
  // HM_STATUS_SOLVE_FAILED; the exact fp32 arithmetic (precision 0) has no such limit.
  f16x2 xm2 = {(_Float16)0.f, (_Float16)0.f};   // range guard: running max of |hi|, see track_max
  bl[8 * HID + tid] = a.dec.w8[tid];

  if (mode != 2) {
    // stage 0 input: rows 0..2 = xyz, rows 3..15 = 0  (groups 0 and 1)
    if (tid < TQ) {
      const f32x4 p = pts4[qbase + tid];
      const f32x2 zz = {0.f, 0.f};
      f16x2 untracked = {(_Float16)0.f, (_Float16)0.f};     // query coordinates: far inside the fp16 range by contract
      split_store<false>(xh4, xl4, (0 * TQ + tid) * 2 + 0, f32x2{p[0], p[1]}, f32x2{p[2], 0.f}, untracked);
      split_store<false>(xh4, xl4, (0 * TQ + tid) * 2 + 1, zz, zz, untracked);
      split_store<false>(xh4, xl4, (1 * TQ + tid) * 2 + 0, zz, zz, untracked);
      split_store<false>(xh4, xl4, (1 * TQ + tid) * 2 + 1, zz, zz, untracked);
    }
    const float* cbias0 = a.c0 + (size_t)b * HID;
    const float* cbias4 = a.c4 + (size_t)b * HID;
    // stage all forward biases in LDS: the epilogues then never wait on global memory
    for (int i = tid; i < 8 * HID; i += 512) {
      const StageDesc& sb = a.dec.st[i >> 9];
      const float* src = sb.inst_bias == 1 ? cbias0 : (sb.inst_bias == 2 ? cbias4 : sb.bias);
      bl[i] = src[i & (HID - 1)];
    }
  } else {
    // backward only: sdf and ReLU masks of this tile's queries come from the forward job they were gathered from
    const int* slot_p = sg.src_slot + qbase;
    const int tiles_src = sg.src_stride / TQ;
    float dy[2] = {0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int q = nb * 32 + qa;
      if (q >= cnt) continue;                                      // padding queries: masks 0 -> zero gradients
      const int slot = slot_p[q];
      const float yv = sg.y_in[(size_t)b * sg.src_stride + slot];
      dy[nb] = 1.f - yv * yv;
      if (hi == 0 && w == 0) sc[2048 + q] = yv;
      const int nb_src = (slot >> 5) & 1;
      const uint2* mp = sg.mask_in + (((size_t)b * tiles_src + (slot >> 6)) * 8) * 512 + (w * 64 + hi * 32 + (slot & 31));
#define HM_GATHER(M, LAYER)                                                     \
      { const uint2 v = mp[(LAYER) * 512];                                      \
        M.x |= mask_nibbles(v.x, nb_src, nb); M.y |= mask_nibbles(v.y, nb_src, nb); }
      HM_GATHER(mk0, 0) HM_GATHER(mk1, 1) HM_GATHER(mk2, 2) HM_GATHER(mk3, 3)
      HM_GATHER(mk4, 4) HM_GATHER(mk5, 5) HM_GATHER(mk6, 6) HM_GATHER(mk7, 7)
#undef HM_GATHER
    }
    __syncthreads();                                               // bl (lin8's row) and the sdf values are staged
    y_keep = lane < cnt ? sc[2048 + lane] : 0.f;                   // only wave 0's value is used (final row store)
    write_g7<BW1>(xh4, xl4, bl, mk7, mb0, mb1, hi, qa, dy[0], dy[1], xm2);
  }

  const int s_begin = mode == 2 ? 8 : 0;
  const int s_end = mode == 0 ? 8 : NSTAGE;
  for (int s = s_begin; s < s_end; ++s) {
    const StageDesc& sd = a.dec.st[s];
    const StageDescH& sh = a.dec.sth[s];
    const int epi = sd.epi;
    const bool u0 = (mb0 >= sd.mb_lo) && (mb0 < sd.mb_hi);
    const bool u1 = (mb1 >= sd.mb_lo) && (mb1 < sd.mb_hi);
    const float us = sh.unscale;
    __syncthreads();
    const bool tr = a.trace != nullptr && blockIdx.x == 0 && tid == a.trace_tid;
    if (tr) a.trace[s * 4 + 0] = clock64();

    if (epi == EPI_BWD4 || epi == EPI_BWD0) {
      // stage the 512 x 4 xyz columns in LDS scratch (one 16-byte load per thread), then broadcast-read them
      reinterpret_cast<f32x4*>(sc)[tid] = reinterpret_cast<const f32x4*>(epi == EPI_BWD4 ? a.dec.w4x : a.dec.w0x)[tid];
      __syncthreads();
      const f32x4* wx = reinterpret_cast<const f32x4*>(sc);
#pragma unroll 2
      for (int g = 0; g < 8; ++g) {
        float x[8];
        load_group<BW1>(xh, xl, 8 * w + g, lane, x);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const f32x4 wv = wx[64 * w + 8 * g + j];
          gx0 = fmaf(x[j], wv[0], gx0);
          gx1 = fmaf(x[j], wv[1], gx1);
          gx2 = fmaf(x[j], wv[2], gx2);
        }
      }
    }

    acc[0][0] = zero16h(); acc[0][1] = zero16h();
    acc[1][0] = zero16h(); acc[1][1] = zero16h();
    if (epi == EPI_BWD0 && u1) {
      // d sdf/d z so far (lin4's transpose) was parked in this tile's J rows (true units); this stage accumulates at
      // 2^shift, so seed the accumulators with it.  Parking it in L2 instead of 32 VGPRs across three stages is what
      // keeps the K loop free of spills.
      const float rs = 1.f / us;
      const int jz = (mb1 - mb_zx) * 32;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int q = nb * 32 + qa;
        if (q < cnt) {
          const float* row = Jout + (qbase + q) * (size_t)a.ldJ;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + jz + 8 * g + 4 * hi);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[1][nb][4 * g + j] = v[j] * rs;
          }
        }
      }
    }

    {
      const WSrc wp0 = make_wsrc(sh.wp, (mb0 - sd.mb_lo) * sh.mb_stride + lane);
      const WSrc wp1 = make_wsrc(sh.wp, (mb1 - sd.mb_lo) * sh.mb_stride + lane);
      if (BW1 && s >= 8) {
        if (u0 && u1) gemm_loop_h1<true, true>(acc, wp0, wp1, sh.n_k16, xh, lane);
        else if (u0) gemm_loop_h1<true, false>(acc, wp0, wp1, sh.n_k16, xh, lane);
        else if (u1) gemm_loop_h1<false, true>(acc, wp0, wp1, sh.n_k16, xh, lane);
      } else {
        if (u0 && u1) {
#ifdef HM_EXPERIMENTAL
          if (!BW1 && (a.tune & 8)) gemm_loop_h4<true, true>(acc, wp0, wp1, sh.n_k16, xh, xl, lane);
          else if (!BW1 && (a.tune & 16)) gemm_loop_h<true, true, true>(acc, wp0, wp1, sh.n_k16, xh, xl, lane, w >> 2, (a.tune >> 5) & 1);
          else
#endif
          gemm_loop_h<true, true>(acc, wp0, wp1, sh.n_k16, xh, xl, lane);
        } else if (u0) gemm_loop_h<true, false>(acc, wp0, wp1, sh.n_k16, xh, xl, lane);
        else if (u1) gemm_loop_h<false, true>(acc, wp0, wp1, sh.n_k16, xh, xl, lane);
      }
    }
    if (tr) a.trace[s * 4 + 1] = clock64();
    __syncthreads();
    if (tr) a.trace[s * 4 + 2] = clock64();

    if (epi <= EPI_FWD7) {
      const float* bias = bl + s * HID;
      uint2 mk = {0, 0};
      // lin3's rows m..m+2 (always rows 29..31 of their block: m = 509 - L, L % 32 == 0) carry xyz into the skip layer:
      // one (slot, g = 3, hi = 1) group of one wave, handled where it occurs instead of a test in every group
      const int mbx = m >> 5;
      // all bias vectors of this wave's rows first (one wait), then pure arithmetic + stores
      f32x4 bv[2][4];
#pragma unroll
      for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          bv[sl][g] = *reinterpret_cast<const f32x4*>(bias + (sl == 0 ? mb0 : mb1) * 32 + 8 * g + 4 * hi);
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const bool use = sl == 0 ? u0 : u1;
        if (!use) continue;
        const int mb = sl == 0 ? mb0 : mb1;
        uint32_t bits = 0;
        const f32x2 us2 = {us, us};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int f4 = mb * 32 + 8 * g + 4 * hi;
          const f32x2 b01 = {bv[sl][g][0], bv[sl][g][1]}, b23 = {bv[sl][g][2], bv[sl][g][3]};
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const f32x2 a01 = {acc[sl][nb][4 * g + 0], acc[sl][nb][4 * g + 1]};
            const f32x2 a23 = {acc[sl][nb][4 * g + 2], acc[sl][nb][4 * g + 3]};
            const f32x2 v01 = __builtin_elementwise_fma(a01, us2, b01), v23 = __builtin_elementwise_fma(a23, us2, b23);
            mask_push(bits, v01[0]); mask_push(bits, v01[1]); mask_push(bits, v23[0]); mask_push(bits, v23[1]);
            f32x2 r01 = {fmaxf(v01[0], 0.f), fmaxf(v01[1], 0.f)}, r23 = {fmaxf(v23[0], 0.f), fmaxf(v23[1], 0.f)};
            if (g == 3 && epi == EPI_FWD3 && mb == mbx && hi == 1) {
              const f32x4 p = pts4[qbase + nb * 32 + qa];
              r01[1] = p[0]; r23[0] = p[1]; r23[1] = p[2];
            }
            split_store<false>(xh4, xl4, ((f4 >> 3) * TQ + nb * 32 + qa) * 2 + hi, r01, r23, xm2);
          }
        }
        if (sl == 0) mk.x = bits; else mk.y = bits;
      }
#define HM_SET(M) M = mk
      if (mode == 1) switch (sd.layer) { HM_MASK_CASES(HM_SET) }
#undef HM_SET
      if (mode == 0 && sg.mask_out != nullptr)       // saved for a later backward-only job over a subset of these queries
        sg.mask_out[(((size_t)b * (sg.n_stride / TQ) + tile) * 8 + sd.layer) * 512 + tid] = mk;

      if (epi == EPI_FWD7) {
        __syncthreads();
        float part = 0.f;
#pragma unroll 2
        for (int g = 0; g < 8; ++g) {
          float x[8];
          load_group(xh, xl, 8 * w + g, lane, x);
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(bl + 8 * HID + 64 * w + 8 * g);
          const f32x4 w1 = *reinterpret_cast<const f32x4*>(bl + 8 * HID + 64 * w + 8 * g + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { part = fmaf(x[j], w0[j], part); }
#pragma unroll
          for (int j = 0; j < 4; ++j) { part = fmaf(x[4 + j], w1[j], part); }
        }
        if (__any(!(fmaxf((float)xm2[0], (float)xm2[1]) < 65504.f))) part = __builtin_nanf("");
        sc[w * 64 + lane] = part;
        __syncthreads();
        float a8 = 0.f;
#pragma unroll
        for (int i = 0; i < NWAVE; ++i) a8 += sc[i * 64 + lane];
        a8 += a.dec.b8;
        const float yv = tanhf(a8);
        y_keep = yv;
        if (w == 0) {
          if (lane < cnt) sg.y[qbase + lane] = yv;
          sc[2048 + lane] = 1.f - yv * yv;
        }
        if (mode == 0) return;
        __syncthreads();
        write_g7<BW1>(xh4, xl4, bl, mk, mb0, mb1, hi, qa, sc[2048 + qa], sc[2048 + 32 + qa], xm2);
      }
    } else if (epi == EPI_BWD || epi == EPI_BWD4) {
      uint2 mk;
#define HM_GET(M) mk = M
      switch (sd.layer) { HM_MASK_CASES(HM_GET) }
#undef HM_GET
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const bool use = sl == 0 ? u0 : u1;
        if (!use) continue;
        const int mb = sl == 0 ? mb0 : mb1;
        if (epi == EPI_BWD4 && mb >= mb_zx) {      // latent rows: park in J (same thread re-reads them in BWD0)
          if (sl == 1) {
            const int jz = (mb1 - mb_zx) * 32;
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
              const int q = nb * 32 + qa;
              if (q < cnt) {
                float* row = Jout + (qbase + q) * (size_t)a.ldJ;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                  f32x4 v;
#pragma unroll
                  for (int j = 0; j < 4; ++j) v[j] = acc[1][nb][4 * g + j] * us;
                  *reinterpret_cast<f32x4*>(row + jz + 8 * g + 4 * hi) = v;
                }
              }
            }
          }
          continue;
        }
        const uint32_t bits = sl == 0 ? mk.x : mk.y;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int f4 = mb * 32 + 8 * g + 4 * hi;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const f32x2 us2 = {us, us};
            const f32x2 p01 = f32x2{acc[sl][nb][4 * g + 0], acc[sl][nb][4 * g + 1]} * us2;
            const f32x2 p23 = f32x2{acc[sl][nb][4 * g + 2], acc[sl][nb][4 * g + 3]} * us2;
            const f32x2 v01 = {mask_keep(p01[0], bits, mask_pos(g, nb, 0)), mask_keep(p01[1], bits, mask_pos(g, nb, 1))};
            const f32x2 v23 = {mask_keep(p23[0], bits, mask_pos(g, nb, 2)), mask_keep(p23[1], bits, mask_pos(g, nb, 3))};
            if (BW1) {
              const float v[4] = {v01[0], v01[1], v23[0], v23[1]};
              f16x4 hq;
#pragma unroll
              for (int j = 0; j < 4; ++j) hq[j] = (_Float16)v[j];
              uint2 hu;
              __builtin_memcpy(&hu, &hq, 8);
              track_max(xm2, hu.x, hu.y, true);
              xh4[((f4 >> 3) * TQ + nb * 32 + qa) * 2 + hi] = hq;
            } else {
              split_store<true>(xh4, xl4, ((f4 >> 3) * TQ + nb * 32 + qa) * 2 + hi, v01, v23, xm2);
            }
          }
        }
      }
    } else {  // EPI_BWD0
      if (u1) {
        const int jz = (mb1 - mb_zx) * 32;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const int q = nb * 32 + qa;
          if (q < cnt) {
            float* row = Jout + (qbase + q) * (size_t)a.ldJ;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              f32x4 v;
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = acc[1][nb][4 * g + j] * us;
              *reinterpret_cast<f32x4*>(row + jz + 8 * g + 4 * hi) = v;
            }
          }
        }
      }
    }
  }

  if (a.trace != nullptr && blockIdx.x == 0 && tid == 0) a.trace[NSTAGE * 4] = clock64();
  if (mode == 0) return;
  if (__any(!(fmaxf((float)xm2[0], (float)xm2[1]) < 65504.f))) gx0 = __builtin_nanf("");
  sc[(w * 4 + 0) * 64 + lane] = gx0;
  sc[(w * 4 + 1) * 64 + lane] = gx1;
  sc[(w * 4 + 2) * 64 + lane] = gx2;
  __syncthreads();
  if (w == 0 && lane < cnt) {
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
    for (int i = 0; i < NWAVE; ++i) {
      g0 += sc[(i * 4 + 0) * 64 + lane];
      g1 += sc[(i * 4 + 1) * 64 + lane];
      g2 += sc[(i * 4 + 2) * 64 + lane];
    }
    const f32x4 p = pts4[qbase + lane];
    float* row = Jout + (qbase + lane) * (size_t)a.ldJ + L;
    row[7] = y_keep;
    row[0] = g0; row[1] = g1; row[2] = g2;
    if (sg.pose_dim != 0) {
      row[3] = g2 * p[1] - g1 * p[2];
      row[4] = g0 * p[2] - g2 * p[0];
      row[5] = g1 * p[0] - g0 * p[1];
      if (sg.pose_dim == 7) row[6] = g0 * p[0] + g1 * p[1] + g2 * p[2];
    }
  }
}


#ifdef HM_EXPERIMENTAL      // scripts/build_variant.sh exp -DHM_EXPERIMENTAL: the round-4 schedule experiments (measured slower)
#include "experimental/hm_decoder_g.inc"
#endif

}  // namespace

static long long* g_trace = nullptr;
static int g_trace_tid = 0;
extern "C" void hm_debug_set_trace_thread(int tid) { g_trace_tid = tid; }
#ifdef HM_EXPERIMENTAL
// Experimental builds only: f16x3 kernel variant 0 = k_decoder_h (product), 1 = k_decoder_g<., true> (ping-pong wave groups),
// 2 = k_decoder_g<., false> (lockstep, raw barriers, primed four-set weight ring); tune = DecodeArgsH::tune.  Same bits.
static int g_k1h_tune = 0;
extern "C" void hm_debug_k1h_tune(int t) { g_k1h_tune = t; }
static int g_k1h_variant = 0;
extern "C" void hm_debug_k1h_variant(int v) { g_k1h_variant = (v >= 0 && v <= 2) ? v : 0; }
#endif
extern "C" void hm_debug_set_trace(long long* d_buf) { g_trace = d_buf; }

namespace hm {

namespace {

template <int TAG>
int launch_h(const hm_decoder_s* dec, DecodeArgsH& a, int grid, hipStream_t stream) {
  if (grid == 0) return 0;
  a.dec = dec->dev;
  a.trace = g_trace;
  a.trace_tid = g_trace_tid;
#ifdef HM_EXPERIMENTAL
  a.tune = g_k1h_tune;
#endif
  if (dec->precision == 2) hipLaunchKernelGGL((k_decoder_h<TAG, true>), dim3(grid), dim3(512), 0, stream, a);
#ifdef HM_EXPERIMENTAL
  else if (g_k1h_variant == 1) hipLaunchKernelGGL((k_decoder_g<TAG, true>), dim3(grid), dim3(512), 0, stream, a);
  else if (g_k1h_variant == 2) hipLaunchKernelGGL((k_decoder_g<TAG, false>), dim3(grid), dim3(512), 0, stream, a);
#endif
  else hipLaunchKernelGGL((k_decoder_h<TAG, false>), dim3(grid), dim3(512), 0, stream, a);
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}

DecSeg plain_seg(const float* d_pts, const int* d_nq, int n_stride, float* d_y, float* d_J, int pose_dim, int mode) {
  DecSeg s;
  memset(&s, 0, sizeof(s));
  s.pts = d_pts; s.n_q = d_nq; s.y = d_y; s.J = d_J; s.n_stride = n_stride; s.mode = mode; s.pose_dim = pose_dim;
  return s;
}

}  // namespace

int launch_decoder_h(const hm_decoder_s* dec, int B, const float* d_pts, const int* d_nq, const int* d_active,
                     int n_stride, const float* d_c0, const float* d_c4, float* d_y, float* d_J, int ldJ,
                     int pose_dim, int mode, hipStream_t stream, int tag) {
  DecodeArgsH a;
  memset(&a, 0, sizeof(a));
  a.active = d_active; a.c0 = d_c0; a.c4 = d_c4; a.B = B; a.ldJ = ldJ;
  a.seg[0] = plain_seg(d_pts, d_nq, n_stride, d_y, d_J, pose_dim, mode);
  a.seg[1] = a.seg[0];
  a.n_blocks0 = B * (n_stride / TQ);
  if (tag == 0) return launch_h<0>(dec, a, a.n_blocks0, stream);
  if (tag == 1) return launch_h<1>(dec, a, a.n_blocks0, stream);
  return launch_h<2>(dec, a, a.n_blocks0, stream);
}

// The LM iteration's main launch: SDF-term forward+backward tiles (long) first, then the forward-only tiles of the
// ball-valid ray samples (short), which also save their ReLU masks for the render Jacobian launch.
int launch_decoder_h_main(const hm_decoder_s* dec, int B, const int* d_active, const float* d_c0, const float* d_c4,
                          int ldJ, const float* d_ptsS, const int* d_nS, int nS_stride, float* d_yS, float* d_JS,
                          int pose_dim, const float* d_ptsR, const int* d_nR, int nR_stride, float* d_yR,
                          void* d_maskR, hipStream_t stream) {
  DecodeArgsH a;
  memset(&a, 0, sizeof(a));
  a.active = d_active; a.c0 = d_c0; a.c4 = d_c4; a.B = B; a.ldJ = ldJ;
  a.seg[0] = plain_seg(d_ptsS, d_nS, nS_stride, d_yS, d_JS, pose_dim, 1);
  a.seg[1] = plain_seg(d_ptsR, d_nR, nR_stride, d_yR, nullptr, 0, 0);
  a.seg[1].mask_out = static_cast<uint2*>(d_maskR);
  a.n_blocks0 = B * (nS_stride / TQ);
  return launch_h<0>(dec, a, a.n_blocks0 + B * (nR_stride / TQ), stream);
}

// forward-only over the ray samples with mask saving (the functional render API, which has no SDF term beside it)
int launch_decoder_h_fwd_masks(const hm_decoder_s* dec, int B, const int* d_active, const float* d_c0,
                               const float* d_c4, const float* d_ptsR, const int* d_nR, int nR_stride, float* d_yR,
                               void* d_maskR, hipStream_t stream) {
  DecodeArgsH a;
  memset(&a, 0, sizeof(a));
  a.active = d_active; a.c0 = d_c0; a.c4 = d_c4; a.B = B; a.ldJ = 0;
  a.seg[0] = plain_seg(d_ptsR, d_nR, nR_stride, d_yR, nullptr, 0, 0);
  a.seg[0].mask_out = static_cast<uint2*>(d_maskR);
  a.seg[1] = a.seg[0];
  a.n_blocks0 = B * (nR_stride / TQ);
  return launch_h<2>(dec, a, a.n_blocks0, stream);
}

// The render Jacobian launch: backward only over the gathered with-grad samples (masks + sdf from the main launch).
int launch_decoder_h_bwd(const hm_decoder_s* dec, int B, const int* d_active, const float* d_c0, const float* d_c4,
                         int ldJ, const float* d_ptsG, const int* d_nG, int nG_stride, float* d_JG, int pose_dim,
                         const int* d_srcG, const float* d_yR, const void* d_maskR, int nR_stride,
                         hipStream_t stream) {
  DecodeArgsH a;
  memset(&a, 0, sizeof(a));
  a.active = d_active; a.c0 = d_c0; a.c4 = d_c4; a.B = B; a.ldJ = ldJ;
  a.seg[0] = plain_seg(d_ptsG, d_nG, nG_stride, nullptr, d_JG, pose_dim, 2);
  a.seg[0].mask_in = static_cast<const uint2*>(d_maskR);
  a.seg[0].src_slot = d_srcG; a.seg[0].y_in = d_yR; a.seg[0].src_stride = nR_stride;
  a.seg[1] = a.seg[0];
  a.n_blocks0 = B * (nG_stride / TQ);
  return launch_h<1>(dec, a, a.n_blocks0, stream);
}

}  // namespace hm
