// K1p: the fused decoder forward + input-gradient backward in PLAIN fp16 MFMA arithmetic (precision 3, "f16").
//
// This is the arithmetic BASELINE.json names for configs[4] ("fp16 MFMA decoder"): every GEMM of the network is ONE
// v_mfma_f32_32x32x16_f16 pass on fp16-rounded weights (per-stage power-of-two scale, hm_pack.hip) and fp16-rounded
// activations, fp32 accumulation.  Results are fp16-class (~1e-3 relative), NOT the reference's fp32 -- never the
// default; the parity tables in profiles/ say what that costs at the metric level.
//
// What the single plane buys: the activations of 128 queries fit the LDS (one fp16 plane [k/8][q][8] = 128 KiB, the
// f16x3 kernel needs 128 KiB for 64), so every weight byte fetched from L2 serves 128 queries instead of 64 and only
// the hi plane (half the bytes) is fetched at all: a quarter of the weight stream per query.  At the matrix rate of
// its single pass the stream would still have to run at 32 B/clk/CU, and the measured K phase of a 512 x 512 stage is
// ~26 k clocks against 16.4 k of MFMA (DESIGN.md section 8 has the trace and the micro-benchmarks): 0.31-0.33 of the
// fp16 peak.  Tiling: 512 threads, wave w owns the
// 32-row blocks {w, w+8} x four 32-query blocks (8 accumulators of 32x32 = 128 registers, plus 32 for the ReLU masks of
// the 8 layers); A operands stream L2 -> VGPR two K-steps ahead (ring of three), B operands are refilled in place
// from LDS right after their last use (one set), 8 MFMAs per K-step.
// Stage list, ReLU masks in registers, latent folding, lin4^T latent partial parked in the J rows, VALU side paths
// and the fp16 range guard are those of hm_decoder_h.hip (reference: deepsdf/networks/deep_sdf_decoder.py:75-110,
// wild_completion/utils.py:112-193, loss.py:229-241).
#include <stdlib.h>

#include "hm_common.h"
#include "hm_internal.h"

using namespace hm;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TQP = 128;       // queries per workgroup tile
constexpr int NQB = 4;         // 32-query blocks per tile
constexpr int NWP = 8;         // waves per workgroup (two per SIMD)
constexpr int NRB = 2;         // 32-row blocks per wave: w, w + 8

struct DecodeArgsP {
  DecoderDev dec;
  const float* pts;
  const int* n_q;
  const int* active;
  const float* c0;
  const float* c4;
  float* y;
  float* J;
  int n_stride;
  int B;
  int ldJ;
  int pose_dim;
};

__device__ __forceinline__ f32x16 zero16p() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

__device__ __forceinline__ void store4(f16x4* x4, int idx, const float (&v)[4]) {
  f16x4 h;
#pragma unroll
  for (int j = 0; j < 4; ++j) h[j] = (_Float16)v[j];
  x4[idx] = h;
}

struct ASetP { f16x8 r[NRB]; };
struct BSetP { f16x8 q[NQB]; };

#define HM_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
#define HM_FENCE() __builtin_amdgcn_sched_barrier(0)

// One K-step (16 k): 8 MFMAs (4 when only one of the wave's row blocks is valid in this stage), ordered by query block
// so that each B operand dies after its second use and is REFILLED IN PLACE for the next step right there (the read
// lands ~6 MFMAs = 200+ cycles before its first use): one B set instead of a ring of two -- with 128 accumulator and
// 32 mask registers per lane there is no room for a second.  The weight fetch for two steps on rides in the last gaps.
template <bool U0, bool U1>
__device__ __forceinline__ void step_p(f32x16 (&acc)[NRB][NQB], const ASetP& a, BSetP& b, ASetP& an,
                                       const f16x8* __restrict__ wp0, const f16x8* __restrict__ wp1, int ka,
                                       const f16x8* xp, int kb, int xo) {
  const f16x8* ph = xp + kb * 2 * TQP + xo;
  HM_FENCE();
#pragma unroll
  for (int nb = 0; nb < NQB; ++nb) {
    if (U0) HM_MFMA(a.r[0], b.q[nb], acc[0][nb]);
    if (U1) HM_MFMA(a.r[1], b.q[nb], acc[1][nb]);
    HM_FENCE();
    b.q[nb] = ph[32 * nb];
    if (nb == 2 && U0) an.r[0] = wp0[ka * 128];
    if (nb == 3 && U1) an.r[1] = wp1[ka * 128];
    if (nb == 3 && U0 && !U1) {}
    HM_FENCE();
  }
}

template <bool U0, bool U1, bool DRAIN>
__device__ __forceinline__ void gemm_loop_p(f32x16 (&acc)[NRB][NQB], const f16x8* __restrict__ wp0,
                                            const f16x8* __restrict__ wp1, int n_k16, const f16x8* xp) {
  // The lane id is re-derived HERE (volatile asm: not CSE'd with the kernel's copy) so that the LDS read pointer of the
  // loop is computed in its preheader instead of being reloaded from a spill slot: a scratch reload still in flight
  // at loop entry makes hipcc's wait-count pass put s_waitcnt vmcnt(0) in front of the first ds_read of EVERY
  // iteration, which drains the weight prefetch ring (effective prefetch distance: one K-step).
  int lane;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
  const int xo = (lane >> 5) * TQP + (lane & 31);
  const int last = n_k16 - 1;
  ASetP a0 = {}, a1 = {}, a2 = {};
  BSetP bq;
  if (DRAIN) {   // forward-only kernel: something (a scalar load) is still pending here and poisons the loop head the
    __builtin_amdgcn_s_waitcnt(0x0070);   // same way; draining once (vmcnt(0) lgkmcnt(0), gfx9 encoding) is free there.
    HM_FENCE();                           // In the fwd+bwd kernel the drain would wait for epilogue stores: left out.
  }
  auto lda = [&](ASetP& a, int k) {
    k = k < last ? k : last;
    if (U0) a.r[0] = wp0[k * 128];
    if (U1) a.r[1] = wp1[k * 128];
  };
  lda(a0, 0); lda(a1, 1);
#pragma unroll
  for (int i = 0; i < NQB; ++i) bq.q[i] = xp[xo + 32 * i];
#define HM_STEPP(AS, ANEXT, I)                                                                              \
  if (HM_COND(I)) {                                                                                         \
    step_p<U0, U1>(acc, AS, bq, ANEXT, wp0, wp1, (ks + (I) + 2 < n_k16) ? ks + (I) + 2 : last, xp,          \
                   (ks + (I) + 1 < n_k16) ? ks + (I) + 1 : last, xo);                                       \
  }
  // groups of three run branch-free (statically named ring sets => counted vmcnt / lgkmcnt waits, no register rotation)
  int ks = 0;
#define HM_COND(I) true
  for (; ks + 3 <= n_k16; ks += 3) {
    HM_STEPP(a0, a2, 0)
    HM_STEPP(a1, a0, 1)
    HM_STEPP(a2, a1, 2)
  }
#undef HM_COND
#define HM_COND(I) (ks + (I) < n_k16)
  if (ks < n_k16) {
    HM_STEPP(a0, a2, 0)
    HM_STEPP(a1, a0, 1)
  }
#undef HM_COND
#undef HM_STEPP
}

// X[k][q] as float for k = 8*grp + j (VALU side paths)
__device__ __forceinline__ void load_group_p(const f16x8* xp, int grp, int q, float (&x)[8]) {
  const f16x8 h = xp[grp * TQP + q];
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = (float)h[j];
}

// ReLU masks: 128 bits per lane and layer (2 row blocks x 4 query blocks x 16 accumulator registers)
struct Mask { uint32_t w[4]; };   // w[2 * r + (nb >> 1)], bit (nb & 1) * 16 + reg

// Shader-clock stamps of wave 0 of workgroup 0 (hm_debug_set_k1p_trace, scripts/gpu_trace_k1p.py), per stage s:
// [5 s + 0] stage entered (after the X barrier), +1 K loop starts, +2 K loop done, +3 barrier passed, +4 epilogue done.
// Compiled in only with -DHM_K1P_TRACE: even switched off at run time the stamps cost 3 % of the launch (more live
// values, different spills), stamps of every wave 8 %.
__device__ long long* g_k1p_trace = nullptr;
#ifdef HM_K1P_TRACE
#define HM_STAMP(SLOT) if (trc) g_k1p_trace[SLOT] = clock64()
#else
#define HM_STAMP(SLOT)
#endif

#define HM_MASK_CASES(OP) \
  case 0: OP(mk0); break; case 1: OP(mk1); break; case 2: OP(mk2); break; case 3: OP(mk3); break; \
  case 4: OP(mk4); break; case 5: OP(mk5); break; case 6: OP(mk6); break; default: OP(mk7); break;

template <int MODE, int TAG>
__global__ __launch_bounds__(512, 2) void k_decoder_p(const DecodeArgsP a) {
  __shared__ f16x8 xp[64 * TQP];   // 128 KiB: X[k/8][q][8], fp16
  __shared__ float sc[3072];       // 12 KiB scratch: xyz weight columns / lin8 partials + dy / final xyz-gradient sums
  __shared__ float bl[9 * HID];    // 18 KiB: biases of the 8 forward stages (per-instance c0 / c4 included) + lin8's row

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = (blockIdx.x + blockIdx.x / a.B) % a.B;   // tile-major block order, instance rotated by the tile (see hm_decoder.hip)
  const int q0 = (blockIdx.x / a.B) * TQP;
  if (a.active != nullptr && a.active[b] == 0) return;
  const int nq = a.n_q[b];
  if (q0 >= nq) return;
  const int cnt = (nq - q0 < TQP) ? nq - q0 : TQP;

  const int L = a.dec.L, m = a.dec.m, mb_zx = a.dec.mb_zx;
  const size_t qbase = (size_t)b * a.n_stride + q0;
  const int qa = lane & 31;
  const int hi = lane >> 5;
  const f32x4* pts4 = reinterpret_cast<const f32x4*>(a.pts);
  f16x4* xp4 = reinterpret_cast<f16x4*>(xp);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  // stage 0 input: rows 0..2 = xyz, rows 3..15 = 0 (groups 0 and 1); queries past the end of the instance read as 0
  if (tid < TQP) {
    const f32x4 p = tid < cnt ? pts4[qbase + tid] : zero4;
    const float v0[4] = {p[0], p[1], p[2], 0.f};
    const float vz[4] = {0.f, 0.f, 0.f, 0.f};
    store4(xp4, (0 * TQP + tid) * 2 + 0, v0);
    store4(xp4, (0 * TQP + tid) * 2 + 1, vz);
    store4(xp4, (1 * TQP + tid) * 2 + 0, vz);
    store4(xp4, (1 * TQP + tid) * 2 + 1, vz);
  }

  Mask mk0 = {}, mk1 = {}, mk2 = {}, mk3 = {}, mk4 = {}, mk5 = {}, mk6 = {}, mk7 = {};
  f32x16 acc[NRB][NQB];
  float gx[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};     // d sdf / d xyz of queries (lane, lane + 64), this wave's share
  float y_keep = 0.f;
  float xmax = 0.f;                                         // fp16 range guard, see hm_decoder_h.hip
  const float* cbias0 = a.c0 + (size_t)b * HID;
  const float* cbias4 = a.c4 + (size_t)b * HID;
  bl[8 * HID + tid] = a.dec.w8[tid];
  for (int i = tid; i < 8 * HID; i += 512) {
    const StageDesc& sb = a.dec.st[i >> 9];
    const float* src = sb.inst_bias == 1 ? cbias0 : (sb.inst_bias == 2 ? cbias4 : sb.bias);
    bl[i] = src[i & (HID - 1)];
  }

  constexpr int n_stage = MODE == 0 ? 8 : NSTAGE;
  for (int s = 0; s < n_stage; ++s) {
    const StageDesc& sd = a.dec.st[s];
    const StageDescH& sh = a.dec.sth[s];
    const int epi = sd.epi;
    bool u[NRB];
#pragma unroll
    for (int r = 0; r < NRB; ++r) u[r] = (w + 8 * r >= sd.mb_lo) && (w + 8 * r < sd.mb_hi);
    const float us = sh.unscale;
    __syncthreads();
#ifdef HM_K1P_TRACE
    const bool trc = g_k1p_trace != nullptr && blockIdx.x == 0 && tid == 0;
#endif
    HM_STAMP(5 * s + 0);

    if (MODE == 1 && (epi == EPI_BWD4 || epi == EPI_BWD0)) {
      // xyz columns of lin4 / lin0 (512 x 4) through LDS scratch, then this wave's 64 rows against both query halves
      reinterpret_cast<f32x4*>(sc)[tid] = reinterpret_cast<const f32x4*>(epi == EPI_BWD4 ? a.dec.w4x : a.dec.w0x)[tid];
      __syncthreads();
      const f32x4* wx = reinterpret_cast<const f32x4*>(sc);
#pragma unroll 2
      for (int g = 0; g < 8; ++g) {
        float xa[8], xb[8];
        load_group_p(xp, 8 * w + g, lane, xa);
        load_group_p(xp, 8 * w + g, lane + 64, xb);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const f32x4 wv = wx[64 * w + 8 * g + j];
          gx[0][0] = fmaf(xa[j], wv[0], gx[0][0]); gx[0][1] = fmaf(xa[j], wv[1], gx[0][1]); gx[0][2] = fmaf(xa[j], wv[2], gx[0][2]);
          gx[1][0] = fmaf(xb[j], wv[0], gx[1][0]); gx[1][1] = fmaf(xb[j], wv[1], gx[1][1]); gx[1][2] = fmaf(xb[j], wv[2], gx[1][2]);
        }
      }
    }

#pragma unroll
    for (int r = 0; r < NRB; ++r)
#pragma unroll
      for (int nb = 0; nb < NQB; ++nb) acc[r][nb] = zero16p();
    if (MODE == 1 && epi == EPI_BWD0) {
      // d sdf/d z so far (lin4's transpose) was parked in this tile's J rows (true units): seed the accumulators
      const float rs = 1.f / us;
#pragma unroll
      for (int r = 0; r < NRB; ++r) {
        if (!u[r]) continue;
        const int jz = (w + 8 * r - mb_zx) * 32;
#pragma unroll
        for (int nb = 0; nb < NQB; ++nb) {
          const int q = nb * 32 + qa;
          if (q < cnt) {
            const float* row = a.J + (qbase + q) * (size_t)a.ldJ;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(row + jz + 8 * g + 4 * hi);
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[r][nb][4 * g + j] = v[j] * rs;
            }
          }
        }
      }
    }

    HM_STAMP(5 * s + 1);
    {
      const f16x8* wp = reinterpret_cast<const f16x8*>(sh.wp);
      const f16x8* wp0 = wp + (size_t)(w - sd.mb_lo) * sh.mb_stride + lane;
      const f16x8* wp1 = wp + (size_t)(w + 8 - sd.mb_lo) * sh.mb_stride + lane;
      if (u[0] && u[1]) gemm_loop_p<true, true, MODE == 0>(acc, wp0, wp1, sh.n_k16, xp);
      else if (u[0]) gemm_loop_p<true, false, MODE == 0>(acc, wp0, wp1, sh.n_k16, xp);
      else if (u[1]) gemm_loop_p<false, true, MODE == 0>(acc, wp0, wp1, sh.n_k16, xp);
    }
    HM_STAMP(5 * s + 2);
    __syncthreads();
    HM_STAMP(5 * s + 3);

    if (MODE == 0 || epi <= EPI_FWD7) {
      const float* bias = bl + s * HID;
      Mask mk = {};
      // lin3's rows m..m+2 (always rows 29..31 of their block: m = 509 - L, L % 32 == 0) carry xyz into the skip layer
      const int mbx = m >> 5;
#pragma unroll
      for (int r = 0; r < NRB; ++r) {
        if (!u[r]) continue;
        const int mb = w + 8 * r;
        f32x4 bv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const f32x4*>(bias + mb * 32 + 8 * g + 4 * hi);
#pragma unroll
        for (int nb = 0; nb < NQB; ++nb) {
          uint32_t bits = 0;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int f4 = mb * 32 + 8 * g + 4 * hi;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float val = fmaf(acc[r][nb][4 * g + j], us, bv[g][j]);
              const bool pos = val > 0.f;
              bits |= (pos ? 1u : 0u) << (4 * g + j);
              v[j] = pos ? val : 0.f;
            }
            xmax = fmaxf(xmax, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
            if (g == 3 && epi == EPI_FWD3 && mb == mbx && hi == 1) {
              const int q = nb * 32 + qa;
              const f32x4 p = q < cnt ? pts4[qbase + q] : zero4;
              v[1] = p[0]; v[2] = p[1]; v[3] = p[2];
            }
            store4(xp4, ((f4 >> 3) * TQP + nb * 32 + qa) * 2 + hi, v);
          }
          mk.w[2 * r + (nb >> 1)] |= bits << ((nb & 1) * 16);
        }
      }
#define HM_SET(M) M = mk
      if (MODE == 1) switch (sd.layer) { HM_MASK_CASES(HM_SET) }
#undef HM_SET

      if (epi == EPI_FWD7) {
        __syncthreads();
        float pa = 0.f, pb = 0.f;
#pragma unroll 2
        for (int g = 0; g < 8; ++g) {
          float xa[8], xb[8];
          load_group_p(xp, 8 * w + g, lane, xa);
          load_group_p(xp, 8 * w + g, lane + 64, xb);
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(bl + 8 * HID + 64 * w + 8 * g);
          const f32x4 w1 = *reinterpret_cast<const f32x4*>(bl + 8 * HID + 64 * w + 8 * g + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) { pa = fmaf(xa[j], w0[j], pa); pb = fmaf(xb[j], w0[j], pb); }
#pragma unroll
          for (int j = 0; j < 4; ++j) { pa = fmaf(xa[4 + j], w1[j], pa); pb = fmaf(xb[4 + j], w1[j], pb); }
        }
        if (__any(!(xmax < 65504.f))) { pa = __builtin_nanf(""); pb = pa; }
        sc[w * TQP + lane] = pa;
        sc[w * TQP + 64 + lane] = pb;
        __syncthreads();
        if (tid < TQP) {
          float a8 = 0.f;
#pragma unroll
          for (int i = 0; i < NWP; ++i) a8 += sc[i * TQP + tid];
          a8 += a.dec.b8;
          const float yv = tanhf(a8);
          y_keep = yv;
          if (tid < cnt) a.y[qbase + tid] = yv;
          sc[NWP * TQP + tid] = 1.f - yv * yv;
        }
        if (MODE == 0) return;
        __syncthreads();
        float dy[NQB];
#pragma unroll
        for (int nb = 0; nb < NQB; ++nb) dy[nb] = sc[NWP * TQP + nb * 32 + qa];
#pragma unroll
        for (int r = 0; r < NRB; ++r) {
          const int mb = w + 8 * r;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int f4 = mb * 32 + 8 * g + 4 * hi;
            const f32x4 wv = *reinterpret_cast<const f32x4*>(bl + 8 * HID + f4);
#pragma unroll
            for (int nb = 0; nb < NQB; ++nb) {
              const uint32_t bits = mk.w[2 * r + (nb >> 1)] >> ((nb & 1) * 16);
              float v[4];
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = ((bits >> (4 * g + j)) & 1u) ? dy[nb] * wv[j] : 0.f;
              store4(xp4, ((f4 >> 3) * TQP + nb * 32 + qa) * 2 + hi, v);
            }
          }
        }
      }
    } else if (MODE == 1 && (epi == EPI_BWD || epi == EPI_BWD4)) {
      Mask mk;
#define HM_GET(M) mk = M
      switch (sd.layer) { HM_MASK_CASES(HM_GET) }
#undef HM_GET
#pragma unroll
      for (int r = 0; r < NRB; ++r) {
        if (!u[r]) continue;
        const int mb = w + 8 * r;
        if (epi == EPI_BWD4 && mb >= mb_zx) {      // latent rows: park in J (same thread re-reads them in BWD0)
          const int jz = (mb - mb_zx) * 32;
#pragma unroll
          for (int nb = 0; nb < NQB; ++nb) {
            const int q = nb * 32 + qa;
            if (q < cnt) {
              float* row = a.J + (qbase + q) * (size_t)a.ldJ;
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[r][nb][4 * g + j] * us;
                *reinterpret_cast<f32x4*>(row + jz + 8 * g + 4 * hi) = v;
              }
            }
          }
          continue;
        }
#pragma unroll
        for (int nb = 0; nb < NQB; ++nb) {
          const uint32_t bits = mk.w[2 * r + (nb >> 1)] >> ((nb & 1) * 16);
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int f4 = mb * 32 + 8 * g + 4 * hi;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ((bits >> (4 * g + j)) & 1u) ? acc[r][nb][4 * g + j] * us : 0.f;
            xmax = fmaxf(xmax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
            store4(xp4, ((f4 >> 3) * TQP + nb * 32 + qa) * 2 + hi, v);
          }
        }
      }
    } else if (MODE == 1) {  // EPI_BWD0
#pragma unroll
      for (int r = 0; r < NRB; ++r) {
        if (!u[r]) continue;
        const int jz = (w + 8 * r - mb_zx) * 32;
#pragma unroll
        for (int nb = 0; nb < NQB; ++nb) {
          const int q = nb * 32 + qa;
          if (q < cnt) {
            float* row = a.J + (qbase + q) * (size_t)a.ldJ;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              f32x4 v;
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = acc[r][nb][4 * g + j] * us;
              *reinterpret_cast<f32x4*>(row + jz + 8 * g + 4 * hi) = v;
            }
          }
        }
      }
    }
    HM_STAMP(5 * s + 4);
  }

  if (MODE == 0) return;
  if (__any(!(xmax < 65504.f))) { gx[0][0] = __builtin_nanf(""); gx[1][0] = gx[0][0]; }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    sc[(w * 3 + c) * TQP + lane] = gx[0][c];
    sc[(w * 3 + c) * TQP + 64 + lane] = gx[1][c];
  }
  __syncthreads();
  if (tid < cnt) {
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
      g0 += sc[(i * 3 + 0) * TQP + tid];
      g1 += sc[(i * 3 + 1) * TQP + tid];
      g2 += sc[(i * 3 + 2) * TQP + tid];
    }
    const f32x4 p = pts4[qbase + tid];
    float* row = a.J + (qbase + tid) * (size_t)a.ldJ + L;
    row[7] = y_keep;
    row[0] = g0; row[1] = g1; row[2] = g2;
    if (a.pose_dim != 0) {
      row[3] = g2 * p[1] - g1 * p[2];
      row[4] = g0 * p[2] - g2 * p[0];
      row[5] = g1 * p[0] - g0 * p[1];
      if (a.pose_dim == 7) row[6] = g0 * p[0] + g1 * p[1] + g2 * p[2];
    }
  }
}

}  // namespace

extern "C" void hm_debug_set_k1p_trace(long long* d_buf) {
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_k1p_trace), &d_buf, sizeof(d_buf));
}

namespace hm {

int launch_decoder_p(const hm_decoder_s* dec, int B, const float* d_pts, const int* d_nq, const int* d_active,
                     int n_stride, const float* d_c0, const float* d_c4, float* d_y, float* d_J, int ldJ,
                     int pose_dim, int mode, hipStream_t stream, int tag) {
  DecodeArgsP a;
  a.dec = dec->dev;
  a.pts = d_pts; a.n_q = d_nq; a.active = d_active; a.c0 = d_c0; a.c4 = d_c4;
  a.y = d_y; a.J = d_J; a.n_stride = n_stride; a.B = B; a.ldJ = ldJ; a.pose_dim = pose_dim;
  const int grid = B * ((n_stride + TQP - 1) / TQP);
  if (grid == 0) return 0;
  if (mode == 0) hipLaunchKernelGGL((k_decoder_p<0, 0>), dim3(grid), dim3(512), 0, stream, a);
  else if (tag == 0) hipLaunchKernelGGL((k_decoder_p<1, 0>), dim3(grid), dim3(512), 0, stream, a);
  else hipLaunchKernelGGL((k_decoder_p<1, 1>), dim3(grid), dim3(512), 0, stream, a);
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hm
