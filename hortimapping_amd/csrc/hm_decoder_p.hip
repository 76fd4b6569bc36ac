// K1p: the fused decoder forward + input-gradient backward in PLAIN fp16 MFMA arithmetic (precision 3, "f16"), and the
// screening pass of the f16x3 render chain (hm_optimize.hip: screen_pass).
//
// This is the arithmetic BASELINE.json names for configs[4] ("fp16 MFMA decoder"): every GEMM of the network is ONE
// v_mfma_f32_32x32x16_f16 pass on fp16-rounded weights (per-stage power-of-two scale, hm_pack.hip) and fp16-rounded
// activations, fp32 accumulation.  Results are fp16-class (~1e-3 relative), NOT the reference's fp32 -- never the
// default; the parity tables in profiles/ say what that costs at the metric level.
//
// Tiling: 128 queries per workgroup (one fp16 activation plane = 128 KiB of LDS), 512 threads; wave w owns the 32-row
// blocks {w, w + 8} x four 32-query blocks (8 accumulators of 32x32 = 128 registers).  Round 6 rewrite:
//  * WEIGHT STREAMS.  Each wave reads one contiguous stream (hm_pack.hip: pack_stream_p) in the order it consumes it, so
//    the ring of FOUR A-operand sets (fetched three K-steps ahead, L2 -> VGPR, never through LDS: no sharing inside the
//    workgroup) runs ACROSS stage boundaries: the first three steps of the next stage are in flight while the epilogue
//    runs, no stage starts with an exposed L2 round trip.
//  * K PERMUTATION.  Inside a K-step lane half h holds k = 16 t + 4 h + {0..3} and 16 t + 8 + 4 h + {0..3}: exactly the
//    rows one lane owns in the 32x32 accumulator layout, so the epilogue writes whole 16-byte activation units
//    X[unit 2 t + h][query][8] with one conflict-free ds_write_b128 per 8 values (the round-5 layout wrote 8-byte halves,
//    2-way bank conflicts on every store), and that unit is the next stage's B operand as it stands.
//  * CONVERSION BEFORE THE BARRIER.  bias / ReLU / fp16 conversion / mask building run on registers before the barrier
//    that ends the stage's LDS reads: the four waves that finish their K loop first (the older wave of each SIMD owns the
//    matrix pipe) convert in the shadow of their partners' MFMAs; after the barrier only the stores are left.
//  * ReLU masks are the SIGN BITS of the fp16 pre-activations, shifted into 128 bits per lane and layer and applied to
//    the packed fp16 gradients (ReLU'(+0) = 1 instead of 0: a unit at exactly zero has a zero incoming row or measure zero).
//  * d sdf / d xyz through lin4 comes out of the matrix pipe (rows m..m+2 of the transposed stage carry lin4's xyz columns).
// Stage list, latent folding, lin4^T latent partial parked in the J rows and the fp16 range guard are those of
// hm_decoder_h.hip (reference: deepsdf/networks/deep_sdf_decoder.py:75-110, wild_completion/utils.py:112-193,
// loss.py:229-241).
#include <stdlib.h>

#include "hm_common.h"
#include "hm_internal.h"
#include "hm_gemm_p.h"

using namespace hm;
using namespace hm_p;

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

struct DecodeArgsP {
  DecoderDev dec;
  const float* pts;
  const int* n_q;
  const int* active;
  const float* c0;
  const float* c4;
  float* y;
  float* J;
  u32x4* mscr;         // forward + backward kernel: ReLU-mask scratch [workgroup][8 layers][512 threads] x 16 bytes
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

// X unit u = 2 t + h of query q holds rows 16 t + 4 h + {0..3} (elements 0..3) and 16 t + 8 + 4 h + {0..3} (elements 4..7)
__device__ __forceinline__ int unit_row0(int u) { return 16 * (u >> 1) + 4 * (u & 1); }

__device__ __forceinline__ void load_unit_p(const f16x8* xp, int u, int q, float (&x)[8]) {
  const f16x8 h = xp[u * TQP + q];
#pragma unroll
  for (int j = 0; j < 8; ++j) x[j] = (float)h[j];
}

// Zero the fp16 halves of d whose ReLU-mask bit says "was negative".  word: the mask word of the block, s: position of
// dword i of unit p of query block nb in it (see Mask): the sign of the low / high half sits at bit s / 16 + s.
__device__ __forceinline__ uint32_t apply_mask_p(uint32_t d, uint32_t word, int s) {
  const uint32_t t = word << (15 - s);
  const s16x2 neg = __builtin_bit_cast(s16x2, t) >> 15;       // 0xffff per half where masked
  return d & ~__builtin_bit_cast(uint32_t, neg);
}

// The lane id, derived afresh (volatile: never hoisted, never merged with an earlier copy).  Everything the epilogues
// address with -- query column, lane half, LDS / J-row / scratch offsets -- is recomputed from it where it is used, a few
// VALU instructions per stage; derived once at the top of the kernel, hipcc keeps two dozen such loop-invariant values in
// registers across all 16 stages (and spills them in the forward + backward kernel).
__device__ __forceinline__ int fresh_lane_p() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

__device__ __forceinline__ bool guard_tripped_p(const f16x8 gmax, const f16x8 gmin) {
  bool bad = false;
#pragma unroll
  for (int e = 0; e < 8; ++e) bad = bad || !((float)gmax[e] < 65504.f) || !((float)gmin[e] > -65504.f);
  return bad;
}

// The converted units of a block wait for the barrier INSIDE the block's accumulator registers (unit p in elements
// 4 p .. 4 p + 3, as bits): no register beyond the accumulators is held across the barrier.  Unit p is formed from
// elements 8 p .. 8 p + 7, so p = 0 is converted first.
__device__ __forceinline__ void stash_unit_p(f32x16& acc, int p, const f16x8 h) {
  const u32x4 d = __builtin_bit_cast(u32x4, h);
#pragma unroll
  for (int i = 0; i < 4; ++i) { const uint32_t t = d[i]; acc[4 * p + i] = __uint_as_float(t); }   // (__builtin_bit_cast applied
                                               // to a vector ELEMENT reads element 0 whatever the index: hipcc 7.2; go through a scalar)
}
__device__ __forceinline__ f16x8 stashed_unit_p(const f32x16& acc, int p) {
  u32x4 d;
#pragma unroll
  for (int i = 0; i < 4; ++i) { const float f = acc[4 * p + i]; d[i] = __float_as_uint(f); }
  return __builtin_bit_cast(f16x8, d);
}

// ReLU masks: 128 bits per lane and layer.  Word 2 r + (nb >> 1); with D_0..D_7 the eight dwords (fp16 pairs) of the
// block (r, nb) in store order (D = 4 p + dword of unit p): the sign of D_i's low / high half sits at bit s / 16 + s,
// s = i + 8 (nb & 1).  Built by the chain m = (m >> 1) | (D & 0x80008000) over nb even (i = 0..7) then nb odd (i = 0..7).
// A layer's four words are written to the launch's scratch block by the forward stage and read back by the backward stage
// that needs them (64 KiB per tile through L2, 16 bytes per thread and stage, issued ahead of the K loop): 4 registers
// instead of 32 held across the whole tile, which is what lets the forward + backward kernel run without spills.
struct Mask { uint32_t w[4]; };

// Shader-clock stamps of the 8 waves of workgroup 0 (80 slots per wave; hm_debug_set_k1p_trace, scripts/gpu_trace_k1p.py),
// per stage s: [5 s + 0] stage entered (after the X barrier), +1 K loop starts, +2 K loop done, +3 conversion done and
// barrier passed, +4 stores done.  Compiled in only with -DHM_K1P_TRACE.
__device__ long long* g_k1p_trace = nullptr;
#ifdef HM_K1P_TRACE
#define HM_STAMP(SLOT) if (trc) g_k1p_trace[80 * w + (SLOT)] = clock64()
#else
#define HM_STAMP(SLOT)
#endif

// -DHM_K1P_DEBUG: hm_debug_set_k1p_probe(stage, row) makes the kernel return X[row][.] as stored by that stage instead of
// the sdf (scripts/gpu_probe_k1p.py compares it with a numpy forward pass layer by layer).  Development builds only.
#ifdef HM_K1P_DEBUG
__device__ int g_k1p_probe[2] = {-1, 0};
#endif


#ifndef HM_P_AHEAD_FWD
#define HM_P_AHEAD_FWD 3         // forward-only kernel: three K-steps ahead (same-box A/B: 1 / 2 / 3 within 1 %)
#endif
#ifndef HM_P_ONE_LOOP
#define HM_P_ONE_LOOP 0
#endif
#ifndef HM_P_AHEAD_BWD
#define HM_P_AHEAD_BWD 3
#endif

// MODE 0: forward (sdf only).  MODE 2: forward that also writes the ReLU masks to the launch's scratch block.  MODE 3:
// backward only -- seed from the sdf and lin7's masks, the eight transposed stages, Jacobian rows.  Forward + backward
// = a MODE 2 launch followed by a MODE 3 launch on the same stream: one fused kernel held the masks, the xyz-gradient
// shares and the forward biases next to 128 accumulator registers and spilled 80 ... 400 registers whatever was tried
// (round 6; 1.35 ms for 131,072 queries against 0.40 + 0.5 for the pair).
template <int MODE, int TAG>
__global__ __launch_bounds__(512, 2) void k_decoder_p(const DecodeArgsP a) {
  constexpr bool FWD = MODE != 3, BWD = MODE == 3, MASKS = MODE == 2;
  constexpr int AH = FWD ? HM_P_AHEAD_FWD : HM_P_AHEAD_BWD;
  __shared__ f16x8 xp[64 * TQP];   // 128 KiB: X[unit][q][8], fp16
  __shared__ float sc[3072];       // 12 KiB scratch: the tile's xyz (forward) / xyz weight columns / lin8 partials + dy / final xyz-gradient sums
  __shared__ int prog[NWP];        // pacing of the wave pairs of a SIMD (hm_gemm_p.h: PaceP)
  __shared__ float bl[9 * HID];    // 18 KiB: biases of the 8 forward stages (per-instance c0 / c4 included) + lin8's row;
                                   // after the forward stages bl[0 .. 3 * TQP) takes d sdf / d xyz through lin4

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
  const int xo = hi * TQP + qa;
  const f32x4* pts4 = reinterpret_cast<const f32x4*>(a.pts);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  // the wave's weight stream; the ring is primed with the first three K-steps before anything else
  WStreamP ws;
  {
    const int stream_bytes = a.dec.ps_steps * 2048;
    char* base = const_cast<char*>(static_cast<const char*>(a.dec.ps)) + (size_t)w * stream_bytes;
    ws.rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, stream_bytes, 0x00020000);
    ws.v0 = lane * 16; ws.v1 = lane * 16 + 1024;
  }
  int sq = 0;                      // byte position of the current stage's step 0 in the stream
  if (BWD) {
#pragma unroll
    for (int s = 0; s < 8; ++s) sq += a.dec.pgrp[s] * 8192;
  }
  ASetP a0, a1, a2, a3;
  a0.r[0] = wload_p(ws, 0, sq);    a0.r[1] = wload_p(ws, 1, sq);
  a1.r[0] = zero8; a1.r[1] = zero8; a2.r[0] = zero8; a2.r[1] = zero8; a3.r[0] = zero8; a3.r[1] = zero8;
  if (AH >= 2) { a1.r[0] = wload_p(ws, 0, sq + 2048); a1.r[1] = wload_p(ws, 1, sq + 2048); }
  if (AH >= 3) { a2.r[0] = wload_p(ws, 0, sq + 4096); a2.r[1] = wload_p(ws, 1, sq + 4096); }

  // stage 0 input: rows 0..2 = xyz (unit 0, elements 0..2), every other row of the padded K-steps 0..3 (units 0..7) = 0;
  // queries past the end of the instance read as 0.  The tile's xyz also waits in sc for lin3's splice.
  if (FWD && tid < TQP) {
    const f32x4 p = tid < cnt ? pts4[qbase + tid] : zero4;
    f16x8 u0 = zero8;
    u0[0] = (_Float16)p[0]; u0[1] = (_Float16)p[1]; u0[2] = (_Float16)p[2];
    xp[tid] = u0;
#pragma unroll
    for (int i = 1; i < 8; ++i) xp[i * TQP + tid] = zero8;
    sc[0 * TQP + tid] = p[0]; sc[1 * TQP + tid] = p[1]; sc[2 * TQP + tid] = p[2];
  }

  if (tid < NWP) prog[tid] = 0;
  PaceP pc;
  pc.prog = (lds_int_p*)prog; pc.w = w; pc.done = 0;
  Mask mk = {};                    // masks of the stage's layer: built (forward) or fetched ahead of the K loop (backward)
  f32x16 acc[NRB][NQB];
  // fp16 range guard (see hm_decoder_h.hip), kept on the PACKED fp16 values that go to X: largest stored activation /
  // gradient and smallest gradient; a tile in which one of them reaches fp16's largest finite value (or inf) is poisoned
  f16x8 gmax = zero8, gmin = zero8;
  bl[8 * HID + tid] = a.dec.w8[tid];
  if (FWD) {
    const float* cbias0 = a.c0 + (size_t)b * HID;
    const float* cbias4 = a.c4 + (size_t)b * HID;
    for (int i = tid; i < 8 * HID; i += 512) {
      const StageDesc& sb = a.dec.st[i >> 9];
      const float* src = sb.inst_bias == 1 ? cbias0 : (sb.inst_bias == 2 ? cbias4 : sb.bias);
      bl[i] = src[i & (HID - 1)];
    }
  }
  const int mbx = m >> 5;          // lin3's rows m..m+2 (always rows 29..31 of their block: m = 509 - L, L % 32 == 0) carry xyz
  if (BWD) {
    reinterpret_cast<f32x4*>(sc + 1024)[tid] = reinterpret_cast<const f32x4*>(a.dec.w0x)[tid];     // lin0's xyz columns, for the last stage
    // backward seed: G7 = mask7 . (dy w8) with dy = 1 - sdf^2 (tanh'), from the forward launch's sdf and lin7's masks
    if (tid < TQP) { const float yv = tid < cnt ? a.y[qbase + tid] : 0.f; sc[tid] = 1.f - yv * yv; }
    const u32x4 mv = a.mscr[((size_t)blockIdx.x * 8 + 7) * 512 + tid];
    __syncthreads();
    float dy[NQB];
#pragma unroll
    for (int nb = 0; nb < NQB; ++nb) dy[nb] = sc[nb * 32 + qa];
#pragma unroll
    for (int r = 0; r < NRB; ++r) {
      const int mb = w + 8 * r;
#pragma unroll
      for (int nb = 0; nb < NQB; ++nb) {
        const uint32_t word = mv[2 * r + (nb >> 1)];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          f16x8 h;
#pragma unroll
          for (int gg = 0; gg < 2; ++gg) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(bl + 8 * HID + mb * 32 + 8 * (2 * p + gg) + 4 * hi);
#pragma unroll
            for (int j = 0; j < 4; ++j) h[4 * gg + j] = (_Float16)(dy[nb] * wv[j]);
          }
          u32x4 d = __builtin_bit_cast(u32x4, h);
#pragma unroll
          for (int i = 0; i < 4; ++i) d[i] = apply_mask_p(d[i], word, 4 * p + i + 8 * (nb & 1));
          xp[(2 * (2 * mb + p) + hi) * TQP + nb * 32 + qa] = __builtin_bit_cast(f16x8, d);
        }
      }
    }
  }

  for (int s = FWD ? 0 : 8; s < (FWD ? 8 : NSTAGE); ++s) {
    const StageDesc& sd = a.dec.st[s];
    const int epi = sd.epi;
    const int n_grp = a.dec.pgrp[s];
    // the wave's two 32-row blocks of this stage: w and w + 8, swapped in stages whose valid blocks all lie in the upper
    // half (lin0's transpose) so that a wave with ONE valid block always has it in slot 0: two K-loop variants, not three
    const int sw8 = a.dec.pswap[s] * 8;
    const int mbr[NRB] = {w + sw8, w + 8 - sw8};
    bool u[NRB];
#pragma unroll
    for (int r = 0; r < NRB; ++r) u[r] = (mbr[r] >= sd.mb_lo) && (mbr[r] < sd.mb_hi);
    const float us = a.dec.pus[s];
    __syncthreads();
#ifdef HM_K1P_TRACE
    const bool trc = g_k1p_trace != nullptr && blockIdx.x == 0 && lane == 0;
#endif
    HM_STAMP(5 * s + 0);
#ifdef HM_K1P_TRACE
    // the constant 100 MHz counter beside the shader clock, at the first and the last stage of the launch: shader clocks per
    // microsecond = the clock the chip actually sustains under this kernel (DVFS; slots 704 .. 707 of the trace buffer)
    if (trc && w == 0 && s == (FWD ? 0 : 8)) { g_k1p_trace[704] = wall_clock64(); g_k1p_trace[705] = clock64(); }
    if (trc && w == 0 && s == (FWD ? 7 : NSTAGE - 1)) { g_k1p_trace[706] = wall_clock64(); g_k1p_trace[707] = clock64(); }
#endif

    if (BWD && (epi == EPI_BWD || epi == EPI_BWD4)) {      // the layer's masks, on their way while the K loop runs
      const u32x4 mv = a.mscr[((size_t)blockIdx.x * 8 + sd.layer) * 512 + 64 * w + fresh_lane_p()];
      mk.w[0] = mv[0]; mk.w[1] = mv[1]; mk.w[2] = mv[2]; mk.w[3] = mv[3];
    }

#pragma unroll
    for (int r = 0; r < NRB; ++r)
#pragma unroll
      for (int nb = 0; nb < NQB; ++nb) acc[r][nb] = zero16p();
    HM_STAMP(5 * s + 1);
#ifdef HM_K1P_TRACE
    long long* gst = (trc && s == 1) ? g_k1p_trace + 640 + 8 * w : nullptr;    // K-loop group stamps of stage 1
#else
    long long* gst = nullptr;
#endif
    // A wave whose second block is not valid in this stage (lin3's 16 - L/32 blocks, lin0's transpose) runs the 4-MFMA form
    // of the loop (HM_P_ONE_LOOP = 1: it runs the full loop on the zero weights its stream holds there instead -- one loop
    // instantiation, for builds in which the second one costs registers: +9 k clocks per such stage at L = 256)
    if (u[0] && (u[1] || HM_P_ONE_LOOP)) {
      k_loop_p<true, true, AH>(acc, a0, a1, a2, a3, ws, sq, n_grp, xp, xo, gst, &pc);
    } else if (u[0]) {
      k_loop_p<true, false, AH>(acc, a0, a1, a2, a3, ws, sq, n_grp, xp, xo, nullptr, &pc);
      // the second block's accumulators are not read in this stage: tell the register allocator (no instruction), or 64
      // registers of zeros ride through this loop
#pragma unroll
      for (int nb = 0; nb < NQB; ++nb) asm volatile("" : "=v"(acc[1][nb]));
    } else {   // no valid row block in this stage: keep the ring in step (the next stage's first three steps)
      const int sn = sq + n_grp * 8192;
      a0.r[0] = wload_p(ws, 0, sn);        a0.r[1] = wload_p(ws, 1, sn);
      if (AH >= 2) { a1.r[0] = wload_p(ws, 0, sn + 2048); a1.r[1] = wload_p(ws, 1, sn + 2048); }
      if (AH >= 3) { a2.r[0] = wload_p(ws, 0, sn + 4096); a2.r[1] = wload_p(ws, 1, sn + 4096); }
    }
    sq += n_grp * 8192;
    pc.done += n_grp;
    // the sets beyond the first AH are refilled by the next stage's first steps before anyone reads them: end
    // their live ranges here (no instruction)
    asm volatile("" : "=v"(a3.r[0]), "=v"(a3.r[1]));
    if (AH < 3) asm volatile("" : "=v"(a2.r[0]), "=v"(a2.r[1]));
    if (AH < 2) asm volatile("" : "=v"(a1.r[0]), "=v"(a1.r[1]));
    if (BWD && epi == EPI_BWD0) {
      const int lane = fresh_lane_p();
      // d sdf / d xyz through lin0: the xyz columns of lin0 (staged in sc by the prologue) against X = the gradient at lin0's
      // output, which this stage does not overwrite -- this wave's 64 rows (units 8 w .. 8 w + 7), both query halves.  AFTER the
      // K loop: the wave that finishes first does it in the shadow of its SIMD partner's MFMAs
      const f32x4* wx = reinterpret_cast<const f32x4*>(sc + 1024);
      float gx[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};     // d sdf / d xyz through lin0 of queries (lane, lane + 64), this wave's share
#pragma unroll 2
      for (int g = 0; g < 8; ++g) {
        float xa[8], xb[8];
        load_unit_p(xp, 8 * w + g, lane, xa);
        load_unit_p(xp, 8 * w + g, lane + 64, xb);
        const int r0 = unit_row0(8 * w + g);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const f32x4 wv = wx[r0 + (j < 4 ? j : j + 4)];
          gx[0][0] = fmaf(xa[j], wv[0], gx[0][0]); gx[0][1] = fmaf(xa[j], wv[1], gx[0][1]); gx[0][2] = fmaf(xa[j], wv[2], gx[0][2]);
          gx[1][0] = fmaf(xb[j], wv[0], gx[1][0]); gx[1][1] = fmaf(xb[j], wv[1], gx[1][1]); gx[1][2] = fmaf(xb[j], wv[2], gx[1][2]);
        }
      }
      // this wave's share waits in bl[512 ..) (the biases are not needed any more) for the sum at the end of the tile
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        bl[512 + (w * 3 + c) * TQP + lane] = gx[0][c];
        bl[512 + (w * 3 + c) * TQP + 64 + lane] = gx[1][c];
      }
    }
    HM_STAMP(5 * s + 2);

    // ---- conversion on registers (no access to X): unit p of block (r, nb) = the 16 bytes X[2 (2 mb + p) + hi][nb * 32 + qa] ----
    const int lane = fresh_lane_p(), tid = 64 * w + lane, qa = lane & 31, hi = lane >> 5;
    bool st_x[NRB] = {false, false};
#ifdef ABL_NOEPI       // timing ablation (wrong results): K loops and barriers only
#pragma unroll
    for (int r = 0; r < NRB; ++r)
#pragma unroll
      for (int nb = 0; nb < NQB; ++nb) asm volatile("" :: "v"(acc[r][nb]));
    if (false) {
#else
    if (FWD) {
#endif
      const float* bias = bl + s * HID;
      mk = Mask{};
      const f32x8 us8 = {us, us, us, us, us, us, us, us};
#pragma unroll
      for (int r = 0; r < NRB; ++r) {
        if (!u[r]) continue;
        st_x[r] = true;
        const int mb = mbr[r];
        f32x4 bv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const f32x4*>(bias + mb * 32 + 8 * g + 4 * hi);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          uint32_t chain = 0;
#pragma unroll
          for (int nbi = 0; nbi < 2; ++nbi) {
            const int nb = 2 * half + nbi;
            // per unit p (accumulator elements 8 p .. 8 p + 7): four packed fmas (scale, bias), four packed conversions, [mask
            // chain], four packed ReLUs, four packed range-guard maxima (four independent chains); the unit lands in
            // elements 4 p .. 4 p + 3
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              f32x8 av, bb;
#pragma unroll
              for (int e = 0; e < 8; ++e) { av[e] = acc[r][nb][8 * p + e]; bb[e] = bv[2 * p + (e >> 2)][e & 3]; }
              f16x8 h = __builtin_convertvector(__builtin_elementwise_fma(av, us8, bb), f16x8);
              if (MASKS) {
                const u32x4 d = __builtin_bit_cast(u32x4, h);
#pragma unroll
                for (int i = 0; i < 4; ++i) chain = (chain >> 1) | (d[i] & 0x80008000u);
              }
              h = __builtin_elementwise_max(h, zero8);
              gmax = __builtin_elementwise_max(gmax, h);
              stash_unit_p(acc[r][nb], p, h);
            }
            HM_FENCE();       // keep the program order
          }
          mk.w[2 * r + half] = chain;
        }
        if (epi == EPI_FWD3 && mb == mbx && hi == 1) {
          // lin3's rows m..m+2 (rows 29..31 of this block: elements 5..7 of unit 1 of the upper lane half) carry xyz into the skip layer
#pragma unroll
          for (int nb = 0; nb < NQB; ++nb) {
            const int q = nb * 32 + qa;
            const f16x2 x01 = {(_Float16)0, (_Float16)sc[0 * TQP + q]};
            const f16x2 x23 = {(_Float16)sc[1 * TQP + q], (_Float16)sc[2 * TQP + q]};
            const uint32_t d2 = (__float_as_uint(acc[r][nb][6]) & 0x0000ffffu) | (__builtin_bit_cast(uint32_t, x01) & 0xffff0000u);
            acc[r][nb][6] = __uint_as_float(d2);
            acc[r][nb][7] = __uint_as_float(__builtin_bit_cast(uint32_t, x23));
          }
        }
      }
      if (MASKS) {
        const u32x4 mv = {mk.w[0], mk.w[1], mk.w[2], mk.w[3]};
        a.mscr[((size_t)blockIdx.x * 8 + sd.layer) * 512 + tid] = mv;
      }
    } else if (epi == EPI_BWD || epi == EPI_BWD4) {
#pragma unroll
      for (int r = 0; r < NRB; ++r) {
        if (!u[r]) continue;
        const int mb = mbr[r];
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
        st_x[r] = true;
        const bool xyz_rows = epi == EPI_BWD4 && mb == mbx && hi == 1;
        const f32x8 us8 = {us, us, us, us, us, us, us, us};
#pragma unroll
        for (int nb = 0; nb < NQB; ++nb) {
          const uint32_t word = mk.w[2 * r + (nb >> 1)];
          if (xyz_rows) {
            // rows m..m+2 of lin4's transpose (elements 13..15): d sdf / d xyz through lin4 (no mask: inputs, not ReLU outputs)
            const int q = nb * 32 + qa;
            bl[0 * TQP + q] = acc[r][nb][13] * us; bl[1 * TQP + q] = acc[r][nb][14] * us; bl[2 * TQP + q] = acc[r][nb][15] * us;
          }
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            f32x8 av;
#pragma unroll
            for (int e = 0; e < 8; ++e) av[e] = acc[r][nb][8 * p + e];
            const f16x8 h = __builtin_convertvector(av * us8, f16x8);
            gmax = __builtin_elementwise_max(gmax, h);
            gmin = __builtin_elementwise_min(gmin, h);
            u32x4 d = __builtin_bit_cast(u32x4, h);
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = apply_mask_p(d[i], word, 4 * p + i + 8 * (nb & 1));
            stash_unit_p(acc[r][nb], p, __builtin_bit_cast(f16x8, d));
          }
          if (xyz_rows) {   // the three xyz rows are no activations of lin3: zero in X (lin3's transpose has zero columns there anyway)
            acc[r][nb][6] = __uint_as_float(__float_as_uint(acc[r][nb][6]) & 0x0000ffffu);
            acc[r][nb][7] = 0.f;
          }
          HM_FENCE();
        }
      }
    } else {  // EPI_BWD0
#pragma unroll
      for (int r = 0; r < NRB; ++r) {
        if (!u[r]) continue;
        const int jz = (mbr[r] - mb_zx) * 32;
#pragma unroll
        for (int nb = 0; nb < NQB; ++nb) {
          const int q = nb * 32 + qa;
          if (q < cnt) {
            float* row = a.J + (qbase + q) * (size_t)a.ldJ;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              // d sdf / d z = lin4's share (parked in this row by the same thread, stage 11) + lin0's
              f32x4 v = *reinterpret_cast<const f32x4*>(row + jz + 8 * g + 4 * hi);
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = fmaf(acc[r][nb][4 * g + j], us, v[j]);
              *reinterpret_cast<f32x4*>(row + jz + 8 * g + 4 * hi) = v;
            }
          }
        }
      }
    }

    __syncthreads();               // every wave is done reading X
    HM_STAMP(5 * s + 3);
#pragma unroll
    for (int r = 0; r < NRB; ++r) {
      if (!st_x[r]) continue;
      const int mb = mbr[r];
#pragma unroll
      for (int nb = 0; nb < NQB; ++nb)
#pragma unroll
        for (int p = 0; p < 2; ++p) xp[(2 * (2 * mb + p) + hi) * TQP + nb * 32 + qa] = stashed_unit_p(acc[r][nb], p);
    }
    HM_STAMP(5 * s + 4);
#ifdef HM_K1P_DEBUG
    if (g_k1p_probe[0] == s) {
      __syncthreads();
      const int row = g_k1p_probe[1];
      const int t16 = row >> 4, rr = row & 15;
      const int hh = (rr >> 2) & 1, e = (rr & 3) + 4 * (rr >> 3);
      if (tid < cnt) a.y[qbase + tid] = (float)xp[(2 * t16 + hh) * TQP + tid][e];
      return;
    }
#endif

    if (FWD && epi == EPI_FWD7) {
      __syncthreads();
      float pa = 0.f, pb = 0.f;
#pragma unroll 2
      for (int g = 0; g < 8; ++g) {
        float xa[8], xb[8];
        load_unit_p(xp, 8 * w + g, lane, xa);
        load_unit_p(xp, 8 * w + g, lane + 64, xb);
        const int r0 = unit_row0(8 * w + g);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(bl + 8 * HID + r0);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(bl + 8 * HID + r0 + 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) { pa = fmaf(xa[j], w0[j], pa); pb = fmaf(xb[j], w0[j], pb); }
#pragma unroll
        for (int j = 0; j < 4; ++j) { pa = fmaf(xa[4 + j], w1[j], pa); pb = fmaf(xb[4 + j], w1[j], pb); }
      }
      if (__any(guard_tripped_p(gmax, gmin))) { pa = __builtin_nanf(""); pb = pa; }
      sc[w * TQP + lane] = pa;
      sc[w * TQP + 64 + lane] = pb;
      __syncthreads();
      if (tid < TQP) {
        float a8 = 0.f;
#pragma unroll
        for (int i = 0; i < NWP; ++i) a8 += sc[i * TQP + tid];
        a8 += a.dec.b8;
        const float yv = tanhf(a8);
        if (tid < cnt) a.y[qbase + tid] = yv;
      }
      return;
    }
  }

  if (FWD) return;
  const bool poisoned = __any(guard_tripped_p(gmax, gmin));
  __syncthreads();
  const int tid_e = 64 * w + fresh_lane_p();
  if (tid_e < cnt) {
    const int tid = tid_e;
    float g0 = bl[0 * TQP + tid], g1 = bl[1 * TQP + tid], g2 = bl[2 * TQP + tid];     // through lin4 (matrix pipe)
#pragma unroll
    for (int i = 0; i < NWP; ++i) {                                                    // through lin0 (the eight waves' shares)
      g0 += bl[512 + (i * 3 + 0) * TQP + tid];
      g1 += bl[512 + (i * 3 + 1) * TQP + tid];
      g2 += bl[512 + (i * 3 + 2) * TQP + tid];
    }
    if (poisoned) g0 = __builtin_nanf("");
    const f32x4 p = pts4[qbase + tid];
    float* row = a.J + (qbase + tid) * (size_t)a.ldJ + L;
    row[7] = a.y[qbase + tid];       // the residual column = sdf (written by this thread after lin8)
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

#ifdef HM_K1P_DEBUG
extern "C" void hm_debug_set_k1p_probe(int stage, int row) {
  const int v[2] = {stage, row};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_k1p_probe), v, sizeof(v));
}
#endif

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
  a.mscr = nullptr;
  if (mode == 0) {
    hipLaunchKernelGGL((k_decoder_p<0, 0>), dim3(grid), dim3(512), 0, stream, a);
    HM_CHECK_HIP(hipGetLastError());
    return 0;
  }
  // forward + backward = two launches: forward with the ReLU masks written to a scratch block (64 KiB per workgroup), then
  // the backward-only kernel.  The block belongs to the STREAM (hm_internal.h: scratch_get): concurrent launches on one
  // decoder handle -- instance groups, several workspaces, host threads -- are on different streams and never share it
  void* scr = nullptr;
  { const int rc = scratch_get(&scr, (size_t)grid * 8 * 512 * sizeof(u32x4), stream); if (rc) return rc; }
  a.mscr = static_cast<u32x4*>(scr);
  if (tag == 0) {
    hipLaunchKernelGGL((k_decoder_p<2, 0>), dim3(grid), dim3(512), 0, stream, a);
    hipLaunchKernelGGL((k_decoder_p<3, 0>), dim3(grid), dim3(512), 0, stream, a);
  } else {
    hipLaunchKernelGGL((k_decoder_p<2, 1>), dim3(grid), dim3(512), 0, stream, a);
    hipLaunchKernelGGL((k_decoder_p<3, 1>), dim3(grid), dim3(512), 0, stream, a);
  }
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hm
