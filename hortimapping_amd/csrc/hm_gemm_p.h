// K loop of the plain-fp16 decoder kernel (hm_decoder_p.hip), in a header so that the stand-alone micro-benchmark
// scripts/microbench/k1p_kloop.hip times exactly the code the product runs.
#pragma once
#include "hm_common.h"

namespace hm_p {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int TQP = 128;       // queries per workgroup tile
constexpr int NQB = 4;         // 32-query blocks per tile
constexpr int NWP = 8;         // waves per workgroup (two per SIMD)
constexpr int NRB = 2;         // 32-row blocks per wave: w, w + 8

struct ASetP { f16x8 r[NRB]; };
struct BSetP { f16x8 q[NQB]; };

#define HM_MFMA(A, B, C) C = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0)
#define HM_FENCE() __builtin_amdgcn_sched_barrier(0)

#ifndef HM_P_CLUSTER
#define HM_P_CLUSTER 0
#endif
// AH (template parameter of k_loop_p): K-steps between a weight fetch and its use, 1 .. 3; the sets a0 .. a(AH - 1) cross
// stage boundaries (8 registers each).

// The weight fetches are BUFFER loads: descriptor (SGPRs) of the wave's stream + the lane's fixed byte offset (one VGPR
// per row block) + a scalar stream position.  Nothing per-lane changes from step to step, all stepping is scalar.  Measured
// (scripts/microbench/issue_cost.hip, profiles/r06_issue_cost.txt: 8 MFMAs + 2 weight fetches + 4 ds_read_b128 per
// iteration, two waves per SIMD, the kernel's stream footprint): 544 clocks per iteration pair with this form against
// 576-611 with global_load_dwordx4 on 64-bit per-lane addresses (512 = the matrix pipe alone).
struct WStreamP {
  __amdgpu_buffer_rsrc_t rs;
  int v0, v1;        // lane * 16, lane * 16 + 1024 (row block 1)
};
__device__ __forceinline__ f16x8 wload_p(const WStreamP& ws, int blk, int so) {
  return __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(ws.rs, blk ? ws.v1 : ws.v0, so, 0));
}

// One K-step (16 k) = a MEMORY cluster -- the weight fetch for THREE steps on (into the set the previous step has just
// finished with) and the B operands of the NEXT step (LDS -> the other of two B sets) -- followed by a COMPUTE cluster of
// 8 MFMAs (4 when only one of the wave's row blocks is valid in this stage) at raised priority.  The two waves of a SIMD
// fall into step: one issues its memory cluster (a global_load_dwordx4 costs the issuing wave ~57 clocks, a ds_read_b128
// ~26: round-6 trace) while the other holds the matrix pipe.  (HM_P_CLUSTER = 0: the round-5
// interleaving, one B set refilled in place after each operand's last use.)
// U0 / U1: row blocks computed; L0 / L1: row blocks fetched (the last three steps of a stage fetch the NEXT stage's first
// steps, whose valid blocks may differ: both are fetched).
template <bool U0, bool U1, bool L0, bool L1, bool REFILL>
__device__ __forceinline__ void step_p(f32x16 (&acc)[NRB][NQB], const ASetP& a, BSetP& b, BSetP& bn, ASetP& an,
                                       const WStreamP& ws, int so, const f16x8* ph) {
  HM_FENCE();
#ifndef ABL_NOA        // timing ablations of scripts/microbench/k1p_kloop.hip only (wrong results): no weight fetch ...
  if (L0) an.r[0] = wload_p(ws, 0, so);
  if (L1) an.r[1] = wload_p(ws, 1, so);
#endif
#if HM_P_CLUSTER
  if (REFILL) {
#pragma unroll
    for (int nb = 0; nb < NQB; ++nb) bn.q[nb] = ph[32 * nb];
  }
  HM_FENCE();
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int nb = 0; nb < NQB; ++nb) {
    if (U0) HM_MFMA(a.r[0], b.q[nb], acc[0][nb]);
    if (U1) HM_MFMA(a.r[1], b.q[nb], acc[1][nb]);
  }
  __builtin_amdgcn_s_setprio(0);
  HM_FENCE();
#else
  HM_FENCE();
#pragma unroll
  for (int nb = 0; nb < NQB; ++nb) {
    if (U0) HM_MFMA(a.r[0], b.q[nb], acc[0][nb]);
    if (U1) HM_MFMA(a.r[1], b.q[nb], acc[1][nb]);
    HM_FENCE();
#ifndef ABL_NOB        // ... / no LDS read
    if (REFILL) b.q[nb] = ph[32 * nb];
#endif
    HM_FENCE();
  }
#endif
}

// PACING of the two waves of a SIMD (round 6).  Waves w and w + 4 share a SIMD and its matrix pipe; left alone the arbiter
// serves the older wave first: it finishes a 512 x 512 stage after ~12.3 k clocks, the younger one -- starved while they
// share, then alone on the pipe at ~70 % of its rate (no partner to fill its operand waits) -- after ~19.8 k, against
// 16.4 k of matrix work for the pair (shader-clock trace, profiles/r06_k1p_trace_fwd.txt).  The leader's lead only has to
// cover its own epilogue conversion (~2 k clocks, done in the shadow of the follower's MFMAs); everything beyond that
// is pipe time lost to the lone-wave tail.  So each wave publishes the number of K-step groups it has started (one LDS
// word per wave, monotonic across stages) and the FOLLOWER (w >= 4) raises its priority while it is HM_P_LEAD or more
// groups behind its partner: the pair then advances about one group apart and ends the stage together.
#ifndef HM_P_PACE
#define HM_P_PACE 0
#endif
#ifndef HM_P_LEAD
#define HM_P_LEAD 2
#endif
typedef __attribute__((address_space(3))) int lds_int_p;
struct PaceP {
  lds_int_p* prog;   // LDS: groups started, per wave (relaxed atomics: plain ds_read_b32 / ds_write_b32, counted waits)
  int w;             // this wave
  int done;          // groups started before this stage (same in both waves of a pair: stages are barrier-separated)
};
__device__ __forceinline__ void pace_p(const PaceP* pc, int g) {
#if HM_P_PACE
  if (pc == nullptr) return;
  const int mine = pc->done + g;
  const int other = __builtin_amdgcn_readfirstlane(__hip_atomic_load(pc->prog + (pc->w ^ 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
  __hip_atomic_store(pc->prog + pc->w, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (pc->w >= 4) {
    if (other - mine >= HM_P_LEAD) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(0);
  }
#endif
}

// K loop of one stage: n_grp groups of four steps (statically named ring sets => counted vmcnt waits, no register
// rotation).  On entry a0 .. a(AH - 1) hold the stage's first steps (fetched by the previous stage's last group, or
// by the tile prologue); on exit they hold the next stage's.  sq: byte position of the stage's step 0 in the wave's stream.
template <bool U0, bool U1, int AH>
__device__ __forceinline__ void k_loop_p(f32x16 (&acc)[NRB][NQB], ASetP& a0, ASetP& a1, ASetP& a2, ASetP& a3,
                                         const WStreamP& ws, int sq, int n_grp, const f16x8* xp, int xo,
                                         long long* gstamp = nullptr, const PaceP* pc = nullptr) {
  BSetP b0, b1;
  const f16x8* ph = xp + xo;
#pragma unroll
  for (int i = 0; i < NQB; ++i) b0.q[i] = ph[32 * i];
#if HM_P_CLUSTER
#define HM_B0 b0
#define HM_B1 b1
#else
#define HM_B0 b0
#define HM_B1 b0
#endif
  // step i of a group computes on set i and fetches set (i + AH) & 3 = the step AH on
  ASetP* const S[4] = {&a0, &a1, &a2, &a3};      // constant-folded after inlining: statically named sets
  int so = sq + AH * 2048;
  ph += 2 * TQP;
  for (int g = 1; g < n_grp; ++g) {
#ifdef HM_K1P_TRACE
    if (gstamp) gstamp[g - 1] = clock64();
#endif
    pace_p(pc, g);
    step_p<U0, U1, U0, U1, true>(acc, a0, HM_B0, HM_B1, *S[(0 + AH) & 3], ws, so, ph);
    step_p<U0, U1, U0, U1, true>(acc, a1, HM_B1, HM_B0, *S[(1 + AH) & 3], ws, so + 2048, ph + 2 * TQP);
    step_p<U0, U1, U0, U1, true>(acc, a2, HM_B0, HM_B1, *S[(2 + AH) & 3], ws, so + 4096, ph + 4 * TQP);
    step_p<U0, U1, U0, U1, true>(acc, a3, HM_B1, HM_B0, *S[(3 + AH) & 3], ws, so + 6144, ph + 6 * TQP);
    so += 8192;
    ph += 8 * TQP;
  }
  // last group: the fetches that run past the stage's end are the NEXT stage's first steps (both row blocks)
  pace_p(pc, n_grp);
  step_p<U0, U1, (0 + AH < 4 ? U0 : true), (0 + AH < 4 ? U1 : true), true>(acc, a0, HM_B0, HM_B1, *S[(0 + AH) & 3], ws, so, ph);
  step_p<U0, U1, (1 + AH < 4 ? U0 : true), (1 + AH < 4 ? U1 : true), true>(acc, a1, HM_B1, HM_B0, *S[(1 + AH) & 3], ws, so + 2048, ph + 2 * TQP);
  step_p<U0, U1, (2 + AH < 4 ? U0 : true), (2 + AH < 4 ? U1 : true), true>(acc, a2, HM_B0, HM_B1, *S[(2 + AH) & 3], ws, so + 4096, ph + 4 * TQP);
  step_p<U0, U1, true, true, false>(acc, a3, HM_B1, HM_B0, *S[(3 + AH) & 3], ws, so + 6144, ph);
#if HM_P_PACE
  if (pc != nullptr) __builtin_amdgcn_s_setprio(0);
#endif
#undef HM_B0
#undef HM_B1
}

}  // namespace hm_p
