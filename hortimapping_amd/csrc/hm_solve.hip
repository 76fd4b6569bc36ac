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
#include "hm_device_fn.h"

using namespace hm;

namespace {

constexpr int MAX_E = MAX_L + 7;
constexpr int NT = 512;      // 8 waves: up to 6 of the 45 lower-triangular 32x32 blocks per wave
constexpr int NSLOT = 6;

__device__ __forceinline__ int tri(int i, int j) { return i * (i + 1) / 2 + j; }   // i >= j

// fp64 wave reduction on DPP lane moves (quad swaps, half-mirror, mirror give every lane its 16-lane row total; the
// four row totals are then combined through readlane) -- __shfl_xor on a double costs two ds_bpermute round trips per
// step, which made the reduction, not the loads, the critical path of the residual sweep.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rdlane_d(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_move<0x141>(v);   // row_half_mirror
  v += dpp_move<0x140>(v);   // row_mirror
  return (rdlane_d(v, 0) + rdlane_d(v, 16)) + (rdlane_d(v, 32) + rdlane_d(v, 48));
}

}  // namespace

constexpr int NBK = (MAX_E + 31) / 32;   // 32-wide blocks of the unknown vector (9 for E <= 263)
constexpr int PS = 36;                    // padded row stride of a 32x32 panel block in LDS (16-byte aligned rows)
constexpr int LGS = NBK * 32;             // row stride of the factor in global scratch

__device__ __forceinline__ float rdlane(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}

// Triangular solves L y = r, L^T x = y on the packed factor in LDS whose DIAGONAL 32x32 blocks have been replaced by
// their inverses: every block step is a 32x32 mat-vec (no serial chain) plus a workgroup-wide off-diagonal update.
// `rv` holds the right-hand side on entry and the solution on exit.
// Latency structure: everything a phase needs from the FACTOR (which does not depend on rv) is fetched into registers
// before the preceding barrier, so after the barrier a phase is eight broadcast reads of rv and 32 FMAs on four partial
// sums -- not 64 dependent LDS round trips (the first version spent 6.5k cycles per block step that way).
static_assert(MAX_E - 32 <= NT, "one off-diagonal row per thread");
__device__ void solve_packed(const float* Lp, float* rv, int E, int tid, bool forward = true) {
  const int lane = tid & 63, wv = tid >> 6, c = lane & 31;
  const int nblk = (E + 31) / 32;
  for (int bb = 0; forward && bb < nblk; ++bb) {           // forward
    const int base = bb * 32;
    const int hi = (base + 32 < E) ? 32 : E - base;
    float la[32], lb[32];
    const int ia = base + c;                               // wave 0: row of the inverted diagonal block
    if (wv == 0) {
      const float* lr = Lp + tri(ia < E ? ia : base, base);
#pragma unroll
      for (int k = 0; k < 32; ++k) la[k] = (ia < E && k <= c) ? lr[k] : 0.f;
    }
    const int ib = base + 32 + tid;                        // every thread: one row below the diagonal block
    const bool rowb = ib < E;
    if (rowb) {
      const float* lr = Lp + tri(ib, base);
#pragma unroll
      for (int kk = 0; kk < 32; ++kk) lb[kk] = kk < hi ? lr[kk] : 0.f;
    }
    if (wv == 0) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
      for (int k = 0; k < 32; k += 4) {
        const f32x4 r4 = *reinterpret_cast<const f32x4*>(rv + base + k);
        s0 = fmaf(la[k], r4[0], s0); s1 = fmaf(la[k + 1], r4[1], s1);
        s2 = fmaf(la[k + 2], r4[2], s2); s3 = fmaf(la[k + 3], r4[3], s3);
      }
      __builtin_amdgcn_wave_barrier();
      if (lane < 32 && ia < E) rv[ia] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (rowb) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
      for (int k = 0; k < 32; k += 4) {
        const f32x4 r4 = *reinterpret_cast<const f32x4*>(rv + base + k);
        s0 = fmaf(lb[k], r4[0], s0); s1 = fmaf(lb[k + 1], r4[1], s1);
        s2 = fmaf(lb[k + 2], r4[2], s2); s3 = fmaf(lb[k + 3], r4[3], s3);
      }
      rv[ib] -= (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
  }
  for (int bb = nblk - 1; bb >= 0; --bb) {                 // backward (transposed)
    const int base = bb * 32;
    const int hi = (base + 32 < E) ? 32 : E - base;
    float la[32], lb[32];
    const int ja = base + c;                               // wave 0: column of the inverted diagonal block
    if (wv == 0) {
#pragma unroll
      for (int ii = 0; ii < 32; ++ii) la[ii] = (ja < E && ii >= c && ii < hi) ? Lp[tri(base + ii, ja)] : 0.f;
    }
    const bool colb = tid < base;                          // every thread: one column left of the diagonal block
    if (colb) {
#pragma unroll
      for (int ii = 0; ii < 32; ++ii) lb[ii] = ii < hi ? Lp[tri(base + ii, tid)] : 0.f;
    }
    if (wv == 0) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
      for (int k = 0; k < 32; k += 4) {
        const f32x4 r4 = *reinterpret_cast<const f32x4*>(rv + base + k);
        s0 = fmaf(la[k], r4[0], s0); s1 = fmaf(la[k + 1], r4[1], s1);
        s2 = fmaf(la[k + 2], r4[2], s2); s3 = fmaf(la[k + 3], r4[3], s3);
      }
      __builtin_amdgcn_wave_barrier();
      if (lane < 32 && ja < E) rv[ja] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    if (colb) {
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
      for (int k = 0; k < 32; k += 4) {
        const f32x4 r4 = *reinterpret_cast<const f32x4*>(rv + base + k);
        s0 = fmaf(lb[k], r4[0], s0); s1 = fmaf(lb[k + 1], r4[1], s1);
        s2 = fmaf(lb[k + 2], r4[2], s2); s3 = fmaf(lb[k + 3], r4[3], s3);
      }
      rv[tid] -= (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
  }
}

__device__ long long* g_k5_trace = nullptr;

// ---------------------------------------------------------------------------------------------------------------
// Fast path of the solve: Jacobi-preconditioned conjugate gradients on the damped normal matrix.
//
// The system the reference inverts (optimizer.py:200-234) is  A = H + w_code I_z + s_damp e_s e_s^T + lambda_0 diag(.)
// (or + lambda_0 max(diag) I): symmetric positive definite and, with the shipped lambda_0 = 0.1, WELL conditioned once
// scaled by its diagonal -- measured cond(D^-1/2 A D^-1/2) = 70 ... 210 on the c2_joint systems, so PCG reaches a 1e-7
// relative residual in 22 ... 26 iterations (solution within 1e-6 of the fp64 solve; the reference's own fp32
// inverse + mv is 3e-7 ... 6e-7 off it).  One iteration is a 263 x 263 symmetric mat-vec out of LDS: about 3k cycles,
// against 515k cycles for the blocked Cholesky + two triangular solves + fp64 refinement, whose 288-step pivot chain
// (a wave-wide LDS round trip per pivot) cannot be shortened.  The direct solver stays as the fallback for systems
// on which CG does not reach the tolerance within CG_MAXIT iterations (lm_on = false, tiny lambda_0, degenerate
// observations ...): slow convergence IS the conditioning test -- there a small residual would not bound the error,
// and the refined direct solve is the accurate one.  The choice is made per instance on the device; both paths are
// deterministic.
// Storage: strictly lower triangle, row-major, rows padded to whole float4s (rows 4g .. 4g+3 hold 4 (g + 1) floats),
// diagonal slot and padding zero; the diagonal lives in diagA.  Every element is read ONCE per mat-vec with 16-byte
// loads and used for both the row dot product (reduced inside a 16-lane DPP row) and the column sums (per-lane
// accumulators, merged in a fixed order).
// ---------------------------------------------------------------------------------------------------------------
constexpr int CG_ROWS = (MAX_E + 3) / 4 * 4;                     // 264
constexpr int CG_G = CG_ROWS / 4;                                 // 66 row groups
constexpr int CG_FLOATS = 4 * CG_G * (2 * (CG_G - 1) + 4) + 4;    // 4 (g+1)(2g + r) at g = CG_G - 1, r = 4 (35376) + one zero float4
constexpr int CG_MAXIT = 48;           // the damped systems need 25 ... 31; beyond 48 the direct path runs
constexpr float CG_TOL = 1e-7f;        // relative residual (preconditioned norm) at which the iteration stops

__device__ __forceinline__ int cg_off4(int i) { const int g = i >> 2; return (g + 1) * (2 * g + (i & 3)); }   // float4 units

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_sum_f(float v) {
  v += dpp_f<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);    // row_half_mirror
  v += dpp_f<0x140>(v);    // row_mirror: every lane of a 16-lane row holds the row total
  return (rdlane(v, 0) + rdlane(v, 16)) + (rdlane(v, 32) + rdlane(v, 48));
}

// Workgroup sum of one value per thread, fixed order (deterministic).  `red` : NT / 64 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red, int lane, int wv) {
  v = wave_sum_f(v);
  if (lane == 0) red[wv] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) s += red[i];
  __syncthreads();
  return s;
}

__device__ __forceinline__ void block_sum2(float v0, float v1, float& s0, float& s1, float* red, int lane, int wv) {
  v0 = wave_sum_f(v0);
  v1 = wave_sum_f(v1);
  if (lane == 0) { red[wv] = v0; red[NT / 64 + wv] = v1; }
  __syncthreads();
  s0 = 0.f; s1 = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) { s0 += red[i]; s1 += red[NT / 64 + i]; }
  __syncthreads();
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// v[lane] + v[lane ^ 16] + v[lane ^ 32] + v[lane ^ 48] in every lane: two gfx950 lane-swap instructions (VALU speed; the
// ds_bpermute route costs an LDS round trip per step).  v_permlane16_swap a, b leaves a = [a0 b0 a2 b2], b = [a1 b1 a3 b3]
// (16-lane rows), v_permlane32_swap a = [a_lo b_lo], b = [a_hi b_hi]; with both registers holding v, a + b is the
// pairwise sum.  Inline asm on purpose: hipcc (ROCm 7.2) folds __builtin_amdgcn_permlane16_swap(u, u) into "both results
// equal" and emits a + a (measured on gfx950: every row got 4 x row 0); the s_nop covers the VALU-write -> permlane hazard.
__device__ __forceinline__ float sum_dpp_rows(float v) {
  float a = v, b = v;
  asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));       // not volatile: a pure function of (a, b),
  float s = a + b, t = s;                                                     // free to be scheduled among its siblings
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(s), "+v"(t));
  return s + t;
}
__device__ __forceinline__ float wave_sum_fast(float v) {       // whole-wave sum in every lane
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x141>(v);
  v += dpp_f<0x140>(v);
  return sum_dpp_rows(v);
}

// Returns true when x (LDS, E entries) solves A x = b: CG stopped at a CG_TOL relative residual in the PRECONDITIONED
// norm sqrt(r . M^-1 r) (the 2-norm of the Jacobi-scaled system, in which cond <= (1 + lambda_0) / lambda_0 * lambda_max
// is bounded by the damping: 70 ... 210 measured, error <= cond * tol).  The plain 2-norm of r is dominated by the
// large-gradient unknowns and lets the small-scale ones (pose entries) stop 1e-4 short -- golden case invalid_later.
// `red`: 4 * (NT / 64) floats (two generations of two sums, so one barrier per reduction suffices).
__device__ bool pcg_solve(const float* __restrict__ H, int ld, const float* diagA, const float* bvec, float* x, int E,
                          float* Ap, float* pv, float* rv, float* qv, float* part, float* red, int tid) {
  const int lane = tid & 63, wv = tid >> 6;
  const int G = (E + 3) >> 2;
  f32x4* Ap4 = reinterpret_cast<f32x4*>(Ap);
  const bool trc = g_k5_trace != nullptr && blockIdx.x == 0 && tid == 0;
  if (trc) g_k5_trace[17] = clock64();
  // ---- strictly lower triangle of H -> LDS (diagonal slot and padding zero).  Flat over the padded float4 slots so that
  // every thread has ~18 INDEPENDENT 16-byte global loads in flight (a row-by-row loop waits out one L2 round trip per
  // row chunk: 250k cycles for this copy alone).  Slot f belongs to row group g with 2 g (g + 1) <= f < 2 (g + 1)(g + 2).
  const int nslot = 2 * G * (G + 1);            // slot `nslot` (inside CG_FLOATS) is kept zero: target of masked reads
  {
    constexpr int NLD = (2 * CG_G * (CG_G + 1) + NT - 1) / NT;          // 18
    f32x4 buf[NLD];
#pragma unroll
    for (int t = 0; t < NLD; ++t) {
      const int f = tid + t * NT;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (f < nslot) {
        int g = (int)((sqrtf(1.f + 2.f * (float)f) - 1.f) * 0.5f);
        g += (2 * (g + 1) * (g + 2) <= f) ? 1 : 0;
        g -= (2 * g * (g + 1) > f) ? 1 : 0;
        const int rem = f - 2 * g * (g + 1);
        int r = (int)((float)rem / (float)(g + 1));                      // rem < 4 (g + 1) <= 264: exact after the fix-up
        r -= (r * (g + 1) > rem) ? 1 : 0;
        r += ((r + 1) * (g + 1) <= rem) ? 1 : 0;
        const int f4 = rem - r * (g + 1);
        const int i = 4 * g + r, k0 = 4 * f4;
        if (i < E && k0 < i) {
          v = *reinterpret_cast<const f32x4*>(H + (size_t)i * ld + k0);     // ld % 4 == 0: 16-byte aligned
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (k0 + e < i) ? v[e] : 0.f;
        }
      }
      buf[t] = v;
    }
#pragma unroll
    for (int t = 0; t < NLD; ++t) {
      const int f = tid + t * NT;
      if (f <= nslot) Ap4[f] = buf[t];          // f == nslot: the zero slot (buf is zero there)
    }
  }
  float ri = 0.f, di = 1.f;
  if (tid < CG_ROWS) {
    ri = tid < E ? bvec[tid] : 0.f;
    di = tid < E ? diagA[tid] : 1.f;
    x[tid] = 0.f;
    rv[tid] = ri;
    pv[tid] = ri / di;
  }
  if (trc) g_k5_trace[18] = clock64();
  int gen = 0;
  auto sums2 = [&](float v0, float v1, float& s0, float& s1) {   // two workgroup sums, ONE barrier (red is two-generation)
    v0 = wave_sum_fast(v0);
    v1 = wave_sum_fast(v1);
    float* rg = red + gen * 2 * (NT / 64);
    gen ^= 1;
    if (lane == 0) { rg[wv] = v0; rg[NT / 64 + wv] = v1; }
    __syncthreads();
    s0 = 0.f; s1 = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) { s0 += rg[i]; s1 += rg[NT / 64 + i]; }
  };
  auto sum1 = [&](float v0) {                   // one workgroup sum, one barrier
    v0 = wave_sum_fast(v0);
    float* rg = red + gen * 2 * (NT / 64);
    gen ^= 1;
    if (lane == 0) rg[wv] = v0;
    __syncthreads();
    float s0 = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) s0 += rg[i];
    return s0;
  };
  float rz, bb;
  sums2(ri * (ri / di), ri * ri, rz, bb);       // (the barrier also publishes pv / rv / x and the matrix)
  if (!(bb > 0.f)) return bb == 0.f;            // b = 0: x = 0 is the solution; NaN: let the direct path report it
  const float rz_stop = CG_TOL * CG_TOL * rz;
  // mat-vec geometry: a wave takes FOUR matrix rows at a time, one per 16-lane DPP row; lane (r, j) covers the columns
  // 4 (j + 16 c) .. + 3, c = 0..3, of row 4 g + r (conflict-free 16-byte reads: consecutive lanes, consecutive float4s).
  // The row dot products then reduce INSIDE a DPP row (four v_add_dpp, no cross-row traffic, all four rows at once);
  // the column sums stay in 16 registers per lane and are merged across DPP rows (lane swaps) / waves (LDS, fixed order)
  // once per mat-vec.  Packed fp32 FMAs (v_pk_fma_f32) halve the instruction count of the products.
  const int dr = lane >> 4, dj = lane & 15;
  bool conv = false;
  for (int it = 0; it < CG_MAXIT; ++it) {
    const bool tri = trc && it == 1;
    if (tri) g_k5_trace[19] = clock64();
    if (trc) g_k5_trace[25] = it + 1;
    // ---- q = A p ----
    // c = 0..3: float4 columns dj + 16 c (columns < 256); c = 4: float4 64 + dj for dj < 2 (columns 256 .. 263, which
    // exist only in the row groups g >= 64).  The group loop has a static trip count (CG_G / 8 rounded up) and no
    // branches -- lanes outside a row, and whole rounds past the last group, read the zero slot -- so that hipcc
    // unrolls it and keeps the LDS reads of the following groups in flight under the products of the current one.
    f32x2 plo[5], phi[5], calo[5], cahi[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const f32x4 t4 = (c < 4 || dj < 2) ? reinterpret_cast<const f32x4*>(pv)[dj + 16 * c] : f32x4{0.f, 0.f, 0.f, 0.f};
      plo[c] = f32x2{t4[0], t4[1]}; phi[c] = f32x2{t4[2], t4[3]};
      calo[c] = f32x2{0.f, 0.f}; cahi[c] = f32x2{0.f, 0.f};
    }
    constexpr int NROUND = (CG_G + NT / 64 - 1) / (NT / 64);         // 9
#pragma unroll
    for (int t = 0; t < NROUND; ++t) {
      const int g = wv + t * (NT / 64);
      const bool live = g < G;
      const int gc = live ? g : 0;
      const int i = 4 * gc + dr;
      const int o4 = cg_off4(i);
      const float pi_ = pv[i];
      const f32x2 pi2 = {pi_, pi_};
      f32x2 rs2 = {0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        if (c == 4 && t < 64 / (NT / 64)) continue;                  // columns >= 256 exist from group 64 on (t = 8)
        const int f4 = dj + 16 * c;
        const bool in = live && f4 <= gc && (c < 4 || dj < 2);
        const f32x4 av = Ap4[in ? o4 + f4 : nslot];                  // outside the padded row: the zero slot
        const f32x2 alo = {av[0], av[1]}, ahi = {av[2], av[3]};
        rs2 = alo * plo[c] + rs2;
        rs2 = ahi * phi[c] + rs2;
        calo[c] = alo * pi2 + calo[c];
        cahi[c] = ahi * pi2 + cahi[c];
      }
      float rsum = rs2[0] + rs2[1];
      rsum += dpp_f<0xB1>(rsum);
      rsum += dpp_f<0x4E>(rsum);
      rsum += dpp_f<0x141>(rsum);
      rsum += dpp_f<0x140>(rsum);
      if (live && dj == 0) qv[i] = rsum;
    }
    if (tri) g_k5_trace[20] = clock64();
    // column sums: across the four DPP rows of the wave (lane swaps), then across waves through LDS (fixed order)
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const f32x4 t4 = {sum_dpp_rows(calo[c][0]), sum_dpp_rows(calo[c][1]), sum_dpp_rows(cahi[c][0]), sum_dpp_rows(cahi[c][1])};
      if (dr == 0 && (c < 4 || dj < 2)) reinterpret_cast<f32x4*>(part + wv * CG_ROWS)[dj + 16 * c] = t4;
    }
    __syncthreads();
    if (tri) g_k5_trace[21] = clock64();
    float qi = 0.f, pi = 0.f;
    if (tid < E) {
      pi = pv[tid];
      float cs = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < NT / 64; ++w2) cs += part[w2 * CG_ROWS + tid];
      qi = fmaf(di, pi, qv[tid] + cs);
    }
    const float pq = sum1(pi * qi);
    if (tri) g_k5_trace[22] = clock64();
    if (!(pq > 0.f)) break;                     // not positive definite / not finite: direct path decides
    const float alpha = rz / pq;
    float zi = 0.f;
    if (tid < E) {
      x[tid] = fmaf(alpha, pi, x[tid]);
      ri = fmaf(-alpha, qi, ri);                // the residual entry of this thread lives in a register
      zi = ri / di;
    }
    const float rz1 = sum1(ri * zi);
    if (rz1 <= rz_stop) { conv = true; break; }
    const float beta = rz1 / rz;
    rz = rz1;
    if (tid < E) pv[tid] = fmaf(beta, pi, zi);
    __syncthreads();
    if (tri) g_k5_trace[27] = clock64();
  }
  __syncthreads();
  return conv;                                  // false: slow convergence or breakdown -> direct solver
}

__global__ __launch_bounds__(NT) void k_solve_update(const SolveArgs a) {
  // one LDS array, two lives: the 32-column panel [NBK][32][PS] during the factorisation, then the packed factor
  __shared__ __attribute__((aligned(16))) float smem[CG_FLOATS];   // >= MAX_E (MAX_E + 1) / 2 of the direct path
  __shared__ __attribute__((aligned(16))) float cg_p[CG_ROWS];
  __shared__ __attribute__((aligned(16))) float cg_q[CG_ROWS];
  __shared__ __attribute__((aligned(16))) float cg_part[(NT / 64) * CG_ROWS];
  __shared__ float bvec[NBK * 32];
  __shared__ __attribute__((aligned(16))) float xvec[NBK * 32];
  __shared__ __attribute__((aligned(16))) float rv[NBK * 32];
  __shared__ float diagA[NBK * 32];
  __shared__ double cs[NBK * 32], rs[NBK * 32];   // fp64 column / row sums of the refinement residual
  __shared__ __attribute__((aligned(16))) float colk[32];
  __shared__ float dinv[32];
  __shared__ float red[4 * (NT / 64)];
  __shared__ float red2[NT / 64];
  __shared__ int flag;

  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  if (a.active[b] == 0) return;
  const bool trc = g_k5_trace != nullptr && b == 0 && tid == 0;
  if (trc) g_k5_trace[0] = clock64();
  const int L = a.L, P = a.P, E = L + P, ld = a.ldJ;
  const float* H = a.Hext + (size_t)b * ld * ld;
  float* Lg = a.Lfac + (size_t)b * LGS * LGS;
  float* z = a.latent + (size_t)b * a.ld_latent;
  const int lane = tid & 63, wv = tid >> 6;
  const int nblk = (E + 31) / 32;

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
  // A non-finite normal matrix (e.g. activations beyond fp16 range in the f16x3 arithmetic) is a numerical failure
  // and reported as such -- it must not pass for the reference's ordinary "submap not valid" exit below, which a
  // NaN sdf would otherwise trigger (no with-grad sample survives a NaN comparison).
  {
    bool nf = false;
    for (int i = tid; i < E; i += NT) nf |= !isfinite(diagA[i]) || !isfinite(bvec[i]);
    if (nf || (a.nflag != nullptr && a.nflag[b] != 0)) flag = 1;
  }
  __syncthreads();
  if (flag) {
    if (tid == 0) { a.active[b] = 0; a.status[b] |= HM_STATUS_SOLVE_FAILED; }
    return;
  }
  if (a.V != nullptr && a.V[b] <= 0) {           // optimizer.py:139-141
    if (tid == 0) { a.active[b] = 0; a.status[b] |= HM_STATUS_INVALID; }
    return;
  }
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
  if (a.dbg_A != nullptr) {
    float* Ad = a.dbg_A + (size_t)b * ld * ld;
    for (int i = tid >> 5; i < E; i += NT / 32)
      for (int k = tid & 31; k <= i; k += 32) Ad[(size_t)i * ld + k] = (i == k) ? diagA[i] : H[(size_t)i * ld + k];
  }
  if (a.dbg_b != nullptr)
    for (int i = tid; i < E; i += NT) a.dbg_b[(size_t)b * ld + i] = bvec[i];
  if (trc) g_k5_trace[1] = clock64();

  // ---- solve A x = b: preconditioned CG (fast path), blocked Cholesky + refinement (fallback) ----
  static_assert(CG_FLOATS >= MAX_E * (MAX_E + 1) / 2 && CG_ROWS <= NBK * 32, "LDS plan");
  const bool cg_ok = !a.force_direct && pcg_solve(H, ld, diagA, bvec, xvec, E, smem, cg_p, rv, cg_q, cg_part, red, tid);
  if (trc) g_k5_trace[2] = clock64();
  if (!cg_ok) {
    // ---- blocked right-looking Cholesky, 32-column panels, trailing update on the fp32 matrix cores ----
    // When the last 32-row block has a spare row (E not a multiple of 32, i.e. joint mode) the right-hand side rides
    // along as row E of the bordered matrix [[A, b], [b^T, c]]: its Cholesky factor's row E IS y = L^-1 b, so the
    // forward sweep of the first triangular solve costs one more panel row instead of nine barrier-separated block
    // steps.  c only has to keep the last pivot positive (1e30).
    const bool aug = (E & 31) != 0;
    // Wave w owns the lower-triangular 32x32 blocks p = w, w+8, ... (p = bi(bi+1)/2 + bk) in MFMA C layout.
    const int cc = lane & 31, hh = lane >> 5;
    f32x16 blk[NSLOT];
    int obi[NSLOT], obk[NSLOT];
  #pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) {
      const int p = wv + (NT / 64) * sl;
      int bi = 0;
      while ((bi + 1) * (bi + 2) / 2 <= p) ++bi;
      const int bk = p - bi * (bi + 1) / 2;
      obi[sl] = bi < nblk ? bi : -1;
      obk[sl] = bk;
  #pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh, k = bk * 32 + cc;
        float v = (i == k) ? 1.f : 0.f;                           // padding: identity
        if (bi < nblk && i < E && k < E) v = (i == k) ? diagA[i] : (k < i ? H[(size_t)i * ld + k] : H[(size_t)k * ld + i]);
        if (aug && bi < nblk && (i == E || k == E))               // bordered system [[A, b], [b^T, c]], see below
          v = (i == E && k == E) ? 1e30f : (i == E ? (k < E ? bvec[k] : 0.f) : (i < E ? bvec[i] : 0.f));
        blk[sl][r] = v;
      }
    }
    float* Pn = smem;                                             // panel: Pn[(bi * 32 + row) * PS + col]
    for (int bj = 0; bj < nblk; ++bj) {
      const bool trj = trc && bj == 0;          // phase stamps of the first (largest) block column
      if (trj) g_k5_trace[8] = clock64();
      // 1. owners of block column bj publish their blocks
  #pragma unroll
      for (int sl = 0; sl < NSLOT; ++sl)
        if (obi[sl] >= 0 && obk[sl] == bj) {
  #pragma unroll
          for (int r = 0; r < 16; ++r)
            Pn[(obi[sl] * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh) * PS + cc] = blk[sl][r];
        }
      __syncthreads();
      if (trj) g_k5_trace[9] = clock64();
      // 2. diagonal block: in-wave Cholesky, lane = row, columns in registers, broadcasts by v_readlane
      if (wv == 0) {
        float ar[32];
        const float* src = Pn + (bj * 32 + cc) * PS;
  #pragma unroll
        for (int k = 0; k < 32; ++k) ar[k] = src[k];
        bool bad = false;
  #pragma unroll
        for (int k = 0; k < 32; ++k) {
          colk[cc] = ar[k];
          __builtin_amdgcn_wave_barrier();
          const float piv = colk[k];
          bad |= !(piv > 0.f);
          float y = __builtin_amdgcn_rsqf(piv);
          y = y * fmaf(-0.5f * piv * y, y, 1.5f);
          const float t = -ar[k] * (y * y);
  #pragma unroll
          for (int g = (k + 1) / 4; g < 8; ++g) {
            const f32x4 v = reinterpret_cast<const f32x4*>(colk)[g];
  #pragma unroll
            for (int e = 0; e < 4; ++e)
              if (4 * g + e > k) ar[4 * g + e] = fmaf(t, v[e], ar[4 * g + e]);
          }
          __builtin_amdgcn_wave_barrier();
          ar[k] = (cc == k) ? piv * y : ar[k] * y;
          if (lane == k) dinv[k] = y;
          __builtin_amdgcn_sched_barrier(0);
        }
        if (bad) flag = 1;
        if (lane < 32) {
          float* dst = Pn + (bj * 32 + cc) * PS;
  #pragma unroll
          for (int k = 0; k < 32; ++k) dst[k] = (k <= cc) ? ar[k] : 0.f;
        }
      }
      if (trj) g_k5_trace[10] = clock64();
      __syncthreads();
      if (trj) g_k5_trace[11] = clock64();
      // 3. panel below the diagonal block: row-wise forward substitution  x L11^T = a
      {
        const int nrow = (nblk - 1 - bj) * 32;
        const float* L11 = Pn + bj * 32 * PS;
        for (int t = tid; t < nrow; t += NT) {
          float* row = Pn + ((bj + 1) * 32 + t) * PS;
          float x[32];
  #pragma unroll
          for (int k = 0; k < 32; ++k) x[k] = row[k];
  #pragma unroll
          for (int k = 0; k < 32; ++k) {
            float sacc = x[k];
  #pragma unroll
            for (int m2 = 0; m2 < k; ++m2) sacc = fmaf(-x[m2], L11[k * PS + m2], sacc);
            x[k] = sacc / L11[k * PS + k];
            __builtin_amdgcn_sched_barrier(0);
          }
  #pragma unroll
          for (int k = 0; k < 32; ++k) row[k] = x[k];
        }
      }
      if (trj) g_k5_trace[12] = clock64();
      __syncthreads();
      if (trj) g_k5_trace[13] = clock64();
      // 4. trailing update  A22 -= L21 L21^T  on the matrix cores, and the finished panel goes to global scratch
  #pragma unroll
      for (int sl = 0; sl < NSLOT; ++sl)
        if (obi[sl] >= 0 && obk[sl] > bj) {
          const float* pa = Pn + (obi[sl] * 32 + cc) * PS + 4 * hh;
          const float* pb = Pn + (obk[sl] * 32 + cc) * PS + 4 * hh;
  #pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(pa + 8 * g);
            const f32x4 bv = *reinterpret_cast<const f32x4*>(pb + 8 * g);
  #pragma unroll
            for (int t = 0; t < 4; ++t)
              blk[sl] = __builtin_amdgcn_mfma_f32_32x32x2f32(-av[t], bv[t], blk[sl], 0, 0, 0);
          }
        }
      {
        const int nel = (nblk - bj) * 32 * 32;
        for (int e = tid; e < nel; e += NT) {
          const int r = e >> 5, k = e & 31;
          Lg[(size_t)(bj * 32 + r) * LGS + bj * 32 + k] = Pn[(bj * 32 + r) * PS + k];
        }
      }
      if (trj) g_k5_trace[14] = clock64();
      __syncthreads();
      if (trj) g_k5_trace[15] = clock64();
    }
    if (flag) {
      if (tid == 0) { a.active[b] = 0; a.status[b] |= HM_STATUS_SOLVE_FAILED; }
      return;
    }
    if (trc) g_k5_trace[16] = clock64();
    // packed factor into LDS (the panel area is dead now), then invert its diagonal blocks in place
    __threadfence_block();
    float* Lp = smem;
    for (int i = tid >> 5; i < E; i += NT / 32)
      for (int k = tid & 31; k <= i; k += 32) Lp[tri(i, k)] = Lg[(size_t)i * LGS + k];
    __syncthreads();
    {
      // column `cc` of inverse(L11): x[k] = Linv[k][cc], forward substitution on e_cc with broadcast reads of L11.
      // Each half-wave takes one diagonal block (blocks w and w + 8 in wave w): all nine blocks in a single pass.
      const int bb = wv + (NT / 64) * hh;
      if (bb < nblk) {
        const int base = bb * 32;
        float x[32];
        // row k of L11 (k + 1 broadcast reads) is fetched one row AHEAD of its use: the forward substitution then costs
        // one LDS latency per row overlapped with the previous row's dot product, instead of a round trip per element
        float cur[32], nxt[32];
        {
          const float* lr = Lp + tri(base, base);
          cur[0] = lr[0];
        }
  #pragma unroll
        for (int k = 0; k < 32; ++k) {
          const int i = base + k;
          if (k + 1 < 32) {
            const int in = i + 1 < E ? i + 1 : base;               // rows past E are never used (x stays 0)
            const float* lr = Lp + tri(in, base);
  #pragma unroll
            for (int m2 = 0; m2 <= k + 1; ++m2) nxt[m2] = lr[m2];
          }
          __builtin_amdgcn_sched_barrier(0);
          float s0 = (k == cc) ? 1.f : 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  #pragma unroll
          for (int m2 = 0; m2 < k; ++m2) {
            if ((m2 & 3) == 0) s0 = fmaf(-x[m2], cur[m2], s0);
            else if ((m2 & 3) == 1) s1 = fmaf(-x[m2], cur[m2], s1);
            else if ((m2 & 3) == 2) s2 = fmaf(-x[m2], cur[m2], s2);
            else s3 = fmaf(-x[m2], cur[m2], s3);
          }
          x[k] = (i < E && k >= cc) ? ((s0 + s1) + (s2 + s3)) / cur[k] : 0.f;
          __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
          for (int m2 = 0; m2 <= k + 1 && m2 < 32; ++m2) cur[m2] = nxt[m2];
        }
        __builtin_amdgcn_wave_barrier();
  #pragma unroll
        for (int k = 0; k < 32; ++k)
          if (k >= cc && base + k < E && base + cc < E) Lp[tri(base + k, base + cc)] = x[k];
      }
    }
    __syncthreads();

    // ---- solve L L^T x = b (fp32), then one refinement step with an fp64 residual from the fp32 system ----
    for (int i = tid; i < NBK * 32; i += NT) {
      rv[i] = i < E ? (aug ? Lg[(size_t)E * LGS + i] : bvec[i]) : 0.f;     // aug: y = row E of the bordered factor
      xvec[i] = 0.f;
    }
    __syncthreads();
    solve_packed(Lp, rv, E, tid, !aug);
    if (trc) g_k5_trace[3] = clock64();
    for (int i = tid; i < E; i += NT) xvec[i] = rv[i];
    __syncthreads();
    // r = b - A x in fp64 from the fp32 system.  Only the lower triangle of H is valid; one sweep over it with
    // row-contiguous (coalesced) reads serves both halves of the symmetric product: element H[i][k] (k < i) adds
    // H[i][k] x[k] to row i (wave reduction) and H[i][k] x[i] to column k (per-lane accumulators, merged with LDS fp64
    // atomics at the end).  Eleven rows per wave (a third of its share) are in flight at a time to cover the load latency.
    for (int k = tid; k < NBK * 32; k += NT) { cs[k] = k < E ? (double)diagA[k] * (double)xvec[k] : 0.0; rs[k] = 0.0; }
    __syncthreads();
    {
      constexpr int NCH = (MAX_E + 63) / 64, RU = 11;
      double colacc[NCH];
  #pragma unroll
      for (int c = 0; c < NCH; ++c) colacc[c] = 0.0;
      for (int i0 = wv; i0 < E; i0 += (NT / 64) * RU) {
        float h[RU][NCH];
  #pragma unroll
        for (int u = 0; u < RU; ++u) {
          const int i = i0 + (NT / 64) * u;
  #pragma unroll
          for (int c = 0; c < NCH; ++c) {
            const int k = lane + 64 * c;
            h[u][c] = (i < E && k < i) ? H[(size_t)i * ld + k] : 0.f;
          }
        }
  #pragma unroll
        for (int u = 0; u < RU; ++u) {
          const int i = i0 + (NT / 64) * u;
          const double xi = i < E ? (double)xvec[i] : 0.0;
          double sd = 0.0;
  #pragma unroll
          for (int c = 0; c < NCH; ++c) {
            sd += (double)h[u][c] * (double)xvec[lane + 64 * c];
            colacc[c] += (double)h[u][c] * xi;
          }
          sd = wave_sum(sd);
          if (lane == 0 && i < E) rs[i] = sd;
        }
      }
  #pragma unroll
      for (int c = 0; c < NCH; ++c)
        if (lane + 64 * c < E) atomicAdd(&cs[lane + 64 * c], colacc[c]);
    }
    __syncthreads();
    for (int i = tid; i < E; i += NT) rv[i] = (float)((double)bvec[i] - (rs[i] + cs[i]));
    __syncthreads();
    if (trc) g_k5_trace[4] = clock64();
    solve_packed(Lp, rv, E, tid);
    if (trc) g_k5_trace[5] = clock64();
    for (int i = tid; i < E; i += NT) xvec[i] += rv[i];
    __syncthreads();
  }
  if (trc) { g_k5_trace[6] = clock64(); g_k5_trace[7] = cg_ok ? 1 : 0; }
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

extern "C" void hm_debug_set_k5_trace(long long* d_buf) {
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_k5_trace), &d_buf, sizeof(d_buf));
}

namespace hm {

int launch_solve_update(const SolveArgs& args, int B, hipStream_t stream) {
  hipLaunchKernelGGL(k_solve_update, dim3(B), dim3(NT), 0, stream, args);
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hm
