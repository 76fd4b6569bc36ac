// K1a: DeepSDF decoder of ANY architecture the reference's `Decoder` class can build -- arbitrary `dims`, `latent_in`,
// `xyz_in_all`, LayerNorm (`norm_layers` without `weight_norm`), `use_tanh` -- forward and input-gradient backward, in
// exact fp32 on the f32-input matrix cores (k_decoder_any) and in f16x3 on the fp16 matrix cores with split operands
// (k_decoder_any_h, second half of this file; same structure, activations as two fp16 planes, the range guard of
// hm_decoder_h.hip).  The two shipped models (8 x 512, latent_in = [4], weight norm) never come here: they run on the
// specialised kernels (hm_decoder.hip / hm_decoder_h.hip / hm_decoder_p.hip); these kernels exist so that a `specs.json`
// with another layer table is decoded and optimised on the GPU instead of being refused (VERDICT r04 "decoder
// generality").  They implement the launch contract of k_decoder (hm_decoder.hip), so the LM iteration of
// hm_optimize.hip runs on them unchanged (render chain as separate launches).  Each kernel is compiled with and
// without the LayerNorm code paths (template parameter LN): tables without LayerNorm run spill-free.
//
// Replaces, per tile of 64 queries of one fruit instance (paths relative to /root/reference):
//   Decoder.__init__ layer table        deepsdf/networks/deep_sdf_decoder.py:29-72
//   Decoder.forward                     deepsdf/networks/deep_sdf_decoder.py:75-110
//   decode_sdf                          wild_completion/utils.py:144-172          (mode 0)
//   get_batch_sdf_jacobian/get_gradient wild_completion/utils.py:112-122,175-193  (mode 1)
//   pose chain rule of compute_sdf_loss wild_completion/loss.py:229-241
//
// Layout.  A workgroup (8 waves) owns 64 queries.  The activations (forward) / back-propagated gradients (backward) of
// the current layer live in LDS as X[k/4][q][k%4] (128 KiB, widths up to 512) and are the B operand of the exact-fp32
// K loop of the fixed-architecture kernel (hm_gemm_f32.h: v_mfma_f32_32x32x2_f32, bitwise an fmaf chain per output);
// the weights of every layer are pre-packed twice in A-operand order (forward: W, backward: W^T) and stream from L2.
// Wave w owns the 32-row blocks w and w + 8 of every layer; row f of a layer's output therefore sits in the same
// (wave, lane, register) in forward layer l and in backward layer l + 1, so the ReLU masks (64 bits per layer and lane)
// never leave the thread.  LayerNorm layers save their normalised activations and 1/sigma in a global slab indexed by
// workgroup (persistent grid); their statistics are per-query sums over all waves (LDS reduction).
#include "hm_common.h"
#include "hm_internal.h"
#include "hm_gemm_f32.h"
#include "hm_gemm_h.h"

using namespace hm;

namespace {

constexpr int ANY_W = HM_ANY_MAX_WIDTH;   // 512
constexpr int ANY_GRID = 512;             // persistent workgroups (= LayerNorm slab slots)
constexpr int SLAB = (ANY_W + 1) * 64;    // floats per LayerNorm layer and slot: x_hat [512][64] + 1/sigma [64]

struct AnyArgs {
  AnyDev dec;
  const float* pts;    // [B][n_stride][4]
  const int* n_q;      // [B]
  const int* active;   // [B] or nullptr
  const float* zc;     // [B][512]: latent of instance b in the first L floats (the c0 buffer of the fixed architecture)
  float* y;            // [B][n_stride]
  float* J;            // [B][n_stride][ldJ]
  float* ln_slab;      // [ANY_GRID][n_ln][SLAB] (unused without LayerNorm layers)
  float* gi_slab;      // [ANY_GRID][MAX_L][64]: d sdf / d z of the tile in flight, query-minor
  unsigned long long* mk_slab;   // [ANY_GRID][HM_ANY_MAX_LIN][512]: ReLU masks of the tile in flight, one word per thread and layer
  int n_stride, B, ldJ, pose_dim, n_tiles;
};

__device__ __forceinline__ int xidx(int k, int q) { return (((k >> 2) * TQ + q) << 2) + (k & 3); }

// per-query totals over all rows: lane (qa, hi) of every wave contributes one partial for query qa (p0) and one for
// query qa + 32 (p1); returns the two totals of this lane's queries
__device__ __forceinline__ void query_sum2(float p0, float p1, float* red, int w, int lane, float& t0, float& t1) {
  red[w * 64 + lane] = p0;
  red[(NWAVE + w) * 64 + lane] = p1;
  __syncthreads();
  const int qa = lane & 31;
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int i = 0; i < NWAVE; ++i) {
    s0 += red[i * 64 + qa] + red[i * 64 + 32 + qa];
    s1 += red[(NWAVE + i) * 64 + qa] + red[(NWAVE + i) * 64 + 32 + qa];
  }
  __syncthreads();
  t0 = s0; t1 = s1;
}

__device__ __forceinline__ void run_gemm(f32x16 (&acc)[2][2], const float* wp, int n_kg, int n_rb, int w,
                                         const f32x4* xs4, int lane) {
  acc[0][0] = zero16(); acc[0][1] = zero16(); acc[1][0] = zero16(); acc[1][1] = zero16();
  const bool u0 = w < n_rb, u1 = w + NWAVE < n_rb;
  const f32x4* wp4 = reinterpret_cast<const f32x4*>(wp);
  const f32x4* wp0 = wp4 + (size_t)w * n_kg * 64 + lane;
  const f32x4* wp1 = wp4 + (size_t)(w + NWAVE) * n_kg * 64 + lane;
  if (u0 && u1) gemm_loop<true, true>(acc, wp0, wp1, n_kg, xs4, lane);
  else if (u0) gemm_loop<true, false>(acc, wp0, wp1, n_kg, xs4, lane);
}

template <int MODE, bool LN>
__global__ __launch_bounds__(512, 2) void k_decoder_any(const AnyArgs a) {
  __shared__ f32x4 xs4[(ANY_W / 4) * TQ];   // 128 KiB
  __shared__ float red[2 * NWAVE * 64];
  __shared__ float gxs[3 * 64];             // d sdf / d xyz, accumulated over the layers that see xyz
  __shared__ float ys[64];
  float* xs = reinterpret_cast<float*>(xs4);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int qa = lane & 31, hi = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = a.dec.L, D0 = L + 3, n_lin = a.dec.n_lin;
  const f32x4* pts4 = reinterpret_cast<const f32x4*>(a.pts);
  float* slab = a.ln_slab + (size_t)blockIdx.x * a.dec.n_ln * SLAB;
  // d sdf / d z is the sum of the transposed latent columns of EVERY layer that sees z (lin0 and the latent_in layers);
  // the contributions land in different (wave, lane) positions, so they meet in a query-minor scratch block of this
  // workgroup (coalesced adds, L2-resident) and are transposed into the Jacobian rows at the end of the tile
  float* gi = a.gi_slab + (size_t)blockIdx.x * (MAX_L * 64);

  for (int t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
    // same tile -> (instance, first query) map as k_decoder: tile-major, instance rotated by the tile index
    const int b = (t + t / a.B) % a.B;
    const int q0 = (t / a.B) * TQ;
    if (a.active != nullptr && a.active[b] == 0) continue;
    const int nq = a.n_q[b];
    if (q0 >= nq) continue;
    const int cnt = (nq - q0 < TQ) ? nq - q0 : TQ;
    const size_t qbase = (size_t)b * a.n_stride + q0;
    const f32x4 p = pts4[qbase + lane];      // lane = query: input rows, concatenated rows, pose chain rule
    const float* z = a.zc + (size_t)b * HID;

    __syncthreads();   // the previous tile is done with X / gxs / ys
    // input rows: [z (L) | xyz (3)], padded to a K group of 8 with zeros (deep_sdf_decoder.py:76-83, eval mode)
    for (int j = w; j < ((D0 + 7) & ~7); j += NWAVE)
      xs[xidx(j, lane)] = j < L ? z[j] : (j < D0 ? p[j - L] : 0.f);
    if (MODE == 1 && tid < 3 * 64) gxs[tid] = 0.f;
    bool gi_first = true;

    // ReLU masks (64 bits per thread and layer): through the launch's scratch block -- one coalesced 8-byte store per layer in
    // the forward pass, one load per layer in the backward pass -- instead of a 16-entry array indexed by the layer loop's
    // counter (32 registers, or scratch memory: what the backward instantiations spilled; round 6).  Forward-only launches
    // (MODE 0) have no block and store nothing.
    unsigned long long* const mkp = MODE == 1 ? a.mk_slab + ((size_t)blockIdx.x * HM_ANY_MAX_LIN) * 512 + tid : nullptr;
    f32x16 acc[2][2];
    float yA = 0.f, yB = 0.f, tA = 0.f, tB = 0.f;   // sdf / inner tanh of queries qa and qa + 32 (wave 0, hi = 0)
    int ln_i = 0;

    // ---------------- forward (deep_sdf_decoder.py:85-110) ----------------
    for (int l = 0; l < n_lin; ++l) {
      const AnyLayer& ly = a.dec.lay[l];
      const int n_rb = (ly.out_dim + 31) >> 5;
      __syncthreads();   // X of this layer complete
      run_gemm(acc, ly.wf, (ly.in_dim + 7) >> 3, n_rb, w, xs4, lane);
      __syncthreads();   // every wave is done reading X

      if (l == n_lin - 1) {        // :93-94 optional tanh, :107-108 the final tanh (always)
        if (w == 0 && hi == 0) {
          const float preA = acc[0][0][0] + ly.bias[0], preB = acc[0][1][0] + ly.bias[0];
          tA = a.dec.use_tanh ? tanhf(preA) : preA;
          tB = a.dec.use_tanh ? tanhf(preB) : preB;
          yA = tanhf(tA); yB = tanhf(tB);
          ys[qa] = yA; ys[32 + qa] = yB;
          if (qa < cnt) a.y[qbase + qa] = yA;
          if (32 + qa < cnt) a.y[qbase + 32 + qa] = yB;
        }
        break;
      }
      // bias (+ LayerNorm :96-101) + ReLU :102
      float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f};
      if (LN && ly.ln) {
        float s[2] = {0.f, 0.f};
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          const int mb = w + sl * NWAVE;
          if (mb >= n_rb) continue;
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int r = mb * 32 + 8 * g + 4 * hi + j;
              const float bv = ly.bias[r];
#pragma unroll
              for (int nb = 0; nb < 2; ++nb) {
                acc[sl][nb][4 * g + j] += bv;
                if (r < ly.out_dim) s[nb] += acc[sl][nb][4 * g + j];
              }
            }
        }
        query_sum2(s[0], s[1], red, w, lane, mean[0], mean[1]);
        mean[0] /= (float)ly.out_dim; mean[1] /= (float)ly.out_dim;
        float s2[2] = {0.f, 0.f};
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          const int mb = w + sl * NWAVE;
          if (mb >= n_rb) continue;
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int r = mb * 32 + 8 * g + 4 * hi + j;
              if (r < ly.out_dim) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                  const float d = acc[sl][nb][4 * g + j] - mean[nb];
                  s2[nb] = fmaf(d, d, s2[nb]);
                }
              }
            }
        }
        query_sum2(s2[0], s2[1], red, w, lane, rstd[0], rstd[1]);
        rstd[0] = 1.f / sqrtf(rstd[0] / (float)ly.out_dim + 1e-5f);   // nn.LayerNorm default eps, biased variance
        rstd[1] = 1.f / sqrtf(rstd[1] / (float)ly.out_dim + 1e-5f);
      }
      float* xh = (LN && MODE == 1 && ly.ln) ? slab + (size_t)ln_i * SLAB : nullptr;
      if (xh != nullptr && w == 0 && hi == 0) { xh[ANY_W * 64 + qa] = rstd[0]; xh[ANY_W * 64 + 32 + qa] = rstd[1]; }
      unsigned long long bits = 0;
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const int mb = w + sl * NWAVE;
        if (mb >= n_rb) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int f4 = mb * 32 + 8 * g + 4 * hi;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int r = f4 + j;
              float val;
              if (LN && ly.ln) {
                const float h = (acc[sl][nb][4 * g + j] - mean[nb]) * rstd[nb];
                if (xh != nullptr) xh[r * 64 + nb * 32 + qa] = h;
                val = fmaf(h, ly.gamma[r], ly.beta[r]);
              } else {
                val = acc[sl][nb][4 * g + j] + ly.bias[r];
              }
              const bool pos = (val > 0.f) && (r < ly.out_dim);
              bits |= (pos ? 1ull : 0ull) << (sl * 32 + nb * 16 + 4 * g + j);
              v[j] = pos ? val : 0.f;
            }
            xs4[(f4 >> 2) * TQ + nb * 32 + qa] = v;
          }
        }
      }
      if (MODE == 1) mkp[(size_t)l * 512] = bits;
      if (LN && ly.ln) ++ln_i;
      const int cat = a.dec.lay[l + 1].cat;
      if (cat != 0) {    // :87-90 x = cat[x, input] (latent_in) or cat[x, xyz] (xyz_in_all)
        __syncthreads();
        const int wcat = cat == 1 ? D0 : 3;
        for (int j = w; j < ((ly.out_dim + wcat + 7) & ~7) - ly.out_dim; j += NWAVE) {
          float v = 0.f;
          if (j < wcat) v = cat == 1 ? (j < L ? z[j] : p[j - L]) : p[j];
          xs[xidx(ly.out_dim + j, lane)] = v;
        }
      }
    }
    if (MODE == 0) continue;

    // ---------------- backward: d sdf / d input (utils.py:112-122 restated) ----------------
    __syncthreads();
    if (w == 0 && hi == 0) {     // gradient w.r.t. the single output row of the last layer; rows 1..7 of the K group: 0
      float dA = 1.f - yA * yA, dB = 1.f - yB * yB;
      if (a.dec.use_tanh) { dA *= 1.f - tA * tA; dB *= 1.f - tB * tB; }
      xs4[qa] = f32x4{dA, 0.f, 0.f, 0.f};
      xs4[32 + qa] = f32x4{dB, 0.f, 0.f, 0.f};
      xs4[TQ + qa] = f32x4{0.f, 0.f, 0.f, 0.f};
      xs4[TQ + 32 + qa] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int l = n_lin - 1; l >= 0; --l) {
      const AnyLayer& ly = a.dec.lay[l];
      const int n_rb = (ly.in_dim + 31) >> 5;
      __syncthreads();
      run_gemm(acc, ly.wb, (ly.out_dim + 7) >> 3, n_rb, w, xs4, lane);
      __syncthreads();
      // columns [base, in_dim) of this layer are the concatenated input (for l = 0 everything is input)
      const int wcat = l == 0 ? ly.in_dim : (ly.cat == 1 ? D0 : (ly.cat == 2 ? 3 : 0));
      const int base = ly.in_dim - wcat;
      const int joff = (l == 0 || ly.cat == 1) ? 0 : L;        // first input index the concatenated part maps to
      if (wcat > 0) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          const int mb = w + sl * NWAVE;
          if (mb >= n_rb || mb * 32 + 32 <= base) continue;
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int col = mb * 32 + 8 * g + 4 * hi + j;
              if (col >= base && col < ly.in_dim) {
                const int ji = joff + col - base;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                  const int q = nb * 32 + qa;
                  if (ji >= L) gxs[(ji - L) * 64 + q] += acc[sl][nb][4 * g + j];
                  else if (gi_first) gi[ji * 64 + q] = acc[sl][nb][4 * g + j];
                  else gi[ji * 64 + q] += acc[sl][nb][4 * g + j];
                }
              }
            }
        }
      }
      if (wcat > 3) gi_first = false;     // (every contribution that carries z covers all L columns)
      if (l == 0) break;
      // gradient w.r.t. the pre-activation of layer l - 1: ReLU mask, then LayerNorm backward
      const AnyLayer& lp = a.dec.lay[l - 1];
      const unsigned long long bits = mkp[(size_t)(l - 1) * 512];
#pragma unroll
      for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            acc[sl][nb][i] = ((bits >> (sl * 32 + nb * 16 + i)) & 1ull) ? acc[sl][nb][i] : 0.f;
      if (LN && lp.ln) {
        --ln_i;
        const float* xh = slab + (size_t)ln_i * SLAB;
        const float rs[2] = {xh[ANY_W * 64 + qa], xh[ANY_W * 64 + 32 + qa]};
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          const int mb = w + sl * NWAVE;
          if (mb >= n_rb) continue;
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int r = mb * 32 + 8 * g + 4 * hi + j;
              if (r < lp.out_dim) {
                const float gm = lp.gamma[r];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                  acc[sl][nb][4 * g + j] *= gm;
                  s1[nb] += acc[sl][nb][4 * g + j];
                  s2[nb] = fmaf(acc[sl][nb][4 * g + j], xh[r * 64 + nb * 32 + qa], s2[nb]);
                }
              }
            }
        }
        float m1[2], m2[2];
        query_sum2(s1[0], s1[1], red, w, lane, m1[0], m1[1]);
        query_sum2(s2[0], s2[1], red, w, lane, m2[0], m2[1]);
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          const int mb = w + sl * NWAVE;
          if (mb >= n_rb) continue;
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int r = mb * 32 + 8 * g + 4 * hi + j;
              if (r < lp.out_dim) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                  acc[sl][nb][4 * g + j] = rs[nb] * (acc[sl][nb][4 * g + j] - m1[nb] / (float)lp.out_dim -
                                                     xh[r * 64 + nb * 32 + qa] * (m2[nb] / (float)lp.out_dim));
              }
            }
        }
      }
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const int mb = w + sl * NWAVE;
        if (mb >= n_rb) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int f4 = mb * 32 + 8 * g + 4 * hi;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (f4 + j < lp.out_dim) ? acc[sl][nb][4 * g + j] : 0.f;
            xs4[(f4 >> 2) * TQ + nb * 32 + qa] = v;
          }
        }
      }
    }
    __syncthreads();
    for (int q = w; q < cnt; q += NWAVE) {           // transpose: Jacobian row of query q <- column q of the scratch block
      float* row = a.J + (qbase + q) * (size_t)a.ldJ;
      for (int j = lane; j < L; j += 64) row[j] = gi[j * 64 + q];
    }
    // pose chain rule  J_pose = g_x [ I | -[p]x | p ]  (loss.py:236-239, utils.py:197-217,257-276); column L + 7 = sdf
    if (w == 0 && lane < cnt) {
      const float g0 = gxs[lane], g1 = gxs[64 + lane], g2 = gxs[128 + lane];
      float* row = a.J + (qbase + lane) * (size_t)a.ldJ + L;
      row[7] = ys[lane];
      row[0] = g0; row[1] = g1; row[2] = g2;
      if (a.pose_dim != 0) {
        row[3] = g2 * p[1] - g1 * p[2];
        row[4] = g0 * p[2] - g2 * p[0];
        row[5] = g1 * p[0] - g0 * p[1];
        if (a.pose_dim == 7) row[6] = g0 * p[0] + g1 * p[1] + g2 * p[2];
      }
    }
  }
}

// ---- f16x3 variant (hm_decoder_set_precision 1): the same kernel on the fp16 matrix cores with hi / lo split operands ----
// (Deliberately a second copy of the kernel text rather than one kernel templated on the arithmetic: the unified form -- a
// Planes<H> policy for layout / store / K loop, 270 lines shorter -- was built and measured on the same tests: identical
// results, but hipcc's register allocation of the f16x3 instance got worse (13 -> 55 spilled registers in the backward
// epilogues) and the shipped table ran 306-311 instead of 351-364 TFLOP/s.  Keep the two in step by hand.)
// element (k, q) of the two fp16 activation planes X[k/8][q][8] (hi, and lo scaled by 2^11)
__device__ __forceinline__ void put_h(_Float16* xhs, _Float16* xls, int k, int q, float v) {
  const int i = (((k >> 3) * TQ + q) << 3) + (k & 7);
  const _Float16 h = (_Float16)v;
  xhs[i] = h;
  xls[i] = (_Float16)((v - (float)h) * LO_SCALE);
}

__device__ __forceinline__ void run_gemm_h(f32x16 (&acc)[2][2], const void* wp, int n_k16, int n_rb, int w, const f16x8* xh,
                                           const f16x8* xl, int lane) {
  acc[0][0] = zero16(); acc[0][1] = zero16(); acc[1][0] = zero16(); acc[1][1] = zero16();
  const bool u0 = w < n_rb, u1 = w + NWAVE < n_rb;
  const WSrc wp0 = make_wsrc(wp, w * n_k16 * 128 + lane);
  const WSrc wp1 = make_wsrc(wp, (w + NWAVE) * n_k16 * 128 + lane);
  if (u0 && u1) gemm_loop_h<true, true>(acc, wp0, wp1, n_k16, xh, xl, lane);
  else if (u0) gemm_loop_h<true, false>(acc, wp0, wp1, n_k16, xh, xl, lane);
}

template <int MODE, bool LN>
__global__ __launch_bounds__(512, 2) void k_decoder_any_h(const AnyArgs a) {
  __shared__ f16x8 xh[64 * TQ];             // 64 KiB: hi plane  X[k/8][q][8]
  __shared__ f16x8 xl[64 * TQ];             // 64 KiB: lo plane (scaled by 2^11)
  __shared__ int ovf;                       // some lane of this tile stored a value beyond the fp16 range
  __shared__ float red[2 * NWAVE * 64];
  __shared__ float gxs[3 * 64];             // d sdf / d xyz, accumulated over the layers that see xyz
  __shared__ float ys[64];
  f16x4* xh4 = reinterpret_cast<f16x4*>(xh);
  f16x4* xl4 = reinterpret_cast<f16x4*>(xl);
  _Float16* xhs = reinterpret_cast<_Float16*>(xh);
  _Float16* xls = reinterpret_cast<_Float16*>(xl);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int qa = lane & 31, hi = lane >> 5;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = a.dec.L, D0 = L + 3, n_lin = a.dec.n_lin;
  const f32x4* pts4 = reinterpret_cast<const f32x4*>(a.pts);
  float* slab = a.ln_slab + (size_t)blockIdx.x * a.dec.n_ln * SLAB;
  // d sdf / d z is the sum of the transposed latent columns of EVERY layer that sees z (lin0 and the latent_in layers);
  // the contributions land in different (wave, lane) positions, so they meet in a query-minor scratch block of this
  // workgroup (coalesced adds, L2-resident) and are transposed into the Jacobian rows at the end of the tile
  float* gi = a.gi_slab + (size_t)blockIdx.x * (MAX_L * 64);

  for (int t = blockIdx.x; t < a.n_tiles; t += gridDim.x) {
    // same tile -> (instance, first query) map as k_decoder: tile-major, instance rotated by the tile index
    const int b = (t + t / a.B) % a.B;
    const int q0 = (t / a.B) * TQ;
    if (a.active != nullptr && a.active[b] == 0) continue;
    const int nq = a.n_q[b];
    if (q0 >= nq) continue;
    const int cnt = (nq - q0 < TQ) ? nq - q0 : TQ;
    const size_t qbase = (size_t)b * a.n_stride + q0;
    const f32x4 p = pts4[qbase + lane];      // lane = query: input rows, concatenated rows, pose chain rule
    const float* z = a.zc + (size_t)b * HID;

    __syncthreads();   // the previous tile is done with X / gxs / ys
    // input rows: [z (L) | xyz (3)], padded to a K group of 8 with zeros (deep_sdf_decoder.py:76-83, eval mode)
    for (int j = w; j < ((D0 + 15) & ~15); j += NWAVE)
      put_h(xhs, xls, j, lane, j < L ? z[j] : (j < D0 ? p[j - L] : 0.f));
    if (tid == 0) ovf = 0;
    f16x2 xm2 = {(_Float16)0.f, (_Float16)0.f};   // range guard: running max of |hi| this lane stored (hm_gemm_h.h: track_max)
    if (MODE == 1 && tid < 3 * 64) gxs[tid] = 0.f;
    bool gi_first = true;

    // ReLU masks (64 bits per thread and layer): through the launch's scratch block -- one coalesced 8-byte store per layer in
    // the forward pass, one load per layer in the backward pass -- instead of a 16-entry array indexed by the layer loop's
    // counter (32 registers, or scratch memory: what the backward instantiations spilled; round 6).  Forward-only launches
    // (MODE 0) have no block and store nothing.
    unsigned long long* const mkp = MODE == 1 ? a.mk_slab + ((size_t)blockIdx.x * HM_ANY_MAX_LIN) * 512 + tid : nullptr;
    f32x16 acc[2][2];
    float yA = 0.f, yB = 0.f, tA = 0.f, tB = 0.f;   // sdf / inner tanh of queries qa and qa + 32 (wave 0, hi = 0)
    int ln_i = 0;

    // ---------------- forward (deep_sdf_decoder.py:85-110) ----------------
    for (int l = 0; l < n_lin; ++l) {
      const AnyLayer& ly = a.dec.lay[l];
      const int n_rb = (ly.out_dim + 31) >> 5;
      __syncthreads();   // X of this layer complete
      if (l == n_lin - 1 && __any(!(fmaxf((float)xm2[0], (float)xm2[1]) < 65504.f))) ovf = 1;
      run_gemm_h(acc, ly.wfh, ly.kf16, n_rb, w, xh, xl, lane);
      const float us = ly.usf;
      __syncthreads();   // every wave is done reading X

      if (l == n_lin - 1) {        // :93-94 optional tanh, :107-108 the final tanh (always)
        if (w == 0 && hi == 0) {
          // a tile whose activations left the fp16 range is poisoned (NaN sdf, NaN Jacobian rows through the seed below),
          // never silent garbage: the policy of hm_decoder_h.hip
          const float poison = ovf ? __builtin_nanf("") : 0.f;
          const float preA = fmaf(acc[0][0][0], us, ly.bias[0]) + poison, preB = fmaf(acc[0][1][0], us, ly.bias[0]) + poison;
          tA = a.dec.use_tanh ? tanhf(preA) : preA;
          tB = a.dec.use_tanh ? tanhf(preB) : preB;
          yA = tanhf(tA); yB = tanhf(tB);
          ys[qa] = yA; ys[32 + qa] = yB;
          if (qa < cnt) a.y[qbase + qa] = yA;
          if (32 + qa < cnt) a.y[qbase + 32 + qa] = yB;
        }
        break;
      }
      // bias (+ LayerNorm :96-101) + ReLU :102
      float mean[2] = {0.f, 0.f}, rstd[2] = {1.f, 1.f};
      if (LN && ly.ln) {
        float s[2] = {0.f, 0.f};
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          const int mb = w + sl * NWAVE;
          if (mb >= n_rb) continue;
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int r = mb * 32 + 8 * g + 4 * hi + j;
              const float bv = ly.bias[r];
#pragma unroll
              for (int nb = 0; nb < 2; ++nb) {
                acc[sl][nb][4 * g + j] = fmaf(acc[sl][nb][4 * g + j], us, bv);
                if (r < ly.out_dim) s[nb] += acc[sl][nb][4 * g + j];
              }
            }
        }
        query_sum2(s[0], s[1], red, w, lane, mean[0], mean[1]);
        mean[0] /= (float)ly.out_dim; mean[1] /= (float)ly.out_dim;
        float s2[2] = {0.f, 0.f};
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          const int mb = w + sl * NWAVE;
          if (mb >= n_rb) continue;
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int r = mb * 32 + 8 * g + 4 * hi + j;
              if (r < ly.out_dim) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                  const float d = acc[sl][nb][4 * g + j] - mean[nb];
                  s2[nb] = fmaf(d, d, s2[nb]);
                }
              }
            }
        }
        query_sum2(s2[0], s2[1], red, w, lane, rstd[0], rstd[1]);
        rstd[0] = 1.f / sqrtf(rstd[0] / (float)ly.out_dim + 1e-5f);   // nn.LayerNorm default eps, biased variance
        rstd[1] = 1.f / sqrtf(rstd[1] / (float)ly.out_dim + 1e-5f);
      }
      float* xh = (LN && MODE == 1 && ly.ln) ? slab + (size_t)ln_i * SLAB : nullptr;
      if (xh != nullptr && w == 0 && hi == 0) { xh[ANY_W * 64 + qa] = rstd[0]; xh[ANY_W * 64 + 32 + qa] = rstd[1]; }
      unsigned long long bits = 0;
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const int mb = w + sl * NWAVE;
        if (mb >= n_rb) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int f4 = mb * 32 + 8 * g + 4 * hi;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int r = f4 + j;
              float val;
              if (LN && ly.ln) {
                const float h = (acc[sl][nb][4 * g + j] - mean[nb]) * rstd[nb];
                if (xh != nullptr) xh[r * 64 + nb * 32 + qa] = h;
                val = fmaf(h, ly.gamma[r], ly.beta[r]);
              } else {
                val = fmaf(acc[sl][nb][4 * g + j], us, ly.bias[r]);
              }
              const bool pos = (val > 0.f) && (r < ly.out_dim);
              bits |= (pos ? 1ull : 0ull) << (sl * 32 + nb * 16 + 4 * g + j);
              v[j] = pos ? val : 0.f;
            }
            split_store<false>(xh4, xl4, ((f4 >> 3) * TQ + nb * 32 + qa) * 2 + hi, f32x2{v[0], v[1]}, f32x2{v[2], v[3]}, xm2);
          }
        }
      }
      if (MODE == 1) mkp[(size_t)l * 512] = bits;
      if (LN && ly.ln) ++ln_i;
      const int cat = a.dec.lay[l + 1].cat;
      if (cat != 0) {    // :87-90 x = cat[x, input] (latent_in) or cat[x, xyz] (xyz_in_all)
        __syncthreads();
        const int wcat = cat == 1 ? D0 : 3;
        for (int j = w; j < ((ly.out_dim + wcat + 15) & ~15) - ly.out_dim; j += NWAVE) {
          float v = 0.f;
          if (j < wcat) v = cat == 1 ? (j < L ? z[j] : p[j - L]) : p[j];
          put_h(xhs, xls, ly.out_dim + j, lane, v);
        }
      }
    }
    if (MODE == 0) continue;

    // ---------------- backward: d sdf / d input (utils.py:112-122 restated) ----------------
    __syncthreads();
    if (w == 0 && hi == 0) {     // gradient w.r.t. the single output row of the last layer; rows 1..7 of the K group: 0
      float dA = 1.f - yA * yA, dB = 1.f - yB * yB;
      if (a.dec.use_tanh) { dA *= 1.f - tA * tA; dB *= 1.f - tB * tB; }
      const f32x2 zz = {0.f, 0.f};      // rows 1..15 of the K step: 0
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        split_store<true>(xh4, xl4, ((u >> 1) * TQ + qa) * 2 + (u & 1), u == 0 ? f32x2{dA, 0.f} : zz, zz, xm2);
        split_store<true>(xh4, xl4, ((u >> 1) * TQ + 32 + qa) * 2 + (u & 1), u == 0 ? f32x2{dB, 0.f} : zz, zz, xm2);
      }
    }
    for (int l = n_lin - 1; l >= 0; --l) {
      const AnyLayer& ly = a.dec.lay[l];
      const int n_rb = (ly.in_dim + 31) >> 5;
      __syncthreads();
      run_gemm_h(acc, ly.wbh, ly.kb16, n_rb, w, xh, xl, lane);
      {
        const float usb = ly.usb;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[sl][nb] *= usb;
      }
      __syncthreads();
      // columns [base, in_dim) of this layer are the concatenated input (for l = 0 everything is input)
      const int wcat = l == 0 ? ly.in_dim : (ly.cat == 1 ? D0 : (ly.cat == 2 ? 3 : 0));
      const int base = ly.in_dim - wcat;
      const int joff = (l == 0 || ly.cat == 1) ? 0 : L;        // first input index the concatenated part maps to
      if (wcat > 0) {
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          const int mb = w + sl * NWAVE;
          if (mb >= n_rb || mb * 32 + 32 <= base) continue;
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int col = mb * 32 + 8 * g + 4 * hi + j;
              if (col >= base && col < ly.in_dim) {
                const int ji = joff + col - base;
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                  const int q = nb * 32 + qa;
                  if (ji >= L) gxs[(ji - L) * 64 + q] += acc[sl][nb][4 * g + j];
                  else if (gi_first) gi[ji * 64 + q] = acc[sl][nb][4 * g + j];
                  else gi[ji * 64 + q] += acc[sl][nb][4 * g + j];
                }
              }
            }
        }
      }
      if (wcat > 3) gi_first = false;     // (every contribution that carries z covers all L columns)
      if (l == 0) break;
      // gradient w.r.t. the pre-activation of layer l - 1: ReLU mask, then LayerNorm backward
      const AnyLayer& lp = a.dec.lay[l - 1];
      const unsigned long long bits = mkp[(size_t)(l - 1) * 512];
#pragma unroll
      for (int sl = 0; sl < 2; ++sl)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int i = 0; i < 16; ++i)
            acc[sl][nb][i] = ((bits >> (sl * 32 + nb * 16 + i)) & 1ull) ? acc[sl][nb][i] : 0.f;
      if (LN && lp.ln) {
        --ln_i;
        const float* xh = slab + (size_t)ln_i * SLAB;
        const float rs[2] = {xh[ANY_W * 64 + qa], xh[ANY_W * 64 + 32 + qa]};
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          const int mb = w + sl * NWAVE;
          if (mb >= n_rb) continue;
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int r = mb * 32 + 8 * g + 4 * hi + j;
              if (r < lp.out_dim) {
                const float gm = lp.gamma[r];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                  acc[sl][nb][4 * g + j] *= gm;
                  s1[nb] += acc[sl][nb][4 * g + j];
                  s2[nb] = fmaf(acc[sl][nb][4 * g + j], xh[r * 64 + nb * 32 + qa], s2[nb]);
                }
              }
            }
        }
        float m1[2], m2[2];
        query_sum2(s1[0], s1[1], red, w, lane, m1[0], m1[1]);
        query_sum2(s2[0], s2[1], red, w, lane, m2[0], m2[1]);
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          const int mb = w + sl * NWAVE;
          if (mb >= n_rb) continue;
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int r = mb * 32 + 8 * g + 4 * hi + j;
              if (r < lp.out_dim) {
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
                  acc[sl][nb][4 * g + j] = rs[nb] * (acc[sl][nb][4 * g + j] - m1[nb] / (float)lp.out_dim -
                                                     xh[r * 64 + nb * 32 + qa] * (m2[nb] / (float)lp.out_dim));
              }
            }
        }
      }
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const int mb = w + sl * NWAVE;
        if (mb >= n_rb) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int f4 = mb * 32 + 8 * g + 4 * hi;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (f4 + j < lp.out_dim) ? acc[sl][nb][4 * g + j] : 0.f;
            split_store<true>(xh4, xl4, ((f4 >> 3) * TQ + nb * 32 + qa) * 2 + hi, f32x2{v[0], v[1]}, f32x2{v[2], v[3]}, xm2);
          }
        }
      }
    }
    if (__any(!(fmaxf((float)xm2[0], (float)xm2[1]) < 65504.f))) ovf = 1;     // a back-propagated gradient left the fp16 range
    __syncthreads();
    const float poison = ovf ? __builtin_nanf("") : 0.f;
    for (int q = w; q < cnt; q += NWAVE) {           // transpose: Jacobian row of query q <- column q of the scratch block
      float* row = a.J + (qbase + q) * (size_t)a.ldJ;
      for (int j = lane; j < L; j += 64) row[j] = gi[j * 64 + q] + poison;
    }
    // pose chain rule  J_pose = g_x [ I | -[p]x | p ]  (loss.py:236-239, utils.py:197-217,257-276); column L + 7 = sdf
    if (w == 0 && lane < cnt) {
      const float g0 = gxs[lane] + poison, g1 = gxs[64 + lane] + poison, g2 = gxs[128 + lane] + poison;
      float* row = a.J + (qbase + lane) * (size_t)a.ldJ + L;
      row[7] = ys[lane];
      row[0] = g0; row[1] = g1; row[2] = g2;
      if (a.pose_dim != 0) {
        row[3] = g2 * p[1] - g1 * p[2];
        row[4] = g0 * p[2] - g2 * p[0];
        row[5] = g1 * p[0] - g0 * p[1];
        if (a.pose_dim == 7) row[6] = g0 * p[0] + g1 * p[1] + g2 * p[2];
      }
    }
  }
}

// the generic decoder has no per-instance folded biases: its "c0" is the latent itself
__global__ void k_latent_copy(const float* __restrict__ latent, int ld_latent, const int* __restrict__ active, int L,
                              float* __restrict__ zc) {
  const int b = blockIdx.x;
  if (active != nullptr && active[b] == 0) return;
  for (int i = threadIdx.x; i < L; i += blockDim.x) zc[(size_t)b * HID + i] = latent[(size_t)b * ld_latent + i];
}

}  // namespace

namespace hm {

int launch_latent_copy_any(const hm_decoder_s* dec, const float* d_latent, int ld_latent, const int* d_active, int B,
                           float* d_zc, hipStream_t stream) {
  hipLaunchKernelGGL(k_latent_copy, dim3(B), dim3(256), 0, stream, d_latent, ld_latent, d_active, dec->L, d_zc);
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_decoder_any(const hm_decoder_s* dec, int B, const float* d_pts, const int* d_nq, const int* d_active,
                       int n_stride, const float* d_zc, float* d_y, float* d_J, int ldJ, int pose_dim, int mode,
                       hipStream_t stream) {
  AnyArgs a;
  a.dec = dec->any;
  a.pts = d_pts; a.n_q = d_nq; a.active = d_active; a.zc = d_zc; a.y = d_y; a.J = d_J;
  a.n_stride = n_stride; a.B = B; a.ldJ = ldJ; a.pose_dim = pose_dim;
  a.n_tiles = B * (n_stride / TQ);
  if (a.n_tiles == 0) return 0;
  const int grid = a.n_tiles < ANY_GRID ? a.n_tiles : ANY_GRID;
  // The per-workgroup scratch of the backward pass (LayerNorm saves, the d sdf / d z block, the ReLU masks) belongs to the
  // launch's STREAM (hm_internal.h: scratch_get).  (Up to round 5 it was one block per decoder handle, indexed by
  // blockIdx.x only: the two instance groups of hm_optimize_batch, two workspaces or two host threads on one handle
  // then read-modify-wrote each other's slots.)  The forward-only kernels do not touch it.
  if (dec->precision != 0 && dec->precision != 1) {
    hm_set_error("any-architecture decoder: precision %d not available (0 = exact fp32, 1 = f16x3)", dec->precision);
    return -1;
  }
  void* slab = nullptr;
  if (mode != 0) {
    const int rc = scratch_get(&slab, (size_t)grid * (((size_t)dec->any.n_ln * SLAB + MAX_L * 64) * sizeof(float) +
                                                        (size_t)HM_ANY_MAX_LIN * 512 * sizeof(unsigned long long)), stream);
    if (rc) return rc;
  }
  a.ln_slab = static_cast<float*>(slab);
  a.gi_slab = a.ln_slab + (size_t)grid * dec->any.n_ln * SLAB;
  a.mk_slab = reinterpret_cast<unsigned long long*>(a.gi_slab + (size_t)grid * MAX_L * 64);
  const bool ln = dec->any.n_ln > 0;      // tables without LayerNorm run kernels compiled without its code paths
#define HM_ANY_LAUNCH(K, M)                                                                     \
  do {                                                                                          \
    if (ln) hipLaunchKernelGGL((K<M, true>), dim3(grid), dim3(512), 0, stream, a);              \
    else hipLaunchKernelGGL((K<M, false>), dim3(grid), dim3(512), 0, stream, a);                \
  } while (0)
  if (dec->precision == 1) {
    if (mode == 0) HM_ANY_LAUNCH(k_decoder_any_h, 0); else HM_ANY_LAUNCH(k_decoder_any_h, 1);
  } else {
    if (mode == 0) HM_ANY_LAUNCH(k_decoder_any, 0); else HM_ANY_LAUNCH(k_decoder_any, 1);
  }
#undef HM_ANY_LAUNCH
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hm
