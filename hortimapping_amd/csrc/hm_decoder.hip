// K1: fused DeepSDF decoder forward + input-gradient backward on CDNA4 (gfx950), exact fp32 on the
// f32-input matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces, per tile of 64 queries of one fruit instance (reference paths relative to /root/reference):
//   Decoder.forward                     deepsdf/networks/deep_sdf_decoder.py:75-110
//   decode_sdf                          wild_completion/utils.py:144-172          (mode 0)
//   get_batch_sdf_jacobian/get_gradient wild_completion/utils.py:112-122,175-193  (mode 1)
//   pose chain rule of compute_sdf_loss wild_completion/loss.py:229-241 (+ utils.py:197-217,257-276)
//
// Layout.  A workgroup (8 waves) owns 64 queries.  Activations live in LDS as X[k/4][q][k%4] (128 KiB);
// every wave owns output rows {32w..32w+31} and {32(w+8)..} of each layer, reads the whole X as the MFMA
// B operand (ds_read_b128, conflict free) and streams its own slice of the pre-packed weights straight
// from L2 into VGPRs as the A operand (global_load_dwordx4, fully coalesced; weights are shared by all
// instances and never staged through LDS because no two waves of a workgroup need the same element).
// The accumulator fragment of one layer is written back (bias/ReLU or ReLU-mask applied) as the next
// layer's X; ReLU masks are kept as bitmasks in registers (64 bits per layer per lane) because forward
// layer l and backward layer l+1 assign the same (row, query) to the same (wave, lane, register).
// The latent part of lin0/lin4 is folded into a per-instance bias (c0, c4) for the forward pass; the
// backward pass still produces the per-query d sdf / d z through the transposed latent columns.
#include "hm_common.h"
#include "hm_internal.h"
#include "hm_gemm_f32.h"

using namespace hm;

struct DecodeArgs {
  DecoderDev dec;
  const float* pts;    // [B][n_stride][4]  object-frame query points (xyz, pad)
  const int* n_q;      // [B] valid queries per instance (device memory)
  const int* active;   // [B] or nullptr
  const float* c0;     // [B][512] per-instance lin0 bias  (W0[:, :L] z + b0)
  const float* c4;     // [B][512] per-instance lin4 bias  (W4[:, m:m+L] z + b4)
  float* y;            // [B][n_stride] sdf
  float* J;            // [B][n_stride][ldJ] rows [d sdf/d z (L) | d sdf/d pose (P) | pad]
  int n_stride;
  int B;
  int ldJ;
  int pose_dim;        // 0 (shape only), 6 (SE3), 7 (Sim3)
  int mode;            // 0 forward only, 1 forward + backward
};

#define HM_MASK_CASES(OP) \
  case 0: OP(mk0); break; case 1: OP(mk1); break; case 2: OP(mk2); break; case 3: OP(mk3); break; \
  case 4: OP(mk4); break; case 5: OP(mk5); break; case 6: OP(mk6); break; default: OP(mk7); break;

// MODE 0: forward only (decode_sdf).  MODE 1: forward + input-gradient backward.  TAG only gives the two
// fwd+bwd call sites of an LM iteration (0: SDF-term rows, 1: render-term samples) distinct kernel names so that
// rocprofv3 --stats reports them separately; the code is identical.
template <int MODE, int TAG>
__global__ __launch_bounds__(512, 2) void k_decoder(const DecodeArgs a) {
  __shared__ f32x4 xs[128 * TQ];   // 128 KiB: X[k/4][q] as float4 over k%4
  __shared__ float sc[2048 + 128];
  __shared__ float bl[8 * HID];    // biases of the 8 forward stages (per-instance c0/c4 included), staged once per tile // partial sums of the VALU side paths

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // tile-major block order, instance rotated by the tile index: blockIdx = tile * B + (b - tile) mod B.  Real tiles of a
  // ragged launch (render samples: a few tiles per instance, the rest of the capacity empty) are then contiguous in
  // blockIdx and spread round-robin over all 8 XCDs; instance-major order put them all on XCDs 0..2 (block i runs on XCD
  // i % 8) and quadrupled the launch time.  The rotation (round 4) spreads the tiles of ONE instance over the XCDs as
  // well: without it instance b sits on XCD b % 8 whenever B % 8 == 0, and a batch in which most instances have already
  // converged (early exits on: the shipped configurations) ran its few active instances on one or two XCDs -- the main
  // launch of wild_pepper.yaml took 51 ms with 4 of 64 instances active against 60 ms with all of them.
  const int b = (blockIdx.x + blockIdx.x / a.B) % a.B;
  const int q0 = (blockIdx.x / a.B) * TQ;
  if (a.active != nullptr && a.active[b] == 0) return;
  const int nq = a.n_q[b];
  if (q0 >= nq) return;
  const int cnt = (nq - q0 < TQ) ? nq - q0 : TQ;

  const int L = a.dec.L, m = a.dec.m, mb_zx = a.dec.mb_zx;
  const size_t qbase = (size_t)b * a.n_stride + q0;
  const int qa = lane & 31;     // this lane's MFMA columns: qa (nb = 0) and qa + 32 (nb = 1)
  const int hi = lane >> 5;
  const f32x4* pts4 = reinterpret_cast<const f32x4*>(a.pts);
  const f32x4 pA = pts4[qbase + qa];
  const f32x4 pB = pts4[qbase + qa + 32];

  // stage 0 input: rows 0..2 = xyz, rows 3..7 = 0
  if (tid < TQ) {
    f32x4 p = pts4[qbase + tid];
    p[3] = 0.f;
    xs[tid] = p;
    xs[TQ + tid] = f32x4{0, 0, 0, 0};
  }

  uint2 mk0 = {0, 0}, mk1 = {0, 0}, mk2 = {0, 0}, mk3 = {0, 0}, mk4 = {0, 0}, mk5 = {0, 0}, mk6 = {0, 0},
        mk7 = {0, 0};
  f32x16 acc[2][2];
  f32x16 accz[2] = {zero16(), zero16()};
  float gx0 = 0.f, gx1 = 0.f, gx2 = 0.f;   // d sdf / d xyz partial (query = lane, rows of this wave)
  float y_keep = 0.f;                      // sdf of query = lane (every wave computes the same value)
  const float* cbias0 = a.c0 + (size_t)b * HID;
  const float* cbias4 = a.c4 + (size_t)b * HID;
  // stage all forward biases in LDS: the epilogues then never wait on global memory
  for (int i = tid; i < 8 * HID; i += 512) {
    const StageDesc& sb = a.dec.st[i >> 9];
    const float* src = sb.inst_bias == 1 ? cbias0 : (sb.inst_bias == 2 ? cbias4 : sb.bias);
    bl[i] = src[i & (HID - 1)];
  }

  constexpr int n_stage = MODE == 0 ? 8 : NSTAGE;
  for (int s = 0; s < n_stage; ++s) {
    const StageDesc& sd = a.dec.st[s];
    const int epi = sd.epi;
    const int mb0 = w, mb1 = w + 8;
    const bool u0 = (mb0 >= sd.mb_lo) && (mb0 < sd.mb_hi);
    const bool u1 = (mb1 >= sd.mb_lo) && (mb1 < sd.mb_hi);
    __syncthreads();   // X of this stage complete

    if (MODE == 1 && (epi == EPI_BWD4 || epi == EPI_BWD0)) {
      // xyz columns of lin4 / lin0, transposed: 3 x 512 dot per query on the VALU (rows of this wave's K slice)
      // stage the 512 x 4 xyz columns in LDS scratch (one 16-byte load per thread), then broadcast-read them
      reinterpret_cast<f32x4*>(sc)[tid] = reinterpret_cast<const f32x4*>(epi == EPI_BWD4 ? a.dec.w4x : a.dec.w0x)[tid];
      __syncthreads();
      const f32x4* wx = reinterpret_cast<const f32x4*>(sc);
#pragma unroll 4
      for (int g = 0; g < 16; ++g) {
        const f32x4 xv = xs[(16 * w + g) * TQ + lane];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 wv = wx[64 * w + 4 * g + j];
          gx0 = fmaf(xv[j], wv[0], gx0);
          gx1 = fmaf(xv[j], wv[1], gx1);
          gx2 = fmaf(xv[j], wv[2], gx2);
        }
      }
    }

    acc[0][0] = zero16(); acc[0][1] = zero16();
    if (MODE == 1 && epi == EPI_BWD0) { acc[1][0] = accz[0]; acc[1][1] = accz[1]; }
    else { acc[1][0] = zero16(); acc[1][1] = zero16(); }

    {
      const f32x4* wp = reinterpret_cast<const f32x4*>(sd.wp);
      const f32x4* wp0 = wp + (size_t)(mb0 - sd.mb_lo) * sd.n_kg * 64 + lane;
      const f32x4* wp1 = wp + (size_t)(mb1 - sd.mb_lo) * sd.n_kg * 64 + lane;
      if (u0 && u1) gemm_loop<true, true>(acc, wp0, wp1, sd.n_kg, xs, lane);
      else if (u0) gemm_loop<true, false>(acc, wp0, wp1, sd.n_kg, xs, lane);
      else if (u1) gemm_loop<false, true>(acc, wp0, wp1, sd.n_kg, xs, lane);
    }
    __syncthreads();   // every wave is done reading X

    if (MODE == 0 || epi <= EPI_FWD7) {
      const float* bias = bl + s * HID;
      uint2 mk = {0, 0};
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const bool use = sl == 0 ? u0 : u1;
        if (!use) continue;
        const int mb = sl == 0 ? mb0 : mb1;
        uint32_t bits = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int f4 = mb * 32 + 8 * g + 4 * hi;
          const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + f4);
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float val = acc[sl][nb][4 * g + j] + bv[j];
              const bool pos = val > 0.f;
              bits |= (pos ? 1u : 0u) << (nb * 16 + 4 * g + j);
              v[j] = pos ? val : 0.f;
            }
            if (epi == EPI_FWD3) {
              const f32x4 p = nb == 0 ? pA : pB;
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const int r = f4 + j - m;
                if (r >= 0 && r < 3) v[j] = (r == 0 ? p[0] : (r == 1 ? p[1] : p[2]));
              }
            }
            xs[(f4 >> 2) * TQ + nb * 32 + qa] = v;
          }
        }
        if (sl == 0) mk.x = bits; else mk.y = bits;
      }
#define HM_SET(M) M = mk
      if (MODE == 1) switch (sd.layer) { HM_MASK_CASES(HM_SET) }
#undef HM_SET

      if (epi == EPI_FWD7) {
        // lin8 + tanh (deep_sdf_decoder.py:107-108): 512-long dot per query on the VALU
        __syncthreads();
        float part = 0.f;
        const f32x4* w8v = reinterpret_cast<const f32x4*>(a.dec.w8);
#pragma unroll 4
        for (int g = 0; g < 16; ++g) {
          const f32x4 xv = xs[(16 * w + g) * TQ + lane];
          const f32x4 wv = w8v[16 * w + g];
          part = fmaf(xv[0], wv[0], part); part = fmaf(xv[1], wv[1], part);
          part = fmaf(xv[2], wv[2], part); part = fmaf(xv[3], wv[3], part);
        }
        sc[w * 64 + lane] = part;
        __syncthreads();
        float a8 = 0.f;
#pragma unroll
        for (int i = 0; i < NWAVE; ++i) a8 += sc[i * 64 + lane];
        a8 += a.dec.b8;
        const float yv = tanhf(a8);
        y_keep = yv;
        if (w == 0) {
          if (lane < cnt) a.y[qbase + lane] = yv;
          sc[2048 + lane] = 1.f - yv * yv;
        }
        if (MODE == 0) return;
        __syncthreads();
        const float dyA = sc[2048 + qa], dyB = sc[2048 + 32 + qa];
        // seed of the backward pass: G7 = (1 - y^2) * W8, masked by lin7's ReLU mask
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
          const int mb = sl == 0 ? mb0 : mb1;
          const uint32_t bits = sl == 0 ? mk.x : mk.y;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int f4 = mb * 32 + 8 * g + 4 * hi;
            const f32x4 wv = *reinterpret_cast<const f32x4*>(a.dec.w8 + f4);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
              const float dy = nb == 0 ? dyA : dyB;
              f32x4 v;
#pragma unroll
              for (int j = 0; j < 4; ++j)
                v[j] = ((bits >> (nb * 16 + 4 * g + j)) & 1u) ? dy * wv[j] : 0.f;
              xs[(f4 >> 2) * TQ + nb * 32 + qa] = v;
            }
          }
        }
      }
    } else if (MODE == 1 && (epi == EPI_BWD || epi == EPI_BWD4)) {
      uint2 mk;
#define HM_GET(M) mk = M
      switch (sd.layer) { HM_MASK_CASES(HM_GET) }
#undef HM_GET
#pragma unroll
      for (int sl = 0; sl < 2; ++sl) {
        const bool use = sl == 0 ? u0 : u1;
        if (!use) continue;
        const int mb = sl == 0 ? mb0 : mb1;
        if (epi == EPI_BWD4 && mb >= mb_zx) {     // latent rows: keep, lin0's transpose adds onto them
          if (sl == 1) { accz[0] = acc[1][0]; accz[1] = acc[1][1]; }
          continue;
        }
        const uint32_t bits = sl == 0 ? mk.x : mk.y;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int f4 = mb * 32 + 8 * g + 4 * hi;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j)
              v[j] = ((bits >> (nb * 16 + 4 * g + j)) & 1u) ? acc[sl][nb][4 * g + j] : 0.f;
            xs[(f4 >> 2) * TQ + nb * 32 + qa] = v;
          }
        }
      }
    } else if (MODE == 1) {  // EPI_BWD0: d sdf / d z complete for this wave's latent rows
      if (u1) {
        const int jz = (mb1 - mb_zx) * 32;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
          const int q = nb * 32 + qa;
          if (q < cnt) {
            float* row = a.J + (qbase + q) * (size_t)a.ldJ;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              f32x4 v;
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] = acc[1][nb][4 * g + j];
              *reinterpret_cast<f32x4*>(row + jz + 8 * g + 4 * hi) = v;
            }
          }
        }
      }
    }
  }

  if (MODE == 0) return;
  // d sdf / d xyz: reduce the 8 per-wave partials, then the pose chain rule
  //   J_pose = g_x [ I | -[p]x | p ]   (loss.py:236-239, utils.py:197-217,257-276)
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
    float* row = a.J + (qbase + lane) * (size_t)a.ldJ + L;
    row[7] = y_keep;                  // residual column of the extended row (consumed by K4)
    if (a.pose_dim == 0) {           // raw xyz gradient (get_batch_sdf_jacobian layout: [.., -3:])
      row[0] = g0; row[1] = g1; row[2] = g2;
    } else {
      row[0] = g0; row[1] = g1; row[2] = g2;
      row[3] = g2 * p[1] - g1 * p[2];   // g_x . (-[p]x) column 0 = (p x g)_0
      row[4] = g0 * p[2] - g2 * p[0];
      row[5] = g1 * p[0] - g0 * p[1];
      if (a.pose_dim == 7) row[6] = g0 * p[0] + g1 * p[1] + g2 * p[2];
    }
  }
}

// Per-instance latent biases: c0 = W0[:, :L] z + b0, c4 = W4[:, m:m+L] z + b4 (the forward pass of every
// query of an instance shares them; deep_sdf_decoder.py:87-90 concatenates z to each query instead).
__global__ void k_latent_bias(const DecoderDev dec, const float* __restrict__ latent, int ld_latent,
                              const int* __restrict__ active, float* __restrict__ c0,
                              float* __restrict__ c4) {
  // grid (B, 4), 256 threads: quarter q of an instance's 2 x 512 outputs per workgroup -- the kernel is bound by each
  // CU's L2 -> VGPR rate on the 1 MB of transposed weights, so it wants all 256 CUs, not one per instance
  __shared__ float z[MAX_L];
  const int b = blockIdx.x;
  if (active != nullptr && active[b] == 0) return;
  const int L = dec.L;
  for (int i = threadIdx.x; i < L; i += blockDim.x) z[i] = latent[(size_t)b * ld_latent + i];
  __syncthreads();
  const int f = (blockIdx.y & 1) * 256 + threadIdx.x;
  const bool second = blockIdx.y >= 2;
  const float* wT = second ? dec.w4z : dec.w0z;   // stored transposed [L][512] for coalescing
  // one serial fma chain per output (the summation order is part of the bitwise contract); what costs time is the
  // latency of the 256 dependent-looking loads, so fetch 32 weights ahead of the chain that consumes them
  float s = 0.f;
  int j = 0;
  for (; j + 32 <= L; j += 32) {
    float wv[32];
#pragma unroll
    for (int u = 0; u < 32; ++u) wv[u] = wT[(size_t)(j + u) * HID + f];
#pragma unroll
    for (int u = 0; u < 32; ++u) s = fmaf(wv[u], z[j + u], s);
  }
  for (; j < L; ++j) s = fmaf(wT[(size_t)j * HID + f], z[j], s);
  s += second ? dec.b4[f] : dec.b0[f];
  (second ? c4 : c0)[(size_t)b * HID + f] = s;
}

namespace hm {

int launch_latent_bias(const hm_decoder_s* dec, const float* d_latent, int ld_latent, const int* d_active,
                       int B, float* d_c0, float* d_c4, hipStream_t stream) {
  if (dec->generic) return launch_latent_copy_any(dec, d_latent, ld_latent, d_active, B, d_c0, stream);
  hipLaunchKernelGGL(k_latent_bias, dim3(B, 4), dim3(256), 0, stream, dec->dev, d_latent, ld_latent,
                     d_active, d_c0, d_c4);
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_decoder(const hm_decoder_s* dec, int B, const float* d_pts, const int* d_nq, const int* d_active,
                   int n_stride, const float* d_c0, const float* d_c4, float* d_y, float* d_J, int ldJ,
                   int pose_dim, int mode, hipStream_t stream, int tag) {
  if (n_stride % TQ != 0) { hm_set_error("n_stride must be a multiple of %d", TQ); return -1; }
  if (dec->generic)
    return launch_decoder_any(dec, B, d_pts, d_nq, d_active, n_stride, d_c0, d_y, d_J, ldJ, pose_dim, mode, stream);
  if (dec->precision == 3)
    return launch_decoder_p(dec, B, d_pts, d_nq, d_active, n_stride, d_c0, d_c4, d_y, d_J, ldJ, pose_dim, mode,
                            stream, tag);
  if (dec->precision == 1 || dec->precision == 2)
    return launch_decoder_h(dec, B, d_pts, d_nq, d_active, n_stride, d_c0, d_c4, d_y, d_J, ldJ, pose_dim, mode,
                            stream, tag);
  DecodeArgs a;
  a.dec = dec->dev;
  a.pts = d_pts; a.n_q = d_nq; a.active = d_active; a.c0 = d_c0; a.c4 = d_c4;
  a.y = d_y; a.J = d_J; a.n_stride = n_stride; a.B = B; a.ldJ = ldJ; a.pose_dim = pose_dim; a.mode = mode;
  const int grid = B * (n_stride / TQ);
  if (grid == 0) return 0;
  if (mode == 0) hipLaunchKernelGGL((k_decoder<0, 0>), dim3(grid), dim3(512), 0, stream, a);
  else if (tag == 0) hipLaunchKernelGGL((k_decoder<1, 0>), dim3(grid), dim3(512), 0, stream, a);
  else hipLaunchKernelGGL((k_decoder<1, 1>), dim3(grid), dim3(512), 0, stream, a);
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hm
