// K2/K3/K6: point transforms, ray sampling and the occupancy ray-rendering scan.
//
// Restates `compute_render_loss` (wild_completion/loss.py:8-217) and its caller's frame loop
// (wild_completion/optimizer.py:102-118) as a dense per-ray computation -- one wavefront per ray, one lane per
// depth sample (M <= 64) -- instead of the reference's where/boolean-index/unique/scatter_add sequence:
//   k_frame_setup      optimizer.py:66,103-111   T_oc, depth window, per-frame constants
//   k_sample_rays      loss.py:30-40             p_c = dir * d_j, p_o = R_oc p_c + t_oc, ball filter (+ count for :43-45)
//   (K1 forward-only on the ball-valid samples)   loss.py:48-49
//   [k_promote         linear occupancy only: one-pass fp16 screening of the ball-valid samples; only the ones that may
//                      lie in the +-cutoff band (utils.py:125-133 clamps the others to exactly 0 / 1) go through the
//                      fp32-class forward -- round 5]
//   k_ray_scan         loss.py:55-176            occupancy, transmittance scan, d_u, occ_ray, de/do, dm/do, do/ds,
//                                                min-grad and occlusion filters, residuals
//   k_ray_offsets      loss.py:160-166           torch.unique(ray ids) == ascending ray order -> prefix sums
//   k_ray_scatter      loss.py:185               gather the surviving sample points for the Jacobian pass
//   (K1 forward+backward on the survivors)       loss.py:186
//   k_ray_reduce       loss.py:188-215           per-ray sum of (de/ds, dm/ds) * d sdf/d(pose, code)
// and the surface-point transform of optimizer.py:168 / :343 (k_transform_points).
#include "hm_common.h"
#include "hm_internal.h"

using namespace hm;

namespace {

__device__ __forceinline__ float det3f(const float* M, int ld) {
  const double a = M[0], b = M[1], c = M[2], d = M[ld], e = M[ld + 1], f = M[ld + 2], g = M[2 * ld],
               h = M[2 * ld + 1], i = M[2 * ld + 2];
  return (float)(a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g));
}

// torch.linspace(start, end, steps)[j] in fp32 (symmetric two-sided formula of ATen's CPU/CUDA kernels)
__device__ __forceinline__ float linspace_at(float start, float end, int steps, int j) {
  const float step = (end - start) / (float)(steps - 1);
  return (j < steps / 2) ? start + step * (float)j : end - step * (float)(steps - 1 - j);
}

}  // namespace

// x_o = R x_w + t  (optimizer.py:168: (points[..., None, :] * T[:3,:3]).sum(-1) + T[:3,3])
__global__ void k_transform_points(const float* __restrict__ pw, int n_in_stride, const int* __restrict__ n,
                                   const float* __restrict__ T_ow, const int* __restrict__ active, int n_stride,
                                   float* __restrict__ out) {
  const int b = blockIdx.y;
  if (active != nullptr && active[b] == 0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_stride) return;
  f32x4 o = {0, 0, 0, 0};
  if (i < n[b]) {
    const float* T = T_ow + (size_t)b * 16;
    const float* p = pw + ((size_t)b * n_in_stride + i) * 3;
    const float x = p[0], y = p[1], z = p[2];
    o[0] = (x * T[0] + y * T[1]) + z * T[2] + T[3];
    o[1] = (x * T[4] + y * T[5]) + z * T[6] + T[7];
    o[2] = (x * T[8] + y * T[9]) + z * T[10] + T[11];
    o[3] = 1.f;
  }
  reinterpret_cast<f32x4*>(out)[(size_t)b * n_stride + i] = o;
}

__global__ void k_frame_setup(const RenderCfg cfg, const RenderBuffers rb, const float* __restrict__ T_ow,
                              const int* __restrict__ active, const float* __restrict__ frame_override) {
  const int b = blockIdx.x;
  if (active != nullptr && active[b] == 0) return;
  const int f = threadIdx.x;
  const int nf = rb.n_frames[b];
  if (f == 0) { rb.nRq[b] = 0; rb.nflag[b] = 0; if (rb.nRp != nullptr) rb.nRp[b] = 0; }   // k_sample_rays counts the ball-valid samples into nRq
  if (f >= cfg.F) return;
  rb.valid_count[b * cfg.F + f] = 0;
  if (f >= nf) return;
  if (frame_override != nullptr) {   // functional hook: caller supplies T_oc | d_min | d_max | ball radius directly
    float* fp = rb.frame + ((size_t)b * cfg.F + f) * 16;
    for (int i = 0; i < 16; ++i) fp[i] = frame_override[((size_t)b * cfg.F + f) * 16 + i];
    return;
  }
  const float* T = T_ow + (size_t)b * 16;
  const float* C = rb.T_wc + ((size_t)b * cfg.F + f) * 16;
  float Toc[12];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) {
      float s = 0.f;
      for (int k = 0; k < 4; ++k) s += T[i * 4 + k] * C[k * 4 + j];            // optimizer.py:104
      Toc[i * 4 + j] = s;
    }
  // T_co[2,3] of the inverse (optimizer.py:105,110): row 2 of M^-1 is (c0 x c1)/det, so T_co[2,3] = -(c0 x c1).t/det
  const double c0[3] = {Toc[0], Toc[4], Toc[8]}, c1[3] = {Toc[1], Toc[5], Toc[9]};
  const double t[3] = {Toc[3], Toc[7], Toc[11]};
  const double cx = c0[1] * c1[2] - c0[2] * c1[1], cy = c0[2] * c1[0] - c0[0] * c1[2],
               cz = c0[0] * c1[1] - c0[1] * c1[0];
  const double det = (double)det3f(Toc, 4);
  const float tco23 = (float)(-(cx * t[0] + cy * t[1] + cz * t[2]) / det);
  const float cur_scale = powf(det3f(T, 4), -1.f / 3.f);                        // optimizer.py:66,250
  const float range = rb.cube_radius[b] * cur_scale;                            // :107
  float* fp = rb.frame + ((size_t)b * cfg.F + f) * 16;
  for (int i = 0; i < 12; ++i) fp[i] = Toc[i];
  fp[12] = tco23 - 1.0f * range;                                                // :110
  fp[13] = tco23 + 0.8f * range;
  fp[14] = range;
  fp[15] = 0.f;
}

// Sample points of every ray, ball filter, and compaction of the ball-valid ones: like the reference (loss.py:38-49)
// only those go through the decoder.  Slots are handed out per 256-sample block (wave ballots + one atomic on the
// instance counter); the slot order between blocks depends on scheduling, which is harmless: the decoder treats every
// query column independently, so a sample's sdf does not depend on its slot.
__global__ __launch_bounds__(256) void k_sample_rays(const RenderCfg cfg, const RenderBuffers rb,
                                                     const int* __restrict__ active) {
  __shared__ int wcount[4];
  __shared__ int base_s;
  const int b = blockIdx.z, f = blockIdx.y;
  if (active != nullptr && active[b] == 0) return;
  if (f >= rb.n_frames[b]) return;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;     // (ray, sample)
  const int total = cfg.R * cfg.M;
  const int r = idx / cfg.M, j = idx - r * cfg.M;
  const int nray = rb.n_fg[b * cfg.F + f] + rb.n_bg[b * cfg.F + f];
  const float* fp = rb.frame + ((size_t)b * cfg.F + f) * 16;
  f32x4 o = {0, 0, 0, 0};
  bool valid = false;
  if (idx < total && r < nray) {
    const float* dir = rb.rays + (((size_t)b * cfg.F + f) * cfg.R + r) * 3;
    const float d = linspace_at(fp[12], fp[13], cfg.M, j);                      // optimizer.py:111
    const float x = dir[0] * d, y = dir[1] * d, z = dir[2] * d;                 // loss.py:30
    o[0] = (x * fp[0] + y * fp[1]) + z * fp[2] + fp[3];                         // loss.py:32-33
    o[1] = (x * fp[4] + y * fp[5]) + z * fp[6] + fp[7];
    o[2] = (x * fp[8] + y * fp[9]) + z * fp[10] + fp[11];
    valid = sqrtf(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]) < fp[14];            // loss.py:38
    o[3] = valid ? 1.f : 0.f;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned long long m = __ballot(valid);
  const int before = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) wcount[wv] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nblk = wcount[0] + wcount[1] + wcount[2] + wcount[3];
    base_s = nblk > 0 ? atomicAdd(&rb.nRq[b], nblk) : 0;
    if (nblk > 0) atomicAdd(&rb.valid_count[b * cfg.F + f], nblk);
  }
  __syncthreads();
  if (idx >= total) return;
  int slot = -1;
  if (valid) {
    slot = base_s + before;
    for (int w2 = 0; w2 < wv; ++w2) slot += wcount[w2];
    reinterpret_cast<f32x4*>(rb.ptsRc)[(size_t)b * rb.nR_stride + slot] = o;
  }
  const size_t at = (size_t)b * rb.nR_stride + (size_t)f * total + idx;
  reinterpret_cast<f32x4*>(rb.ptsR)[at] = o;
  rb.cpos[at] = slot;
}

// LINEAR-occupancy screening (RenderCfg::screen; round 5).  sdf_to_occupancy (utils.py:125-133) clamps: a sample with
// sdf >= +th has occupancy exactly 0, one with sdf <= -th exactly 1, and neither is a with-grad sample (loss.py:66) --
// whatever its exact value.  So the exact (fp32-class, three-pass) forward is only needed for samples that MAY lie inside
// the band.  Every ball-valid sample has been decoded by the ONE-pass fp16 forward (K1p, half the cost per pass and a
// third of the passes) into sdfS; with |s_fp16 - s| <= eps (measured bound with margin, scripts/measure_screen_eps.py):
//   s_fp16 >  th + eps  ->  s >  th: far outside, occupancy 0
//   s_fp16 < -th - eps  ->  s < -th: far inside,  occupancy 1
//   otherwise (or not finite)      : PROMOTED to the f16x3 forward (+ ReLU masks), exactly as before.
// Behind the first far-inside sample of a ray the transmittance is exactly 0 (loss.py:81: cumprod of 1 - o with an
// o == 1 factor), so every later sample of that ray has term_prob == 0, contributes 0 to d_u / occ_ray / the suffix sums
// and fails `de_do > min_grad_thre` (0 or 0/0): its value cannot reach any output bit and it is not promoted either
// ("dead").  One wavefront per ray, lane = depth sample; cpos is rewritten from "slot in ptsRc" to "slot in the promoted
// list ptsRp" or a CPOS_FAR_* code that k_ray_scan turns into a saturated sdf.  Bit-identical results by construction;
// tests/test_gpu_round5.py compares whole trajectories with the screening off and counts violations in verify mode.
// Sixty-four rays per 1024-thread workgroup, four per wave (round 6): the four rays' loads are in flight together, their
// ballots stay in registers, and the promoted samples of the whole workgroup take their slots with ONE atomic on the
// instance counter -- 25 atomics per instance and iteration on the challenge configuration (one per wave cost 1.1 ms per
// iteration there: 96 k atomics on 64 words; one per 16 rays 0.55 ms; this form: see profiles/r06_configs2_*).
constexpr int PROMOTE_RPW = 4;
__global__ __launch_bounds__(1024) void k_promote(const RenderCfg cfg, const RenderBuffers rb,
                                                  const int* __restrict__ active) {
  __shared__ int wcnt[16];
  __shared__ int base_s;
  const int b = blockIdx.z, f = blockIdx.y;
  const int wv = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  if (active != nullptr && active[b] == 0) return;                 // (workgroup-uniform exits)
  if (f >= rb.n_frames[b]) return;
  const int nray = rb.n_fg[b * cfg.F + f] + rb.n_bg[b * cfg.F + f];
  // a frame with too few ball-valid samples is skipped by k_ray_scan (loss.py:43-45): none of its samples is needed
  const bool frame_ok = rb.valid_count[b * cfg.F + f] >= cfg.min_valid;
  const int M = cfg.M;
  const float lim = cfg.occ_th + cfg.screen_eps;
  const size_t ibase = (size_t)b * rb.nR_stride;
  const int r0 = (blockIdx.x * 16 + wv) * PROMOTE_RPW;              // this wave's rays: r0 .. r0 + 3
  size_t at[PROMOTE_RPW];
  int slot[PROMOTE_RPW];
  float st[PROMOTE_RPW];
  bool in[PROMOTE_RPW];
#pragma unroll
  for (int i = 0; i < PROMOTE_RPW; ++i) {
    const int r = r0 + i;
    const bool ray_ok = r < cfg.R && r < nray;
    at[i] = ibase + (size_t)(f * cfg.R + (ray_ok ? r : 0)) * M + lane;
    in[i] = ray_ok && lane < M;
    slot[i] = in[i] ? rb.cpos[at[i]] : CPOS_NOT_VALID;
  }
#pragma unroll
  for (int i = 0; i < PROMOTE_RPW; ++i)                    // k_sample_rays: slot >= 0 <=> ball-valid (w of ptsR is 1)
    st[i] = slot[i] >= 0 ? rb.sdfS[ibase + slot[i]] : 0.f;
  unsigned long long pm[PROMOTE_RPW], fm[PROMOTE_RPW];
  int mine = 0;
  int n_valid = 0, n_dead = 0, n_bad = 0;
#pragma unroll
  for (int i = 0; i < PROMOTE_RPW; ++i) {
    const bool valid = slot[i] >= 0;
    const bool far_in = valid && st[i] < -lim, far_out = valid && st[i] > lim;
    const unsigned long long m_in = __ballot(far_in);
    const int first_in = m_in ? __ffsll((long long)m_in) - 1 : 64;
    // (a sample whose screening value is not finite is never dead: it is promoted, so that the range guard of the f16x3
    // forward reports it exactly as it did before the screening existed)
    const bool dead = valid && lane > first_in && isfinite(st[i]);
    const bool promote = valid && frame_ok && !dead && !far_in && !far_out;     // includes non-finite screening values
    pm[i] = __ballot(promote);
    fm[i] = __ballot(far_in || dead);
    mine += __popcll(pm[i]);
    if (rb.screen_stats != nullptr) {
      bool bad = false;
      if (rb.sdfFull != nullptr && valid && frame_ok && !promote) {
        const float ex = rb.sdfFull[ibase + slot[i]];
        if (far_in && lane == first_in) bad = !(ex < -cfg.occ_th);             // must really saturate to occupancy 1
        else if (!dead && !far_in) bad = !(ex > cfg.occ_th);                   // far outside: must really be occupancy 0
      }
      n_valid += __popcll(__ballot(valid && frame_ok));
      n_dead += __popcll(__ballot(dead && frame_ok));
      n_bad += __popcll(__ballot(bad));
    }
  }
  if (lane == 0) wcnt[wv] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) tot += wcnt[i];
    base_s = tot > 0 ? atomicAdd(&rb.nRp[b], tot) : 0;
  }
  __syncthreads();
  int base = base_s;
  for (int i = 0; i < wv; ++i) base += wcnt[i];
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int i = 0; i < PROMOTE_RPW; ++i) {
    if (slot[i] >= 0) {
      int code;
      if ((pm[i] >> lane) & 1ull) {
        code = base + __popcll(pm[i] & below);
        reinterpret_cast<f32x4*>(rb.ptsRp)[ibase + code] = reinterpret_cast<const f32x4*>(rb.ptsR)[at[i]];
      } else {
        code = ((fm[i] >> lane) & 1ull) ? CPOS_FAR_INSIDE : CPOS_FAR_OUTSIDE;  // (!frame_ok: the code is never read)
      }
      rb.cpos[at[i]] = code;
    }
    base += __popcll(pm[i]);
  }
  if (rb.screen_stats != nullptr && lane == 0) {
    if (n_valid) atomicAdd(&rb.screen_stats[0], (unsigned long long)n_valid);
    if (mine) atomicAdd(&rb.screen_stats[1], (unsigned long long)mine);
    if (n_bad) atomicAdd(&rb.screen_stats[2], (unsigned long long)n_bad);
    if (n_dead) atomicAdd(&rb.screen_stats[3], (unsigned long long)n_dead);
  }
}

// one wavefront per ray, lane = depth sample
__global__ __launch_bounds__(256) void k_ray_scan(const RenderCfg cfg, const RenderBuffers rb,
                                                  const int* __restrict__ active) {
  const int b = blockIdx.z, f = blockIdx.y;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= cfg.R) return;
  const int ray = f * cfg.R + r;
  const size_t rix = (size_t)b * cfg.F * cfg.R + ray;
  if (active != nullptr && active[b] == 0) return;
  const int n_fg = rb.n_fg[b * cfg.F + f];
  const int nray = n_fg + rb.n_bg[b * cfg.F + f];
  const bool frame_ok = f < rb.n_frames[b] && rb.valid_count[b * cfg.F + f] >= cfg.min_valid;   // loss.py:43-45
  if (!frame_ok || r >= nray) {
    if (lane == 0) { rb.keepcnt[rix] = 0; rb.keepmask[rix] = 0ull; }
    if (lane == 0 && r == 0 && f < rb.n_frames[b] && !frame_ok && rb.status != nullptr)
      atomicOr(&rb.status[b], HM_STATUS_FRAME_SKIPPED);          // 'This frame is not valid' (optimizer.py:130-132)
    return;
  }
  const int M = cfg.M;
  const float* fp = rb.frame + ((size_t)b * cfg.F + f) * 16;
  const float d_min = fp[12], d_max = fp[13];
  const size_t sbase = (size_t)b * rb.nR_stride + (size_t)ray * M;
  const bool in = lane < M;
  const float th = cfg.occ_th;
  const float sigma = th / 3.f * 0.55f;                                         // loss.py:59-60
  float s = 0.f, o = 0.f, dj = 0.f;
  bool valid = false;
  if (in) {
    valid = rb.ptsR[(sbase + lane) * 4 + 3] != 0.f;
    const int slot = rb.cpos[sbase + lane];
    // screened-far samples (k_promote) carry a code instead of a slot: any sdf beyond the clamp gives the same bits
    s = slot >= 0 ? rb.sdfR[(size_t)b * rb.nR_stride + slot]
                  : (slot == CPOS_FAR_INSIDE ? -1e30f : (slot == CPOS_FAR_OUTSIDE ? 1e30f : 0.f));
    if (valid && !isfinite(s)) rb.nflag[b] = 1;                 // reported by the solver as a numerical failure
    dj = linspace_at(d_min, d_max, M, lane);
    if (valid) {
      if (cfg.log_occ) o = 1.f / (1.f + expf(s / sigma));                       // utils.py:136-142 sigmoid(-s/sigma)
      else o = 0.5f - fminf(fmaxf(s, -th), th) / (2.f * th);                    // utils.py:125-133
    }
  }
  const bool wg = valid && (s > -th) && (s < th);                               // loss.py:66
  const float delta_d = (d_max - d_min) / (float)(M - 1);                       // :75
  const float d_term = d_max + delta_d;                                         // :78
  // inclusive prefix product of (1 - o)  (:81)
  const float one_m = 1.f - o;
  float T = in ? one_m : 1.f;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float up = __shfl_up(T, off);
    if (lane >= off) T *= up;
  }
  float Tprev = __shfl_up(T, 1);
  if (lane == 0) Tprev = 1.f;
  const float Tlast = __shfl(T, M - 1);
  const float prob = in ? o * Tprev : 0.f;                                      // :82-91
  float occ_ray = prob, dsum = dj * prob;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { occ_ray += __shfl_xor(occ_ray, off); dsum += __shfl_xor(dsum, off); }
  const float d_u = dsum + d_term * Tlast;                                      // :96
  // inclusive suffix sum of T over the ray's samples (:103-107)
  float suf = in ? T : 0.f;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float dn = __shfl_down(suf, off);
    if (lane + off < 64) suf += dn;
  }
  const float de_do = suf * delta_d / one_m;
  const float dm_do = Tlast / one_m;                                            // :101-102
  const float do_ds = cfg.log_occ ? -o * (1.f - o) / sigma : -1.f / (2.f * th); // :120-123
  const bool is_bg = r >= n_fg;
  const float obs = rb.depth[((size_t)b * cfg.F + f) * cfg.R + r];
  bool keep = in && wg && (de_do > cfg.min_grad);                               // :111-118
  if (cfg.occlusion_on && is_bg && (obs < d_u - cfg.occlusion_th) && (obs > 0.f)) keep = false;   // :132-139
  const unsigned long long km = __ballot(keep);
  if (in) {
    float* cf = rb.coef + (sbase + lane) * 2;
    cf[0] = keep ? de_do * do_ds : 0.f;                                         // de_ds :126
    cf[1] = keep ? dm_do * do_ds : 0.f;                                         // dm_ds :127
  }
  if (lane == 0) {
    rb.keepcnt[rix] = __popcll(km);
    rb.keepmask[rix] = km;
    rb.res_d[rix] = (is_bg ? d_term : obs) - d_u;                               // :142,151,155
    rb.res_m[rix] = occ_ray - (is_bg ? 0.f : 1.f);                              // :172-176
  }
}

// per instance: exclusive scans over rays (ascending ray index == torch.unique order, loss.py:160-166)
__global__ __launch_bounds__(1024) void k_ray_offsets(const RenderCfg cfg, const RenderBuffers rb,
                                                      const int* __restrict__ active) {
  __shared__ int wsum_k[16], wsum_e[16];
  __shared__ int carry_k, carry_e;
  const int b = blockIdx.x;
  if (active != nullptr && active[b] == 0) return;
  const int nray = cfg.F * cfg.R;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) { carry_k = 0; carry_e = 0; }
  __syncthreads();
  for (int base = 0; base < nray; base += 1024) {
    const int i = base + tid;
    const int k = i < nray ? rb.keepcnt[(size_t)b * nray + i] : 0;
    const int e = k > 0 ? 1 : 0;
    int sk = k, se = e;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int uk = __shfl_up(sk, off), ue = __shfl_up(se, off);
      if (lane >= off) { sk += uk; se += ue; }
    }
    if (lane == 63) { wsum_k[wv] = sk; wsum_e[wv] = se; }
    __syncthreads();
    int pk = carry_k, pe = carry_e;
    for (int w = 0; w < wv; ++w) { pk += wsum_k[w]; pe += wsum_e[w]; }
    if (i < nray) {
      rb.ray_off[(size_t)b * nray + i] = pk + sk - k;
      rb.ray_row[(size_t)b * nray + i] = e ? pe + se - 1 : -1;
    }
    __syncthreads();
    if (tid == 1023) { carry_k = pk + sk; carry_e = pe + se; }
    __syncthreads();
  }
  if (tid == 0) {
    int ng = carry_k;
    if (ng > rb.nG_stride) {     // more Jacobian samples than limits.max_grad_samples: the per-ray sums would be
      ng = rb.nG_stride;         // partial and the normal equations biased, so report it and let K5 stop the instance
      if (rb.status != nullptr) rb.status[b] |= HM_STATUS_LIMIT;
      rb.nflag[b] = 1;
    }
    rb.nG[b] = ng;
    rb.V[b] = carry_e;
  }
}

// flat_jac: exclusive scan of nG over the ACTIVE instances (an inactive one keeps a stale nG from its last iteration)
__global__ __launch_bounds__(1024) void k_ray_gbase(const RenderBuffers rb, const int* __restrict__ active, int B) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < B; base += 1024) {
    const int b = base + tid;
    const int k = (b < B && (active == nullptr || active[b] != 0)) ? rb.nG[b] : 0;
    int sk = k;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(sk, off);
      if (lane >= off) sk += u;
    }
    if (lane == 63) wsum[wv] = sk;
    __syncthreads();
    int pk = carry;
    for (int w2 = 0; w2 < wv; ++w2) pk += wsum[w2];
    if (b < B) rb.gbase[b] = pk + sk - k;
    __syncthreads();
    if (tid == 1023) carry = pk + sk;
    __syncthreads();
  }
  if (tid == 0) *rb.gtotal = carry;
}

__global__ __launch_bounds__(256) void k_ray_scatter(const RenderCfg cfg, const RenderBuffers rb,
                                                     const int* __restrict__ active) {
  const int b = blockIdx.z, f = blockIdx.y;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= cfg.R) return;
  if (active != nullptr && active[b] == 0) return;
  const int ray = f * cfg.R + r;
  const size_t rix = (size_t)b * cfg.F * cfg.R + ray;
  const unsigned long long km = rb.keepmask[rix];
  if (km == 0ull || !((km >> lane) & 1ull)) return;
  const int rank = __popcll(km & ((1ull << lane) - 1ull));
  const int dst = rb.ray_off[rix] + rank;
  if (dst >= rb.nG_stride) return;
  const size_t src = (size_t)b * rb.nR_stride + (size_t)ray * cfg.M + lane;
  f32x4 p = reinterpret_cast<const f32x4*>(rb.ptsR)[src];
  p[3] = 1.f;
  // list position: per-instance lists padded to nG_stride, or ONE flat list over the instances (flat_jac)
  const size_t at = cfg.flat_jac ? (size_t)rb.gbase[b] + dst : (size_t)b * rb.nG_stride + dst;
  reinterpret_cast<f32x4*>(rb.ptsG)[at] = p;
  rb.srcG[at] = cfg.flat_jac ? b * rb.nR_stride + rb.cpos[src] : rb.cpos[src];       // where the forward pass decoded this sample
  rb.coefG[at * 2 + 0] = rb.coef[src * 2 + 0];
  rb.coefG[at * 2 + 1] = rb.coef[src * 2 + 1];
}

// one workgroup per ray: J_d = sum_k de_ds_k * J_k, J_m = sum_k dm_ds_k * J_k over the ray's surviving samples,
// in ascending sample order (the order scatter_add_ visits them, loss.py:209-215)
__global__ __launch_bounds__(128) void k_ray_reduce(const RenderCfg cfg, const RenderBuffers rb,
                                                    const int* __restrict__ active, int L) {
  const int b = blockIdx.y, ray = blockIdx.x;
  if (active != nullptr && active[b] == 0) return;
  const int nray = cfg.F * cfg.R;
  const size_t rix = (size_t)b * nray + ray;
  const int cnt = rb.keepcnt[rix];
  if (cnt <= 0) return;
  const int row = rb.ray_row[rix];
  const int off = rb.ray_off[rix];
  const int ldJ = L + POSE_PAD;
  const int ncol = L + 7;
  const size_t first = (cfg.flat_jac ? (size_t)rb.gbase[b] : (size_t)b * rb.nG_stride) + off;
  const float* Jg = rb.JG + first * ldJ;
  const float* cg = rb.coefG + first * 2;
  float* Jd = rb.JR + ((size_t)b * 2 * nray + row) * ldJ;
  float* Jm = rb.JR + ((size_t)b * 2 * nray + nray + row) * ldJ;
  for (int c = threadIdx.x; c < ncol; c += blockDim.x) {
    float sd = 0.f, sm = 0.f;
    for (int k = 0; k < cnt && off + k < rb.nG_stride; ++k) {
      const float v = Jg[(size_t)k * ldJ + c];
      sd += cg[2 * k] * v;
      sm += cg[2 * k + 1] * v;
    }
    Jd[c] = sd;
    Jm[c] = sm;
  }
  if (threadIdx.x == 0) {
    Jd[L + 7] = rb.res_d[rix];
    Jm[L + 7] = rb.res_m[rix];
  }
}

namespace hm {

int launch_transform_points(const float* d_points_w, int n_in_stride, const int* d_n, const float* d_T_ow,
                            const int* d_active, int B, int n_stride, float* d_pts4, hipStream_t stream) {
  dim3 grid((n_stride + 255) / 256, B);
  hipLaunchKernelGGL(k_transform_points, grid, dim3(256), 0, stream, d_points_w, n_in_stride, d_n, d_T_ow,
                     d_active, n_stride, d_pts4);
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_render_front(const RenderCfg& cfg, const RenderBuffers& rb, const float* d_T_ow, const int* d_active,
                        int B, hipStream_t stream, const float* d_frame_override) {
  hipLaunchKernelGGL(k_frame_setup, dim3(B), dim3(64), 0, stream, cfg, rb, d_T_ow, d_active, d_frame_override);
  HM_CHECK_HIP(hipGetLastError());
  dim3 grid((cfg.R * cfg.M + 255) / 256, cfg.F, B);
  hipLaunchKernelGGL(k_sample_rays, grid, dim3(256), 0, stream, cfg, rb, d_active);
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_render_promote(const RenderCfg& cfg, const RenderBuffers& rb, const int* d_active, int B,
                          hipStream_t stream) {
  dim3 grid((cfg.R + 16 * PROMOTE_RPW - 1) / (16 * PROMOTE_RPW), cfg.F, B);
  hipLaunchKernelGGL(k_promote, grid, dim3(1024), 0, stream, cfg, rb, d_active);
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_render_scan(const RenderCfg& cfg, const RenderBuffers& rb, const int* d_active, int B,
                       hipStream_t stream) {
  dim3 grid((cfg.R + 3) / 4, cfg.F, B);
  hipLaunchKernelGGL(k_ray_scan, grid, dim3(256), 0, stream, cfg, rb, d_active);
  HM_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(k_ray_offsets, dim3(B), dim3(1024), 0, stream, cfg, rb, d_active);
  HM_CHECK_HIP(hipGetLastError());
  if (cfg.flat_jac) {
    hipLaunchKernelGGL(k_ray_gbase, dim3(1), dim3(1024), 0, stream, rb, d_active, B);
    HM_CHECK_HIP(hipGetLastError());
  }
  hipLaunchKernelGGL(k_ray_scatter, grid, dim3(256), 0, stream, cfg, rb, d_active);
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_render_reduce(const RenderCfg& cfg, const RenderBuffers& rb, const int* d_active, int B, int L,
                         hipStream_t stream) {
  dim3 grid(cfg.F * cfg.R, B);
  hipLaunchKernelGGL(k_ray_reduce, grid, dim3(128), 0, stream, cfg, rb, d_active, L);
  HM_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace hm
