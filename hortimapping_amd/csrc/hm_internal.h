// Internal launch helpers shared between translation units of libhortihip.
#pragma once
#include "hm_common.h"

// per-instance status bits (mirrors include/hortimapping_amd.h)
#define HM_STATUS_CONV_G 1        // optimizer.py:276  gradient convergence
#define HM_STATUS_CONV_C 2        // optimizer.py:280  latent-code convergence
#define HM_STATUS_CONV_P 4        // optimizer.py:285  pose convergence
#define HM_STATUS_MAX_ITER 8      // optimizer.py:289
#define HM_STATUS_INVALID 16      // optimizer.py:139-141 "This submap is not valid"
#define HM_STATUS_SOLVE_FAILED 32 // non-finite / non-SPD system (the reference would propagate NaN)
#define HM_STATUS_FRAME_SKIPPED 64 // a frame returned None in some iteration (optimizer.py:130-132), informational
#define HM_STATUS_LIMIT 128       // instance exceeds a workspace capacity: refused / stopped, never silently truncated

namespace hm {

struct RowSegment {
  const float* rows;       // base of the row buffer
  size_t inst_stride;      // floats between instances
  int row_offset;          // first row of the segment inside an instance's buffer
  const int* count_dev;    // [B] rows in this segment (device) or nullptr -> count_const
  int count_const;
  const int* norm_dev;     // [B] normaliser n_t (device) or nullptr -> the row count
  float weight;            // w_t
  float robust_th;         // Huber threshold, <= 0: off
};

struct SolveArgs {
  const float* Hext;       // [B][ldJ][ldJ]
  float* Lfac;             // [B][288][288] scratch: Cholesky factor panels
  float* latent;           // [B][ld_latent]
  float* T_ow;             // [B][16]
  const int* pose_known;   // [B] or nullptr
  const int* V;            // [B] depth-render residual count (joint mode) or nullptr (shape-only)
  const int* nflag;        // [B] set when a ball-valid ray sample decoded to a non-finite sdf (joint mode) or nullptr
  int* active;             // [B]
  int* iter_count;         // [B]
  int* status;             // [B]
  float* cur_scale;        // [B] or nullptr
  float* dbg_A;            // [B][ldJ][ldJ] damped system (lower) or nullptr
  float* dbg_b;            // [B][ldJ] or nullptr
  float* dbg_delta;        // [B][ldJ] or nullptr
  int L, P, ldJ, ld_latent;
  int iter, max_iter;
  int lm_on, lm_eye, scale_on;
  int force_direct;        // 1: skip the PCG fast path of the solve (tests of the Cholesky fallback)
  float w_code, s_damp, lam0;
  float eps_g, eps_c, eps_t, eps_r, eps_s;
};

struct RenderCfg {
  int F, R, M;             // capacities: frames, rays per frame, samples per ray
  int log_occ, occlusion_on, scale_on;
  float occ_th, occlusion_th, min_grad;
  int min_valid;
  // LINEAR-occupancy screening (round 5; hm_render.hip k_promote): 0 off; 1 on; 2 on + verify against a full f16x3
  // forward (rb.sdfFull).  A ray sample whose one-pass fp16 sdf lies beyond occ_th + screen_eps is "far": its occupancy is
  // exactly 0 or 1 whatever its exact sdf (utils.py:125-133 clamps), so only the others go through the f16x3 forward.
  int screen;
  float screen_eps;
  // 1: the Jacobian samples of ALL instances form ONE flat list (rb.gbase / rb.gtotal) instead of one list per instance
  // padded to whole 64-query tiles -- the backward-only f16x3 launch then runs ceil(total / 64) tiles instead of
  // sum_b ceil(nG[b] / 64) (C2-joint: 137 samples per instance = 2.14 tiles, launched as 3).  f16x3 render chain only.
  int flat_jac;
};

// cpos codes of screened-far samples (instead of a slot in the promoted list)
constexpr int CPOS_NOT_VALID = -1;   // outside the ball (loss.py:38)
constexpr int CPOS_FAR_INSIDE = -2;  // sdf < -occ_th for certain (or behind such a sample on its ray): occupancy 1
constexpr int CPOS_FAR_OUTSIDE = -3; // sdf > +occ_th for certain: occupancy 0

struct RenderBuffers {
  // caller inputs
  const float* T_wc;       // [B][F][16]
  const float* rays;       // [B][F][R][3]  fg rays first, then bg
  const float* depth;      // [B][F][R]
  const int* n_fg;         // [B][F]
  const int* n_bg;         // [B][F]
  const int* n_frames;     // [B]
  const float* cube_radius;// [B]
  // workspace
  float* frame;            // [B][F][16]  T_oc (12) | d_min | d_max | range | pad
  int* nflag;              // [B] numerical-failure flag of the render pass (non-finite sdf of a ball-valid sample)
  int* status;             // [B] per-instance status word of the running optimisation (nullptr outside of it)
  int* valid_count;        // [B][F]
  int* nRq;                // [B] ball-valid samples to decode (K_v, loss.py:38-49)
  float* ptsR;             // [B][nR_stride][4]  all samples, (frame, ray, depth) order, w = ball-valid flag
  float* ptsRc;            // [B][nR_stride][4]  the ball-valid ones, compacted (what the decoder reads)
  int* cpos;               // [B][nR_stride]     sample -> slot in ptsRc / sdfR, -1 if not ball-valid
  float* sdfR;             // [B][nR_stride]     decoder output, compacted order
  int* keepcnt;            // [B][F*R]
  unsigned long long* keepmask;  // [B][F*R]
  float* res_d;            // [B][F*R]
  float* res_m;            // [B][F*R]
  float* coef;             // [B][nR_stride][2]  (de_ds, dm_ds) per sample
  int* ray_off;            // [B][F*R]
  int* ray_row;            // [B][F*R]
  int* nG;                 // [B] samples that need the Jacobian
  int* V;                  // [B] emitted rays
  float* ptsG;             // [B][nG_stride][4]
  float* coefG;            // [B][nG_stride][2]
  float* JG;               // [B][nG_stride][ldJ]
  float* yG;               // [B][nG_stride]
  int* srcG;               // [B][nG_stride]   slot (index in ptsRc / sdfR order) each Jacobian sample was gathered from
                           //                  (flat_jac: GLOBAL slot b * nR_stride + slot, list position gbase[b] + k)
  int* gbase;              // [B] flat_jac: first list position of instance b = sum of nG over the active instances before it
  int* gtotal;             // [1] flat_jac: total Jacobian samples of the launch
  void* maskR;             // [B][nR_stride / 64][8][512] uint2: ReLU masks saved by the f16x3 forward pass over ptsRc
  // screening (RenderCfg::screen): fp16 sdf of every ball-valid sample, the promoted samples the f16x3 forward decodes
  // instead of ptsRc (cpos then indexes THIS list), their count, and the per-group statistics
  float* sdfS;             // [B][nR_stride]     one-pass fp16 sdf, ptsRc slot order
  float* ptsRp;            // [B][nR_stride][4]  promoted samples
  int* nRp;                // [B]
  const float* sdfFull;    // verify mode: f16x3 sdf of EVERY ball-valid sample, ptsRc slot order (else nullptr)
  unsigned long long* screen_stats;  // [4]: screened, promoted, violations (verify mode), dead (behind a far-inside sample)
  float* JR;               // [B][2*F*R][ldJ]   depth rows then mask rows
  int nR_stride, nG_stride;
};

int launch_latent_bias(const hm_decoder_s* dec, const float* d_latent, int ld_latent, const int* d_active,
                       int B, float* d_c0, float* d_c4, hipStream_t stream);

int launch_decoder(const hm_decoder_s* dec, int B, const float* d_pts, const int* d_nq, const int* d_active,
                   int n_stride, const float* d_c0, const float* d_c4, float* d_y, float* d_J, int ldJ,
                   int pose_dim, int mode, hipStream_t stream, int tag = 0);

int launch_decoder_h(const hm_decoder_s* dec, int B, const float* d_pts, const int* d_nq, const int* d_active,
                     int n_stride, const float* d_c0, const float* d_c4, float* d_y, float* d_J, int ldJ,
                     int pose_dim, int mode, hipStream_t stream, int tag);

// f16x3 decoder, job-list launches of the LM iteration (hm_decoder_h.hip)
int launch_decoder_h_main(const hm_decoder_s* dec, int B, const int* d_active, const float* d_c0, const float* d_c4,
                          int ldJ, const float* d_ptsS, const int* d_nS, int nS_stride, float* d_yS, float* d_JS,
                          int pose_dim, const float* d_ptsR, const int* d_nR, int nR_stride, float* d_yR,
                          void* d_maskR, hipStream_t stream);
int launch_decoder_h_fwd_masks(const hm_decoder_s* dec, int B, const int* d_active, const float* d_c0,
                               const float* d_c4, const float* d_ptsR, const int* d_nR, int nR_stride, float* d_yR,
                               void* d_maskR, hipStream_t stream);
int launch_decoder_h_bwd(const hm_decoder_s* dec, int B, const int* d_active, const float* d_c0, const float* d_c4,
                         int ldJ, const float* d_ptsG, const int* d_nG, int nG_stride, float* d_JG, int pose_dim,
                         const int* d_srcG, const float* d_yR, const void* d_maskR, int nR_stride,
                         hipStream_t stream);

// Scratch of a launch (K1p's ReLU-mask block, the any-architecture kernels' backward slabs): ONE block per (device, stream), kept by
// the library and grown on demand (hm_pack.hip).  Launches on one stream run one after the other, so they can share the block;
// launches on different streams -- the instance groups of hm_optimize_batch, concurrent calls, host threads -- never do (the
// round-5 per-decoder block, indexed by blockIdx.x only, was shared by concurrent groups: ADVICE r05).  Round 6 first used a
// stream-ordered allocation per launch (hipMallocAsync / hipFreeAsync): correct, but the pool hands a block freed on one stream to
// the next allocation on ANOTHER stream behind an internal dependency, which tied the two instance groups of a call together
// (whole any-architecture jobs 129 -> 119 instances/s) and, with the default release threshold, re-allocated after every
// synchronisation (40-80 ms outliers).  No allocation call is left on the launch path once a stream has seen its largest launch.
int scratch_get(void** p, size_t bytes, hipStream_t stream);

int launch_decoder_p(const hm_decoder_s* dec, int B, const float* d_pts, const int* d_nq, const int* d_active,
                     int n_stride, const float* d_c0, const float* d_c4, float* d_y, float* d_J, int ldJ,
                     int pose_dim, int mode, hipStream_t stream, int tag);

// any-architecture decoder (hm_decoder_any.hip): same contract as launch_decoder; d_zc = the c0 buffer, holding the latent
int launch_latent_copy_any(const hm_decoder_s* dec, const float* d_latent, int ld_latent, const int* d_active, int B,
                           float* d_zc, hipStream_t stream);
int launch_decoder_any(const hm_decoder_s* dec, int B, const float* d_pts, const int* d_nq, const int* d_active,
                       int n_stride, const float* d_zc, float* d_y, float* d_J, int ldJ, int pose_dim, int mode,
                       hipStream_t stream);

int launch_normal_eq(const RowSegment* segs, int n_seg, int L, int B, const int* d_active, float* d_Hext,
                     hipStream_t stream, int split_f16 = 0);    // split_f16: K4h (fp16 MFMA on split operands) for the f16x3 arithmetics

int launch_solve_update(const SolveArgs& args, int B, hipStream_t stream);

int launch_transform_points(const float* d_points_w, int n_in_stride, const int* d_n, const float* d_T_ow,
                            const int* d_active, int B, int n_stride, float* d_pts4, hipStream_t stream);

int launch_render_front(const RenderCfg& cfg, const RenderBuffers& rb, const float* d_T_ow, const int* d_active,
                        int B, hipStream_t stream, const float* d_frame_override = nullptr);   // frame setup + ray sampling
int launch_render_promote(const RenderCfg& cfg, const RenderBuffers& rb, const int* d_active, int B,
                          hipStream_t stream);               // screening: far / promoted split of the ball-valid samples
int launch_render_scan(const RenderCfg& cfg, const RenderBuffers& rb, const int* d_active, int B,
                       hipStream_t stream);                  // ray scan + offsets + scatter
int launch_render_reduce(const RenderCfg& cfg, const RenderBuffers& rb, const int* d_active, int B, int L,
                         hipStream_t stream);                // per-ray sum of sample Jacobians

}  // namespace hm
